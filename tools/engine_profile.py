"""Dev tool: where the host time of an engine decode step goes (cProfile over 4 x 128 decode tokens)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from vllm_neuron_amd._vllm_compat import SamplingParams
from vllm_neuron_amd.engine import MI355XEngine

hf = SimpleNamespace(**bench.MODELS["llama31_8b"])
override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": bench.BUCKETS,
            "pa_num_blocks": bench.PA_NUM_BLOCKS, "quantized": True, "quantization_dtype": "f8e4m3",
            "quantization_type": "per_channel_symmetric"}
if len(sys.argv) > 1 and sys.argv[1] == "device":
    override["on_device_sampling_config"] = {"dynamic": True}
eng = MI355XEngine(hf, max_model_len=bench.MAX_MODEL_LEN, max_num_seqs=bench.MAX_NUM_SEQS,
                   block_size=bench.BLOCK_SIZE, num_gpu_blocks_override=bench.PA_NUM_BLOCKS,
                   enable_prefix_caching=True, override_mi355x_config=override)
g = torch.Generator().manual_seed(3)
def run(n=128):
    prompts = [torch.randint(0, hf.vocab_size, (900,), generator=g).tolist() for _ in range(4)]
    t = time.perf_counter()
    outs = eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=n))
    dt = time.perf_counter() - t
    first = max(o.ttft_s for o in outs)
    return (sum(len(o.token_ids) for o in outs) - 4) / (dt - first)
print("warm rate", run(32))
pr = cProfile.Profile()
pr.enable()
r = run(128)
pr.disable()
print("profiled rate", r)
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
print("unprofiled rate", run(128))
