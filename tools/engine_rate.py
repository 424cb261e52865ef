"""Dev tool: end-to-end decode rate through the engine loop (scheduler -> runner -> library -> sampler)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from vllm_neuron_amd._vllm_compat import SamplingParams
from vllm_neuron_amd.engine import MI355XEngine

hf = SimpleNamespace(**bench.MODELS["llama31_8b"])
override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": bench.BUCKETS,
            "pa_num_blocks": bench.PA_NUM_BLOCKS, "quantized": True, "quantization_dtype": "f8e4m3",
            "quantization_type": "per_channel_symmetric", "prefill_fp8_activations": True}
override.update(eval(sys.argv[1]) if len(sys.argv) > 1 else {})
eng = MI355XEngine(hf, max_model_len=bench.MAX_MODEL_LEN, max_num_seqs=bench.MAX_NUM_SEQS,
                   block_size=bench.BLOCK_SIZE, num_gpu_blocks_override=bench.PA_NUM_BLOCKS,
                   enable_prefix_caching=True, tensor_parallel_size=1, override_mi355x_config=override)
g = torch.Generator().manual_seed(0)
NEW = 256
def run(profile=False):
    prompts = [torch.randint(0, hf.vocab_size, (900,), generator=g).tolist() for _ in range(4)]
    sp = SamplingParams(temperature=0.0, max_tokens=NEW, ignore_eos=True) if "ignore_eos" in SamplingParams.__init__.__code__.co_varnames else SamplingParams(temperature=0.0, max_tokens=NEW)
    t = time.perf_counter()
    outs = eng.generate(prompts, sp)
    dt = time.perf_counter() - t
    ntok = sum(len(o.token_ids) for o in outs)
    ttft = max(o.ttft_s for o in outs)
    print(f"generated {ntok} tokens in {dt*1e3:.1f} ms; last first-token at {ttft*1e3:.1f} ms -> decode phase "
          f"{(ntok - 4) / (dt - ttft):.0f} tok/s ({(dt - ttft) / (ntok / 4 - 1) * 1e3:.3f} ms/step)", flush=True)
run(); run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
