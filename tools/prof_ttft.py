"""Dev tool: host-side profile of one context-encoding request through the engine.
    python tools/prof_ttft.py [a8=0|1]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from vllm_neuron_amd._vllm_compat import SamplingParams
from vllm_neuron_amd.engine import MI355XEngine

hf = SimpleNamespace(**bench.MODELS["llama31_8b"])
override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": bench.BUCKETS,
            "pa_num_blocks": bench.PA_NUM_BLOCKS, "quantized": True, "quantization_dtype": "f8e4m3",
            "quantization_type": "per_channel_symmetric", "prefill_fp8_activations": len(sys.argv) > 1 and sys.argv[1] == "1"}
eng = MI355XEngine(hf, max_model_len=bench.MAX_MODEL_LEN, max_num_seqs=bench.MAX_NUM_SEQS,
                   block_size=bench.BLOCK_SIZE, num_gpu_blocks_override=bench.PA_NUM_BLOCKS,
                   enable_prefix_caching=True, tensor_parallel_size=1, override_mi355x_config=override)
import gc
native = eng.worker.model_runner.model.model
_fw = native.forward
acc = {"fw": 0.0, "gc": 0.0}
def timed_forward(*a, **kw):
    t = time.perf_counter(); r = _fw(*a, **kw); acc["fw"] += time.perf_counter() - t; return r
native.forward = timed_forward
_g = {}
def gccb(phase, info):
    if phase == "start": _g["t"] = time.perf_counter()
    else: acc["gc"] += time.perf_counter() - _g["t"]; print("   gc gen", info["generation"], flush=True)
gc.callbacks.append(gccb)
g = torch.Generator().manual_seed(0)
for n in (256, 2048):
    for i in range(9):
        prompt = torch.randint(0, hf.vocab_size, (n - 17,), generator=g).tolist()
        acc["fw"] = acc["gc"] = 0.0
        t = time.perf_counter()
        out = eng.generate([prompt], SamplingParams(temperature=0.0, max_tokens=1))[0]
        print(f"bucket {n}: ttft {out.ttft_s*1e3:.2f} total {(time.perf_counter()-t)*1e3:.2f} native {acc['fw']*1e3:.2f} gc {acc['gc']*1e3:.2f}", flush=True)

pr = cProfile.Profile()
prompt = torch.randint(0, hf.vocab_size, (2048 - 17,), generator=g).tolist()
pr.enable()
out = eng.generate([prompt], SamplingParams(temperature=0.0, max_tokens=1))[0]
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
