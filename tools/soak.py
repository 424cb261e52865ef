"""Dev tool: soak the whole serving loop for N seconds at Llama-3.1-8B shapes -- random prompt
lengths (some sharing prefixes), staggered arrivals into the continuous batch, greedy and random
sampling, CPU and on-device samplers -- and check that nothing leaks or drifts: device free memory
flat, the same request set reproduces the same greedy ids at the start and at the end."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from vllm_neuron_amd._vllm_compat import SamplingParams
from vllm_neuron_amd.engine import MI355XEngine

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
hf = SimpleNamespace(**bench.MODELS["llama31_8b"])
override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": bench.BUCKETS,
            "pa_num_blocks": bench.PA_NUM_BLOCKS, "quantized": True, "quantization_dtype": "f8e4m3",
            "quantization_type": "per_channel_symmetric", "prefill_fp8_activations": True}
eng = MI355XEngine(hf, max_model_len=bench.MAX_MODEL_LEN, max_num_seqs=bench.MAX_NUM_SEQS,
                   block_size=bench.BLOCK_SIZE, num_gpu_blocks_override=bench.PA_NUM_BLOCKS,
                   enable_prefix_caching=True, tensor_parallel_size=1, override_mi355x_config=override)
native = eng.worker.model_runner.model.model
adapter = eng.worker.model_runner.model
rng = random.Random(0)
g = torch.Generator().manual_seed(0)
base = torch.randint(0, hf.vocab_size, (1500,), generator=g).tolist()
probe = [base[:700], base[:300] + torch.randint(0, hf.vocab_size, (200,), generator=g).tolist()]
greedy = SamplingParams(temperature=0.0, max_tokens=24)
first = [o.token_ids for o in eng.generate(probe, greedy)]
free0 = native.kv_stats()["device_free_bytes"]
t0, nreq, ntok = time.time(), 0, 0
while time.time() - t0 < SECONDS:
    batch = []
    for _ in range(rng.randint(1, 6)):
        n = rng.choice([1, 5, 31, 32, 33, rng.randint(2, 1900)])
        p = (base[:rng.randint(0, min(n, 1400))] + torch.randint(0, hf.vocab_size, (n,), generator=g).tolist())[:n]
        batch.append(p or [1])
    sp = greedy if rng.random() < 0.5 else SamplingParams(temperature=0.9, top_k=50, top_p=0.9, max_tokens=rng.randint(1, 40))
    adapter.mi355x_config.on_device_sampling_config = {"dynamic": True} if rng.random() < 0.5 else None
    outs = eng.generate(batch, sp)
    assert all(o.finished and all(0 <= t < hf.vocab_size for t in o.token_ids) for o in outs)
    nreq += len(outs); ntok += sum(len(o.token_ids) for o in outs)
adapter.mi355x_config.on_device_sampling_config = None
outs_last = eng.generate(probe, greedy)
last = [o.token_ids for o in outs_last]
again = [o.token_ids for o in eng.generate(probe, greedy)]
free1 = native.kv_stats()["device_free_bytes"]
print("cached tokens of the probe at the end:", [o.num_cached_tokens for o in outs_last])
for a, b in zip(first, last):
    n = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(a))
    print("  first", a[:8], "last", b[:8], "agree for", n, "of", len(a))
print("two runs in a row at the end identical:", last == again)
print(f"soak {time.time() - t0:.0f} s: {nreq} requests, {ntok} generated tokens; probe ids reproduced: {first == last}; "
      f"device free {free0} -> {free1} bytes")
assert last == again and abs(free1 - free0) < (64 << 20)
