"""Dev tool: engine decode rate under vLLM's native scheduler (chunked prefill on): 4 requests of 900 prompt
tokens, 128 new tokens each.    python tools/chunked_rate.py [max_num_batched_tokens=2048]
A small budget (256) makes most steps MIXED records (a prompt chunk + requests that generate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from vllm_neuron_amd._vllm_compat import SamplingParams
from vllm_neuron_amd.engine import MI355XEngine

hf = SimpleNamespace(**bench.MODELS["llama31_8b"])
override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": bench.BUCKETS,
            "pa_num_blocks": bench.PA_NUM_BLOCKS, "quantized": True, "quantization_dtype": "f8e4m3",
            "quantization_type": "per_channel_symmetric", "chunked_prefill_config": {"max_num_seqs": 4}}
eng = MI355XEngine(hf, max_model_len=bench.MAX_MODEL_LEN, max_num_seqs=4, block_size=32,
                   num_gpu_blocks_override=bench.PA_NUM_BLOCKS, enable_prefix_caching=True,
                   enable_chunked_prefill=True, max_num_batched_tokens=int(sys.argv[1]) if len(sys.argv) > 1 else 2048, override_mi355x_config=override)
g = torch.Generator().manual_seed(3)
for rep in range(2):
    prompts = [torch.randint(0, hf.vocab_size, (900,), generator=g).tolist() for _ in range(4)]
    t = time.perf_counter()
    outs = eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=128))
    dt = time.perf_counter() - t
    first = max(o.ttft_s for o in outs)
    print(f"chunked-prefill engine: {(sum(len(o.token_ids) for o in outs) - 4) / (dt - first):.1f} tok/s (decode), first token {first * 1e3:.1f} ms, "
          f"whole job {dt * 1e3:.1f} ms", flush=True)
