"""Dev tool: context-encoding time per bucket through mi_forward (weight-only and FP8 x FP8), Llama-3.1-8B shapes.
    python tools/ttft_ab.py [a8=0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import prefill_inputs
from tests.test_fullsize_properties_gpu import LLAMA31_8B
a8 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
BS, MAXLEN, NSEQ, NB = 32, 2048, 4, 4097
MB = MAXLEN // BS
m = NativeModel(**LLAMA31_8B, num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=a8)
m.init_synthetic_weights(1, 0.02)
m.finalize()
g = torch.Generator().manual_seed(0)
blocks = list(range(1, MB + 1))
res = {}
for N in (256, 512, 1024, 2048):
    p = torch.randint(0, 128256, (N - 17,), generator=g).tolist()
    inp = prefill_inputs(p, blocks, BS, MAXLEN, 0)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); m.forward(**inp); ts.append((time.perf_counter() - t) * 1e3)
    res[N] = round(min(ts[1:]), 2)
print(f"a8={a8} GEMM_WIDE={os.environ.get('MI355X_GEMM_WIDE', 'auto')}: prefill ms {res}", flush=True)
