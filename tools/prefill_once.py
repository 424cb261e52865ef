"""Dev tool: a few context-encoding calls of ONE bucket through mi_forward (Llama-3.1-8B shapes), for rocprofv3.
    python tools/prefill_once.py bucket [a8=0|1] [layers=32] [reps=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import prefill_inputs
from tests.test_fullsize_properties_gpu import LLAMA31_8B
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
a8 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
geo = dict(LLAMA31_8B, num_layers=layers)
BS, MAXLEN, NSEQ, NB = 32, 2048, 4, 4097
MB = MAXLEN // BS
m = NativeModel(**geo, num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=a8)
m.init_synthetic_weights(1, 0.02)
m.finalize()
p = torch.randint(0, 128256, (N - 17,), generator=torch.Generator().manual_seed(0)).tolist()
inp = prefill_inputs(p, list(range(1, MB + 1)), BS, MAXLEN, 0)
ts = []
for _ in range(reps + 1):
    t = time.perf_counter(); m.forward(**inp); ts.append((time.perf_counter() - t) * 1e3)
print(f"bucket {N} a8={a8} layers={layers}: prefill ms min {min(ts[1:]):.3f} all {[round(t, 2) for t in ts]}", flush=True)
m.close()
