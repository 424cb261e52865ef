"""Dev tool: device-resident decode step time (graph replay) of the Llama-3.1-8B shapes.
    python tools/decode_ab.py [wd=f8e4m3] [ctx=1024] [steps=200] [model=llama|qwen]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import decode_inputs
from tests.test_fullsize_properties_gpu import LLAMA31_8B, QWEN25_7B

wd = sys.argv[1] if len(sys.argv) > 1 else "f8e4m3"
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
geo = QWEN25_7B if (len(sys.argv) > 4 and sys.argv[4] == "qwen") else LLAMA31_8B
BS, MAXLEN, NSEQ, NB = 32, 2048, int(os.environ.get('ROWS', 4)), 4097   # ROWS=n: rows per step
MB = MAXLEN // BS
m = NativeModel(**geo, num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=int(os.environ.get('MI355X_GRAPHS', '1')), ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=0)
m.init_synthetic_weights(1, 0.02)
m.finalize()
perm = (torch.randperm(NB - 1, generator=torch.Generator().manual_seed(2)) + 1).tolist()
blocks = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
inp = decode_inputs(list(range(1, NSEQ + 1)), [ctx - 1] * NSEQ, blocks, BS, MAXLEN)
for _ in range(3):
    m.forward(**inp)
m.replay_decode(20)
best = min(m.replay_decode(steps) / steps for _ in range(3))
import time
t = time.perf_counter()
for _ in range(200):
    m.forward(**inp)
fw = (time.perf_counter() - t) / 200 * 1e3
t = time.perf_counter()
for _ in range(200):
    m.forward_tokens(**inp)
ft = (time.perf_counter() - t) / 200 * 1e3
print(f"forward loop {fw:.4f} ms/call, forward_tokens loop {ft:.4f} ms/call (graphs={os.environ.get('MI355X_GRAPHS', '1')})")
print(f"{wd} ctx={ctx}: {best:.4f} ms/step -> {NSEQ / best * 1e3:.0f} tok/s  (env PRIV={os.environ.get('MI355X_GEMV_PRIV', '1')})", flush=True)
