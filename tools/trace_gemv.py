"""Dev tool: in-kernel phase timeline of the decode GEMVs (needs a -DMI_TRACE build of the library,
passed as MI355X_VLLM_LIB=...; see the MI_TRACE block in csrc/linear_kernels.hip)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vllm_neuron_amd import _native
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import decode_inputs

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CTX = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
BS, MAXLEN, NSEQ, NB = 32, 2048, 4, 4097
MB = MAXLEN // BS
m = NativeModel(num_layers=L, hidden_size=4096, num_heads=32, num_kv_heads=8, head_dim=128,
                intermediate_size=14336, vocab_size=128256, rms_norm_eps=1e-5, rope_theta=500000.0,
                rope_type=1, rope_factor=8.0, rope_low_freq_factor=1.0, rope_high_freq_factor=4.0,
                rope_original_max_position=8192, qkv_bias=0, tie_word_embeddings=0,
                num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=1)
m.init_synthetic_weights(1, 0.02)
m.finalize()
perm = (torch.randperm(NB - 1, generator=torch.Generator().manual_seed(2)) + 1).tolist()
blocks = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
inp = decode_inputs([1, 2, 3, 4], [CTX - 1] * NSEQ, blocks, BS, MAXLEN)
for _ in range(3):
    m.forward(**inp)
m.replay_decode(20)
NL, NBLK, NST = 160, 512, 8
buf = torch.zeros(NL * NBLK * NST, dtype=torch.int64, device="cuda")
lib = _native.load_library()
lib.mi_debug_trace.argtypes = [ctypes.c_void_p]
assert lib.mi_debug_trace(buf.data_ptr()) == 0
m.replay_decode(1)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NL, NBLK, NST).astype(np.float64) * 0.01   # 100 MHz -> us
names = ["qkv", "o", "gate_up", "down"]
rows = {n: [] for n in names}
nk = 4 * L + 1
for k in range(nk):
    live = t[k, :, 0] > 0
    if not live.any():
        continue
    a = t[k, live]
    t0 = a[:, 0].min()
    nxt = t[k + 1, t[k + 1, :, 0] > 0] if k + 1 < nk else None
    rec = dict(grid=int(live.sum()), start_spread=a[:, 0].max() - t0,
               issue=np.median(a[:, 1] - a[:, 0]), stage=np.median(a[:, 2] - a[:, 1]),
               bar=np.median(a[:, 3] - a[:, 2]), first=np.median(a[:, 4] - a[:, 3]),
               stream=np.median(a[:, 5] - a[:, 4]), med_total=np.median(a[:, 5] - a[:, 0]),
               span=a[:, 5].max() - t0,
               gap_next=(nxt[:, 0].min() - a[:, 5].max()) if nxt is not None and len(nxt) else float("nan"))
    if k == nk - 1:
        print("lm_head", {k_: round(v, 2) for k_, v in rec.items()})
    elif k >= 8:
        rows[names[k % 4]].append(rec)
for n in names:
    rs = rows[n]
    print(n, {k_: round(float(np.mean([r[k_] for r in rs])), 2) for k_ in rs[0]})
step_span = t[nk - 1][t[nk - 1, :, 0] > 0][:, 5].max() - t[0][t[0, :, 0] > 0][:, 0].min()
print("first gemv start -> lm_head end: %.1f us" % step_span)
