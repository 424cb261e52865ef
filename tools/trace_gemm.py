"""Dev tool: in-kernel phase timeline of the context-encoding GEMMs (a -DMI_TRACE build of the
library via MI355X_VLLM_LIB=...).  Stamps: 0 start, 1..5 around K-step 4 (LDS write+barrier, load
issue, MFMA phase, closing barrier), 6 top of K-step 5 after the register rotate, 7 end."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vllm_neuron_amd import _native
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import prefill_inputs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = 4
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
blocks = list(range(1, MB + 1))
prompt = torch.randint(0, 128256, (N - 17,), generator=torch.Generator().manual_seed(0)).tolist()
inp = prefill_inputs(prompt, blocks, BS, MAXLEN, 0)
for _ in range(2):
    m.forward(**inp)
NL, NBLK, NST = 160, 512, 8
buf = torch.zeros(NL * NBLK * NST, dtype=torch.int64, device="cuda")
lib = _native.load_library()
lib.mi_debug_trace.argtypes = [ctypes.c_void_p]
assert lib.mi_debug_trace(buf.data_ptr()) == 0
m.forward(**inp)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NL, NBLK, NST).astype(np.float64) * 0.01
names = ["qkv", "o", "gate_up", "down"]
for k in range(4, 8):   # second layer
    live = (t[k, :, 0] > 0) & (t[k, :, 7] > 0)
    a = t[k, live]
    d = lambda i, j: float(np.median(a[:, j] - a[:, i]))
    print(names[k % 4], dict(blocks=int(live.sum()), to_step4=round(d(0, 1), 2), lds_bar=round(d(1, 2), 2),
                             issue=round(d(2, 3), 2), mfma=round(d(3, 4), 2), bar2=round(d(4, 5), 2),
                             rotate=round(d(5, 6), 2), kstep=round(d(1, 6), 2), total=round(d(0, 7), 2),
                             span=round(float(a[:, 7].max() - a[:, 0].min()), 1)))
