"""Quick direct bench of the native model (dev tool; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import prefill_inputs, decode_inputs

wd = sys.argv[1] if len(sys.argv) > 1 else "f8e4m3"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
A8 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
BS, MAXLEN, NSEQ = 32, 2048, 4
MB = MAXLEN // BS
NB = 4096 + 1
t0 = time.time()
m = NativeModel(num_layers=L, hidden_size=4096, num_heads=32, num_kv_heads=8, head_dim=128,
                intermediate_size=14336, vocab_size=128256, rms_norm_eps=1e-5, rope_theta=500000.0,
                rope_type=1, rope_factor=8.0, rope_low_freq_factor=1.0, rope_high_freq_factor=4.0,
                rope_original_max_position=8192, qkv_bias=0, tie_word_embeddings=0,
                num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=A8)
m.init_synthetic_weights(1, 0.02)
m.finalize()
print("init s", time.time() - t0, m.kv_stats(), flush=True)
g = torch.Generator().manual_seed(0)
perm = (torch.randperm(NB - 1, generator=torch.Generator().manual_seed(2)) + 1).tolist()
blocks = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
for N in (256, 512, 1024, 2048):
    plen = N - 17
    prompt = torch.randint(0, 128256, (plen,), generator=g).tolist()
    ts = []
    for rep in range(3):
        inp = prefill_inputs(prompt, blocks[0], BS, MAXLEN, 0)
        t = time.time(); lg = m.forward(**inp); ts.append(time.time() - t)
    print(f"prefill bucket {N} (len {plen}): {min(ts)*1e3:.2f} ms  (runs {[round(x*1e3,2) for x in ts]})", flush=True)
# decode at ctx
for ctx in (256, 1024, 2040):
    pos = [ctx - 1] * NSEQ
    toks = [1, 2, 3, 4]
    inp = decode_inputs(toks, pos, blocks, BS, MAXLEN)
    for _ in range(3): m.forward(**inp)
    n = 50
    t = time.time()
    for _ in range(n): lg = m.forward(**inp)
    dt = (time.time() - t) / n
    print(f"decode B=4 ctx={ctx}: {dt*1e3:.3f} ms/step -> {NSEQ/dt:.0f} tok/s", flush=True)
m.profile_enable(True)
for _ in range(5): m.forward(**inp)
p = m.profile_read(); m.profile_enable(False)
print("profile (5 eager steps):", p)
gb = p["gemv_weight_bytes"] / 1e9
print(f"gemv: {gb/ (p['ms']['gemv']/1e3):.0f} GB/s over {p['launches']['gemv']} launches")
