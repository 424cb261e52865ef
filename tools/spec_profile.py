"""Dev tool: per-kernel-class time of a token-generation pass at M rows (the target's pass of a speculation step).
    python tools/spec_profile.py [rows=16] [k=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from tests.helpers import decode_inputs
from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
from vllm_neuron_amd.worker.mi355x_model_loader import _decoder_geometry

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = rows // k
BS, MAXLEN = 32, 2048
mb = MAXLEN // BS
m = NativeModel(num_blocks=1 + B * mb, block_size=BS, max_num_seqs=rows, max_model_len=MAXLEN, ctx_buckets=bench.BUCKETS,
                weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1, tp_degree=1,
                tp_rank=0, device_id=0, use_graphs=1, prefill_fp8_activations=0,
                **_decoder_geometry(SimpleNamespace(**bench.MODELS["llama31_8b"])))
m.init_synthetic_weights(1, 0.02)
m.finalize()
blocks = [[1 + b * mb + j for j in range(mb)] for b in range(B)]
inp = decode_inputs([1] * rows, [1023 - (i % k) for i in range(rows)], [blocks[i // k] for i in range(rows)], BS, MAXLEN)
for _ in range(3):
    m.forward(**inp)
m.replay_decode(8)
print(f"rows {rows}: {m.replay_decode(100) / 100:.4f} ms per pass (graph)")
for cls in (["gemv"], ["attn_decode"], ["gemv", "attn_decode"]):
    m.forward(**inp)
    try:
        print(cls, f"{m.replay_decode_classes(100, cls) / 100:.4f} ms")
    except Exception as e:
        print(cls, e)
m.forward(**inp)
m.profile_enable(True)
m.forward(**inp)
p = m.profile_read()
m.profile_enable(False)
print({c: (p["launches"][c], round(p["ms"][c], 3)) for c in p["launches"] if p["launches"][c]})
