"""Dev tool: the concurrent single-GPU loopback (one stream per shard, device flag waits) at a given TP degree.
    MI355X_TP_LOOPBACK_CONCURRENT=1 GPU_MAX_HW_QUEUES=16 python tools/tp_concurrent_probe.py 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_tp_group_gpu import _group_model, BS
from oracle.paged_decoder import DecoderConfig
tp = int(sys.argv[1])
cfg = DecoderConfig(num_layers=1, hidden_size=256, num_heads=8, num_kv_heads=8, head_dim=64, intermediate_size=512,
                    vocab_size=512, rms_norm_eps=1e-5, rope_theta=10000.0)
try:
    m = _group_model(cfg, tp, "bf16", "per_tensor_symmetric", max_model_len=2048, ctx_buckets=[2048], num_blocks=2 * (2048 // BS) + 1, max_num_seqs=2)
    print("OK", m.tp_info()["mode"], flush=True)
    bufs = [torch.ones(1024).cuda() * (r + 1) for r in range(tp)]
    for _ in range(20):
        m.tp_all_reduce(bufs)
    print("20 exchanges OK", flush=True)
except Exception as e:
    print("FAIL", str(e)[:300], flush=True)
