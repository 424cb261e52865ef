"""Dev tool: time of the on-device sampler for 4 rows of a 128256-word vocabulary, greedy and top-k / top-p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd import _native

lib = _native.load_library()
B, V = 4, 128256
logits = torch.randn(B, V, device="cuda") * 1.3
out = torch.empty(B, dtype=torch.int32, device="cuda")
for name, params in (("greedy", None), ("top_k=50 top_p=0.9", [[50.0, 0.9, 0.8]] * B), ("top_k=256 top_p=1", [[256.0, 1.0, 1.0]] * B)):
    p = torch.tensor(params, dtype=torch.float32, device="cuda") if params else None
    def call():
        _native.check(lib.mi_op_sample(logits.data_ptr(), B, V, p.data_ptr() if p is not None else None, 7, out.data_ptr(), None))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t) / 200 * 1e6
    nb = lib.mi_op_sample_scratch_bytes(B)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def call_ws():
        _native.check(lib.mi_op_sample_ws(logits.data_ptr(), B, V, p.data_ptr() if p is not None else None, 7, out.data_ptr(), scratch.data_ptr(), nb, None))
    for _ in range(5):
        call_ws()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        call_ws()
    torch.cuda.synchronize()
    print(f"{name}: engine form (scratch, spread over the chip) {(time.perf_counter() - t) / 200 * 1e6:.1f} us; one work-group per row {t1:.1f} us")
    continue
    print(f"{name}: {(time.perf_counter() - t) / 200 * 1e6:.1f} us per call of {B} rows (per-op entry: one work-group per row)")
