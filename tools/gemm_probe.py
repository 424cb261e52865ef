"""Dev tool: one context-encoding GEMM shape through mi_op_qlinear (path 2 = 128x128 kernel, 3 = wide-N
LDS-DMA kernel), timed with events; run under rocprofv3 --pmc for counters.
    python tools/gemm_probe.py M N K path [wd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd import _native as lib
L = lib.load_library()
M, N, K, path = (int(v) for v in sys.argv[1:5])
WD = int(sys.argv[5]) if len(sys.argv) > 5 else 1
w = torch.randint(0, 0x70, (N * K,), dtype=torch.uint8, device="cuda")      # random fp8 codes (no NaN)
scale = torch.ones(N, dtype=torch.float32, device="cuda")
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.float32, device="cuda")
def launch():
    lib.check(L.mi_op_qlinear(x.data_ptr(), M, w.data_ptr(), scale.data_ptr(), None, N, K, WD, y.data_ptr(), path, None))
for _ in range(3): launch()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", 20))
a.record()
for _ in range(reps): launch()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / reps
print(f"M={M} N={N} K={K} path={path}: {us:.1f} us  {2*M*N*K/us/1e9:.3f} PF/s", flush=True)
