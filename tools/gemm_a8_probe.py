"""Dev tool: one FP8 x FP8 context-encoding GEMM shape through mi_op_qlinear_a8, timed with events.
    MI355X_A8_WIDE=0|1 python tools/gemm_a8_probe.py N K M [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd import _native as lib
L = lib.load_library()
N, K = int(sys.argv[1]), int(sys.argv[2])
w = torch.randint(0, 0x70, (N * K,), dtype=torch.uint8, device="cuda")      # random fp8 codes (no NaN)
scale = torch.ones(N, dtype=torch.float32, device="cuda")
for M in (int(v) for v in sys.argv[3:]):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.float32, device="cuda")
    def launch():
        lib.check(L.mi_op_qlinear_a8(x.data_ptr(), M, w.data_ptr(), scale.data_ptr(), None, N, K, y.data_ptr(), None))
    for _ in range(3): launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps): launch()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"A8_WIDE={os.environ.get('MI355X_A8_WIDE', 'auto')} M={M} N={N} K={K}: {us:.1f} us (incl. row quantization)  {2*M*N*K/us/1e9:.3f} PF/s", flush=True)
