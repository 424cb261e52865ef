"""Dev tool: context-encoding attention alone (mi_op_paged_attn_prefill) at the Llama-3.1-8B head geometry, per bucket.
    python tools/attn_prefill_time.py            (MI355X_ATTN_PREFILL_V1=1 for the first-generation kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd import _native as lib
L = lib.load_library()
nh, nkv, hd, bs = 32, 8, 128, 32
cases = [(239, 0), (495, 0), (1007, 0), (2031, 0)]
if len(sys.argv) > 1:   # T:pos0 pairs
    cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]
for T, pos0 in cases:
    MB = (pos0 + T + bs - 1) // bs + 1
    nb = 1 + MB
    pool = torch.randn(2, nb, nkv, bs, hd, device="cuda").to(torch.bfloat16)
    bt = torch.zeros(MB, dtype=torch.int32)
    bt[:MB - 1] = (torch.randperm(nb - 1) + 1)[:MB - 1].int()
    btd = bt.cuda()
    q = torch.randn(T, nh, hd, device="cuda").to(torch.bfloat16)
    out = torch.empty(T, nh * hd, dtype=torch.bfloat16, device="cuda")
    def launch():
        lib.check(L.mi_op_paged_attn_prefill(q.data_ptr(), T, pos0, pool.data_ptr(), nb, bs, btd.data_ptr(), MB, nh, nkv, hd,
                                             out.data_ptr(), None))
    for _ in range(5): launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    a.record()
    for _ in range(reps): launch()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    fl = 4 * nh * hd * (T * T / 2 + T * pos0)
    print(f"T={T} pos0={pos0}: {us:.1f} us  {fl / us / 1e9:.3f} PF/s  (v1={os.environ.get('MI355X_ATTN_PREFILL_V1', '0')})", flush=True)
