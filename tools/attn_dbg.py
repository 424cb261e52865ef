"""Dev tool: structured inputs for the context-encoding attention op (which index is mapped wrong?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_neuron_amd import _native as lib
from tests.helpers import ref_attention, pool_to_native
L = lib.load_library()
hd, nh, nkv, bs = int(os.environ.get("HD", 128)), 8, 2, 32
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pos0 = 0
kv_len = T
MB = (kv_len + bs - 1) // bs + 1
nb = 1 + MB
bt = torch.zeros(MB, dtype=torch.int32)
bt[:MB - 1] = torch.arange(1, MB).int()

def run(q, k, v, tag):
    pool = pool_to_native(k.to(torch.bfloat16), v.to(torch.bfloat16), nb, bs).cuda()
    qd = q.to(torch.bfloat16).cuda()
    out = torch.empty(T, nh * hd, dtype=torch.bfloat16, device="cuda")
    btd = bt.cuda()
    lib.check(L.mi_op_paged_attn_prefill(qd.data_ptr(), T, pos0, pool.data_ptr(), nb, bs, btd.data_ptr(), MB, nh, nkv, hd,
                                         out.data_ptr(), None))
    torch.cuda.synchronize()
    idx = torch.arange(kv_len)
    blk = bt[idx // bs].long()
    kf = k.to(torch.bfloat16).float().reshape(nb, bs, nkv, hd)
    vf = v.to(torch.bfloat16).float().reshape(nb, bs, nkv, hd)
    ref = ref_attention(q.to(torch.bfloat16).float(), kf[blk, idx % bs], vf[blk, idx % bs], pos0 + torch.arange(T))
    o = out.cpu().float().reshape(T, nh, hd)
    r = ref.reshape(T, nh, hd)
    err = (o - r).abs()
    print(f"[{tag}] max err {err.max().item():.4f}; per head {[round(e, 3) for e in err.amax(dim=(0, 2)).tolist()]}")
    print(f"   per row (first 8) {[round(e, 3) for e in err.amax(dim=(1, 2))[:8].tolist()]}  per dim-block of 8: {[round(e, 2) for e in err.amax(dim=(0, 1)).reshape(-1, 8).amax(1).tolist()]}")
    if err.max() > 0.05:
        t, h = divmod(int(err.amax(dim=2).argmax()), nh)
        print("   got", [round(x, 2) for x in o[t, h, :16].tolist()])
        print("   ref", [round(x, 2) for x in r[t, h, :16].tolist()])

N = nb * bs
zq = torch.zeros(T, nh, hd)
rq = torch.randn(T, nh, hd)
rk, rv = torch.randn(N, nkv, hd), torch.randn(N, nkv, hd)
# A: uniform attention, V[key] = position in the pool row (block b, offset o -> b * 32 + o): out = mean of visible positions
vpos = torch.arange(N).float()[:, None, None].expand(N, nkv, hd) / 16
run(zq, rk, vpos, "Q=0, V=key/16")
# B: V[key][d] = d / 16: the softmax cannot matter
vd = (torch.arange(hd).float() / 16)[None, None, :].expand(N, nkv, hd)
run(rq, rk, vd.clone(), "V=d/16")
run(zq, rk, rv, "Q=0, V random")
run(rq, rk, vpos, "random QK, V=key/16")
run(rq, rk, rv, "random")
