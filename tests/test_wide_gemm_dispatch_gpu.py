"""The wide context-encoding GEMMs (gemm_wide_kernel incl. its K-split, gemm_a8_wide_kernel) with their REAL epilogues
(RoPE + K/V scatter, SwiGLU, residual, slabs summed by the next norm) against the 128 x 128 kernels they replaced, at
the Llama-3.1-8B shapes the dispatch plan was measured on (the zoo models are too small to reach the wide tile).

Both runs compute the same arithmetic with the same rounding points; only the fp32 summation order inside a projection
differs (tile shapes, K-split slabs).  The dispatch switches are read once per process, so each side runs in a child.
Bar: rms of the logit difference below 2 % of the logit standard deviation and below the smallest top-2 gap that the
weight-only run decides a token by ... i.e. the same first token wherever the reference run's own top-2 gap exceeds 4 x rms.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from vllm_neuron_amd._native import NativeModel, MI_W, MI_Q
from tests.helpers import prefill_inputs
from tests.test_fullsize_properties_gpu import LLAMA31_8B, QWEN25_7B
a8, out, which = int(sys.argv[1]), sys.argv[2], sys.argv[3]
geo = dict(QWEN25_7B if which == "qwen25_7b_int8" else LLAMA31_8B, num_layers=2)
wd = "int8" if which == "qwen25_7b_int8" else "f8e4m3"
BS, MAXLEN = 32, 2048
m = NativeModel(**geo, num_blocks=129, block_size=BS, max_num_seqs=4, max_model_len=MAXLEN,
                weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                prefill_fp8_activations=a8)
m.init_synthetic_weights(1, 0.02)
m.finalize()
g = torch.Generator().manual_seed(5)
res = {}
for n in (239, 300, 700, 1100, 2031):
    prompt = torch.randint(0, geo["vocab_size"], (n,), generator=g).tolist()
    blocks = list(range(1, 1 + (n + BS - 1) // BS))
    res[n] = m.forward(**prefill_inputs(prompt, blocks, BS, MAXLEN)).float().cpu().clone()
torch.save(res, out)
m.close()
print("CHILD_OK", flush=True)
"""


def _run(tmp_path, a8, tag, env_extra, which="llama31_8b_fp8"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"logits_{tag}.pt")
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": root}, str(a8), out, which], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    return torch.load(out, weights_only=True)


@pytest.mark.parametrize("which,a8", [("llama31_8b_fp8", 0), ("llama31_8b_fp8", 1), ("qwen25_7b_int8", 0)])
def test_wide_tiles_agree_with_the_128x128_kernels(tmp_path, which, a8):
    """Qwen2.5-7B INT8 (BASELINE config 4): the plan's INT8 constants, q/k/v biases, 7 q heads per kv head (the one-block,
    two-pass form of the context-encoding attention)."""
    new = _run(tmp_path, a8, "wide", {}, which)
    old = _run(tmp_path, a8, "plain", {"MI355X_GEMM_WIDE": "0", "MI355X_A8_WIDE": "0"}, which)
    for n, want in old.items():
        got = new[n]
        assert got.shape == want.shape and torch.isfinite(got).all()
        rms = (got - want).pow(2).mean().sqrt().item()
        std = want.std().item()
        assert rms < 0.02 * std, (a8, n, rms, std)
        top2 = want.topk(2, dim=-1).values
        decided = (top2[:, 0] - top2[:, 1]) > 4 * rms
        assert (got.argmax(-1)[decided] == want.argmax(-1)[decided]).all(), (a8, n)
