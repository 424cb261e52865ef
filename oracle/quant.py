"""Oracle: symmetric weight quantizer (TEST INFRASTRUCTURE ONLY).

Restates the semantics selected by the reference's override keys
``quantized`` / ``quantization_type`` (default ``per_tensor_symmetric``) /
``quantization_dtype`` (default ``int8``)
(/root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:886-898).
The quantizer itself lives in NxDI (``save_quantized_state_dict``,
loader.py:238-239), which is absent here and whose outputs no reference test
checks -> PARITY UNPINNED.  The statement below is therefore the definition the
HIP path is held to, bit-exactly:

    amax  = max |W|  over the whole tensor (per_tensor) or each output row
            (per_channel);  rows that are all zero use scale 1
    scale = amax / QMAX                (fp32 division; QMAX = 127 int8, 448 e4m3fn)
    q     = rne(clamp(W / scale, -QMAX, QMAX))   (fp32 division, round-nearest-even)
    W'    = q * scale                   (weight-only: activations stay bf16)

fp8 is OCP e4m3fn (what gfx950 MFMA consumes), not the fnuz encoding.
"""

from __future__ import annotations

import torch

QMAX = {"int8": 127.0, "f8e4m3": 448.0}
QUANT_TYPES = ("per_tensor_symmetric", "per_channel_symmetric")


def _amax(w: torch.Tensor, quantization_type: str) -> torch.Tensor:
    a = w.abs()
    if quantization_type == "per_tensor_symmetric":
        return a.max().reshape(1).expand(w.shape[0]).contiguous()
    if quantization_type == "per_channel_symmetric":
        return a.amax(dim=1)
    raise ValueError(f"unknown quantization_type {quantization_type!r}")


def quantize_weight(w: torch.Tensor, quantization_dtype: str = "int8",
                    quantization_type: str = "per_tensor_symmetric"):
    """w: [N, K] float -> (q [N, K] int8 | float8_e4m3fn, scale [N] fp32).

    The scale is always returned per output channel; per-tensor mode
    replicates the single scalar, which is numerically identical.
    """
    if quantization_dtype not in QMAX:
        raise ValueError(f"unknown quantization_dtype {quantization_dtype!r}")
    w = w.to(torch.float32)
    qmax = QMAX[quantization_dtype]
    amax = _amax(w, quantization_type)
    scale = torch.where(amax > 0, amax / qmax, torch.ones_like(amax))
    t = (w / scale[:, None]).clamp_(-qmax, qmax)
    if quantization_dtype == "int8":
        q = torch.round(t).to(torch.int8)           # torch.round is half-to-even
    else:
        q = t.to(torch.float8_e4m3fn)                # RNE; |t| <= 448 so no overflow
    return q, scale.to(torch.float32)


def dequantize_weight(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.to(torch.float32) * scale[:, None]
