"""Seeded synthetic decoder weights + the tiny model zoo the goldens are made on
(TEST INFRASTRUCTURE ONLY).

No checkpoints exist in the build image or on the GPU box (no network), so parity
is anchored on random-weight models whose HF-transformers outputs were recorded
once (``oracle/gen_golden.py``).  Weights are regenerated from the seed on both
sides instead of being shipped; ``weights_checksum`` guards against RNG drift.
Every matrix is bf16-representable so that the device's bf16 copies are exact.
"""

from __future__ import annotations

import torch

from .paged_decoder import DecoderConfig

# name -> (DecoderConfig kwargs, hf model_type).  Head dims / GQA ratios mirror the
# BASELINE configs: TinyLlama (hd 64, 8:1), Llama-3.1-8B (hd 128, 4:1, llama3 rope),
# Qwen2.5-7B (hd 128... here hd 64 to stay small, 7:1, qkv bias).
ZOO = {
    "tinyllama_like": dict(model_type="llama", num_layers=2, hidden_size=256, num_heads=8,
                           num_kv_heads=1, head_dim=64, intermediate_size=512, vocab_size=512,
                           rms_norm_eps=1e-5, rope_theta=10000.0),
    "llama31_like": dict(model_type="llama", num_layers=2, hidden_size=256, num_heads=8,
                         num_kv_heads=2, head_dim=128, intermediate_size=768, vocab_size=640,
                         rms_norm_eps=1e-5, rope_theta=500000.0,
                         rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0,
                                       "original_max_position_embeddings": 64}),
    "qwen25_like": dict(model_type="qwen2", num_layers=2, hidden_size=448, num_heads=7,
                        num_kv_heads=1, head_dim=64, intermediate_size=512, vocab_size=512,
                        rms_norm_eps=1e-6, rope_theta=1000000.0, qkv_bias=True),
}


def zoo_config(name: str) -> DecoderConfig:
    kw = dict(ZOO[name])
    kw.pop("model_type")
    return DecoderConfig(**kw)


def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def make_weights(cfg: DecoderConfig, seed: int = 1, std: float | None = None) -> dict:
    """HF-named state dict, fp32 tensors holding bf16-representable values.

    ``std=None``: fan-in scaled matrices (diverse, well-conditioned logits for parity
    work).  ``std=0.02``: the BASELINE synthetic distribution."""
    g = torch.Generator().manual_seed(seed)

    def mat(n, k):
        s = std if std is not None else (1.0 / k ** 0.5)
        return _bf16r(torch.randn(n, k, generator=g) * s)

    H, hd = cfg.hidden_size, cfg.head_dim
    w = {"model.embed_tokens.weight": _bf16r(torch.randn(cfg.vocab_size, H, generator=g))}
    for l in range(cfg.num_layers):
        p = f"model.layers.{l}."
        w[p + "self_attn.q_proj.weight"] = mat(cfg.num_heads * hd, H)
        w[p + "self_attn.k_proj.weight"] = mat(cfg.num_kv_heads * hd, H)
        w[p + "self_attn.v_proj.weight"] = mat(cfg.num_kv_heads * hd, H)
        if cfg.qkv_bias:
            for n, d in (("q", cfg.num_heads * hd), ("k", cfg.num_kv_heads * hd),
                         ("v", cfg.num_kv_heads * hd)):
                w[p + f"self_attn.{n}_proj.bias"] = _bf16r(torch.randn(d, generator=g) * 0.1)
        w[p + "self_attn.o_proj.weight"] = mat(H, cfg.num_heads * hd)
        w[p + "mlp.gate_proj.weight"] = mat(cfg.intermediate_size, H)
        w[p + "mlp.up_proj.weight"] = mat(cfg.intermediate_size, H)
        w[p + "mlp.down_proj.weight"] = mat(H, cfg.intermediate_size)
        w[p + "input_layernorm.weight"] = _bf16r(1.0 + 0.1 * torch.randn(H, generator=g))
        w[p + "post_attention_layernorm.weight"] = _bf16r(1.0 + 0.1 * torch.randn(H, generator=g))
    w["model.norm.weight"] = _bf16r(1.0 + 0.1 * torch.randn(H, generator=g))
    if not cfg.tie_word_embeddings:
        w["lm_head.weight"] = mat(cfg.vocab_size, H)
    return w


def weights_checksum(w: dict) -> float:
    return float(sum(t.double().abs().sum() for t in w.values()))


def make_prompts(vocab_size: int, seed: int = 0) -> list[list[int]]:
    """Four prompts shaped like the reference's tiny-test prompts (short, short, short,
    long: test/tiny/test_prefix_caching_inference.py:60-75); prompts 1 and 3 share a
    70-token prefix (2 full blocks of 32 + a partial one) to exercise prefix-cache hits."""
    g = torch.Generator().manual_seed(seed)

    def ids(n):
        return torch.randint(0, vocab_size, (n,), generator=g).tolist()

    shared = ids(70)
    return [ids(6), shared + ids(9), ids(7), shared + ids(70)]
