"""Oracle: paged-KV Llama/Qwen2 decoder step on the CPU (TEST INFRASTRUCTURE ONLY).

One call of :meth:`PagedDecoderOracle.forward` is one call of the reference's
model adapter, ``NeuronCausalLM.forward``
(/root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:336-365),
with the argument record the runner builds
(/root/reference/vllm_neuron/worker/neuronx_distributed_model_runner.py):

  context encoding (prefill, runner.py:704-763, 853-885)
      input_ids/position_ids [B, L]   the FULL prompt, pad 0
      slot_mapping [B, max_model_len] slots of tokens computed..L-1, then -1
                                      (already sliced at num_computed_tokens, :762-763)
      block_table [B, MB]             pad 0 (= the null block, a VALID index)
      full_context_lens [B,1] = L     computed_context_lens [B,1] = cached prefix
  token generation (decode, runner.py:765-832, 887-917)
      input_ids/position_ids [B, 1]   last sampled token, pos = len(prompt)+len(out)-1
      slot_mapping [B, 1]             bt[pos // bs] * bs + pos % bs
      block_table [B, MB]             pad 0 or -1 (:805-817)
      full = pos + 1, computed = pos

and returns what the CPU-sampling branch returns: the last-token logits
``output.logits[:, -1, :]`` (loader.py:363), fp32 ``[B, V]``.

Rows beyond ``full_context_lens`` are never read: masking is by length, never by
the block-table pad value.  Slot -1 means "no write".

The decoder equations are those of HF ``modeling_llama.py`` /
``modeling_qwen2.py`` (RMSNorm, rotate-half RoPE incl. the llama3 frequency
rescale, GQA causal attention, SwiGLU), which the reference's e2e test uses as
its oracle of record (test/e2e/online/online_server_runner.py:95-146).

``compute`` selects where values are rounded:
  "fp32"  no rounding anywhere: must equal HF fp32 (this is how the oracle is pinned)
  "bf16"  the rounding points of the HIP path: fp32 residual
          stream; bf16 at every GEMM input (post-norm activations, attention
          output, SwiGLU output), bf16 q and bf16 K/V cache; fp32 accumulation,
          fp32 softmax, fp32 logits.

``prefill_fp8_activations`` (with fp8-quantized weights): a linear whose input has MORE THAN
16 rows and whose K is a multiple of 128 — i.e. the context-encoding GEMMs, never the
token-generation GEMVs nor the 1-row lm_head — quantizes its bf16 input per token to OCP
e4m3 (scale = amax/448, RNE) before the contraction, like the device's MX-scaled MFMA path.
PARITY UNPINNED upstream (no reference test covers quantized outputs); this is the definition.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

from .quant import dequantize_weight, quantize_weight

_LINEARS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


@dataclass
class DecoderConfig:
    num_layers: int
    hidden_size: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    # None | {"rope_type": "llama3", "factor", "low_freq_factor", "high_freq_factor",
    #         "original_max_position_embeddings"}
    rope_scaling: dict | None = None
    qkv_bias: bool = False
    tie_word_embeddings: bool = False
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_hf(cls, hf) -> "DecoderConfig":
        """Same field derivation as the reference's ``_get_model_configs``
        (loader.py:612-631): head_dim falls back to hidden_size // num_heads."""
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        rp = getattr(hf, "rope_parameters", None) or {}
        rs = getattr(hf, "rope_scaling", None) or rp
        scaling = None
        if rs and rs.get("rope_type", rs.get("type", "default")) == "llama3":
            scaling = {k: rs[k] for k in ("factor", "low_freq_factor", "high_freq_factor",
                                          "original_max_position_embeddings")}
            scaling["rope_type"] = "llama3"
        theta = rp.get("rope_theta", getattr(hf, "rope_theta", 10000.0))
        return cls(
            num_layers=hf.num_hidden_layers, hidden_size=hf.hidden_size,
            num_heads=hf.num_attention_heads, num_kv_heads=hf.num_key_value_heads,
            head_dim=int(head_dim), intermediate_size=hf.intermediate_size,
            vocab_size=hf.vocab_size, rms_norm_eps=hf.rms_norm_eps, rope_theta=float(theta),
            rope_scaling=scaling,
            qkv_bias=(getattr(hf, "model_type", "") == "qwen2") or bool(getattr(hf, "attention_bias", False)),
            tie_word_embeddings=bool(getattr(hf, "tie_word_embeddings", False)),
        )


def rope_inv_freq(cfg: DecoderConfig) -> torch.Tensor:
    """inv_freq[d/2] in fp32; llama3 rescale per HF ``_compute_llama3_parameters``."""
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim))
    rs = cfg.rope_scaling
    if not rs:
        return inv
    factor, lo, hi = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
    old = rs["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    mid = (1 - smooth) * scaled / factor + smooth * scaled
    is_mid = ~(wavelen < old / hi) & ~(wavelen > old / lo)
    return torch.where(is_mid, mid, scaled)


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class PagedDecoderOracle:
    """Holds weights + the paged KV pool ``[L][2][NB][bs][nkv][hd]`` (NB counts the null
    block 0, i.e. the reference's ``pa_num_blocks`` after the +1 of
    platform.py:150-159 / loader.py:806-815)."""

    def __init__(self, cfg: DecoderConfig, weights: dict, num_blocks: int, block_size: int,
                 compute: str = "fp32", quant: dict | None = None, prefill_fp8_activations: bool = False):
        assert compute in ("fp32", "bf16")
        self.cfg, self.compute = cfg, compute
        self.a8 = bool(prefill_fp8_activations)
        self.a8_weights: set = set()
        self.block_size, self.num_blocks = block_size, num_blocks
        self.r = _bf16 if compute == "bf16" else (lambda t: t)
        self.quant = quant
        self.w = self._prepare_weights(weights)
        self.inv_freq = rope_inv_freq(cfg)
        self.kv = torch.zeros(cfg.num_layers, 2, num_blocks, block_size, cfg.num_kv_heads,
                              cfg.head_dim, dtype=torch.float32)

    # ---- weights -------------------------------------------------------------------
    def _prepare_weights(self, weights: dict) -> dict:
        q = self.quant
        skip = tuple(q.get("modules_to_not_convert") or ()) if q else ()
        out = {}
        for name, t in weights.items():
            t = t.to(torch.float32)
            is_linear = name.endswith(".weight") and (
                any(f".{p}." in name for p in _LINEARS) or name.startswith("lm_head"))
            if q and q.get("quantized") and is_linear and not any(s in name for s in skip):
                # quantize-at-load from the tensor as handed over (loader.py:238-239)
                qt, sc = quantize_weight(t, q.get("quantization_dtype", "int8"),
                                         q.get("quantization_type", "per_tensor_symmetric"))
                if q.get("quantization_dtype") == "f8e4m3" and t.shape[1] % 128 == 0:
                    self.a8_weights.add(name)
                t = dequantize_weight(qt, sc)
            elif self.compute == "bf16" and (is_linear or name == "model.embed_tokens.weight"):
                t = _bf16(t)            # device copies of unquantized matrices are bf16;
                                        # norm gains and biases stay fp32
            out[name] = t
        if self.cfg.tie_word_embeddings and "lm_head.weight" not in out:
            out["lm_head.weight"] = out["model.embed_tokens.weight"]
        return out

    # ---- pieces ---------------------------------------------------------------------
    def _rmsnorm(self, x, g):
        var = x.pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(var + self.cfg.rms_norm_eps) * g

    def _rope(self, x, pos):
        """x [T, nheads, hd] fp32, pos [T]; HF rotate-half convention."""
        ang = pos.to(torch.float32)[:, None] * self.inv_freq[None, :]      # [T, hd/2]
        cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
        h = x.shape[-1] // 2
        x1, x2 = x[..., :h], x[..., h:]
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)

    def _lin(self, x, name, layer=None):
        p = f"model.layers.{layer}." if layer is not None else ""
        if self.a8 and x.shape[0] > 16 and f"{p}{name}.weight" in self.a8_weights:
            amax = x.abs().amax(dim=1, keepdim=True)
            s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
            x = (x / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s
        y = x @ self.w[f"{p}{name}.weight"].t()
        b = self.w.get(f"{p}{name}.bias")
        return y if b is None else y + b

    # ---- the model call ----------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids, position_ids, seq_ids, block_table, slot_mapping,
                full_context_lens, computed_context_lens) -> torch.Tensor:
        cfg, bs, r = self.cfg, self.block_size, self.r
        B, S = input_ids.shape
        full = full_context_lens.reshape(-1).tolist()
        comp = computed_context_lens.reshape(-1).tolist()
        logits = torch.empty(B, cfg.vocab_size, dtype=torch.float32)
        scale = 1.0 / math.sqrt(cfg.head_dim)
        group = cfg.num_heads // cfg.num_kv_heads
        for b in range(B):
            if S == 1:                       # token generation: the one token sits at full-1
                ids, pos, slots = input_ids[b, :1], position_ids[b, :1], slot_mapping[b, :1]
            else:                            # context encoding: skip the cached prefix ourselves
                n_new = full[b] - comp[b]
                ids, pos = input_ids[b, comp[b]:full[b]], position_ids[b, comp[b]:full[b]]
                slots = slot_mapping[b, :n_new]
            T = ids.shape[0]
            assert T >= 1, "nothing to compute (computed_context_lens == full_context_lens)"
            kpos = torch.arange(full[b])
            kblk = block_table[b, kpos // bs].long()
            koff = kpos % bs
            mask = kpos[None, :] > pos[:, None]                      # causal, [T, full]
            h = self.w["model.embed_tokens.weight"][ids]             # fp32 residual stream
            for l in range(cfg.num_layers):
                xn = r(self._rmsnorm(h, self.w[f"model.layers.{l}.input_layernorm.weight"]))
                q = self._lin(xn, "self_attn.q_proj", l).view(T, cfg.num_heads, cfg.head_dim)
                k = self._lin(xn, "self_attn.k_proj", l).view(T, cfg.num_kv_heads, cfg.head_dim)
                v = self._lin(xn, "self_attn.v_proj", l).view(T, cfg.num_kv_heads, cfg.head_dim)
                q, k = r(self._rope(q, pos)), r(self._rope(k, pos))
                v = r(v)
                w = slots >= 0                                        # slot -1: no write
                sl = slots[w].long()
                self.kv[l, 0, sl // bs, sl % bs] = k[w]
                self.kv[l, 1, sl // bs, sl % bs] = v[w]
                K = self.kv[l, 0, kblk, koff]                         # [full, nkv, hd] via the block table
                V = self.kv[l, 1, kblk, koff]
                K = K.repeat_interleave(group, dim=1)
                V = V.repeat_interleave(group, dim=1)
                s = torch.einsum("thd,jhd->htj", q, K) * scale
                s = s.masked_fill(mask[None], float("-inf"))
                p = torch.softmax(s, dim=-1)
                o = r(torch.einsum("htj,jhd->thd", p, V).reshape(T, cfg.num_heads * cfg.head_dim))
                h = h + self._lin(o, "self_attn.o_proj", l)
                xn = r(self._rmsnorm(h, self.w[f"model.layers.{l}.post_attention_layernorm.weight"]))
                g, u = self._lin(xn, "mlp.gate_proj", l), self._lin(xn, "mlp.up_proj", l)
                a = r(torch.nn.functional.silu(g) * u)
                h = h + self._lin(a, "mlp.down_proj", l)
            xn = r(self._rmsnorm(h[-1:], self.w["model.norm.weight"]))   # logits[:, -1, :]
            logits[b] = (xn @ self.w["lm_head.weight"].t())[0]
        return logits


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    """CPU-sampling tail for greedy requests: vLLM's Sampler reduces to argmax
    (runner.py:1218; temperature 0 == top_k 1, test/tiny/test_dynamic_sampling.py:55,125)."""
    return logits.argmax(dim=-1)
