#!/usr/bin/env python3
"""Generate tests/golden/hf_decoder_golden.safetensors (TEST INFRASTRUCTURE ONLY).

Runs in the BUILD CONTAINER only (needs ``transformers``; the fixture, not this
script's dependency, travels to the GPU box).  For every model in
``oracle.synth.ZOO`` it builds the HF-transformers model from a locally constructed
config (no hub access), loads the seeded synthetic weights, and records for each
prompt the greedy continuation and the fp32 next-token logits of every step,
computed WITHOUT any KV cache (a full forward over the whole sequence per step) so
the fixture is independent of every caching scheme.

HF-transformers greedy is the reference's own oracle of record
(/root/reference/test/e2e/online/online_server_runner.py:95-146).

    python -m oracle.gen_golden
"""

from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.synth import ZOO, make_prompts, make_weights, weights_checksum, zoo_config  # noqa: E402

NEW_TOKENS = 12


def build_hf(name: str):
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    z = ZOO[name]
    common = dict(vocab_size=z["vocab_size"], hidden_size=z["hidden_size"],
                  intermediate_size=z["intermediate_size"], num_hidden_layers=z["num_layers"],
                  num_attention_heads=z["num_heads"], num_key_value_heads=z["num_kv_heads"],
                  max_position_embeddings=4096, rms_norm_eps=z["rms_norm_eps"],
                  rope_theta=z["rope_theta"], tie_word_embeddings=False,
                  attn_implementation="eager")
    if z["model_type"] == "llama":
        hf_cfg = LlamaConfig(head_dim=z["head_dim"], rope_scaling=z.get("rope_scaling"), **common)
        model = LlamaForCausalLM(hf_cfg)
    else:
        hf_cfg = Qwen2Config(use_sliding_window=False, **common)
        model = Qwen2ForCausalLM(hf_cfg)
        assert hf_cfg.hidden_size // hf_cfg.num_attention_heads == z["head_dim"]
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing
    return model.eval().to(torch.float32), cfg, w


@torch.no_grad()
def main(out_path: str) -> None:
    tensors, meta = {}, {}
    for name in ZOO:
        model, cfg, w = build_hf(name)
        meta[f"{name}.weights_checksum"] = repr(weights_checksum(w))
        for i, prompt in enumerate(make_prompts(cfg.vocab_size, seed=0)):
            seq = list(prompt)
            step_logits = []
            for _ in range(NEW_TOKENS):
                lg = model(torch.tensor([seq])).logits[0, -1].to(torch.float32)
                step_logits.append(lg)
                seq.append(int(lg.argmax()))
            tensors[f"{name}.prompt.{i}"] = torch.tensor(prompt, dtype=torch.int64)
            tensors[f"{name}.generated.{i}"] = torch.tensor(seq[len(prompt):], dtype=torch.int64)
            tensors[f"{name}.logits.{i}"] = torch.stack(step_logits)
            top2 = torch.stack(step_logits).topk(2, dim=-1).values
            print(f"{name} prompt {i} (len {len(prompt)}): gen {seq[len(prompt):]}  "
                  f"min top1-top2 gap {float((top2[:, 0] - top2[:, 1]).min()):.4f}")
    import transformers
    meta["transformers_version"] = transformers.__version__
    meta["torch_version"] = torch.__version__
    save_file(tensors, out_path, metadata=meta)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(os.path.join(os.path.dirname(here), "tests", "golden", "hf_decoder_golden.safetensors"))
