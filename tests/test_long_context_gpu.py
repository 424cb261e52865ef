"""Long contexts through the model call: an 8192-token window (256 blocks of 32) on a small model,
prompt of ~5000 tokens encoded in one call, a prefix-cache hit that leaves 900 tokens to encode
against 4096 cached ones, then token generation at context > 4096 (the decode attention's deepest
context split).  Checked against the CPU oracle."""
import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.synth import make_weights, zoo_config
from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,wd", [("tinyllama_like", "bf16"), ("llama31_like", "f8e4m3")])
def test_long_context_matches_oracle(name, wd):
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    bs, maxlen, nseq = 32, 8192, 2
    mb = maxlen // bs
    nb = 1 + 2 * mb
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    rs = cfg.rope_scaling or {}
    m = NativeModel(
        num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads,
        num_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
        vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        rope_type=1 if rs else 0, rope_factor=rs.get("factor", 1.0),
        rope_low_freq_factor=rs.get("low_freq_factor", 1.0), rope_high_freq_factor=rs.get("high_freq_factor", 4.0),
        rope_original_max_position=rs.get("original_max_position_embeddings", 0),
        qkv_bias=int(cfg.qkv_bias), tie_word_embeddings=int(cfg.tie_word_embeddings),
        num_blocks=nb, block_size=bs, max_num_seqs=nseq, max_model_len=maxlen,
        weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
        tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[1024, 8192])
    m.load_state_dict(w)
    m.finalize()
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type="per_channel_symmetric")
    oracle = PagedDecoderOracle(cfg, w, nb, bs, compute="bf16", quant=quant)
    g = torch.Generator().manual_seed(5)
    perm = (torch.randperm(nb - 1, generator=g) + 1).tolist()
    blocks = [perm[:mb], perm[mb:2 * mb]]
    prompt = torch.randint(0, cfg.vocab_size, (4996,), generator=g).tolist()
    worst = 0.0

    def check(inp):
        nonlocal worst
        got, ref = m.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        return ref

    ref0 = check(prefill_inputs(prompt, blocks[0], bs, maxlen, 0))
    # second request: shares the first 4096 tokens (128 full blocks) of the first one
    p2 = prompt[:4096] + torch.randint(0, cfg.vocab_size, (900,), generator=g).tolist()
    blocks[1][:128] = blocks[0][:128]
    ref1 = check(prefill_inputs(p2, blocks[1], bs, maxlen, 4096))
    toks = [prompt + [int(ref0.argmax())], p2 + [int(ref1.argmax())]]
    for _ in range(3):
        ref = check(decode_inputs([t[-1] for t in toks], [len(t) - 1 for t in toks], blocks, bs, maxlen))
        for t, row in zip(toks, ref):
            t.append(int(row.argmax()))
    assert worst < 0.06, worst
    m.close()
