"""Maximum sizes: contexts far beyond the BASELINE's max_model_len 2048 (the reference's models go to 128k;
/root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:725-793 sizes everything from
max_model_len).  Small decoder, max_model_len 16384 (512 blocks per sequence), parity against the oracle:

  * context encoding of a 9000-token prompt (16384 bucket), and of a second request that hits 8192 cached
    tokens of it (prefix caching, 808 new tokens at positions 8192..8999);
  * token generation at context 9001.. beside a short sequence (ragged batch: 9001 and 301 keys);
  * a context that ends at the last position of the model length (16383 cached keys + the token at 16383).

tests/test_long_context_gpu.py holds the 8192 case with scattered blocks.

Tolerance as tests/test_model_gpu.py (|logits - oracle_bf16| <= 0.06).
"""

import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.synth import make_weights, zoo_config
from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu

BS, MAXLEN, NSEQ = 32, 16384, 2
MB = MAXLEN // BS
NB = 1 + 3 * MB


def _model(cfg, w, wd="bf16"):
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    rs = cfg.rope_scaling or {}
    m = NativeModel(
        num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads,
        num_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
        vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        rope_type=1 if rs else 0, rope_factor=rs.get("factor", 1.0),
        rope_low_freq_factor=rs.get("low_freq_factor", 1.0), rope_high_freq_factor=rs.get("high_freq_factor", 4.0),
        rope_original_max_position=rs.get("original_max_position_embeddings", 0),
        qkv_bias=int(cfg.qkv_bias), tie_word_embeddings=int(cfg.tie_word_embeddings),
        num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN, ctx_buckets=[1024, 16384],
        weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
        tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, prefill_fp8_activations=0)
    m.load_state_dict(w)
    m.finalize()
    return m


@pytest.mark.parametrize("name,wd", [("llama31_like", "bf16"), ("qwen25_like", "f8e4m3")])
def test_long_prompt_prefix_hit_and_ragged_decode(name, wd):
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(11)
    long_prompt = torch.randint(1, cfg.vocab_size, (9000,), generator=g).tolist()
    short_prompt = torch.randint(1, cfg.vocab_size, (300,), generator=g).tolist()
    blocks = [[1 + i * MB + j for j in range(MB)] for i in range(3)]
    model = _model(cfg, w, wd)
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type="per_channel_symmetric")
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16", quant=quant)
    worst = 0.0

    def check(inp, what):
        nonlocal worst
        got, ref = model.forward(**inp), oracle.forward(**inp)
        d = (got - ref).abs().max().item()
        worst = max(worst, d)
        assert d < 0.06, (what, d)
        return got

    first = check(prefill_inputs(long_prompt, blocks[0], BS, MAXLEN), "9000-token prompt")
    check(prefill_inputs(short_prompt, blocks[1], BS, MAXLEN), "300-token prompt")
    # a third request with the same first 8192 tokens: its block table starts with request 0's 256 full blocks
    hit_blocks = blocks[0][:8192 // BS] + blocks[2][8192 // BS:]
    hit = check(prefill_inputs(long_prompt, hit_blocks, BS, MAXLEN, 8192), "prefix hit at 8192")
    assert (hit - first).abs().max().item() < 0.06
    toks = [int(first[0].argmax()), 7]
    pos = [9000, 300]
    for step in range(3):
        out = check(decode_inputs(toks, pos, blocks[:2], BS, MAXLEN), f"decode step {step}")
        toks = [int(out[0].argmax()), int(out[1].argmax())]
        pos = [p + 1 for p in pos]
    model.close()


def test_context_up_to_the_last_position():
    """The last block of the table and the last position of the model length: a 16383-token context (cached
    K/V written by a prefix-hit call that computes only the tail) and the token at position 16383."""
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(13)
    prompt = torch.randint(1, cfg.vocab_size, (MAXLEN - 1,), generator=g).tolist()
    blocks = [1 + j for j in range(MB)]
    model = _model(cfg, w)
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16")
    inp = prefill_inputs(prompt, blocks, BS, MAXLEN)
    got, ref = model.forward(**inp), oracle.forward(**inp)
    assert (got - ref).abs().max().item() < 0.06
    d = decode_inputs([int(ref[0].argmax())], [MAXLEN - 1], [blocks], BS, MAXLEN)
    got, ref = model.forward(**d), oracle.forward(**d)
    assert (got - ref).abs().max().item() < 0.06
    model.close()


def test_128k_window_prefix_equivalence_and_the_block_limit():
    """131072-token model length (4096 blocks of 32: the most the context-encoding attention stages per sequence).
    No oracle at this size (its score matrix is T x T); the size-independent property instead: a 70001-token
    prompt encoded in one call gives the logits of the same prompt encoded as 65536 cached + 4465 new tokens,
    and the two sequences then generate the same token-generation logits side by side."""
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, seed=7)
    maxlen = 131072
    mb = maxlen // BS
    nb = 1 + 2 * mb

    def make(max_model_len):
        return NativeModel(
            num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads,
            num_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
            vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, rope_type=0,
            rope_factor=1.0, rope_low_freq_factor=1.0, rope_high_freq_factor=4.0, rope_original_max_position=0,
            qkv_bias=0, tie_word_embeddings=0, num_blocks=nb, block_size=BS, max_num_seqs=2,
            max_model_len=max_model_len, ctx_buckets=[8192, max_model_len], weight_dtype=MI_W["bf16"],
            quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1, tp_degree=1, tp_rank=0, device_id=0,
            use_graphs=1, prefill_fp8_activations=0)

    with pytest.raises(Exception, match="4096 blocks"):
        make(maxlen + BS)
    m = make(maxlen)
    m.load_state_dict(w)
    m.finalize()
    g = torch.Generator().manual_seed(17)
    prompt = torch.randint(1, cfg.vocab_size, (70001,), generator=g).tolist()
    blocks = [[1 + i * mb + j for j in range(mb)] for i in range(2)]
    one = m.forward(**prefill_inputs(prompt, blocks[0], BS, maxlen)).clone()
    m.forward(**prefill_inputs(prompt[:65536], blocks[1], BS, maxlen))
    two = m.forward(**prefill_inputs(prompt, blocks[1], BS, maxlen, 65536)).clone()
    assert torch.isfinite(one).all()
    assert (one - two).abs().max().item() < 0.06
    tok = int(one[0].argmax())
    out = m.forward(**decode_inputs([tok, tok], [70001, 70001], blocks, BS, maxlen))
    assert torch.isfinite(out).all()
    assert (out[0] - out[1]).abs().max().item() < 0.06
    # the last position of the window
    out = m.forward(**decode_inputs([tok], [maxlen - 1], blocks[:1], BS, maxlen))
    assert torch.isfinite(out).all()
    m.close()
