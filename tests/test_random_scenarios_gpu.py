"""Randomised call sequences through the C ABI against the CPU oracle: random model family /
weight dtype / block size, ragged prompt lengths (1 token .. several blocks), prefix-cache hits of
whole blocks, scattered and re-used physical blocks, batches that shrink and reorder between
token-generation steps, pads of 0 and -1.  Same tolerance as tests/test_model_gpu.py."""
import random

import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.synth import make_weights, zoo_config
from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu


def _native(cfg, weights, wd, qt, bs, maxlen, nseq, nb):
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
        num_blocks=nb, block_size=bs, max_num_seqs=nseq, max_model_len=maxlen,
        weight_dtype=MI_W[wd], quant_type=MI_Q[qt], quantize_lm_head=1,
        tp_degree=1, tp_rank=0, device_id=0, use_graphs=1)
    m.load_state_dict(weights)
    m.finalize()
    return m


@pytest.mark.parametrize("seed", range(16))
def test_random_call_sequences_match_oracle(seed):
    rng = random.Random(seed)
    name = rng.choice(["tinyllama_like", "llama31_like", "qwen25_like"])
    wd, qt = rng.choice([("bf16", "per_tensor_symmetric"), ("f8e4m3", "per_channel_symmetric"),
                         ("int8", "per_channel_symmetric"), ("f8e4m3", "per_tensor_symmetric")])
    bs = rng.choice([32, 64])
    maxlen, nseq = 384, 4
    mb = maxlen // bs
    nb = 1 + 3 * nseq * mb
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1 + seed % 3)
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type=qt)
    oracle = PagedDecoderOracle(cfg, w, nb, bs, compute="bf16", quant=quant)
    model = _native(cfg, w, wd, qt, bs, maxlen, nseq, nb)
    free = list(range(1, nb))
    rng.shuffle(free)
    g = torch.Generator().manual_seed(seed)
    worst = 0.0

    def check(inp):
        nonlocal worst
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        return ref

    seqs = []       # dicts: tokens (all so far), blocks
    shared = None   # a finished prompt whose full blocks later requests may hit
    for r in range(nseq):
        plen = rng.choice([1, 2, bs - 1, bs, bs + 1, rng.randint(3, 200), rng.randint(100, 250)])
        toks = torch.randint(0, cfg.vocab_size, (plen,), generator=g).tolist()
        blocks = [free.pop() for _ in range(mb)]
        comp = 0
        if shared is not None and rng.random() < 0.6:
            nhit = rng.randint(1, max(1, (min(len(shared["tokens"]), plen) - 1) // bs)) if min(len(shared["tokens"]), plen) > bs else 0
            if nhit:
                toks[:nhit * bs] = shared["tokens"][:nhit * bs]
                blocks[:nhit] = shared["blocks"][:nhit]
                comp = nhit * bs
        ref = check(prefill_inputs(toks, blocks, bs, maxlen, comp))
        seqs.append(dict(tokens=toks + [int(ref.argmax())], blocks=blocks))
        if shared is None and plen > bs:
            shared = dict(tokens=list(toks), blocks=list(blocks))
    live = list(range(nseq))
    for step in range(7):
        if len(live) > 1 and rng.random() < 0.3:
            live.remove(rng.choice(live))           # a request finishes
        rng.shuffle(live)                            # rows arrive in any order
        rows = [seqs[i] for i in live if len(seqs[i]["tokens"]) < maxlen]
        if not rows:
            break
        last = [s["tokens"][-1] for s in rows]
        pos = [len(s["tokens"]) - 1 for s in rows]
        ref = check(decode_inputs(last, pos, [s["blocks"] for s in rows], bs, maxlen,
                                  pad_block=rng.choice([0, -1])))
        for s, row in zip(rows, ref):
            s["tokens"].append(int(row.argmax()))
    assert worst < 0.06, (name, wd, qt, bs, worst)
    model.close()


@pytest.mark.parametrize("name,wd", [("llama31_like", "f8e4m3"), ("qwen25_like", "int8"), ("tinyllama_like", "bf16")])
def test_wide_decode_batches_match_oracle(name, wd):
    """Token-generation batches of 5..16 rows (the GEMV's general staging path: more than four
    rows per norm prologue) and of 17..32 rows (the platform's default max_num_seqs is 32: two
    16-column MFMA groups per wave over one weight stream -- gemv_bigk_kernel<MG = 2> for the norm
    prologues, gemv_kstream_kernel<MG = 2> for the bf16 activations) against the oracle."""
    bs, maxlen, nseq = 32, 256, 32
    mb = maxlen // bs
    nb = 1 + nseq * mb
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=2)
    qt = "per_channel_symmetric"
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type=qt)
    oracle = PagedDecoderOracle(cfg, w, nb, bs, compute="bf16", quant=quant)
    model = _native(cfg, w, wd, qt, bs, maxlen, nseq, nb)
    g = torch.Generator().manual_seed(3)
    blocks = [[1 + i * mb + j for j in range(mb)] for i in range(nseq)]
    seqs, worst = [], 0.0
    for i in range(nseq):
        p = torch.randint(0, cfg.vocab_size, (3 + (11 * i) % 180,), generator=g).tolist()
        inp = prefill_inputs(p, blocks[i], bs, maxlen, 0)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        seqs.append(p + [int(ref.argmax())])
    for B in (32, 16, 9, 24, 5, 17, 13):
        rows = list(range(B))
        inp = decode_inputs([seqs[i][-1] for i in rows], [len(seqs[i]) - 1 for i in rows], [blocks[i] for i in rows], bs, maxlen)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        for i, row in zip(rows, ref):
            seqs[i].append(int(row.argmax()))
    assert worst < 0.06, worst
    model.close()


@pytest.mark.parametrize("nseq", [4, 12])
def test_wide_hidden_size_matches_oracle(nseq):
    """hidden_size 8192 (the Llama-3.3-70B width of BASELINE config 5): more 16-byte chunks per
    activation row than threads in a GEMV work-group, a 64 KiB LDS image at B = 4, K = 8192 GEMMs.
    At 12 rows the image (192 KiB) exceeds the LDS: the norm-prologue GEMVs stage it in K-chunks
    (gemv_bigk_kernel), sums of squares accumulated across the chunks."""
    from oracle.paged_decoder import DecoderConfig
    cfg = DecoderConfig(num_layers=1, hidden_size=8192, num_heads=8, num_kv_heads=1, head_dim=128,
                        intermediate_size=512, vocab_size=512, rms_norm_eps=1e-5, rope_theta=500000.0)
    bs, maxlen = 32, 128
    mb = maxlen // bs
    nb = 1 + nseq * mb
    w = make_weights(cfg, seed=4)
    wd, qt = "f8e4m3", "per_channel_symmetric"
    oracle = PagedDecoderOracle(cfg, w, nb, bs, compute="bf16",
                                quant=dict(quantized=True, quantization_dtype=wd, quantization_type=qt))
    model = _native(cfg, w, wd, qt, bs, maxlen, nseq, nb)
    g = torch.Generator().manual_seed(6)
    blocks = [[1 + i * mb + j for j in range(mb)] for i in range(nseq)]
    seqs, worst, scale = [], 0.0, 0.0
    for i in range(nseq):
        p = torch.randint(0, cfg.vocab_size, (5 + (20 * i) % 100,), generator=g).tolist()
        inp = prefill_inputs(p, blocks[i], bs, maxlen, 0)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst, scale = max(worst, (got - ref).abs().max().item()), max(scale, ref.abs().max().item())
        seqs.append(p + [int(ref.argmax())])
    for _ in range(3):
        inp = decode_inputs([s[-1] for s in seqs], [len(s) - 1 for s in seqs], blocks, bs, maxlen)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        for s, row in zip(seqs, ref):
            s.append(int(row.argmax()))
    assert worst < 0.015 * max(scale, 4.0), (worst, scale)      # the 0.06-on-O(4) bar, scaled to these logits
    model.close()
