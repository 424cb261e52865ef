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
