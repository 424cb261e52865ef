"""End-to-end parity of the HIP model call (mi_forward through the C ABI) on the MI355X:

  * against the committed HF-transformers goldens (tests/golden, bf16 weights), and
  * against the CPU oracle in "bf16" mode on the same call sequence, for bf16 / fp8 / int8
    weights (quantized outputs: parity unpinned upstream, so the oracle IS the definition).

The call sequence follows the reference contract: 4 prefills of B=1 (one of them a
prefix-cache hit with computed_context_lens = 64), then token generation at B=4.
Tolerances: logits are O(4); the device path rounds activations to bf16 at the same points
as the oracle, so what remains is accumulation order and bf16 tie flips:
  |logits - oracle_bf16| <= 0.06,  |logits - HF fp32| <= 0.12,
greedy ids must match wherever the golden top-1/top-2 gap exceeds 0.15.
"""

import os

import pytest
import torch
from safetensors import safe_open

from oracle import PagedDecoderOracle
from oracle.synth import ZOO, make_prompts, make_weights, weights_checksum, zoo_config
from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_decoder_golden.safetensors")
BS, MAXLEN, NSEQ = 32, 256, 4
MB = MAXLEN // BS
NB = 1 + NSEQ * MB


def native_model(cfg, weights, weight_dtype="bf16", quant_type="per_tensor_symmetric", use_graphs=1, a8=0,
                 artifacts=None):
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
        num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
        weight_dtype=MI_W[weight_dtype], quant_type=MI_Q[quant_type], quantize_lm_head=1,
        tp_degree=1, tp_rank=0, device_id=0, use_graphs=use_graphs, prefill_fp8_activations=a8)
    if artifacts is not None:
        try:
            m.load_artifacts(artifacts)           # weight images saved by NativeModel.save_artifacts
        except Exception:
            m.close()
            raise
    else:
        m.load_state_dict(weights)
    m.finalize()
    return m


def scenario(prompts, golden_gen):
    """Yields (kind, inputs, [(req, step)]) for the reference-contract call sequence."""
    blocks = [[1 + i * MB + j for j in range(MB)] for i in range(len(prompts))]
    blocks[3][0:2] = blocks[1][0:2]          # prompt 3 hits prompt 1's first two full blocks
    for i, p in enumerate(prompts):
        comp = 64 if i == 3 else 0
        yield "prefill", prefill_inputs(p, blocks[i], BS, MAXLEN, comp), [(i, 0)]
    n_new = len(golden_gen[0])
    for s in range(1, n_new):
        last = [golden_gen[i][s - 1] for i in range(len(prompts))]       # teacher forcing
        pos = [len(prompts[i]) + s - 1 for i in range(len(prompts))]
        yield "decode", decode_inputs(last, pos, blocks, BS, MAXLEN, pad_block=(-1 if s % 2 else 0)), \
            [(i, s) for i in range(len(prompts))]


def load_golden(name):
    f = safe_open(GOLD, "pt")
    gen = [f.get_tensor(f"{name}.generated.{i}").tolist() for i in range(4)]
    logits = [f.get_tensor(f"{name}.logits.{i}") for i in range(4)]
    return gen, logits, float(f.metadata()[f"{name}.weights_checksum"])


@pytest.mark.parametrize("name", list(ZOO))
def test_bf16_matches_hf_golden_and_oracle(name):
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, gold, csum = load_golden(name)
    assert abs(weights_checksum(w) - csum) < 1e-6 * csum, "synthetic weights drifted from the golden's"
    prompts = make_prompts(cfg.vocab_size, 0)
    model = native_model(cfg, w, "bf16")
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16")
    worst_o = worst_g = 0.0
    for kind, inp, rows in scenario(prompts, gen):
        got, ref = model.forward(**inp), oracle.forward(**inp)
        for r, (req, step) in enumerate(rows):
            g = gold[req][step]
            worst_o = max(worst_o, (got[r] - ref[r]).abs().max().item())
            worst_g = max(worst_g, (got[r] - g).abs().max().item())
            top2 = g.topk(2).values
            if top2[0] - top2[1] > 0.15:
                assert int(got[r].argmax()) == gen[req][step], (kind, req, step)
    assert worst_o < 0.06, worst_o
    assert worst_g < 0.12, worst_g
    model.close()


@pytest.mark.parametrize("name", ["llama31_like", "qwen25_like"])
@pytest.mark.parametrize("wdtype", ["f8e4m3", "int8"])
@pytest.mark.parametrize("qtype", ["per_tensor_symmetric", "per_channel_symmetric"])
def test_quantized_matches_oracle(name, wdtype, qtype):
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    model = native_model(cfg, w, wdtype, qtype)
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16",
                                quant=dict(quantized=True, quantization_dtype=wdtype, quantization_type=qtype))
    worst = 0.0
    for kind, inp, rows in scenario(prompts, gen):
        got, ref = model.forward(**inp), oracle.forward(**inp)
        for r, (req, step) in enumerate(rows):
            worst = max(worst, (got[r] - ref[r]).abs().max().item())
            top2 = ref[r].topk(2).values
            if top2[0] - top2[1] > 0.15:
                assert int(got[r].argmax()) == int(ref[r].argmax()), (kind, req, step)
    assert worst < 0.06, worst
    model.close()


@pytest.mark.parametrize("name", ["llama31_like", "tinyllama_like"])
def test_prefill_fp8_activations_matches_oracle(name):
    """Context-encoding GEMMs with per-token FP8 activations on the MX-scaled MFMA vs the oracle's
    statement of the same rule (rows > 16, K % 128 == 0, fp8 weights).  fp8 activations carry a
    3-bit mantissa, so a borderline rounding of one activation moves a logit more than in the
    bf16 path: tolerance 0.15 on O(4) logits, ids compared where the oracle's gap > 0.3.  (0.12 until
    round 3: the second-generation context-encoding attention sums in another order and defers the
    softmax rescale -- its own op test holds the same 0.03 against an fp64 reference as the first
    kernel -- and on the head_dim-64 model one activation now rounds to the neighbouring fp8 code:
    0.1215 against 0.11 before.  The weight-only parity tests keep their 0.06.)"""
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    model = native_model(cfg, w, "f8e4m3", "per_channel_symmetric", a8=1)
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16", prefill_fp8_activations=True,
                                quant=dict(quantized=True, quantization_dtype="f8e4m3",
                                           quantization_type="per_channel_symmetric"))
    plain = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16",
                               quant=dict(quantized=True, quantization_dtype="f8e4m3",
                                          quantization_type="per_channel_symmetric"))
    worst = moved = 0.0
    for kind, inp, rows in scenario(prompts, gen):
        got, ref, base = model.forward(**inp), oracle.forward(**inp), plain.forward(**inp)
        for r, (req, step) in enumerate(rows):
            worst = max(worst, (got[r] - ref[r]).abs().max().item())
            moved = max(moved, (ref[r] - base[r]).abs().max().item())
            top2 = ref[r].topk(2).values
            if top2[0] - top2[1] > 0.3:
                assert int(got[r].argmax()) == int(ref[r].argmax()), (kind, req, step)
    assert worst < 0.15, worst
    assert moved > 0.01, "the FP8-activation rule never fired: the test is not testing it"
    model.close()


def test_graph_replay_equals_eager():
    """hipGraph replay of the token-generation step must be bit-identical to eager launches."""
    name = "llama31_like"
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    outs = []
    for graphs in (0, 1):
        model = native_model(cfg, w, "f8e4m3", "per_channel_symmetric", use_graphs=graphs)
        outs.append([model.forward(**inp) for _, inp, _ in scenario(prompts, gen)])
        model.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_forward_rejects_bad_inputs():
    cfg = zoo_config("tinyllama_like")
    model = native_model(cfg, make_weights(cfg, 1), "bf16")
    inp = prefill_inputs([1, 2, 3], [1], BS, MAXLEN, 0)
    bad = dict(inp)
    bad["block_table"] = inp["block_table"].clone()
    bad["block_table"][0, 0] = NB + 5
    with pytest.raises(ValueError, match="block_table entry out of range"):
        model.forward(**bad)
    bad = dict(inp)
    bad["input_ids"] = torch.tensor([[1, 2, cfg.vocab_size]])
    with pytest.raises(ValueError, match="token id out of range"):
        model.forward(**bad)
    bad = dict(inp)
    bad["computed_context_lens"] = torch.tensor([[3]])
    with pytest.raises(ValueError, match="nothing to compute"):
        model.forward(**bad)
    model.close()


def test_failed_token_generation_call_does_not_poison_the_resident_block_tables():
    """The block tables of token generation live on the device; the host keeps a shadow of what it sent.  A call
    whose SECOND row is bad has already compared (and recorded) the first row when it fails, without sending it:
    the next good call must send that row again.  Two identical models, one of them fed the bad call first."""
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, 1)
    prompts = make_prompts(cfg.vocab_size, 0)[:2]
    blocks = [[1 + i * MB + j for j in range(MB)] for i in range(2)]
    outs = []
    for poison in (True, False):
        model = native_model(cfg, w, "bf16")
        last = []
        for i, p in enumerate(prompts):
            last.append(int(model.forward(**prefill_inputs(p, blocks[i], BS, MAXLEN, 0)).argmax(dim=1)[0]))
        good = decode_inputs(last, [len(p) for p in prompts], blocks, BS, MAXLEN)
        if poison:
            bad = dict(good)
            bad["block_table"] = good["block_table"].clone()
            bad["block_table"][1, 0] = NB + 7                     # row 0 is fine and is recorded before row 1 fails
            with pytest.raises(ValueError, match="block_table entry out of range"):
                model.forward(**bad)
        outs.append(model.forward(**good))
        model.close()
    assert torch.equal(outs[0], outs[1])


def test_collective_path_single_rank_equals_fused_path():
    """The tensor-parallel code path (fp32 partial -> RCCL all-reduce -> partial folded in by the
    next norm prologue, vocab all-gather) with a ONE-rank communicator must give bit-identical
    logits to the single-GPU path (projection written straight into the residual stream):
    same fp32 additions in the same order.  Exercises the RCCL linkage on a 1-GPU box."""
    name = "llama31_like"
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    outs = []
    for with_comm in (False, True):
        from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
        rs = cfg.rope_scaling or {}
        m = NativeModel(
            num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads,
            num_kv_heads=cfg.num_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
            vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
            rope_type=1, rope_factor=rs["factor"], rope_low_freq_factor=rs["low_freq_factor"],
            rope_high_freq_factor=rs["high_freq_factor"],
            rope_original_max_position=rs["original_max_position_embeddings"], qkv_bias=0, tie_word_embeddings=0,
            num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN, weight_dtype=MI_W["f8e4m3"],
            quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1, tp_degree=1, tp_rank=0, device_id=0,
            use_graphs=0)
        if with_comm:
            m.tp_init(m.tp_unique_id())
        m.load_state_dict(w)
        m.finalize()
        outs.append([m.forward(**inp) for _, inp, _ in scenario(prompts, gen)])
        if with_comm:
            m.profile_enable(True)
            m.forward(**list(scenario(prompts, gen))[-1][1])
            assert m.profile_read()["launches"]["comm"] == 2 * cfg.num_layers + 1
        m.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_in_launch_attention_merge_matches_oracle():
    """MI355X_ATTN_MERGE=1: the context splits of decode attention are merged by the last
    work-group to finish instead of a combine launch (off by default: measured slower).  The
    switch is read once per process, so the oracle comparisons above are re-run in a child."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MI355X_ATTN_MERGE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__, "-k",
                        "quantized_matches_oracle or bf16_matches_hf_golden"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name,wd", [("llama31_like", "f8e4m3"), ("tinyllama_like", "bf16")])
def test_block_size_16_matches_oracle(name, wd):
    """vLLM's default block size: a 32-token attention tile then spans two blocks (each 16-token half
    looks its own block up).  Scattered blocks, a prefix hit on a 16-token boundary, batched decode
    across block boundaries."""
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    from tests.helpers import decode_inputs, prefill_inputs
    bs, maxlen, nseq = 16, 256, 4
    mb = maxlen // bs
    nb = 1 + nseq * mb
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    qt = "per_channel_symmetric"
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type=qt)
    oracle = PagedDecoderOracle(cfg, w, nb, bs, compute="bf16", quant=quant)
    rs = cfg.rope_scaling or {}
    model = NativeModel(
        num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads,
        head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, rope_type=1 if rs else 0,
        rope_factor=rs.get("factor", 1.0), rope_low_freq_factor=rs.get("low_freq_factor", 1.0),
        rope_high_freq_factor=rs.get("high_freq_factor", 4.0),
        rope_original_max_position=rs.get("original_max_position_embeddings", 0), qkv_bias=int(cfg.qkv_bias),
        tie_word_embeddings=int(cfg.tie_word_embeddings), num_blocks=nb, block_size=bs, max_num_seqs=nseq,
        max_model_len=maxlen, weight_dtype=MI_W[wd], quant_type=MI_Q[qt], quantize_lm_head=1, tp_degree=1, tp_rank=0,
        device_id=0, use_graphs=1)
    model.load_state_dict(w)
    model.finalize()
    g = torch.Generator().manual_seed(5)
    perm = (torch.randperm(nb - 1, generator=g) + 1).tolist()
    blocks = [perm[i * mb:(i + 1) * mb] for i in range(nseq)]
    lens = (15, 16, 47, 130)
    seqs = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
    seqs[2][:32] = seqs[3][:32]                         # two shared 16-token blocks
    blocks[2][:2] = blocks[3][:2]
    worst = 0.0
    for i in (3, 0, 1, 2):
        inp = prefill_inputs(seqs[i], blocks[i], bs, maxlen, 32 if i == 2 else 0)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        seqs[i].append(int(ref.argmax()))
    for _ in range(20):                                 # crosses 16- and 32-token boundaries for every row
        inp = decode_inputs([s[-1] for s in seqs], [len(s) - 1 for s in seqs], blocks, bs, maxlen)
        got, ref = model.forward(**inp), oracle.forward(**inp)
        worst = max(worst, (got - ref).abs().max().item())
        for s, row in zip(seqs, ref):
            s.append(int(row.argmax()))
    assert worst < 0.06, worst
    model.close()
