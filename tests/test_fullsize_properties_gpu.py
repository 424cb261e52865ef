"""Parity at BASELINE.json's full sizes, where the CPU oracle would take hours: size-independent
properties of the call contract that hold for ANY weights, checked on seeded synthetic weights
generated on the device.  One model per BASELINE config that fits one GPU:

  config 3 (headline)  Llama-3.1-8B   32 layers, H 4096, 32/8 heads x 128, I 14336, V 128256, FP8 per-channel
  config 2             Llama-3.1-8B   the same shapes, bf16 weights
  config 4             Qwen2.5-7B     28 layers, H 3584, 28/4 heads x 128, I 18944, V 152064, INT8 per-channel,
                                      qkv bias, prefix caching with a 512-token shared prefix;
                                      down_proj (K = 18944) runs the K-streamed GEMV (gemv_kstream_kernel)
  (config 5, Llama-3.3-70B at TP 8, is in tests/test_tp_group_gpu.py)

all at block_size 32, max_num_seqs 4, max_model_len 2048, pa_num_blocks 4096 (+ the null block).

  exact (bit-for-bit)
    * a model call does not depend on WHICH physical blocks hold the context (block permutation);
    * rows of a token-generation batch are independent: the same sequence in four rows gives four
      identical logit rows (the B = 1 call agrees within tolerance: its context split differs);
    * a replayed hipGraph step equals the eagerly launched one; repeating a step is idempotent;
    * on-device greedy ids == argmax of the logits the CPU-sampling call returns.
  within the stated tolerance (different kernels compute the same quantity)
    * teacher forcing: the logits of position N-1 from context encoding of N tokens == those of
      a token-generation step after context encoding of N-1 tokens (GEMM + MFMA flash attention
      vs GEMV + split-context decode attention);
    * a prefix-cache hit (computed_context_lens = 256) == encoding the full prompt;
    * (characterisation) the MX FP8 x FP8 context-encoding mode computes the same function as
      weight-only FP8 up to its activation-quantization noise.
Tolerance: logits here have std 1.28, |max| 5.7.  Two kernel families that round the activations
to bf16 at the same points but sum in different orders drift apart over 32 layers by 0.024 rms /
0.11 max over the 128256 logits (measured); the bar is 0.04 rms and 0.16 max (~3 % of the logit
range), and the top-1 id must agree unless the top-2 gap is inside that band.
"""
import pytest
import torch

from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu

BS, MAXLEN, NSEQ, NB = 32, 2048, 4, 4096 + 1
MB = MAXLEN // BS
TOL_MAX, TOL_RMS = 0.16, 0.04


def _close(a, b):
    d = (a - b).float()
    return d.abs().max().item() <= TOL_MAX and d.pow(2).mean().sqrt().item() <= TOL_RMS


LLAMA31_8B = dict(num_layers=32, hidden_size=4096, num_heads=32, num_kv_heads=8, head_dim=128,
                  intermediate_size=14336, vocab_size=128256, rms_norm_eps=1e-5, rope_theta=500000.0,
                  rope_type=1, rope_factor=8.0, rope_low_freq_factor=1.0, rope_high_freq_factor=4.0,
                  rope_original_max_position=8192, qkv_bias=0, tie_word_embeddings=0)
QWEN25_7B = dict(num_layers=28, hidden_size=3584, num_heads=28, num_kv_heads=4, head_dim=128,
                 intermediate_size=18944, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1000000.0,
                 rope_type=0, rope_factor=1.0, rope_low_freq_factor=1.0, rope_high_freq_factor=4.0,
                 rope_original_max_position=0, qkv_bias=1, tie_word_embeddings=0)
CONFIGS = {"llama31_8b_fp8": (LLAMA31_8B, "f8e4m3"), "llama31_8b_bf16": (LLAMA31_8B, "bf16"),
           "qwen25_7b_int8": (QWEN25_7B, "int8")}


def _model(name="llama31_8b_fp8", a8=0, use_graphs=1):
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    geo, wd = CONFIGS[name]
    m = NativeModel(**geo, num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
                    weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                    tp_degree=1, tp_rank=0, device_id=0, use_graphs=use_graphs, ctx_buckets=[256, 512, 1024, 2048],
                    prefill_fp8_activations=a8)
    m.init_synthetic_weights(1, 0.02)
    m.finalize()
    return m


@pytest.fixture(scope="module", params=list(CONFIGS))
def model(request):
    m = _model(request.param)
    m.vocab = CONFIGS[request.param][0]["vocab_size"]
    yield m
    m.close()


def _blocks(seed, n=MB):
    return (torch.randperm(NB - 1, generator=torch.Generator().manual_seed(seed)) + 1)[:n].tolist()


def _prompt(n, seed=0, vocab=128256):
    return torch.randint(0, vocab, (n,), generator=torch.Generator().manual_seed(seed)).tolist()


def test_block_permutation_invariance_and_idempotence(model):
    p = _prompt(700, vocab=model.vocab)
    a = model.forward(**prefill_inputs(p, _blocks(1), BS, MAXLEN, 0))
    b = model.forward(**prefill_inputs(p, _blocks(2), BS, MAXLEN, 0))
    c = model.forward(**prefill_inputs(p, _blocks(2), BS, MAXLEN, 0))
    assert torch.isfinite(a).all() and a.std() > 0.1
    assert torch.equal(a, b) and torch.equal(b, c)


def test_teacher_forcing_encoding_equals_generation(model):
    for n in (300, 1025):
        p = _prompt(n, seed=n, vocab=model.vocab)
        blocks = _blocks(3)
        full = model.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))           # logits at position n-1
        model.forward(**prefill_inputs(p[:-1], blocks, BS, MAXLEN, 0))             # KV for 0..n-2
        step = model.forward(**decode_inputs([p[-1]], [n - 1], [blocks], BS, MAXLEN))
        assert _close(full, step), (n, (full - step).abs().max().item())
        assert int(full.argmax()) == int(step.argmax()) or full.topk(2).values.diff().abs().item() < TOL_MAX


def test_prefix_cache_hit_equals_full_encoding(model):
    p = _prompt(900, seed=5, vocab=model.vocab)
    blocks = _blocks(4)
    full = model.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))
    hit = model.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 256))              # blocks 0..7 already hold the prefix
    assert _close(full, hit)


def test_batch_rows_are_independent_and_graph_replay_is_exact(model):
    n = 513
    p = _prompt(n, seed=9, vocab=model.vocab)
    perm = _blocks(10, NSEQ * MB)                   # disjoint physical blocks per row
    rows = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
    for blocks in rows:
        model.forward(**prefill_inputs(p[:-1], blocks, BS, MAXLEN, 0))
    inp4 = decode_inputs([p[-1]] * NSEQ, [n - 1] * NSEQ, rows, BS, MAXLEN)
    first = model.forward(**inp4)                   # eager launches (and the capture) for this shape
    again = model.forward(**inp4)                   # graph replay
    assert torch.equal(first, again)
    for r in range(1, NSEQ):
        assert torch.equal(first[0], first[r]), r
    one = model.forward(**decode_inputs([p[-1]], [n - 1], [rows[2]], BS, MAXLEN))
    assert _close(one[0], first[2])                 # B = 1 splits the context differently: same value, other summation order
    assert model.forward_tokens(**inp4).tolist() == first.argmax(dim=1).tolist()


def test_shared_prefix_batch_matches_unshared(model):
    """four sequences sharing a 512-token prefix (16 full blocks, the SAME
    physical blocks in every block table) + unique suffixes of 64..256 tokens.  Encoding each
    suffix on top of the cached prefix (computed_context_lens = 512) must equal encoding the whole
    prompt into private blocks, and so must the following token-generation batch."""
    prefix = _prompt(512, seed=21, vocab=model.vocab)
    sufs = [_prompt(n, seed=30 + i, vocab=model.vocab) for i, n in enumerate((64, 128, 200, 256))]
    perm = _blocks(22, 9 * MB)                      # disjoint: 16 shared blocks, 4 suffix tails, 4 private rows
    shared = perm[:16]
    hit_rows = [shared + perm[(1 + i) * MB:(1 + i) * MB + MB - 16] for i in range(NSEQ)]
    own_rows = [perm[(5 + i) * MB:(6 + i) * MB] for i in range(NSEQ)]
    hit_logits, own_logits = [], []
    model.forward(**prefill_inputs(prefix, shared, BS, MAXLEN, 0))                        # the prefix's KV
    for i, suf in enumerate(sufs):
        hit_logits.append(model.forward(**prefill_inputs(prefix + suf, hit_rows[i], BS, MAXLEN, 512)))
    toks = [int(l.argmax()) for l in hit_logits]
    pos = [512 + len(sf) for sf in sufs]
    step_hit = model.forward(**decode_inputs(toks, pos, hit_rows, BS, MAXLEN))
    for i, suf in enumerate(sufs):
        own_logits.append(model.forward(**prefill_inputs(prefix + suf, own_rows[i], BS, MAXLEN, 0)))
    step_own = model.forward(**decode_inputs(toks, pos, own_rows, BS, MAXLEN))
    for i in range(NSEQ):
        assert _close(hit_logits[i], own_logits[i]), i
    assert _close(step_hit, step_own)


def test_fp8_activation_mode_is_the_same_function_up_to_its_quantization_noise():
    """Characterisation, not parity (the rule itself is pinned against the oracle on the small
    models): per-token e4m3 activations carry 3 mantissa bits (3.6 % rms per element) into every
    GEMM of 32 layers.  On these UNSTRUCTURED random weights that noise reaches the logits almost
    undamped -- measured 0.31-0.49 rms against a logit std of 1.28 (cosine 0.92-0.97), far more
    than on trained checkpoints -- so the check is only that the two modes compute the same
    function: strongly correlated logits, no blow-up, deterministic."""
    m8 = _model(a8=1)
    ref = _model(a8=0)
    p = _prompt(600, seed=11)
    blocks = _blocks(20)
    a = m8.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))
    a2 = m8.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))
    b = ref.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))
    assert torch.equal(a, a2) and torch.isfinite(a).all()
    cos = torch.nn.functional.cosine_similarity(a, b).item()
    rms = (a - b).pow(2).mean().sqrt().item()
    assert cos > 0.9 and rms < 0.5 * b.std().item(), (cos, rms)
    assert abs(a.std().item() / b.std().item() - 1) < 0.1
    m8.close()
    ref.close()


def test_fused_speculation_at_the_headline_shapes():
    """Llama-3.1-8B FP8 target (16 rows: 4 sequences x 4 candidates) with a COPY of itself as the draft
    (same seed: every candidate is what the target would choose, so a step yields 4 tokens): the
    16-row weight-streaming pass at the real shapes (K-streamed down_proj, 16-row prologues, 16
    attention rows per layer), the chained draft steps, the catch-up row, acceptance.  The text must
    be the target's own greedy text; an id may differ only where the plain run's top-2 gap is inside
    the tolerance band of the file header (the 16-row pass splits the attention context differently
    from the 4-row pass)."""
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    K, n_new, L = 4, 13, 300
    nb = 1 + 2 * NSEQ * MB

    def build(rows):
        m = NativeModel(**LLAMA31_8B, num_blocks=nb, block_size=BS, max_num_seqs=rows, max_model_len=MAXLEN,
                        weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
                        tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, ctx_buckets=[256, 512, 1024, 2048],
                        prefill_fp8_activations=0)
        m.init_synthetic_weights(1, 0.02)
        m.finalize()
        return m
    target, draft = build(NSEQ * K), build(2 * NSEQ)
    g = torch.Generator().manual_seed(9)
    prompts = [torch.randint(0, 128256, (L + 7 * i,), generator=g).tolist() for i in range(NSEQ)]
    perm = (torch.randperm(nb - 1, generator=g) + 1).tolist()
    blocks_a = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
    blocks_b = [perm[(NSEQ + i) * MB:(NSEQ + i + 1) * MB] for i in range(NSEQ)]

    # the target alone: one token per step, keeping the top-2 gap of every step
    want, gaps = [[] for _ in prompts], [[] for _ in prompts]

    def take(i, row):
        top2 = row.topk(2)
        want[i].append(int(top2.indices[0]))
        gaps[i].append(float(top2.values[0] - top2.values[1]))
    for i, p in enumerate(prompts):
        take(i, target.forward(**prefill_inputs(p, blocks_a[i], BS, MAXLEN, 0))[0])
    for s in range(1, n_new):
        lg = target.forward(**decode_inputs([w[-1] for w in want], [len(p) + s - 1 for p in prompts], blocks_a, BS, MAXLEN))
        for i in range(NSEQ):
            take(i, lg[i])

    got = [[] for _ in prompts]
    for i, p in enumerate(prompts):
        inp = prefill_inputs(p, blocks_b[i], BS, MAXLEN, 0)
        got[i].append(int(target.forward(**inp).argmax(dim=1)[0]))
        draft.forward(**inp)
    bt = torch.tensor(blocks_b, dtype=torch.long)
    pending = [None] * NSEQ
    steps = produced = 0
    while min(len(x) for x in got) < n_new:
        last = torch.tensor([x[-1] for x in got])
        pos = [len(p) + len(x) - 1 for p, x in zip(prompts, got)]
        cu = torch.tensor([st[1] if st is not None and st[0] == q else -1 for st, q in zip(pending, pos)])
        acc, nxt = target.forward_spec(draft, last, torch.tensor(pos), bt, K, catchup_ids=cu)
        for i in range(NSEQ):
            n = int(nxt[i]) - pos[i]
            assert 1 <= n <= K
            got[i].extend(acc[i, :n].tolist())
            pending[i] = (pos[i] + K, int(acc[i, K - 2])) if n == K else None
            produced += n
        steps += 1
    for i in range(NSEQ):
        for s, (a, b) in enumerate(zip(got[i][:n_new], want[i])):
            if a != b:
                assert gaps[i][s] < TOL_MAX, (i, s, a, b, gaps[i][s])
                break                                   # the texts part ways at a near-tie; nothing to compare behind it
    print(f"8B speculation with a copied draft: {produced / (steps * NSEQ):.2f} tokens per sequence and step")
    assert produced / (steps * NSEQ) > 3.0              # a near-tie may cost a window, not the rule
    draft.close()
    target.close()
