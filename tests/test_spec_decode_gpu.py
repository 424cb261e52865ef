"""Fused speculation (mi_forward_spec) on the MI355X: k chained draft steps + one target pass over
the B * k candidate rows + greedy acceptance, all on the device.

Reference: NxDI's fused draft + target graph behind
/root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:349-355, output contract
:308-333.  No reference test holds a speculative golden output (parity unpinned); what is checkable
is (a) the oracle's restatement of the step (oracle/spec_decode.py) on the same inputs and (b) the
property that greedy acceptance never changes the text: the tokens must be exactly those the target
alone generates greedily -- for ANY draft (a copy of the target: everything accepted; an unrelated
model: almost nothing accepted; a perturbed copy: something in between).
"""
import dataclasses

import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.spec_decode import fused_speculation_step, remask
from oracle.synth import make_prompts, make_weights, zoo_config
from tests.helpers import decode_inputs, prefill_inputs

pytestmark = pytest.mark.gpu

BS, MAXLEN, NSEQ, K = 32, 256, 4, 4
MB = MAXLEN // BS
NB = 1 + 2 * NSEQ * MB


def _model(cfg, weights, max_num_seqs, weight_dtype="bf16"):
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
        num_blocks=NB, block_size=BS, max_num_seqs=max_num_seqs, max_model_len=MAXLEN,
        weight_dtype=MI_W[weight_dtype], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1,
        tp_degree=1, tp_rank=0, device_id=0, use_graphs=1, prefill_fp8_activations=0)
    m.load_state_dict(weights)
    m.finalize()
    return m


def _drafts(cfg, w):
    """name -> (draft config, draft weights)"""
    g = torch.Generator().manual_seed(11)
    noisy = {}
    for name, t in w.items():
        if t.dim() == 2 and "embed" not in name:
            noisy[name] = (t + torch.randn(t.shape, generator=g) * t.std() * 0.25).to(torch.bfloat16).float()
        else:
            noisy[name] = t
    small = dataclasses.replace(zoo_config("tinyllama_like"), vocab_size=cfg.vocab_size)
    return {"copy": (cfg, w), "perturbed": (cfg, noisy), "unrelated": (small, make_weights(small, seed=7))}


class _Catchup:
    """What the adapter keeps per sequence (MI355XCausalLM._forward_fused_speculation): after a step that
    generated all k tokens, the token in front of the last one has not been through the draft."""

    def __init__(self, n):
        self.pending = [None] * n            # (position of the next step, token)

    def ids(self, pos):
        out = []
        for i, p in enumerate(pos):
            st, self.pending[i] = self.pending[i], None
            out.append(st[1] if st is not None and st[0] == int(p) else -1)
        return torch.tensor(out, dtype=torch.long)

    def record(self, pos, acc, nxt, k):
        for i, p in enumerate(pos):
            if k >= 2 and int(nxt[i]) - int(p) == k:
                self.pending[i] = (int(p) + k, int(acc[i, k - 2]))


def _plain_greedy(model, prompts, blocks, n_new):
    out = [[] for _ in prompts]
    for i, p in enumerate(prompts):
        out[i].append(int(model.forward(**prefill_inputs(p, blocks[i], BS, MAXLEN, 0)).argmax(dim=1)[0]))
    for s in range(1, n_new):
        last = [o[-1] for o in out]
        pos = [len(p) + s - 1 for p in prompts]
        ids = model.forward(**decode_inputs(last, pos, blocks, BS, MAXLEN)).argmax(dim=1).tolist()
        for i, t in enumerate(ids):
            out[i].append(int(t))
    return out


@pytest.mark.parametrize("weight_dtype", ["bf16", "f8e4m3"])
@pytest.mark.parametrize("draft_kind", ["copy", "perturbed", "unrelated"])
def test_fused_speculation_is_lossless_and_matches_the_oracle_step(draft_kind, weight_dtype):
    cfg = zoo_config("llama31_like")
    w = make_weights(cfg, seed=1)
    dcfg, dw = _drafts(cfg, w)[draft_kind]
    prompts = make_prompts(cfg.vocab_size, 0)
    n_new = 24
    target = _model(cfg, w, NSEQ * K, weight_dtype)
    draft = _model(dcfg, dw, 2 * NSEQ, weight_dtype)        # every sequence + one catch-up row each
    blocks_a = [[1 + i * MB + j for j in range(MB)] for i in range(NSEQ)]
    blocks_b = [[1 + (NSEQ + i) * MB + j for j in range(MB)] for i in range(NSEQ)]
    want = _plain_greedy(target, prompts, blocks_a, n_new)       # the target alone, one token per step

    # speculation on other blocks: prefill both models, first token from the target's prefill
    got = [[] for _ in prompts]
    for i, p in enumerate(prompts):
        inp = prefill_inputs(p, blocks_b[i], BS, MAXLEN, 0)
        got[i].append(int(target.forward(**inp).argmax(dim=1)[0]))
        draft.forward(**inp)
    bt = torch.tensor([b + [0] * (MB - len(b)) for b in blocks_b], dtype=torch.long)
    steps, produced = 0, 0
    cu = _Catchup(NSEQ)
    while min(len(g) for g in got) < n_new:
        last = torch.tensor([g[-1] for g in got])
        pos = torch.tensor([len(p) + len(g) - 1 for p, g in zip(prompts, got)])
        acc, nxt = target.forward_spec(draft, last, pos, bt, K, catchup_ids=cu.ids(pos))
        cu.record(pos, acc, nxt, K)
        masked = remask(acc, nxt, pos)
        for i in range(NSEQ):
            toks = [int(t) for t in masked[i] if t != -1]
            assert 1 <= len(toks) <= K and int(nxt[i]) == int(pos[i]) + len(toks)
            got[i].extend(toks)
            produced += len(toks)
        steps += 1
    for i in range(NSEQ):
        assert got[i][:n_new] == want[i], (draft_kind, i, got[i][:n_new], want[i])
    rate = produced / (steps * NSEQ)
    print(f"{draft_kind}/{weight_dtype}: {rate:.2f} tokens per sequence and step over {steps} steps")
    if draft_kind == "copy":
        assert rate == K                                         # the draft IS the target: every candidate accepted
    if draft_kind == "unrelated":
        assert rate < 1.5
    draft.close()
    target.close()


def test_fused_speculation_step_against_the_oracle():
    """One speculation step at a time, same inputs to the HIP path and to oracle/spec_decode.py
    (bf16 weights; both sides continue from the ORACLE's accepted tokens, so a near-tie that flips
    one side's argmax cannot snowball)."""
    cfg = zoo_config("llama31_like")
    w = make_weights(cfg, seed=1)
    dcfg, dw = _drafts(cfg, w)["perturbed"]
    prompts = make_prompts(cfg.vocab_size, 0)
    target, draft = _model(cfg, w, NSEQ * K), _model(dcfg, dw, 2 * NSEQ)
    o_target = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16")
    o_draft = PagedDecoderOracle(dcfg, dw, NB, BS, compute="bf16")
    blocks = [[1 + i * MB + j for j in range(MB)] for i in range(NSEQ)]
    seqs = []
    for i, p in enumerate(prompts):
        inp = prefill_inputs(p, blocks[i], BS, MAXLEN, 0)
        first = int(o_target.forward(**inp).argmax(dim=1)[0])
        o_draft.forward(**inp)
        target.forward(**inp)
        draft.forward(**inp)
        seqs.append([first])
    bt = torch.tensor(blocks, dtype=torch.long)
    agree = total = 0
    cu = _Catchup(NSEQ)
    for _ in range(8):
        last = [s[-1] for s in seqs]
        pos = [len(p) + len(s) - 1 for p, s in zip(prompts, seqs)]
        acc_o, nxt_o = fused_speculation_step(o_target, o_draft, last, pos, bt, K, BS, MAXLEN)
        acc, nxt = target.forward_spec(draft, torch.tensor(last), torch.tensor(pos), bt, K, catchup_ids=cu.ids(pos))
        cu.record(pos, acc_o, nxt_o, K)        # both sides continue from the oracle's tokens
        total += NSEQ
        agree += sum(int(torch.equal(acc[i], acc_o[i]) and nxt[i] == nxt_o[i]) for i in range(NSEQ))
        m = remask(acc_o, nxt_o, torch.tensor(pos))
        for i in range(NSEQ):
            seqs[i].extend(int(t) for t in m[i] if t != -1)
    print(f"speculation steps identical to the oracle's: {agree}/{total}")
    assert agree >= total - 2, (agree, total)                    # bf16 near-ties may flip a draft token
    draft.close()
    target.close()


def test_fused_speculation_window_clipped_at_max_model_len_and_block_boundary():
    """Candidates beyond max_model_len are never produced; a window that crosses a block boundary
    takes its slots from the block table (the reference's consecutive slots, runner.py:825-830, hold
    only inside one block)."""
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, seed=3)
    target, draft = _model(cfg, w, NSEQ * K), _model(cfg, w, 2 * NSEQ)
    g = torch.Generator().manual_seed(5)
    L = 218                                                       # windows cross the block boundary 223 | 224, then run into position 255
    p = torch.randint(0, cfg.vocab_size, (L,), generator=g).tolist()
    blocks = [[9, 3, 12, 5, 1, 14, 7, 2]]                         # scattered blocks
    inp = prefill_inputs(p, blocks[0], BS, MAXLEN, 0)
    seq = [int(target.forward(**inp).argmax(dim=1)[0])]
    draft.forward(**inp)
    bt = torch.tensor(blocks, dtype=torch.long)
    cu = _Catchup(1)
    full_windows = 0
    while L + len(seq) - 1 < MAXLEN:                              # as long as the last token has a position to be processed at
        pos = L + len(seq) - 1
        acc, nxt = target.forward_spec(draft, torch.tensor([seq[-1]]), torch.tensor([pos]), bt, K, catchup_ids=cu.ids([pos]))
        cu.record([pos], acc, nxt, K)
        n = int(nxt[0]) - pos
        lim = max(1, min(K, MAXLEN - pos - 1))                    # a window never takes the sequence past MAXLEN tokens
        full_windows += n == lim
        assert 1 <= n <= lim
        seq.extend(acc[0, :n].tolist())
        assert L + len(seq) <= MAXLEN or pos == MAXLEN - 1        # (the step at the last position yields its one token, like the plain step)
    assert L + len(seq) == MAXLEN + 1 and full_windows >= 9       # the draft is a copy and is caught up: full windows, the last ones clipped
    # the same text from the target alone (fresh blocks)
    blocks2 = [[20, 21, 22, 23, 24, 25, 26, 27]]
    want = _plain_greedy(target, [p], blocks2, len(seq))
    assert seq == want[0]
    # the draft runs on the target's stream: closing the target first must leave the draft closable
    target.close()
    draft.close()


# ---- the whole plugin path: MI355XEngine(speculative_config=...) ---------------------------------
def _hf_like(cfg, model_type="llama"):
    from types import SimpleNamespace
    return SimpleNamespace(
        architectures=["LlamaForCausalLM"], model_type=model_type, vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
        intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
        num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads, head_dim=cfg.head_dim,
        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, rope_scaling=cfg.rope_scaling,
        tie_word_embeddings=False)


@pytest.mark.parametrize("draft_kind,quant", [("perturbed", False), ("copy", True), ("unrelated", False)])
def test_engine_with_speculative_config_generates_the_target_text(draft_kind, quant):
    """vLLM's speculative_config through the plugin (reference loader.py:785-791, 349-355;
    runner.py:293-345): same prompts, same greedy text as the engine without a draft model; the HF
    goldens hold for it as for every other greedy path; steps produce several tokens."""
    from tests.test_engine_gpu import check_against_golden
    from vllm_neuron_amd._vllm_compat import SamplingParams, SimpleModelConfig, SimpleSpeculativeConfig
    from vllm_neuron_amd.engine import MI355XEngine
    name = "llama31_like"
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    dcfg, dw = _drafts(cfg, w)[draft_kind]
    prompts = make_prompts(cfg.vocab_size, 0)
    sp = SamplingParams(temperature=0.0, max_tokens=12)
    q = {"quantized": True, "quantization_dtype": "f8e4m3", "quantization_type": "per_channel_symmetric"} if quant else {}

    plain = MI355XEngine(_hf_like(cfg), max_model_len=256, max_num_seqs=4, block_size=32, enable_prefix_caching=True,
                         override_mi355x_config={"state_dict": w, **q})
    want = [o.token_ids for o in plain.generate(prompts, sp)]
    plain.worker.model_runner.model.model.close()

    spec = SimpleSpeculativeConfig(num_speculative_tokens=K, draft_model_config=SimpleModelConfig(model="", hf_config=_hf_like(dcfg)))
    eng = MI355XEngine(_hf_like(cfg), max_model_len=256, max_num_seqs=4, block_size=32, enable_prefix_caching=True,
                       speculative_config=spec, override_mi355x_config={"state_dict": w, "draft_state_dict": dw, **q})
    mcfg = eng.worker.model_runner.model.mi355x_config
    assert mcfg.enable_fused_speculation and mcfg.speculation_length == K and mcfg.on_device_sampling_config
    steps = 0
    for p in prompts:
        eng.add_request(p, sp)
    multi = 0
    while eng.has_unfinished_requests():
        _, out = eng.step()
        steps += 1
        multi += sum(len(t) > 1 for t in out.sampled_token_ids)
    outs = [eng.outputs[f"req-{i}"] for i in range(len(prompts))]
    got = [o.token_ids for o in outs]
    assert got == want, (draft_kind, got, want)
    assert all(o.finished and len(o.token_ids) == 12 for o in outs)
    if not quant:
        check_against_golden(name, outs)
    if draft_kind != "unrelated":
        assert multi > 0                     # some step generated more than one token for a request
    # a request that samples (the reference's EAGLE test runs top_k = 50): served one token per step
    rid = eng.add_request(prompts[0], SamplingParams(temperature=0.8, top_k=20, max_tokens=6))
    while eng.has_unfinished_requests():
        _, out = eng.step()
        assert all(len(t) <= 1 for t in out.sampled_token_ids)
    assert len(eng.outputs[rid].token_ids) == 6 and eng.outputs[rid].finished
    runner = eng.worker.model_runner
    runner.model.draft.close()
    runner.model.model.close()


def test_engine_speculation_runs_a_request_up_to_max_model_len():
    """a greedy request that runs to the model length with an agreeing draft used to leave
    max_model_len + 1 tokens (runner assertion).  The last windows are clipped so that the sequence
    ends at exactly max_model_len tokens, with the plain engine's text."""
    from vllm_neuron_amd._vllm_compat import SamplingParams, SimpleModelConfig, SimpleSpeculativeConfig
    from vllm_neuron_amd.engine import MI355XEngine
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, seed=2)
    MAXL = 96
    g = torch.Generator().manual_seed(11)
    prompt = torch.randint(0, cfg.vocab_size, (70,), generator=g).tolist()
    sp = SamplingParams(temperature=0.0, max_tokens=1000, ignore_eos=True)
    plain = MI355XEngine(_hf_like(cfg), max_model_len=MAXL, max_num_seqs=4, block_size=32, enable_prefix_caching=True,
                         override_mi355x_config={"state_dict": w})
    want = plain.generate([prompt], sp)[0].token_ids
    plain.worker.model_runner.model.model.close()
    assert len(prompt) + len(want) == MAXL
    spec = SimpleSpeculativeConfig(num_speculative_tokens=K, draft_model_config=SimpleModelConfig(model="", hf_config=_hf_like(cfg)))
    eng = MI355XEngine(_hf_like(cfg), max_model_len=MAXL, max_num_seqs=4, block_size=32, enable_prefix_caching=True,
                       speculative_config=spec, override_mi355x_config={"state_dict": w, "draft_state_dict": w})
    got = eng.generate([prompt], sp)[0].token_ids
    assert got == want
    runner = eng.worker.model_runner
    runner.model.draft.close()
    runner.model.model.close()


def test_engine_speculation_with_a_tensor_parallel_target(monkeypatch):
    """The reference's own speculation test runs tensor-parallel (test/tiny/test_eagle_speculative_decoding.py:
    tensor_parallel_size=32).  Here: the target as an in-process group of 2 rank shards (both on GPU 0), the
    draft unsharded on rank 0's GPU; candidates' ids are handed to every shard, the vocabulary-parallel logits
    meet at rank 0's sampler.  Same greedy text as the plain engine."""
    from tests.test_engine_gpu import check_against_golden
    from vllm_neuron_amd._vllm_compat import SamplingParams, SimpleModelConfig, SimpleSpeculativeConfig
    from vllm_neuron_amd.engine import MI355XEngine
    monkeypatch.setenv("MI355X_TP_LOOPBACK", "1")
    name = "llama31_like"
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    dcfg, dw = _drafts(cfg, w)["perturbed"]
    prompts = make_prompts(cfg.vocab_size, 0)
    spec = SimpleSpeculativeConfig(num_speculative_tokens=K, draft_model_config=SimpleModelConfig(model="", hf_config=_hf_like(dcfg)))
    eng = MI355XEngine(_hf_like(cfg), max_model_len=256, max_num_seqs=4, block_size=32, enable_prefix_caching=True,
                       tensor_parallel_size=2, speculative_config=spec,
                       override_mi355x_config={"state_dict": w, "draft_state_dict": dw})
    multi = 0
    for p in prompts:
        eng.add_request(p, SamplingParams(temperature=0.0, max_tokens=12))
    while eng.has_unfinished_requests():
        _, out = eng.step()
        multi += sum(len(t) > 1 for t in out.sampled_token_ids)
    outs = [eng.outputs[f"req-{i}"] for i in range(len(prompts))]
    check_against_golden(name, outs)
    assert multi > 0
    runner = eng.worker.model_runner
    runner.model.draft.close()
    runner.model.model.close()


def test_forward_spec_argument_checks():
    """Bad calls come back as errors with a message (MI_EINVAL -> ValueError), never as a fault on the GPU."""
    cfg = zoo_config("tinyllama_like")
    w = make_weights(cfg, seed=3)
    target, draft = _model(cfg, w, NSEQ * K), _model(cfg, w, 2 * NSEQ)
    small = _model(cfg, w, 2)                                      # too few rows for 4 x 4 candidates
    bt = torch.tensor([[1, 2, 3, 4, 5, 6, 7, 8]] * NSEQ, dtype=torch.long)
    ids, pos = torch.tensor([1, 2, 3, 4]), torch.tensor([10, 11, 12, 13])
    with pytest.raises(ValueError, match="max_num_seqs"):
        small.forward_spec(draft, ids, pos, bt, K)
    with pytest.raises(ValueError, match="two finalized contexts"):
        target.forward_spec(target, ids, pos, bt, K)
    with pytest.raises(ValueError, match="position out of range"):
        target.forward_spec(draft, ids, torch.tensor([10, 11, 12, MAXLEN]), bt, K)
    with pytest.raises(ValueError, match="narrower than the speculation window"):
        target.forward_spec(draft, ids, torch.tensor([10, 11, 12, 63]), bt[:, :2], K)   # positions 63..66 need 3 blocks
    with pytest.raises(ValueError, match="token id out of range"):
        target.forward_spec(draft, torch.tensor([1, 2, 3, cfg.vocab_size]), pos, bt, K)
    with pytest.raises(ValueError, match="block_table entry out of range"):
        bad = bt.clone()
        bad[2, 0] = NB + 5
        target.forward_spec(draft, ids, pos, bad, K)
    with pytest.raises(ValueError, match="catch-up rows"):
        tiny_draft = _model(cfg, w, NSEQ)
        target.forward_spec(tiny_draft, ids, pos, bt, K, catchup_ids=torch.tensor([5, 5, 5, 5]))
    # and a good call still works afterwards
    acc, nxt = target.forward_spec(draft, ids, pos, bt, K)
    assert acc.shape == (NSEQ, K) and bool(((nxt - pos) >= 1).all())
    for m in (small, draft, target):
        m.close()
