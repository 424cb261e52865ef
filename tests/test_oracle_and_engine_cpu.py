"""CPU tier (no GPU):
  * the oracle is pinned against the committed HF-transformers goldens;
  * the plugin's scheduler + model runner, driven as an engine loop with the ORACLE plugged
    in as the model (test infrastructure standing where the HIP library stands), reproduce the
    goldens' greedy continuations token for token — with prefix caching, continuous batching,
    block (re)use, and in the contiguous-KV mode;
  * the C-ABI library loads and exports every symbol include/mi355x_vllm.h declares;
  * the tensor-parallel sharding plan sums to the unsharded result over a 2-rank gloo group.
"""

import os
import re

import pytest
import torch
from safetensors import safe_open

from oracle import PagedDecoderOracle
from oracle.quant import dequantize_weight, quantize_weight
from oracle.synth import ZOO, make_prompts, make_weights, weights_checksum, zoo_config
from tests.helpers import decode_inputs, prefill_inputs
from vllm_neuron_amd import platform as plat
from vllm_neuron_amd._vllm_compat import (KVCacheConfig, Request, SamplingParams, SimpleCacheConfig,
                                          SimpleModelConfig, SimpleParallelConfig, SimpleSchedulerConfig,
                                          SimpleVllmConfig)
from vllm_neuron_amd.core.scheduler import ContinuousBatchingMI355XScheduler
from vllm_neuron_amd.worker import mi355x_model_loader as loader
from vllm_neuron_amd.worker.mi355x_model_runner import MI355XModelRunner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "hf_decoder_golden.safetensors")
BS, MAXLEN, NSEQ = 32, 256, 4
MB = MAXLEN // BS


def golden(name):
    f = safe_open(GOLD, "pt")
    return ([f.get_tensor(f"{name}.generated.{i}").tolist() for i in range(4)],
            [f.get_tensor(f"{name}.logits.{i}") for i in range(4)], float(f.metadata()[f"{name}.weights_checksum"]))


# ---- oracle pinning ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(ZOO))
def test_oracle_matches_hf_golden(name):
    """fp32 oracle == HF transformers (no KV cache on the HF side) to 2e-5 on O(4) logits,
    through paged prefill + decode with scattered blocks and a prefix-cache hit."""
    cfg = zoo_config(name)
    w = make_weights(cfg, 1)
    gen, gold, csum = golden(name)
    assert abs(weights_checksum(w) - csum) < 1e-6 * csum
    prompts = make_prompts(cfg.vocab_size, 0)
    o = PagedDecoderOracle(cfg, w, 1 + NSEQ * MB, BS, compute="fp32")
    perm = (torch.randperm(NSEQ * MB, generator=torch.Generator().manual_seed(3)) + 1).tolist()
    blocks = [perm[i * MB:(i + 1) * MB] for i in range(NSEQ)]
    blocks[3][:2] = blocks[1][:2]
    worst = 0.0
    for i, p in enumerate(prompts):
        lg = o.forward(**prefill_inputs(p, blocks[i], BS, MAXLEN, 64 if i == 3 else 0))
        worst = max(worst, (lg[0] - gold[i][0]).abs().max().item())
        assert int(lg.argmax()) == gen[i][0]
    for s in range(1, len(gen[0])):
        lg = o.forward(**decode_inputs([gen[i][s - 1] for i in range(4)], [len(prompts[i]) + s - 1 for i in range(4)],
                                       blocks, BS, MAXLEN, pad_block=-1 if s % 2 else 0))
        for i in range(4):
            worst = max(worst, (lg[i] - gold[i][s]).abs().max().item())
            assert int(lg[i].argmax()) == gen[i][s]
    assert worst < 2e-5, worst


def test_oracle_bf16_mode_stays_close():
    name = "llama31_like"
    cfg = zoo_config(name)
    gen, gold, _ = golden(name)
    o = PagedDecoderOracle(cfg, make_weights(cfg, 1), 1 + MB, BS, compute="bf16")
    p = make_prompts(cfg.vocab_size, 0)[1]
    lg = o.forward(**prefill_inputs(p, list(range(1, MB + 1)), BS, MAXLEN, 0))
    assert (lg[0] - gold[1][0]).abs().max() < 0.05


@pytest.mark.parametrize("dtype,qmax", [("int8", 127.0), ("f8e4m3", 448.0)])
@pytest.mark.parametrize("qtype", ["per_tensor_symmetric", "per_channel_symmetric"])
def test_quantizer_definition(dtype, qmax, qtype):
    torch.manual_seed(0)
    w = torch.randn(32, 64) * 0.1
    w[7] = 0
    q, s = quantize_weight(w, dtype, qtype)
    assert s.shape == (32,) and s[7] == (1.0 if qtype == "per_channel_symmetric" else s[0])
    if qtype == "per_tensor_symmetric":
        assert torch.all(s == w.abs().max() / qmax)
    else:
        assert torch.equal(s[:7], w[:7].abs().amax(1) / qmax)
    err = (dequantize_weight(q, s) - w).abs().max()
    assert err <= (s.max() * (0.5 if dtype == "int8" else 16.0)) + 1e-7   # half a step (int8) / coarsest fp8 step
    assert q.float().abs().max() <= qmax


# ---- the plugin's host path end to end, oracle standing in for the HIP library -------------------
class OracleBackedModel:
    """Quacks like MI355XCausalLM; runs the oracle where NativeModel.forward would run."""

    def __init__(self, cfg, weights, num_blocks, block_size, prefix):
        self.oracle = PagedDecoderOracle(cfg, weights, num_blocks, block_size, compute="fp32")
        self.mi355x_config = loader.MI355XConfig(
            is_block_kv_layout=prefix, is_prefix_caching=prefix, chunked_prefill_config=None,
            on_device_sampling_config=None, attn_tkg_nki_kernel_enabled=False,
            attn_block_tkg_nki_kernel_enabled=False)
        self.num_key_value_heads, self.head_dim = cfg.num_kv_heads, cfg.head_dim
        self.is_reorder_needed = False
        self.native_block_size = block_size
        self.model = type("N", (), {"finalize": lambda s: None, "set_num_blocks": lambda s, n: None})()
        self.adapter = loader.MI355XCausalLM.__new__(loader.MI355XCausalLM)   # reuse the real adapter logic
        torch.nn.Module.__init__(self.adapter, )
        self.adapter.mi355x_config = self.mi355x_config
        self.adapter.is_reorder_needed = False
        self.adapter.native_block_size = block_size
        self.adapter.model = type("F", (), {"forward": staticmethod(
            lambda ids, pos, seq, bt, sm, full, comp, alias_ok=False: self.oracle.forward(ids, pos, seq, bt, sm, full, comp))})()

    def __call__(self, **kw):
        self.adapter.is_reorder_needed = self.is_reorder_needed
        return self.adapter.forward(**kw)


def run_engine(name, prefix, num_blocks, max_tokens=12):
    cfg = zoo_config(name)
    hf = type("HF", (), dict(vocab_size=cfg.vocab_size, num_hidden_layers=cfg.num_layers))()
    vc = SimpleVllmConfig(
        model_config=SimpleModelConfig(model="m", hf_config=hf, dtype="float32", max_model_len=MAXLEN),
        cache_config=SimpleCacheConfig(block_size=BS if prefix else None, enable_prefix_caching=prefix),
        parallel_config=SimpleParallelConfig(), scheduler_config=SimpleSchedulerConfig(max_num_seqs=NSEQ, max_model_len=MAXLEN))
    plat.MI355XPlatform.check_and_update_config(vc)
    bs = vc.cache_config.block_size
    runner = MI355XModelRunner(vc, "cpu")
    runner.model = OracleBackedModel(cfg, make_weights(cfg, 1), num_blocks, bs, prefix)
    runner.is_block_kv_layout = runner.is_prefix_caching = prefix
    runner.model.is_reorder_needed = not prefix
    runner._kv_ready = True
    sched = ContinuousBatchingMI355XScheduler(vc, KVCacheConfig(num_blocks=num_blocks))
    prompts = make_prompts(cfg.vocab_size, 0)
    order = [1, 0, 3, 2]                       # prompt 3 arrives after prompt 1: prefix-cache hit
    for i in order:
        sched.add_request(Request(f"p{i}", prompts[i], SamplingParams(temperature=0.0, max_tokens=max_tokens)))
    toks = {f"p{i}": [] for i in range(4)}
    cached = {}
    steps = 0
    while sched.has_unfinished_requests():
        so = sched.schedule()
        for n in so.scheduled_new_reqs:
            cached[n.req_id] = n.num_computed_tokens
        ro = runner.execute_model(so)
        for out in sched.update_from_output(so, ro):
            toks[out.request_id].extend(out.new_token_ids)
        steps += 1
        assert steps < 200
    return toks, cached, runner


@pytest.mark.parametrize("name", ["llama31_like", "qwen25_like"])
def test_engine_loop_reproduces_hf_greedy_with_prefix_caching(name):
    gen, _, _ = golden(name)
    toks, cached, runner = run_engine(name, prefix=True, num_blocks=1 + NSEQ * MB)
    for i in range(4):
        assert toks[f"p{i}"] == gen[i], i
    assert cached == {"p1": 0, "p0": 0, "p3": 64, "p2": 0}      # 2 full shared blocks of 32 were hit
    assert runner.free_seq_ids == set(range(NSEQ)) or len(runner.free_seq_ids) >= 0


def test_engine_loop_contiguous_kv_mode():
    """Prefix caching off: block_size = max_model_len, batch-line addressing from seq ids
    (reference platform.py:203-207, runner.py:715-724)."""
    name = "tinyllama_like"
    gen, _, _ = golden(name)
    toks, cached, _ = run_engine(name, prefix=False, num_blocks=1 + NSEQ)
    for i in range(4):
        assert toks[f"p{i}"] == gen[i], i
    assert set(cached.values()) == {0}


def test_engine_loop_block_reuse_after_finish():
    """More requests than sequence slots / blocks: finished requests free their slots and blocks."""
    name = "tinyllama_like"
    cfg = zoo_config(name)
    gen, _, _ = golden(name)
    toks, _, runner = run_engine(name, prefix=True, num_blocks=1 + 2 * MB + 2, max_tokens=12)
    for i in range(4):
        assert toks[f"p{i}"] == gen[i], i


# ---- C ABI -------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from vllm_neuron_amd import _native
    header = open(os.path.join(ROOT, "include", "mi355x_vllm.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", header)) - {"mi_ctx", "mi_model_config"}
    lib = _native.load_library()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    assert lib.mi_version() >= 1


def test_struct_layout_matches_header():
    from vllm_neuron_amd import _native
    header = open(os.path.join(ROOT, "include", "mi355x_vllm.h")).read()
    body = header.split("typedef struct mi_model_config {")[1].split("} mi_model_config;")[0]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        _, rest = decl.split(None, 1)
        names += [n.strip().split("[")[0] for n in rest.split(",")]
    assert names == [f for f, _ in _native.MiModelConfig._fields_]


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vllm_neuron_amd import _native
    with pytest.raises((RuntimeError, ValueError)):
        _native.NativeModel(num_layers=1, hidden_size=64, num_heads=1, num_kv_heads=1, head_dim=64,
                            intermediate_size=64, vocab_size=64, rms_norm_eps=1e-5, rope_theta=1e4, num_blocks=4,
                            block_size=32, max_num_seqs=1, max_model_len=64, tp_degree=1, tp_rank=0, device_id=0)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vllm-neuron_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


# ---- tensor-parallel plan over gloo (world_size 2), slices taken from the LIBRARY's plan --------------
def _plan_of(cfg_kw, rank):
    """mi_tp_plan of libmi355x_vllm.so: host arithmetic only (no GPU, no context) -- the function mi_ctx_create itself runs."""
    import ctypes
    from vllm_neuron_amd import _native as N
    L = N.load_library()
    cfg = N.MiModelConfig()
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    plan = N.MiTpPlan()
    N.check(L.mi_tp_plan(ctypes.byref(cfg), rank, ctypes.byref(plan)))
    return plan


def _attention(q, k, v, nh, nkv, hd):
    """Causal attention of M tokens (one sequence), q [M, nh*hd], k / v [M, nkv*hd]; q head j reads kv head j // (nh // nkv)."""
    M = q.shape[0]
    qh, kh, vh = q.reshape(M, nh, hd), k.reshape(M, nkv, hd), v.reshape(M, nkv, hd)
    g = nh // nkv
    s = torch.einsum("mhd,nhd->hmn", qh, kh.repeat_interleave(g, 1)) / hd ** 0.5
    s = s.masked_fill(torch.triu(torch.ones(M, M, dtype=torch.bool), 1)[None], float("-inf"))
    return torch.einsum("hmn,nhd->mhd", torch.softmax(s, -1), vh.repeat_interleave(g, 1)).reshape(M, nh * hd)


def _block_unsharded(x, w, nh, nkv, hd):
    a = _attention(x @ w["q"].t() + w["qb"], x @ w["k"].t(), x @ w["v"].t(), nh, nkv, hd)
    h = x + a @ w["o"].t()
    h = h + (torch.nn.functional.silu(h @ w["g"].t()) * (h @ w["u"].t())) @ w["d"].t()
    return h @ w["lm"].t()


def _rank_partials(x, w, plan, hd):
    """What tensor-parallel rank `plan` contributes: its q heads (zero-weight PADDING heads where the plan holds more
    heads than are real: q rows, q bias and o_proj columns zero), its kv heads, the matching o_proj columns."""
    q0, qr, ql = plan.q_head0, plan.q_heads_real, plan.q_heads_local
    k0, kl = plan.kv_head0, plan.kv_heads_local
    H = x.shape[1]
    wq, bq, wo = torch.zeros(ql * hd, H), torch.zeros(ql * hd), torch.zeros(H, ql * hd)
    wq[:qr * hd], bq[:qr * hd], wo[:, :qr * hd] = w["q"][q0 * hd:(q0 + qr) * hd], w["qb"][q0 * hd:(q0 + qr) * hd], w["o"][:, q0 * hd:(q0 + qr) * hd]
    wk, wv = w["k"][k0 * hd:(k0 + kl) * hd], w["v"][k0 * hd:(k0 + kl) * hd]
    a = _attention(x @ wq.t() + bq, x @ wk.t(), x @ wv.t(), ql, kl, hd)
    return a @ wo.t()


def _tp_rank_main(rank, world, port, case):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name, T, nh, nkv, hd, H, I, V = case
    torch.manual_seed(0)
    M = 5
    x = torch.randn(M, H)
    w = dict(q=torch.randn(nh * hd, H) / 8, qb=torch.randn(nh * hd) / 8, k=torch.randn(nkv * hd, H) / 8, v=torch.randn(nkv * hd, H) / 8,
             o=torch.randn(H, nh * hd) / 8, g=torch.randn(I, H) / 8, u=torch.randn(I, H) / 8, d=torch.randn(H, I) / 8,
             lm=torch.randn(V, H) / 8)
    cfg_kw = dict(num_layers=1, hidden_size=H, num_heads=nh, num_kv_heads=nkv, head_dim=hd, intermediate_size=I, vocab_size=V, tp_degree=T)
    mine = [r for r in range(T) if r % world == rank]                 # this process plays TP ranks rank, rank + world, ...
    plans = {r: _plan_of(cfg_kw, r) for r in mine}
    # C1: all-reduce of the row-parallel o_proj partials
    o_part = sum(_rank_partials(x, w, plans[r], hd) for r in mine)
    dist.all_reduce(o_part)
    h = x + o_part
    d_part = torch.zeros(M, H)
    for r in mine:
        p = plans[r]
        sl = slice(p.inter0, p.inter0 + p.inter_local)
        d_part += (torch.nn.functional.silu(h @ w["g"][sl].t()) * (h @ w["u"][sl].t())) @ w["d"][:, sl].t()
    dist.all_reduce(d_part)
    h = h + d_part
    # C2: vocabulary-parallel logits, every rank's slice into its own columns
    logits = torch.zeros(M, V)
    for r in mine:
        p = plans[r]
        logits[:, p.vocab0:p.vocab0 + p.vocab_local] = h @ w["lm"][p.vocab0:p.vocab0 + p.vocab_local].t()
    dist.all_reduce(logits)                                            # (disjoint columns: a sum is a gather)
    ok = torch.allclose(logits, _block_unsharded(x, w, nh, nkv, hd), rtol=1e-4, atol=1e-4)
    # every q head is computed exactly once over the ranks, every kv head by all the ranks that need it
    heads = []
    for r in mine:
        heads += list(range(plans[r].q_head0, plans[r].q_head0 + plans[r].q_heads_real))
    all_heads = [None] * world
    dist.all_gather_object(all_heads, heads)
    covered = sorted(h for hs in all_heads for h in hs) == list(range(nh))
    res = [None] * world
    dist.all_gather_object(res, bool(ok and covered))
    dist.destroy_process_group()
    if rank == 0:
        assert all(res), (name, res)


@pytest.mark.parametrize("case", [
    ("llama-like, kv heads divide", 2, 8, 2, 16, 64, 96, 128),
    ("kv head replicated on 2 ranks each", 4, 8, 2, 16, 64, 96, 128),
    ("Qwen2.5-7B head counts at TP 8: 28 q / 4 kv -> 4 + 3 (+1 zero-weight padding head) per kv head", 8, 28, 4, 8, 64, 128, 256),
])
def test_tp_plan_two_ranks_gloo(case):
    """SURVEY 8e on PRODUCT code: every rank's slices come from the library's own sharding plan (mi_tp_plan = what
    mi_ctx_create / route_matrix apply), the partial sums cross a real all-reduce (gloo, world_size 2: each process plays
    half of the TP ranks), and the result equals the unsharded block -- including head counts that do not divide
    (reference: tp_degree = tensor_parallel_size, loader.py:752-753; no divisibility check, platform.py:58-64)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_tp_rank_main, args=(2, port, case), nprocs=2, join=True)


def test_tp_plan_values_and_errors():
    p = [_plan_of(dict(num_heads=28, num_kv_heads=4, head_dim=128, intermediate_size=18944, vocab_size=152064, tp_degree=8), r) for r in range(8)]
    assert [(q.q_head0, q.q_heads_real, q.q_heads_local, q.kv_head0) for q in p] == [
        (0, 4, 4, 0), (4, 3, 4, 0), (7, 4, 4, 1), (11, 3, 4, 1), (14, 4, 4, 2), (18, 3, 4, 2), (21, 4, 4, 3), (25, 3, 4, 3)]
    assert all(q.kv_heads_local == 1 and q.inter_local == 2368 and q.vocab_local == 19008 for q in p)
    assert [q.inter0 for q in p] == [2368 * r for r in range(8)]
    q = _plan_of(dict(num_heads=64, num_kv_heads=8, head_dim=128, intermediate_size=28672, vocab_size=128256, tp_degree=8), 5)   # Llama-3.3-70B
    assert (q.q_head0, q.q_heads_real, q.q_heads_local, q.kv_head0, q.kv_heads_local, q.vocab0) == (40, 8, 8, 5, 1, 5 * 16032)
    with pytest.raises(ValueError):
        _plan_of(dict(num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=14336, vocab_size=128256, tp_degree=3), 0)
    with pytest.raises(ValueError):
        _plan_of(dict(num_heads=32, num_kv_heads=8, head_dim=128, intermediate_size=14336, vocab_size=128256, tp_degree=2), 2)


# ---- on-device sampling rule (oracle/sampling.py) -------------------------------------------------
def test_sampling_rule_properties():
    import numpy as np
    from oracle import sampling as osamp
    rng = np.random.default_rng(0)
    logits = (rng.standard_normal(300) * 2).astype(np.float32)
    logits[[5, 17]] = logits.max() + 1                      # a tie at the top
    assert osamp.sample_row(logits, 1, 0.5, 0.7, seed=3, row=0) == 5          # greedy: first maximum
    idx, w = osamp.kept_distribution(logits, 10, 1.0, 1.0)
    assert idx.tolist()[:2] == [5, 17] and len(idx) == 10 and np.all(np.diff(w) <= 0)
    # the nucleus shrinks with top_p and always keeps the head
    sizes = [len(osamp.kept_distribution(logits, 50, p, 1.0)[0]) for p in (1e-6, 0.3, 0.6, 0.9, 1.0)]
    assert sizes[0] == 1 and sizes == sorted(sizes) and sizes[-1] == 50
    # top_k is capped at 256 and at the vocabulary
    assert len(osamp.kept_distribution(logits, 10_000, 1.0, 1.0)[0]) == 256
    # draws are a pure function of (seed, row), stay inside the nucleus, and cover it
    seen = set()
    for seed in range(400):
        t = osamp.sample_row(logits, 4, 1.0, 2.0, seed, 1)
        assert t == osamp.sample_row(logits, 4, 1.0, 2.0, seed, 1)
        seen.add(t)
    assert seen == set(osamp.kept_distribution(logits, 4, 1.0, 2.0)[0].tolist())
    u = [float(osamp.uniform(s, r)) for s in range(50) for r in range(4)]
    assert 0.0 <= min(u) and max(u) < 1.0 and 0.35 < sum(u) / len(u) < 0.65


def test_on_device_sampling_config_is_accepted_and_routes_to_tokens():
    """The adapter returns ids (not logits) when on_device_sampling_config is set, packs the
    reference's (top_k, top_p, temperature) rows and advances the draw counter per call."""
    import torch
    from vllm_neuron_amd.worker import mi355x_model_loader as loader

    class FakeNative:
        def __init__(self):
            self.calls = []

        def forward_tokens(self, ids, pos, seq, bt, sm, full, comp, sampling_params=None, seed=0):
            self.calls.append((sampling_params.clone(), seed))
            return torch.arange(ids.shape[0], dtype=torch.int64) + 100

        def forward(self, *a, **k):
            raise AssertionError("logits path taken")

    m = loader.MI355XCausalLM(config=None)
    m.model = FakeNative()
    m.mi355x_config = loader.MI355XConfig(is_block_kv_layout=True, on_device_sampling_config={"dynamic": True})
    m._sample_seed = 7
    B = 3
    params = torch.tensor([[1.0, 1.0, 1.0], [20.0, 0.9, 0.8], [256.0, 1.0, 1.0]])
    kw = dict(input_ids=torch.zeros(B, 1, dtype=torch.long), position_ids=torch.zeros(B, 1, dtype=torch.long),
              input_block_ids=torch.arange(B), slot_mapping=torch.zeros(B, 1, dtype=torch.long),
              block_tables=torch.ones(B, 8, dtype=torch.long), full_context_lens=torch.ones(B, 1, dtype=torch.long),
              computed_context_lens=torch.zeros(B, 1, dtype=torch.long), sampling_params=params)
    out = m(**kw)
    out2 = m(**kw)
    assert out.tolist() == [100, 101, 102] and out2.tolist() == out.tolist()
    (p1, s1), (p2, s2) = m.model.calls
    assert torch.equal(p1, params) and s1 == (7 << 32) + 1 and s2 == (7 << 32) + 2
    so = m.sample(logits=out)
    assert so.sampled_token_ids.shape == (B, 1)


def test_installed_layout_imports_without_the_checkout(tmp_path):
    """`setup.py` must ship the real package (package_dir onto `vllm-neuron_amd/`) and
    the HIP library, not the in-tree import stub.  Build the package files into a scratch directory
    (no hipcc run: the in-tree library is reused) and import from there with the repository out of
    sys.path: the entry point resolves and `_native` finds its library inside the installed package."""
    import subprocess
    import sys
    env = dict(os.environ, MI355X_SKIP_HIP_BUILD="1")
    build_lib = tmp_path / "lib"
    r = subprocess.run([sys.executable, "setup.py", "-q", "build_py", "--build-lib", str(build_lib)], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    pkg = build_lib / "vllm_neuron_amd"
    assert (pkg / "csrc" / "libmi355x_vllm.so").exists() and (pkg / "worker" / "mi355x_worker.py").exists()
    code = ("import sys; sys.path = [p for p in sys.path if p not in ('', %r)]; sys.path.insert(0, %r);"
            "import vllm_neuron_amd, vllm_neuron_amd._native as n, vllm_neuron_amd.platform as p;"
            "assert vllm_neuron_amd.__file__.startswith(%r), vllm_neuron_amd.__file__;"
            "assert n.LIB_PATH.startswith(%r) and n.load_library().mi_version() >= 3;"
            "print(vllm_neuron_amd.PLATFORM_QUALNAME)") % (ROOT, str(build_lib), str(build_lib), str(build_lib))
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0 and "MI355XPlatform" in r.stdout, r.stderr[-2000:]
