"""Drop-in boundary behaviour on CPU (no GPU, no vLLM): registration, platform config
rewrites, scheduler policy, loader config math, runner bookkeeping.  The scenarios and the
known-answer tables are the ones the reference's own unit tests pin
(/root/reference/test/unit/: test_init.py, test_platform.py, core/test_scheduler.py,
worker/test_model_loader.py, worker/test_model_runner.py — cited per test)."""

import os
import warnings
from types import SimpleNamespace

import pytest
import torch

import vllm_neuron_amd
from vllm_neuron_amd import platform as plat
from vllm_neuron_amd._vllm_compat import (CachedRequestData, KVCacheConfig, NewRequestData, Request,
                                          RequestStatus, SamplerOutput, SamplingParams,
                                          SchedulerOutput, SimpleCacheConfig, SimpleModelConfig,
                                          SimpleParallelConfig, SimpleSchedulerConfig, SimpleVllmConfig)
from vllm_neuron_amd.core.scheduler import (ContinuousBatchingMI355XScheduler,
                                            check_stop_with_min_tokens)
from vllm_neuron_amd.worker import mi355x_model_loader as loader
from vllm_neuron_amd.worker.mi355x_model_runner import MI355XModelRunner


def hf_cfg(**kw):
    base = dict(architectures=["LlamaForCausalLM"], num_key_value_heads=2, head_dim=64, vocab_size=512,
                num_attention_heads=8, hidden_size=256, num_hidden_layers=2, intermediate_size=512,
                rms_norm_eps=1e-5, rope_theta=10000.0, model_type="llama")
    base.update(kw)
    return SimpleNamespace(**base)


def vcfg(max_model_len=256, max_num_seqs=4, block_size=32, prefix=True, override=None, tp=1, **kw):
    return SimpleVllmConfig(
        model_config=SimpleModelConfig(model="m", hf_config=hf_cfg(), dtype="bfloat16", max_model_len=max_model_len),
        cache_config=SimpleCacheConfig(block_size=block_size, num_gpu_blocks_override=override,
                                       enable_prefix_caching=prefix),
        parallel_config=SimpleParallelConfig(tensor_parallel_size=tp),
        scheduler_config=SimpleSchedulerConfig(max_num_seqs=max_num_seqs, max_model_len=max_model_len), **kw)


# ---- registration (reference test/unit/test_init.py:9-41) -----------------------------------
def test_register_without_device(monkeypatch):
    monkeypatch.setattr(vllm_neuron_amd, "_is_mi355x_dev", lambda: False)
    with pytest.warns(UserWarning, match="Skipping MI355X plugin registration"):
        assert vllm_neuron_amd.register() is None


def test_register_with_device(monkeypatch):
    monkeypatch.setattr(vllm_neuron_amd, "_is_mi355x_dev", lambda: True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert vllm_neuron_amd.register() == "vllm_neuron_amd.platform.MI355XPlatform"
    mod, cls = vllm_neuron_amd.register().rsplit(".", 1)
    assert getattr(__import__(mod, fromlist=[cls]), cls) is plat.MI355XPlatform


# ---- platform (reference test/unit/test_platform.py) ------------------------------------------
def test_platform_attrs():                                      # :20-36
    p = plat.MI355XPlatform()
    assert p.device_type == "cpu" and p.device_name == "cpu"
    assert p.is_out_of_tree() if hasattr(p, "is_out_of_tree") else True
    assert "fbgemm_fp8" in p.supported_quantization and "mi355x_quant" in p.supported_quantization
    assert p.get_device_name() == "mi355x"
    assert p.is_async_output_supported(None) is False
    assert p.is_pin_memory_available() is False and p.use_all_gather() is True
    assert p.supports_v1(None) is True


def test_worker_and_scheduler_cls_rewrite():                    # :156-195
    c = vcfg()
    plat.MI355XPlatform.check_and_update_config(c)
    assert c.parallel_config.worker_cls == plat.WORKER_CLS
    assert c.scheduler_config.scheduler_cls == plat.SCHEDULER_CLS
    assert c.scheduler_config.chunked_prefill_enabled is False
    assert c.scheduler_config.max_num_batched_tokens == 131072        # :531-549
    c2 = vcfg()
    c2.parallel_config.worker_cls = "my.Worker"
    plat.MI355XPlatform.check_and_update_config(c2)
    assert c2.parallel_config.worker_cls == "my.Worker"


def test_block_size_and_null_block():                           # :198-222, :673-762
    c = vcfg(max_model_len=2048, prefix=False, block_size=None, override=1)
    plat.MI355XPlatform.check_and_update_config(c)
    assert c.cache_config.block_size == 2048
    assert c.cache_config.num_gpu_blocks_override == 2
    plat.MI355XPlatform.check_and_update_config(c)               # idempotent per CacheConfig instance
    assert c.cache_config.num_gpu_blocks_override == 2
    c.cache_config = SimpleCacheConfig(block_size=32, num_gpu_blocks_override=5, enable_prefix_caching=True)
    plat.MI355XPlatform.check_and_update_config(c)               # a NEW CacheConfig is adjusted again
    assert c.cache_config.num_gpu_blocks_override == 6
    assert c.cache_config.block_size == 32


def test_uni_executor_and_defaults():                           # :225-248, :531-549
    c = vcfg(tp=8, max_num_seqs=None)
    plat.MI355XPlatform.check_and_update_config(c)
    assert c.parallel_config.distributed_executor_backend == "uni"
    assert c.scheduler_config.max_num_seqs == 32


def test_missing_block_size_asserts(monkeypatch):               # :280-306, :414-428
    c = vcfg(block_size=None, prefix=True)
    with pytest.raises(AssertionError, match="block_size must be set"):
        plat.MI355XPlatform.check_and_update_config(c)
    monkeypatch.setenv("DISABLE_MI355X_CUSTOM_SCHEDULER", "1")
    c = vcfg(block_size=None)
    with pytest.raises(AssertionError, match="block_size must be set"):
        plat.MI355XPlatform.check_and_update_config(c)


def test_empty_vllm_config_is_tolerated():                      # platform.py:142-144
    c = vcfg()
    c.model_config = None
    plat.MI355XPlatform.check_and_update_config(c)
    assert c.parallel_config.worker_cls == "auto"


# ---- scheduler (reference test/unit/core/test_scheduler.py) -----------------------------------
def make_sched(max_num_seqs=4, num_blocks=64):
    c = vcfg(max_num_seqs=max_num_seqs)
    plat.MI355XPlatform.check_and_update_config(c)
    return ContinuousBatchingMI355XScheduler(c, KVCacheConfig(num_blocks=num_blocks))


def req(i, n=5, **sp):
    return Request(f"r{i}", list(range(1, n + 1)), SamplingParams(temperature=0.0, **{"max_tokens": 8, **sp}),
                   eos_token_id=2)


def fake_output(sched_out, tok=7):
    ids = [r.req_id for r in sched_out.scheduled_new_reqs] + list(sched_out.scheduled_cached_reqs.req_ids)
    return SimpleNamespace(req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)},
                           sampled_token_ids=[[tok] for _ in ids])


def test_one_prompt_per_step_then_decode():                     # :404-422, :382-402
    s = make_sched()
    for i in range(3):
        s.add_request(req(i))
    seen = []
    for _ in range(3):
        out = s.schedule()
        assert len(out.scheduled_new_reqs) == 1 and not out.scheduled_cached_reqs.req_ids   # prefill XOR decode
        assert len(s.holdback_queue) == 0
        seen.append(out.scheduled_new_reqs[0].req_id)
        s.update_from_output(out, fake_output(out))
    assert seen == ["r0", "r1", "r2"]
    out = s.schedule()
    assert not out.scheduled_new_reqs and sorted(out.scheduled_cached_reqs.req_ids) == ["r0", "r1", "r2"]


def test_capacity_limit():                                      # :147-164
    s = make_sched(max_num_seqs=2)
    for i in range(3):
        s.add_request(req(i))
    for _ in range(2):
        out = s.schedule()
        s.update_from_output(out, fake_output(out))
    out = s.schedule()                                           # batch is full: r2 stays waiting, decode runs
    assert not out.scheduled_new_reqs and len(out.scheduled_cached_reqs.req_ids) == 2
    assert [r.request_id for r in s.waiting] == ["r2"]


def test_min_tokens_beats_eos_and_stop_tokens():                # :256-365
    r = req(0, min_tokens=3, stop_token_ids=[9])
    r.append_output_token_ids(2)                                 # EOS as first token
    assert check_stop_with_min_tokens(r, 100) is False
    r.append_output_token_ids(9)
    assert check_stop_with_min_tokens(r, 100) is False
    r.append_output_token_ids(9)
    assert check_stop_with_min_tokens(r, 100) is True and r.status == RequestStatus.FINISHED_STOPPED
    assert r.stop_reason == 9
    r2 = req(1)
    r2.append_output_token_ids(2)
    assert check_stop_with_min_tokens(r2, 100) is True
    r3 = req(2, ignore_eos=True)
    r3.append_output_token_ids(2)
    assert check_stop_with_min_tokens(r3, 100) is False


def test_length_cap():                                          # :424-444
    r = req(0, max_tokens=2)
    r.append_output_token_ids(5)
    assert check_stop_with_min_tokens(r, 100) is False
    r.append_output_token_ids(5)
    assert check_stop_with_min_tokens(r, 100) is True and r.status == RequestStatus.FINISHED_LENGTH_CAPPED
    r = req(1, n=9, max_tokens=50)
    r.append_output_token_ids(5)
    assert check_stop_with_min_tokens(r, 10) is True             # num_tokens >= max_model_len


def test_scheduler_stop_trims_tokens():
    s = make_sched()
    s.add_request(req(0, max_tokens=1))
    out = s.schedule()
    outs = s.update_from_output(out, fake_output(out))
    assert outs[0].finished and outs[0].new_token_ids == [7]
    assert not s.has_unfinished_requests()
    assert "r0" in s.schedule().finished_req_ids


# ---- loader config math (reference test/unit/worker/test_model_loader.py) -----------------------
def loader_cfgs(max_model_len=256, max_num_seqs=4, block_size=32, override=None, prefix=True):
    c = vcfg(max_model_len, max_num_seqs, block_size, prefix, override)
    return c.model_config, c.cache_config, c.parallel_config, c.scheduler_config


def test_default_config_fields():                               # :535-588
    m, cache, par, sch = loader_cfgs()
    d = loader._get_default_mi355x_config(m, cache, par, sch, None, None)
    assert d["tp_degree"] == 1 and d["ctx_batch_size"] == 1 and d["batch_size"] == 4
    assert d["max_context_length"] == 256 and d["seq_len"] == 256 and d["enable_bucketing"] is True
    assert d["is_continuous_batching"] is True and d["quantized"] is False and d["torch_dtype"] == "bfloat16"
    assert d["padding_side"] == "right" and d["pa_block_size"] == 32
    assert d["pa_num_blocks"] == (256 // 32) * 4
    assert d["is_block_kv_layout"] is True and d["is_prefix_caching"] is True
    assert d["on_device_sampling_config"] is None               # CPU sampling is the default (parity) path


@pytest.mark.parametrize("max_model_len,max_num_seqs,block_size,is_block,pa_num_blocks,ok", [
    (2048, 32, 16, True, 2000, False), (1024, 16, 8, True, 3000, True), (512, 8, 4, True, 1024, True),
    (1000, 10, 16, True, 630, True), (1000, 10, 16, True, 620, False), (4096, 64, 8, False, 100, True)])
def test_sufficient_blocks_validation(max_model_len, max_num_seqs, block_size, is_block, pa_num_blocks, ok):
    """Known-answer table of reference test_model_loader.py:2831-2842 (min blocks use ceil)."""
    _, cache, _, sch = loader_cfgs(max_model_len, max_num_seqs, block_size, prefix=False)
    cfg = {"is_block_kv_layout": is_block, "pa_num_blocks": pa_num_blocks}
    if ok:
        assert loader._validate_mi355x_config(cache, sch, cfg) is cfg
    else:
        with pytest.raises(AssertionError, match="blocks are required"):
            loader._validate_mi355x_config(cache, sch, cfg)


@pytest.mark.parametrize("override,pa_num_blocks,matching,sufficient", [
    (201, None, True, True), (200, None, True, False), (201, 200, True, True), (200, 199, True, False),
    (201, 150, False, None), (201, 201, False, None), (None, 200, False, None)])
def test_pa_num_blocks_matrix(override, pa_num_blocks, matching, sufficient):
    """Known-answer matrix of reference test_model_loader.py:2915-2949: exactly 200 blocks are
    required (max_model_len 1600 / block_size 8, one sequence)."""
    m, cache, par, sch = loader_cfgs(1600, 1, 8, override)
    ov = {"is_block_kv_layout": True}
    if pa_num_blocks is not None:
        ov["pa_num_blocks"] = pa_num_blocks

    def run():
        cfg = loader._get_mi355x_config_after_override(
            loader._get_default_mi355x_config(m, cache, par, sch, None, None), dict(ov))
        cfg = loader._handle_pa_num_blocks(cache, cfg, ov)
        return loader._validate_mi355x_config(cache, sch, cfg)

    if matching and sufficient:
        assert run()["pa_num_blocks"] == override
    elif not matching:
        with pytest.raises(ValueError, match="pa_num_blocks"):
            run()
    else:
        with pytest.raises(AssertionError):
            run()


def test_override_merge_and_quant_keys():                       # loader.py:870-900
    d = {"batch_size": 4, "quantized": False}
    out = loader._get_mi355x_config_after_override(d, {"quantized": True, "skip_warmup": True,
                                                      "text_neuron_config": {}, "vision_neuron_config": {}})
    assert out["quantized"] is True and out["quantization_dtype"] == "int8"
    assert out["quantization_type"] == "per_tensor_symmetric" and out["quantized_checkpoints_path"] is None
    assert out["skip_warmup"] is True and "text_neuron_config" not in out and "vision_neuron_config" not in out
    out = loader._get_mi355x_config_after_override({"a": 1}, {"quantized": True, "quantization_dtype": "f8e4m3",
                                                            "quantization_type": "per_channel_symmetric"})
    assert out["quantization_dtype"] == "f8e4m3" and out["quantization_type"] == "per_channel_symmetric"
    assert loader._get_mi355x_config_after_override({"a": 1}, None) == {"a": 1}


def test_get_model_configs_errors():                            # loader.py:612-631
    assert loader._get_model_configs(hf_cfg()) == ("LlamaForCausalLM", 2, 64)
    assert loader._get_model_configs(hf_cfg(head_dim=None)) == ("LlamaForCausalLM", 2, 32)
    with pytest.raises(ValueError, match="No architectures"):
        loader._get_model_configs(hf_cfg(architectures=[]))
    with pytest.raises(ValueError, match="Missing required fields"):
        loader._get_model_configs(hf_cfg(num_key_value_heads=None))
    with pytest.raises(ValueError, match="is not supported on MI355X"):
        loader._check_architecture("GPT2LMHeadModel")


def test_override_key_alias():
    assert loader.get_override_config({"override_neuron_config": {"a": 1}}) == {"a": 1}
    assert loader.get_override_config({"override_mi355x_config": {"b": 2}, "override_neuron_config": {"a": 1}}) == {"b": 2}
    assert loader.get_override_config({}) is None and loader.get_override_config(None) is None


def test_sort_inputs_and_reordered():                           # :1245-1274, :1334-1356, :1656-1681
    base = loader.MI355XModelBase(hf_cfg())
    ids = torch.tensor([2, 0, 1])
    inputs = dict(input_ids=torch.tensor([[20], [0], [10]]), names=["c", "a", "b"], empty=torch.tensor([]),
                  other=torch.zeros(5, 1), scalar=3)
    base.is_reorder_needed = True
    with base._reordered(ids, **inputs) as (sorted_ids, srt, restore):
        assert sorted_ids.tolist() == [0, 1, 2]
        assert srt["input_ids"].flatten().tolist() == [0, 10, 20] and srt["names"] == ["a", "b", "c"]
        assert srt["empty"].numel() == 0 and srt["other"].shape == (5, 1) and srt["scalar"] == 3
        assert restore(srt["input_ids"]).flatten().tolist() == [20, 0, 10]
    base.is_reorder_needed = False
    with base._reordered(ids, **inputs) as (same_ids, same, restore):
        assert same_ids is ids and same["input_ids"] is inputs["input_ids"]
        assert restore(inputs["input_ids"]) is inputs["input_ids"]


# ---- runner bookkeeping (reference test/unit/worker/test_model_runner.py) ------------------------
class FakeModel:
    """Stands in for MI355XCausalLM: records the kwargs, returns logits that make token
    (10 + row) the argmax."""

    def __init__(self, vocab=512, prefix=True):
        self.mi355x_config = loader.MI355XConfig(
            is_block_kv_layout=prefix, is_prefix_caching=prefix, chunked_prefill_config=None,
            on_device_sampling_config=None, attn_tkg_nki_kernel_enabled=False,
            attn_block_tkg_nki_kernel_enabled=False)
        self.num_key_value_heads, self.head_dim, self.vocab = 2, 64, vocab
        self.calls = []
        self.is_reorder_needed = False
        self.model = SimpleNamespace(finalize=lambda: None, set_num_blocks=lambda n: None)

    def __call__(self, **kw):
        self.calls.append(kw)
        pcs = kw.get("prefill_completion_state")
        B = len(pcs) if pcs is not None else kw["input_ids"].shape[0]     # chunked prefill: one row per request
        logits = torch.zeros(B, self.vocab)
        logits[torch.arange(B), 10 + torch.arange(B)] = 5.0
        return logits

    def sample(self, logits):
        raise RuntimeError("not used")


def make_runner(prefix=True, **kw):
    c = vcfg(prefix=prefix, **kw)
    plat.MI355XPlatform.check_and_update_config(c)
    r = MI355XModelRunner(c, "cpu")
    r.model = FakeModel(prefix=prefix)
    r.is_block_kv_layout = r.is_prefix_caching = prefix
    r._kv_ready = True
    return r


def sched_out(new=(), cached=None, finished=()):
    cached = cached or CachedRequestData()
    nst = {n.req_id: len(n.prompt_token_ids) - n.num_computed_tokens for n in new}
    nst.update({r: 1 for r in cached.req_ids})
    return SchedulerOutput(scheduled_new_reqs=list(new), scheduled_cached_reqs=cached, num_scheduled_tokens=nst,
                           total_num_scheduled_tokens=sum(nst.values()), finished_req_ids=set(finished))


def new_req(rid, prompt, blocks, computed=0, **sp):
    return NewRequestData(req_id=rid, prompt_token_ids=prompt, mm_features=[],
                          sampling_params=SamplingParams(temperature=0.0, **sp), pooling_params=None,
                          block_ids=(list(blocks),), num_computed_tokens=computed)


def test_prefill_input_contract_with_prefix_hit():              # reference runner.py:704-763, 853-885
    from tests.helpers import prefill_inputs
    r = make_runner()
    prompt = list(range(100, 170))
    out = r.execute_model(sched_out([new_req("a", prompt, [3, 9, 4], computed=64)]))
    kw = r.model.calls[0]
    want = prefill_inputs(prompt, [3, 9, 4], 32, 256, 64)
    assert torch.equal(kw["input_ids"], want["input_ids"]) and torch.equal(kw["position_ids"], want["position_ids"])
    assert torch.equal(kw["block_tables"], want["block_table"])
    assert torch.equal(kw["slot_mapping"], want["slot_mapping"])
    assert kw["full_context_lens"].tolist() == [[70]] and kw["computed_context_lens"].tolist() == [[64]]
    assert kw["input_block_ids"].tolist() == [r.vllm_req_to_seq_id_mapping["a"]]
    assert out.sampled_token_ids == [[10]] and out.req_ids == ["a"]
    assert r.requests["a"].output_token_ids == [10]


def test_decode_input_contract_and_block_append():              # runner.py:765-832; tests :821-872
    from tests.helpers import decode_inputs
    r = make_runner()
    r.execute_model(sched_out([new_req("a", list(range(31)), [5])]))
    r.execute_model(sched_out([new_req("b", list(range(40)), [6, 7])]))
    cached = CachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[False, False], new_token_ids=[[], []],
                               new_block_ids=[([8],), None], num_computed_tokens=[31, 40])
    out = r.execute_model(sched_out(cached=cached))
    kw = r.model.calls[-1]
    want = decode_inputs([10, 10], [31, 40], [[5, 8], [6, 7]], 32, 256)
    for k_mine, k_ref in (("input_ids", "input_ids"), ("position_ids", "position_ids"),
                          ("block_tables", "block_table"), ("slot_mapping", "slot_mapping"),
                          ("full_context_lens", "full_context_lens"),
                          ("computed_context_lens", "computed_context_lens")):
        assert torch.equal(kw[k_mine], want[k_ref]), k_mine
    assert r.requests["a"].block_ids == ([5, 8],)                 # appended, not replaced
    # rows come back in model order (a, b) and are re-ordered to the persistent batch's order
    got = {rid: out.sampled_token_ids[out.req_id_to_index[rid]] for rid in out.req_ids}
    assert got == {"a": [10], "b": [11]}
    # resumed from preemption: block ids are REPLACED (tests :984-1018)
    cached = CachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[True, False], new_token_ids=[[], []],
                               new_block_ids=[([20, 21],), None], num_computed_tokens=[32, 41])
    r.execute_model(sched_out(cached=cached))
    assert r.requests["a"].block_ids == ([20, 21],)
    # ... and the TENSOR handed to the model follows, although the new list is as long as the old
    # one (the persistent row used to be rewritten only on a length change)
    kw = r.model.calls[-1]
    assert kw["block_tables"][0, :3].tolist() == [20, 21, 0] and kw["block_tables"][1, :3].tolist() == [6, 7, 0]
    assert kw["slot_mapping"][0].tolist() == [21 * 32 + 0]         # position 32 -> second block of the new list
    # same length again, not resumed: the row is kept (and stays right)
    cached = CachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[False, False], new_token_ids=[[], []],
                               new_block_ids=[None, None], num_computed_tokens=[33, 42])
    r.execute_model(sched_out(cached=cached))
    assert r.model.calls[-1]["block_tables"][0, :3].tolist() == [20, 21, 0]


def test_incremental_decode_inputs_equal_the_rebuilt_ones():
    """The persistent token-generation inputs (block-table rows rewritten only when a request moves
    into the row or a block is appended) are, field for field, what the reference's per-step rebuild
    produces (runner.py:765-832, 887-917) -- over requests joining, finishing and changing rows."""
    r = make_runner()
    r.execute_model(sched_out([new_req("a", list(range(31)), [5])]))
    r.execute_model(sched_out([new_req("b", list(range(40)), [6, 7])]))
    r.execute_model(sched_out([new_req("c", list(range(70)), [9, 10, 11])]))

    def step(ids, new_blocks, computed, finished=()):
        cached = CachedRequestData(req_ids=ids, resumed_from_preemption=[False] * len(ids), new_token_ids=[[]] * len(ids),
                                   new_block_ids=new_blocks, num_computed_tokens=computed)
        so = sched_out(cached=cached, finished=list(finished))
        r._update_states(so)
        fast = r._prepare_decode_inputs_incremental(so)
        data, is_prefill = r._prepare_continuous_batching_inputs(so)
        slow = r._finalize_continuous_batching_inputs(data, is_prefill)
        assert not is_prefill and fast.request_ids == slow.request_ids
        for f in ("input_tokens", "position_ids", "input_block_ids", "slot_mapping", "block_tables",
                  "full_context_lens", "computed_context_lens"):
            assert torch.equal(getattr(fast, f), getattr(slow, f)), f
        # feed the sampled token back like execute_model does
        for rid in ids:
            r.requests[rid].output_token_ids.append(7)

    step(["a", "b", "c"], [([8],), None, None], [31, 40, 70])            # a crosses into a second block
    step(["a", "b", "c"], [None, None, None], [32, 41, 71])              # nothing changes: rows are kept
    step(["a", "c"], [None, None], [33, 72], finished=["b"])             # c moves up a row
    r.execute_model(sched_out([new_req("d", list(range(10)), [12])]))    # a prefill in between
    step(["a", "c", "d"], [None, None, None], [34, 73, 10])              # d takes a fresh row


def _chunked_runner():
    r = make_runner()
    r.is_chunked_prefill = True
    r.use_custom_seq_id_mapping = False
    r.cache_config.block_size = 8
    r.model.mi355x_config.chunked_prefill_config = SimpleNamespace(max_num_seqs=4)
    return r


def test_chunked_prefill_new_request_chunk():                   # reference test_model_runner.py:1469-1510
    r = _chunked_runner()
    so = SchedulerOutput(scheduled_new_reqs=[new_req("req1", [1, 2, 3, 4, 5], [0, 1, 2])],
                         scheduled_cached_reqs=CachedRequestData(), num_scheduled_tokens={"req1": 3},
                         total_num_scheduled_tokens=3, finished_req_ids=set())
    data = r._prepare_chunked_prefill_inputs(so)
    assert data.request_ids == ["req1"] and data.input_tokens == [1, 2, 3]      # the first 3 tokens
    assert data.position_ids == [0, 1, 2] and data.slot_mapping == [0, 1, 2]
    assert data.full_context_lens == [3] and data.computed_context_lens == [0]
    assert data.prefill_completion_state == [False]                             # more chunks remain


def test_chunked_prefill_cached_requests_and_finalize():        # reference runner.py:964-1051
    """A ragged step: the next chunk of a long prompt + a request that generates.  (The reference's
    KAT at test_model_runner.py:1252-1290 feeds a state no scheduler produces -- output tokens on a
    prompt that is not encoded yet -- and expects 3 tokens for 2 scheduled; here a chunk always has
    exactly num_scheduled_tokens tokens, taken from prompt + outputs at positions start .. end-1.)"""
    r = _chunked_runner()
    r.requests = {
        "long": SimpleNamespace(prompt_token_ids=list(range(100, 120)), output_token_ids=[], block_ids=([3, 4, 5],),
                                sampling_params=SamplingParams(temperature=0.0)),
        "gen": SimpleNamespace(prompt_token_ids=[1, 2, 3, 4, 5, 6, 7], output_token_ids=[8, 9], block_ids=([0, 1],),
                               sampling_params=SamplingParams(temperature=0.0)),
    }
    cached = CachedRequestData(req_ids=["long", "gen"], resumed_from_preemption=[False, False], new_token_ids=[[], []],
                               new_block_ids=[None, None], num_computed_tokens=[8, 8])
    so = SchedulerOutput(scheduled_new_reqs=[], scheduled_cached_reqs=cached, num_scheduled_tokens={"long": 12, "gen": 1},
                         total_num_scheduled_tokens=13, finished_req_ids=set())
    data = r._prepare_chunked_prefill_inputs(so)
    assert data.input_tokens == list(range(108, 120)) + [9]                      # the last output token is fed back
    assert data.position_ids == list(range(8, 20)) + [8]
    assert data.slot_mapping == [4 * 8 + i for i in range(8)] + [5 * 8 + i for i in range(4)] + [1 * 8 + 0]
    assert data.full_context_lens == [20, 9] and data.computed_context_lens == [8, 8]
    assert data.prefill_completion_state == [True, True]
    m = r._finalize_chunked_prefill_inputs(data)
    assert m.input_tokens.shape == (1, 13) and m.position_ids.shape == (1, 13) and m.slot_mapping.shape == (13,)
    assert m.block_tables.tolist() == [[3, 4, 5], [0, 1, 0]]                     # padded with block 0
    assert m.prefill_completion_state.dtype == torch.bool and m.input_block_ids.tolist() == [0]
    assert m.full_context_lens.tolist() == [20, 9] and m.sampling_params is None  # CPU sampling: rows not built


def test_chunked_prefill_incomplete_rows_yield_no_token():      # reference runner.py:1060-1063
    r = _chunked_runner()
    r.execute_model(SchedulerOutput(
        scheduled_new_reqs=[new_req("p", list(range(30)), [0, 1, 2, 3]), new_req("q", [5, 6, 7], [4])],
        scheduled_cached_reqs=CachedRequestData(), num_scheduled_tokens={"p": 16, "q": 3},
        total_num_scheduled_tokens=19, finished_req_ids=set()))
    assert r.model.calls[-1]["prefill_completion_state"].tolist() == [False, True]
    assert r.requests["p"].output_token_ids == [] and len(r.requests["q"].output_token_ids) == 1


def test_finished_requests_free_seq_ids():                      # tests :683-718, :1020-1049
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    slot = r.vllm_req_to_seq_id_mapping["a"]
    assert slot not in r.free_seq_ids
    out = r.execute_model(sched_out(finished=["a"]))
    assert out.sampled_token_ids == [] and slot in r.free_seq_ids
    assert "a" not in r.requests and "a" not in r.vllm_req_to_seq_id_mapping
    assert r.input_batch.req_ids == []


def test_minus_one_pads_are_stripped():                         # tests :773-819, :922-956
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    r.execute_model(sched_out([new_req("b", [1, 2], [2])]))
    cached = CachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[False, False], new_token_ids=[[], []],
                               new_block_ids=[None, None], num_computed_tokens=[3, 2])
    r.execute_model(sched_out(cached=cached))                    # both requests are in the batch again
    first = r.input_batch.req_ids[0]
    n_before = len(r.requests[first].output_token_ids)
    out = r._generate_model_runner_output(SamplerOutput(sampled_token_ids=torch.tensor([[0], [-1]])))
    assert out.sampled_token_ids == [[0], []]                    # 0 is a token, -1 a pad
    assert r.requests[first].output_token_ids[-1] == 0 and len(r.requests[first].output_token_ids) == n_before + 1
    assert r._generate_model_runner_output(None).sampled_token_ids == []


def test_greedy_sampling_params_rewrite():                      # tests :1373-1418
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    r.requests["a"].sampling_params = SamplingParams(temperature=0.0, top_k=50, top_p=0.9)
    p = r.get_mi355x_sampling_params(torch.zeros(1, 1))
    assert p.shape == (1, 3) and p[0].tolist() == pytest.approx([1.0, 0.9, 1.0])
    r.requests["a"].sampling_params = SamplingParams(temperature=0.7, top_k=0, top_p=1.0)
    assert r.get_mi355x_sampling_params(torch.zeros(1, 1))[0].tolist() == pytest.approx([256.0, 1.0, 0.7])


def test_cpu_sampling_validation_errors():                      # reference test_cpu_sampling.py:99-118, 253-276
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    mi = SimpleNamespace(request_ids=["a"])
    with pytest.raises(RuntimeError, match="CPU sampling failed.*2D tensor"):
        r._cpu_sample(torch.zeros(1, 1, 512), mi)
    with pytest.raises(RuntimeError, match="does not match model vocab size"):
        r._cpu_sample(torch.zeros(1, 100), mi)
    calls = []
    r.cpu_sampler = lambda logits, md: calls.append((logits, md)) or SamplerOutput(torch.tensor([[3]]))
    assert r._cpu_sample(torch.zeros(1, 512), mi).sampled_token_ids.tolist() == [[3]]
    assert len(calls) == 1 and calls[0][1] is r.input_batch.sampling_metadata


def test_contiguous_kv_mode_inputs():
    """Prefix caching off: block_size = max_model_len, no slot / block tensors (runner.py:715-724)."""
    r = make_runner(prefix=False, block_size=None)
    assert r.block_size == 256
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    kw = r.model.calls[0]
    assert kw["block_tables"].numel() == 0 and kw["slot_mapping"].numel() == 0
    assert kw["full_context_lens"].tolist() == [[3]]
    spec = r.get_kv_cache_spec()["layer"]
    assert (spec.block_size, spec.num_kv_heads, spec.head_size, spec.dtype) == (256, 2, 64, torch.bfloat16)


def test_unsupported_features_raise():
    r = make_runner()
    with pytest.raises(NotImplementedError):
        r.execute_model(sched_out([NewRequestData("m", [1], ["img"], SamplingParams(), None, ([1],), 0)]))
    c = vcfg()
    c.lora_config = object()
    rr = MI355XModelRunner(c, "cpu")
    with pytest.raises(NotImplementedError, match="Multi-lora"):
        rr.load_model()


# ---- fused speculation (reference loader.py:243-334, 785-791; runner.py:293-345, 488-498, 825-830) -------------
def test_remask_fused_spec_output_known_answer():               # reference test_model_loader.py:1175-1210
    fused = [torch.tensor([[1, 2, 0], [3, 0, 0]]),               # accepted tokens, 0-padded (0 is ALSO a real token id)
             torch.tensor([[5], [4]])]                           # next position ids
    inputs = {"position_ids": torch.tensor([[2], [3]])}
    masked = loader.MI355XCausalLM._remask_fused_spec_output(None, fused, inputs)
    assert masked.tolist() == [[1, 2, 0], [3, -1, -1]]           # row 0 generated 3 tokens (the 0 is one), row 1 one
    # counts are clamped to [0, T]; a 1-D next-position vector is accepted too
    fused = [torch.tensor([[7, 8], [9, 9], [4, 0]]), torch.tensor([9, 3, 6])]
    inputs = {"position_ids": torch.tensor([[0, 1, 2], [1, 2, 3], [3, 4, 5]])}
    assert loader.MI355XCausalLM._remask_fused_spec_output(None, fused, inputs).tolist() == [[7, 8], [-1, -1], [4, -1]]


def test_speculative_defaults_and_eagle_is_rejected():           # reference loader.py:785-791
    from vllm_neuron_amd._vllm_compat import SimpleSpeculativeConfig
    c = vcfg()
    spec = SimpleSpeculativeConfig(num_speculative_tokens=4)
    d = loader._get_default_mi355x_config(c.model_config, c.cache_config, c.parallel_config, c.scheduler_config, None, spec)
    assert d["enable_fused_speculation"] is True and d["speculation_length"] == 4 and "enable_eagle_speculation" not in d
    d = loader._get_default_mi355x_config(c.model_config, c.cache_config, c.parallel_config, c.scheduler_config, None,
                                          SimpleSpeculativeConfig(num_speculative_tokens=2, method="eagle"))
    assert d["enable_eagle_speculation"] is True
    d = loader._get_default_mi355x_config(c.model_config, c.cache_config, c.parallel_config, c.scheduler_config, None, None)
    assert "enable_fused_speculation" not in d
    with pytest.raises(NotImplementedError, match="EAGLE"):
        loader.get_mi355x_model(c.model_config, c.cache_config, c.parallel_config, c.scheduler_config, None,
                                speculative_config=SimpleSpeculativeConfig(method="eagle"))


def test_runner_output_with_speculative_config():                # reference test_model_runner.py:720-770
    from vllm_neuron_amd._vllm_compat import SimpleSpeculativeConfig
    r = make_runner()
    r.speculative_config = SimpleSpeculativeConfig(num_speculative_tokens=3)
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [1])]))
    r.execute_model(sched_out([new_req("b", [1, 2], [2])]))
    cached = CachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[False, False], new_token_ids=[[], []],
                               new_block_ids=[None, None], num_computed_tokens=[3, 2])
    r.execute_model(sched_out(cached=cached))
    ids = r.input_batch.req_ids
    before = {rid: list(r.requests[rid].output_token_ids) for rid in ids}
    # [B, T, 1] as the fused step returns it: row 0 generated 3 tokens (one of them id 0), row 1 one token
    out = r._generate_model_runner_output(SamplerOutput(sampled_token_ids=torch.tensor([[[5], [0], [7]], [[9], [-1], [-1]]])))
    assert out.sampled_token_ids == [[5, 0, 7], [9]]
    assert r.spec_token_ids == [[5, 0], []]                      # all but the last generated token of each row
    assert r.requests[ids[0]].output_token_ids == before[ids[0]] + [5, 0, 7]
    assert r.requests[ids[1]].output_token_ids == before[ids[1]] + [9]
    row = r.input_batch.req_id_to_index[ids[0]]
    n = r.input_batch.num_tokens_no_spec[row]
    assert r.input_batch.token_ids_cpu[row, n - 3:n].tolist() == [5, 0, 7] and r.input_batch.num_tokens[row] == n
    # the next token-generation step feeds the LAST token at the position behind everything generated
    r.execute_model(sched_out(cached=cached))
    kw = r.model.calls[-1]
    a = cached.req_ids.index(ids[0])                             # rows of the call follow the scheduler's order
    pos = len(r.requests[ids[0]].prompt_token_ids) + len(before[ids[0]]) + 3 - 1
    assert kw["position_ids"][a].tolist() == [pos] and kw["input_ids"][a].tolist() == [7]
    # slots of the speculation window: looked up per position; beyond the owned blocks: the pad
    bs = r.cache_config.block_size
    blk = r.requests[ids[0]].block_ids[0]
    assert kw["slot_mapping"].shape[1] == 3
    assert kw["slot_mapping"][a].tolist() == [blk[p // bs] * bs + p % bs if p // bs < len(blk) else -1
                                              for p in range(pos, pos + 3)]


def test_update_states_with_scheduled_spec_tokens():             # reference test_model_runner.py:1163-1215
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3, 4, 5], [1])]))
    cached = CachedRequestData(req_ids=["a"], resumed_from_preemption=[False], new_token_ids=[[]],
                               new_block_ids=[None], num_computed_tokens=[5])
    so = sched_out(cached=cached)
    so.scheduled_spec_decode_tokens = {"a": [10, 11, 12]}
    row = r.input_batch.req_id_to_index["a"]
    n = r.input_batch.num_tokens_no_spec[row]
    r._update_states(so)
    assert r.input_batch.token_ids_cpu[row, n:n + 3].tolist() == [10, 11, 12]
    assert r.input_batch.num_tokens[row] == n + 3 and r.input_batch.num_tokens_no_spec[row] == n


def test_scheduler_reserves_blocks_for_the_speculation_window():
    """vLLM's num_lookahead_tokens: a running request owns blocks for the positions a speculation
    step may write; several tokens per step are appended and the stop rule trims them."""
    from vllm_neuron_amd._vllm_compat import KVCacheConfig, Request, SimpleSpeculativeConfig
    from vllm_neuron_amd.core.scheduler import ContinuousBatchingMI355XScheduler
    c = vcfg(block_size=32, max_model_len=256)
    c.speculative_config = SimpleSpeculativeConfig(num_speculative_tokens=4)
    plat.MI355XPlatform.check_and_update_config(c)
    s = ContinuousBatchingMI355XScheduler(c, KVCacheConfig(num_blocks=33))
    s.add_request(Request("a", list(range(30)), SamplingParams(temperature=0.0, max_tokens=6), eos_token_id=None))
    out = s.schedule()
    assert len(out.scheduled_new_reqs[0].block_ids[0]) == 1                       # 30 prompt tokens: one block
    mro = SimpleNamespace(req_id_to_index={"a": 0}, sampled_token_ids=[[100]])
    s.update_from_output(out, mro)
    out = s.schedule()                                                             # 31 tokens + 4 ahead -> a second block
    assert out.scheduled_cached_reqs.new_block_ids[0] is not None and len(s.requests["a"].block_ids) == 2
    mro = SimpleNamespace(req_id_to_index={"a": 0}, sampled_token_ids=[[101, 102, 103, 104]])
    res = s.update_from_output(out, mro)
    assert res[0].new_token_ids == [101, 102, 103, 104] and not res[0].finished
    out = s.schedule()
    mro = SimpleNamespace(req_id_to_index={"a": 0}, sampled_token_ids=[[105, 106, 107]])
    res = s.update_from_output(out, mro)                                            # max_tokens 6: the window is trimmed
    assert res[0].new_token_ids == [105] and res[0].finished


def test_adapter_fused_speculation_bookkeeping():
    """MI355XCausalLM with fused speculation on, the native library replaced by a recorder: context
    encoding runs target AND draft and returns the first token in column 0; token generation is ONE
    forward_spec call; after a step that generated all k tokens the second-to-last of them is handed
    to the draft's next step (draft_catchup_ids), once, and only for that sequence; a step with a request
    that samples falls back to one token per sequence (reference loader.py:349-355, 308-333)."""
    k = 3

    class Native:
        def __init__(self):
            self.calls = []

        def forward_tokens(self, ids, *a, **kw):
            self.calls.append(("tokens", ids.shape))
            return torch.full((ids.shape[0],), 41, dtype=torch.long)

        def forward_spec(self, draft, ids, pos, bt, kk, catchup_ids=None):
            self.calls.append(("spec", ids.tolist(), pos.tolist(), catchup_ids.tolist()))
            acc = torch.tensor([[7, 8, 9], [5, 0, 0]])[:ids.shape[0]]             # row 0: all k accepted, row 1: one token
            nxt = pos + torch.tensor([3, 1])[:ids.shape[0]]
            return acc, nxt

    m = loader.MI355XCausalLM(SimpleNamespace(vocab_size=512))
    m.mi355x_config = loader.MI355XConfig(is_block_kv_layout=True, is_prefix_caching=True, chunked_prefill_config=None,
                                          on_device_sampling_config={"dynamic": True}, enable_fused_speculation=True,
                                          speculation_length=k)
    m.model, m.draft, m._draft_catchup = Native(), Native(), {}
    greedy = torch.tensor([[1.0, 1.0, 1.0]])
    # context encoding of one prompt (sequence id 3)
    out = m.forward(torch.tensor([[11, 12, 13, 14]]), torch.tensor([3]), position_ids=torch.arange(4)[None],
                    slot_mapping=torch.arange(4)[None], block_tables=torch.tensor([[1, 0]]),
                    full_context_lens=torch.tensor([[4]]), computed_context_lens=torch.tensor([[0]]),
                    sampling_params=greedy, prefill_completion_state=None)
    assert out.tolist() == [[41, -1, -1]]
    assert [c[0] for c in m.model.calls] == ["tokens"] and [c[0] for c in m.draft.calls] == ["tokens"]
    # token generation for sequences 3 and 5
    kw = dict(slot_mapping=torch.zeros(2, k, dtype=torch.long), block_tables=torch.tensor([[1, 0], [2, 0]]),
              full_context_lens=torch.tensor([[5], [9]]), computed_context_lens=torch.tensor([[4], [8]]),
              sampling_params=greedy.repeat(2, 1), prefill_completion_state=None)
    out = m.forward(torch.tensor([[41], [77]]), torch.tensor([3, 5]), position_ids=torch.tensor([[4], [8]]), **kw)
    assert out.tolist() == [[7, 8, 9], [5, -1, -1]]
    assert m.model.calls[-1] == ("spec", [41, 77], [4, 8], [-1, -1])
    assert m._draft_catchup == {3: (7, 8)}                     # sequence 3 continues at position 7; the draft has not seen token 8
    out = m.forward(torch.tensor([[9], [5]]), torch.tensor([3, 5]), position_ids=torch.tensor([[7], [9]]), **kw)
    assert m.model.calls[-1] == ("spec", [9, 5], [7, 9], [8, -1])
    # a catch-up that does not match the position (the request was replaced) is dropped
    m._draft_catchup = {5: (99, 1)}
    m.forward(torch.tensor([[9], [5]]), torch.tensor([3, 5]), position_ids=torch.tensor([[10], [10]]), **kw)
    assert m.model.calls[-1][3] == [-1, -1]
    # a step with a request that samples: one token per sequence from the ordinary sampler, both models fed
    n_t, n_d = len(m.model.calls), len(m.draft.calls)
    out = m.forward(torch.tensor([[9]]), torch.tensor([3]), position_ids=torch.tensor([[13]]),
                    slot_mapping=torch.zeros(1, k, dtype=torch.long), block_tables=torch.tensor([[1, 0]]),
                    full_context_lens=torch.tensor([[14]]), computed_context_lens=torch.tensor([[13]]),
                    sampling_params=torch.tensor([[20.0, 0.9, 0.8]]), prefill_completion_state=None)
    assert out.tolist() == [[41, -1, -1]]
    assert m.model.calls[n_t:] == [("tokens", torch.Size([1, 1]))] and m.draft.calls[n_d:] == [("tokens", torch.Size([1, 1]))]


def test_check_stop_no_condition_and_pooling_params():          # reference test_scheduler.py:598-660
    r = req(0, max_tokens=20)
    for t in (1, 3, 4, 5):
        r.append_output_token_ids(t)
    assert check_stop_with_min_tokens(r, 100) is False and not r.is_finished()
    # pooling requests stop when (and only when) a pooler output arrives
    r = req(1, max_tokens=100)
    r.pooling_params = object()
    assert check_stop_with_min_tokens(r, 100, pooler_output=None) is False and not r.is_finished()
    assert check_stop_with_min_tokens(r, 100, pooler_output=torch.tensor([1.0])) is True
    assert r.status == RequestStatus.FINISHED_STOPPED


def test_model_config_overrides_with_a_stand_in_vllm(monkeypatch):   # reference test_platform.py:251-530, 552-650
    """vLLM is not installed here; a stand-in `vllm.config.ModelConfig` receives the overrides the
    platform installs at registration and they are exercised the way the reference's tests do:
    head-count divisibility no longer checked, external launcher needs a seed, expert parallelism
    delegates, pipeline parallelism checks the model and switches async output off, quantization /
    cuda-graph verification are no-ops, the user's max_model_len is trusted, applying twice is harmless."""
    import sys
    import types

    class ModelConfig:
        def verify_with_parallel_config(self, parallel_config):
            raise AssertionError("upstream verifier: head count must divide the TP degree")

        def _verify_quantization(self):
            raise AssertionError("upstream quantization verifier")

        def _verify_cuda_graph(self):
            raise AssertionError("upstream cuda-graph verifier")

        def get_and_verify_max_len(self, max_model_len):
            raise AssertionError("upstream max_model_len verifier")
    vllm = types.ModuleType("vllm")
    vllm_config = types.ModuleType("vllm.config")
    vllm_config.ModelConfig = ModelConfig
    vllm.config = vllm_config
    monkeypatch.setitem(sys.modules, "vllm", vllm)
    monkeypatch.setitem(sys.modules, "vllm.config", vllm_config)
    monkeypatch.setattr(plat.MI355XPlatform, "_config_overrides_applied", False)
    plat.MI355XPlatform.pre_register_and_update(parser=object())
    assert plat.MI355XPlatform._config_overrides_applied
    plat.MI355XPlatform.pre_register_and_update(None)              # second call: nothing to do

    mc = ModelConfig()
    mc.seed, mc.spec_target_max_model_len, mc.use_async_output_proc = 0, None, True
    pc = SimpleNamespace(distributed_executor_backend="uni", enable_expert_parallel=False, pipeline_parallel_size=1)
    mc.verify_with_parallel_config(pc)                             # 28 heads at TP 8 would pass: nothing is checked
    mc._verify_quantization()
    mc._verify_cuda_graph()
    assert mc.get_and_verify_max_len(4096) == 4096
    mc.spec_target_max_model_len = 1024
    assert mc.get_and_verify_max_len(4096) == 1024
    # external launcher: the seed must be set
    pc.distributed_executor_backend = "external_launcher"
    mc.seed = None
    with pytest.raises(AssertionError, match="Seed must be set"):
        mc.verify_with_parallel_config(pc)
    mc.seed = 1
    mc.verify_with_parallel_config(pc)
    # expert parallelism delegates to the upstream check
    calls = []
    mc._verify_with_expert_parallelism = lambda: calls.append("ep")
    pc.enable_expert_parallel = True
    mc.verify_with_parallel_config(pc)
    assert calls == ["ep"]
    # pipeline parallelism: model support is required; async output processing is switched off
    pc.enable_expert_parallel, pc.pipeline_parallel_size = False, 2
    mc.architectures = ["LlamaForCausalLM"]
    mc.registry = SimpleNamespace(is_pp_supported_model=lambda archs: False)
    with pytest.raises(NotImplementedError, match="Pipeline parallelism is not supported"):
        mc.verify_with_parallel_config(pc)
    mc.registry = SimpleNamespace(is_pp_supported_model=lambda archs: True)
    mc.verify_with_parallel_config(pc)
    assert mc.use_async_output_proc is False


def test_model_base_and_sample_contract():                      # reference test_model_loader.py: base-class NotImplemented, sample()
    base = loader.MI355XModelBase(SimpleNamespace())
    for call in (lambda: base.forward(None, None, None, None), lambda: base.sample(torch.zeros(1, 4)),
                 lambda: base.load_weights("", "LlamaForCausalLM")):
        with pytest.raises(NotImplementedError):
            call()
    m = loader.MI355XCausalLM(SimpleNamespace())
    m.mi355x_config = loader.MI355XConfig(on_device_sampling_config={"dynamic": True})
    out = m.sample(torch.tensor([5, 9]))                         # on-device sampling: the "logits" ARE the sampled ids
    assert out.sampled_token_ids.tolist() == [[5], [9]] and out.logprobs_tensors is None
    m.mi355x_config = loader.MI355XConfig(on_device_sampling_config=None)
    with pytest.raises(RuntimeError, match="CPU sampling should be handled by the model runner"):
        m.sample(torch.zeros(2, 8))


def test_artifact_directory_rules(tmp_path, monkeypatch):       # reference loader.py:160-226, 888-891
    geo = {"num_layers": 2, "hidden_size": 256}
    f = loader.MI355XCausalLM._artifact_dir
    monkeypatch.delenv("MI355X_COMPILED_ARTIFACTS", raising=False)
    # synthetic / in-memory weights are not cached unless a path is named
    assert f("", {"synthetic_weights": {"seed": 1}}, geo, True, "f8e4m3", "per_channel_symmetric", 1) is None
    assert f("", {"state_dict": {}}, geo, False, "int8", "per_tensor_symmetric", 1) is None
    # quantized_checkpoints_path wins (quantized models only), then the environment variable
    q = {"quantized_checkpoints_path": str(tmp_path / "q"), "state_dict": {}}
    assert f("", q, geo, True, "f8e4m3", "per_channel_symmetric", 1) == str(tmp_path / "q")
    assert f("", q, geo, False, "f8e4m3", "per_channel_symmetric", 1) is None          # not quantized: the key is ignored
    monkeypatch.setenv("MI355X_COMPILED_ARTIFACTS", str(tmp_path / "env"))
    assert f("", {"state_dict": {}}, geo, False, "int8", "per_tensor_symmetric", 1) == str(tmp_path / "env")
    monkeypatch.delenv("MI355X_COMPILED_ARTIFACTS")
    # a local checkpoint directory: <model>/mi355x-compiled-artifacts/<md5 of the configuration>
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    (ckpt / "model.safetensors").write_bytes(b"x" * 10)
    a = f(str(ckpt), {}, geo, True, "f8e4m3", "per_channel_symmetric", 1)
    assert a.startswith(str(ckpt / "mi355x-compiled-artifacts")) and len(os.path.basename(a)) == 32
    assert f(str(ckpt), {}, geo, True, "f8e4m3", "per_channel_symmetric", 1) == a      # stable
    assert f(str(ckpt), {}, geo, True, "int8", "per_channel_symmetric", 1) != a        # another configuration
    assert f(str(ckpt), {}, geo, True, "f8e4m3", "per_channel_symmetric", 8) != a      # another TP degree
    assert f(str(ckpt), {"modules_to_not_convert": ["lm_head"]}, geo, True, "f8e4m3", "per_channel_symmetric", 1) != a
    (ckpt / "model.safetensors").write_bytes(b"x" * 11)                                 # another checkpoint
    assert f(str(ckpt), {}, geo, True, "f8e4m3", "per_channel_symmetric", 1) != a
    assert f(str(tmp_path / "missing"), {}, geo, True, "f8e4m3", "per_channel_symmetric", 1) is None


def test_artifact_checkpoint_identity(tmp_path):                # base / instruct of one shape must not share artifacts
    cls = loader.MI355XCausalLM
    ident = cls._checkpoint_identity
    assert ident("", {"state_dict": {}}) is None                                   # nothing to compare with: the artifacts are all there is
    a = {"w": torch.arange(8, dtype=torch.float32).reshape(2, 4)}
    b = {"w": torch.arange(8, dtype=torch.float32).reshape(2, 4) + 1}
    assert ident("", {"state_dict": a}) == ident("", {"state_dict": {"w": a["w"].clone()}}) != ident("", {"state_dict": b})
    assert ident("", {"synthetic_weights": {"seed": 3}}) != ident("", {"synthetic_weights": {"seed": 4}})
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    (ckpt / "model.safetensors").write_bytes(b"x" * 10)
    i1 = ident(str(ckpt), {})
    (ckpt / "model.safetensors").write_bytes(b"x" * 11)
    assert i1 != ident(str(ckpt), {}) and i1["files"][0][:2] == ["model.safetensors", 10]
    art = tmp_path / "art"
    cls._check_artifact_identity(str(art), i1)                                      # no directory yet: load_artifacts reports that
    art.mkdir()
    with pytest.raises(ValueError, match="does not say"):
        cls._check_artifact_identity(str(art), i1)
    cls._write_artifact_identity(str(art), i1)
    cls._check_artifact_identity(str(art), i1)
    cls._check_artifact_identity(str(art), None)
    with pytest.raises(ValueError, match="another checkpoint"):
        cls._check_artifact_identity(str(art), ident(str(ckpt), {}))


def test_update_states_resume_after_preemption_and_kv_init():   # reference test_model_runner.py:823-873 (complex), :569-590
    r = make_runner()
    r.execute_model(sched_out([new_req("a", [1, 2, 3], [4, 5])]))
    assert r.requests["a"].block_ids == ([4, 5],)
    # preempted and resumed: the scheduler hands a NEW block list, which replaces the old one
    cached = CachedRequestData(req_ids=["a"], resumed_from_preemption=[True], new_token_ids=[[]],
                               new_block_ids=[([9, 8, 7],)], num_computed_tokens=[3])
    r.execute_model(sched_out(cached=cached))
    assert r.requests["a"].block_ids == ([9, 8, 7],)
    kw = r.model.calls[-1]
    assert kw["block_tables"][0, :3].tolist() == [9, 8, 7]       # the decode inputs follow the new blocks
    pos = len(r.requests["a"].prompt_token_ids) + len(r.requests["a"].output_token_ids) - 2
    assert kw["slot_mapping"][0].tolist() == [9 * 32 + pos]
    # not resumed, nothing appended: the blocks stay
    cached = CachedRequestData(req_ids=["a"], resumed_from_preemption=[False], new_token_ids=[[]],
                               new_block_ids=[None], num_computed_tokens=[4])
    r.execute_model(sched_out(cached=cached))
    assert r.requests["a"].block_ids == ([9, 8, 7],)
    # KV cache spec and initialisation (the library owns the pool: sized once, when vLLM has decided)
    spec = r.get_kv_cache_spec()["layer"]
    assert (spec.block_size, spec.num_kv_heads, spec.head_size, spec.dtype) == (32, 2, 64, torch.bfloat16)
    calls = []
    r.model.model = SimpleNamespace(finalize=lambda: calls.append("finalize"), set_num_blocks=lambda n: calls.append(("blocks", n)))
    r._kv_ready = False
    r.initialize_kv_cache(SimpleNamespace(num_blocks=77))
    r.initialize_kv_cache(SimpleNamespace(num_blocks=99))        # second call: nothing
    assert calls == [("blocks", 77), "finalize"]
