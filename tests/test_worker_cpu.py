"""MI355XWorker on the CPU, with the model runner replaced by a recorder -- the method set and
behaviour the reference's worker tests pin (/root/reference/test/unit/worker/test_neuron_worker.py:
281-700), plus what is specific to this worker: the choice of the tensor-parallel devices and the
truthful KV sizing."""
from types import SimpleNamespace
from unittest.mock import MagicMock

import pytest
import torch

from vllm_neuron_amd import platform as plat
from vllm_neuron_amd._vllm_compat import (SimpleCacheConfig, SimpleDeviceConfig, SimpleModelConfig,
                                          SimpleParallelConfig, SimpleSchedulerConfig, SimpleVllmConfig)
from vllm_neuron_amd.worker.mi355x_worker import MI355XWorker


def _cfg(tp=1):
    hf = SimpleNamespace(architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=512, hidden_size=256,
                         intermediate_size=512, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
                         head_dim=32, rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=None, tie_word_embeddings=False)
    c = SimpleVllmConfig(
        model_config=SimpleModelConfig(model="", hf_config=hf, max_model_len=256),
        cache_config=SimpleCacheConfig(block_size=32, enable_prefix_caching=True),
        parallel_config=SimpleParallelConfig(tensor_parallel_size=tp),
        scheduler_config=SimpleSchedulerConfig(max_num_seqs=4, max_model_len=256),
        device_config=SimpleDeviceConfig("cpu"))
    plat.MI355XPlatform.check_and_update_config(c)
    return c


@pytest.fixture
def worker():
    w = MI355XWorker(_cfg(), local_rank=0, rank=0, distributed_init_method="", is_driver_worker=True)
    w.model_runner = MagicMock()
    return w


def test_worker_initialization_and_method_set(worker):          # reference :281-311
    assert worker.device == "cpu" and worker.model_config is not None and worker.tp_device_ids == [0]
    for name in ("load_model", "execute_model", "init_device", "initialize_cache", "initialize_from_config",
                 "get_kv_cache_spec", "determine_available_memory", "compile_or_warm_up_model", "check_health",
                 "take_draft_token_ids", "get_supported_tasks"):
        assert callable(getattr(worker, name)), name
    assert worker.compile_or_warm_up_model() is None and worker.check_health() is None
    assert worker.get_supported_tasks() == ["generate"]             # reference :663-681


def test_execute_model_driver_and_non_driver(worker):             # reference :339-360
    out = object()
    worker.model_runner.execute_model.return_value = out
    so = object()
    assert worker.execute_model(so) is out
    worker.model_runner.execute_model.assert_called_once_with(so)
    worker.is_driver_worker = False
    assert worker.execute_model(so) is None


def test_cache_and_model_plumbing(worker):                        # reference :362-421, :644-661
    worker.initialize_cache(100, 7)
    assert worker.cache_config.num_gpu_blocks == 100 and worker.cache_config.num_cpu_blocks == 7
    worker.load_model()
    worker.model_runner.load_model.assert_called_once_with()
    spec = {"layer": object()}
    worker.model_runner.get_kv_cache_spec.return_value = spec
    assert worker.get_kv_cache_spec() is spec
    kvc = object()
    worker.initialize_from_config(kvc)
    worker.model_runner.initialize_kv_cache.assert_called_once_with(kvc)
    worker.model_runner.take_draft_token_ids.return_value = None
    assert worker.take_draft_token_ids() is None


def test_determine_available_memory_is_what_the_pool_can_hold(worker):   # reference :313-337 (fallback 20 GiB)
    native = MagicMock()
    native.kv_stats.return_value = {"device_free_bytes": 100 * 2 ** 30, "workspace_bytes": 5 * 2 ** 30}
    native.kv_bytes_per_block.return_value = 2 * 2 * 32 * 64 * 2 * 4          # 4 layers of K + V, 32 tokens, 2 heads x 32
    worker.model_runner.model.model = native
    worker.model_runner.model.draft = None
    one_layer = 2 * 32 * 64 * 2
    worker.model_runner.get_kv_cache_spec.return_value = {"layer": SimpleNamespace(page_size_bytes=one_layer)}
    got = worker.determine_available_memory()
    blocks = (int(100 * 2 ** 30 * 0.95) - 5 * 2 ** 30) // native.kv_bytes_per_block.return_value
    assert got == blocks * one_layer                                  # vLLM divides by the one-layer page: `blocks` again
    # a draft model (fused speculation) keeps a second pool under the same block ids
    draft = MagicMock()
    draft.kv_bytes_per_block.return_value = native.kv_bytes_per_block.return_value // 4
    draft.kv_stats.return_value = {"workspace_bytes": 2 ** 30}
    worker.model_runner.model.draft = draft
    blocks2 = (int(100 * 2 ** 30 * 0.95) - 6 * 2 ** 30) // (native.kv_bytes_per_block.return_value * 5 // 4)
    assert worker.determine_available_memory() == blocks2 * one_layer
    # no statistics -> the reference's fallback
    native.kv_stats.side_effect = RuntimeError("no device")
    assert worker.determine_available_memory() == 20 * 2 ** 30


def test_lora_and_unsupported_operations(worker):                 # reference :480-526
    assert worker.list_loras() == set()
    for call in (lambda: worker.add_lora(object()), lambda: worker.remove_lora(1), lambda: worker.pin_lora(1),
                 worker.get_model):
        with pytest.raises(NotImplementedError):
            call()


def test_init_device_builds_the_runner_with_the_tensor_parallel_devices(monkeypatch):   # reference :610-630, :683-700
    seen = {}

    class Runner:
        def __init__(self, vllm_config, device, device_id, tp_device_ids):
            seen.update(device=device, device_id=device_id, tp=tp_device_ids)
    import vllm_neuron_amd.worker.mi355x_model_runner as mr
    monkeypatch.setattr(mr, "MI355XModelRunner", Runner)
    monkeypatch.setenv("MI355X_TP_DEVICES", "3,1")
    w = MI355XWorker(_cfg(tp=2), local_rank=0, rank=0, distributed_init_method="", is_driver_worker=True)
    assert w.tp_device_ids == [3, 1]
    w.init_device()
    assert isinstance(w.model_runner, Runner) and seen == {"device": "cpu", "device_id": 0, "tp": [3, 1]}
    monkeypatch.setenv("MI355X_TP_DEVICES", "0,1,2")
    with pytest.raises(ValueError, match="lists 3 devices"):
        MI355XWorker(_cfg(tp=2), local_rank=0, rank=0, distributed_init_method="", is_driver_worker=True)
    # fewer GPUs visible than ranks: an error that names the two ways out, or every shard on one GPU
    monkeypatch.delenv("MI355X_TP_DEVICES")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="MI355X_TP_LOOPBACK"):
        MI355XWorker(_cfg(tp=4), local_rank=0, rank=0, distributed_init_method="", is_driver_worker=True)
    monkeypatch.setenv("MI355X_TP_LOOPBACK", "1")
    assert MI355XWorker(_cfg(tp=4), local_rank=0, rank=0, distributed_init_method="", is_driver_worker=True).tp_device_ids == [0] * 4
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert MI355XWorker(_cfg(tp=4), local_rank=2, rank=0, distributed_init_method="", is_driver_worker=True).tp_device_ids == [2, 3, 4, 5]
