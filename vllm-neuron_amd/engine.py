# SPDX-License-Identifier: Apache-2.0
"""Standalone engine loop: the plugin's scheduler + worker driven without vLLM's EngineCore.

Used where vLLM itself is not installed (this build image, bench.py, the tests): it performs
exactly the calls vLLM's engine performs on a platform plugin — `check_and_update_config`,
worker bring-up (`init_device`, `load_model`, `get_kv_cache_spec`, `determine_available_memory`,
`initialize_from_config`), then `schedule()` -> `execute_model()` -> `update_from_output()`
per step — so the same plugin code paths are exercised and timed.
"""

from __future__ import annotations

import gc
import time
from dataclasses import dataclass, field

from ._vllm_compat import (HAVE_VLLM, KVCacheConfig, Request, SamplingParams, Scheduler, SimpleCacheConfig,
                           SimpleDeviceConfig, SimpleModelConfig, SimpleParallelConfig,
                           SimpleSchedulerConfig, SimpleVllmConfig)
from .core.scheduler import ContinuousBatchingMI355XScheduler
from .platform import MI355XPlatform
from .worker.mi355x_worker import MI355XWorker


@dataclass
class RequestOutput:
    request_id: str
    prompt_token_ids: list
    token_ids: list = field(default_factory=list)
    finished: bool = False
    ttft_s: float | None = None          # arrival -> first token
    arrival: float = 0.0
    num_cached_tokens: int = 0


class MI355XEngine:
    def __init__(self, hf_config, model: str = "", *, max_model_len=2048, max_num_seqs=4, block_size=32,
                 num_gpu_blocks_override=None, enable_prefix_caching=True, tensor_parallel_size=1,
                 dtype="bfloat16", override_mi355x_config: dict | None = None, seed=0, local_rank=0,
                 enable_chunked_prefill=False, max_num_batched_tokens=None, speculative_config=None):
        if HAVE_VLLM:  # pragma: no cover
            raise RuntimeError("vLLM is installed: use vllm.LLM(...) — the plugin registers itself")
        cfg = SimpleVllmConfig(
            model_config=SimpleModelConfig(model=model, hf_config=hf_config, dtype=dtype,
                                           max_model_len=max_model_len, seed=seed),
            cache_config=SimpleCacheConfig(block_size=block_size, num_gpu_blocks_override=num_gpu_blocks_override,
                                           enable_prefix_caching=enable_prefix_caching),
            parallel_config=SimpleParallelConfig(tensor_parallel_size=tensor_parallel_size),
            scheduler_config=SimpleSchedulerConfig(max_num_seqs=max_num_seqs, max_model_len=max_model_len,
                                                   chunked_prefill_enabled=bool(enable_chunked_prefill),
                                                   max_num_batched_tokens=int(max_num_batched_tokens or 131072)),
            device_config=SimpleDeviceConfig("cpu"), speculative_config=speculative_config,
            additional_config={"override_mi355x_config": dict(override_mi355x_config or {})})
        if enable_chunked_prefill:
            # vLLM's native scheduler (the plugin's override is off: DISABLE_MI355X_CUSTOM_SCHEDULER=1,
            # reference platform.py:131-175); `_vllm_compat.Scheduler` stands in for it here
            import os
            prev = os.environ.get("DISABLE_MI355X_CUSTOM_SCHEDULER")
            os.environ["DISABLE_MI355X_CUSTOM_SCHEDULER"] = "1"
            try:
                MI355XPlatform.check_and_update_config(cfg)
            finally:
                if prev is None:
                    del os.environ["DISABLE_MI355X_CUSTOM_SCHEDULER"]
                else:
                    os.environ["DISABLE_MI355X_CUSTOM_SCHEDULER"] = prev
        else:
            MI355XPlatform.check_and_update_config(cfg)
        self.vllm_config = cfg
        # ONE worker whatever tensor_parallel_size is (uni executor): the library context behind it
        # drives every GPU of the tensor-parallel group
        self.worker = MI355XWorker(cfg, local_rank=local_rank, rank=0, distributed_init_method="",
                                   is_driver_worker=True)
        self.worker.init_device()
        self.worker.load_model()
        spec = self.worker.get_kv_cache_spec()["layer"]
        if cfg.cache_config.num_gpu_blocks_override is not None:
            num_blocks = cfg.cache_config.num_gpu_blocks_override
        else:
            # what vLLM does: as many single-layer pages as fit in what the worker reports, capped
            # for the standalone harness at the minimum the validator demands + the null block
            need = -(-max_model_len // cfg.cache_config.block_size) * max_num_seqs + 1
            fit = self.worker.determine_available_memory() // spec.page_size_bytes
            num_blocks = max(2, min(fit, need))
        self.worker.initialize_cache(num_blocks, 0)
        self.worker.initialize_from_config(KVCacheConfig(num_blocks=num_blocks))
        sched_cls = Scheduler if enable_chunked_prefill else ContinuousBatchingMI355XScheduler
        self.scheduler = sched_cls(cfg, KVCacheConfig(num_blocks=num_blocks))
        self.outputs: dict[str, RequestOutput] = {}
        self._next_id = 0
        # What vLLM's engine core does once start-up is over: move everything allocated so far
        # (torch, the model wrappers, config objects) out of the collector's reach, so a full
        # collection in the serving loop does not walk it (measured here: 40-80 ms stalls on
        # every third request, i.e. several TTFTs).
        gc.collect()
        gc.freeze()

    def add_request(self, prompt_token_ids, sampling_params: SamplingParams | None = None, eos_token_id=None,
                    request_id: str | None = None) -> str:
        rid = request_id or f"req-{self._next_id}"
        self._next_id += 1
        sp = sampling_params or SamplingParams(temperature=0.0, max_tokens=16)
        now = time.perf_counter()
        self.scheduler.add_request(Request(rid, prompt_token_ids, sp, eos_token_id=eos_token_id, arrival_time=now))
        self.outputs[rid] = RequestOutput(rid, list(prompt_token_ids), arrival=now)
        return rid

    def has_unfinished_requests(self) -> bool:
        return self.scheduler.has_unfinished_requests()

    def step(self):
        sched_out = self.scheduler.schedule()
        runner_out = self.worker.execute_model(sched_out)
        now = time.perf_counter()
        for new_req in sched_out.scheduled_new_reqs:
            self.outputs[new_req.req_id].num_cached_tokens = new_req.num_computed_tokens
        for out in self.scheduler.update_from_output(sched_out, runner_out):
            ro = self.outputs[out.request_id]
            if out.new_token_ids and ro.ttft_s is None:
                ro.ttft_s = now - ro.arrival
            ro.token_ids.extend(out.new_token_ids)
            ro.finished = ro.finished or out.finished
        return sched_out, runner_out

    def generate(self, prompts: list[list[int]], sampling_params: SamplingParams | None = None,
                 eos_token_id=None) -> list[RequestOutput]:
        ids = [self.add_request(p, sampling_params, eos_token_id) for p in prompts]
        while self.has_unfinished_requests():
            self.step()
        return [self.outputs[i] for i in ids]
