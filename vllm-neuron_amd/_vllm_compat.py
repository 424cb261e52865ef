# SPDX-License-Identifier: Apache-2.0
"""vLLM 0.11 symbols this plugin is written against.

With vLLM installed the real classes are re-exported and the plugin is a drop-in
out-of-tree backend.  This build image has no vLLM (and no network), so the same names
fall back to small stand-ins that keep upstream's field names and call signatures:
enough to run the plugin's scheduler / worker / model runner as a standalone engine
(bench.py, tests) over the HIP library.  The stand-ins are host plumbing only — no model
arithmetic lives here.
"""

from __future__ import annotations

import enum
from collections import OrderedDict, deque
from dataclasses import dataclass, field
from typing import Any, Optional

import torch

try:  # pragma: no cover - exercised only where vLLM exists
    import vllm  # noqa: F401
    HAVE_VLLM = True
except Exception:  # ImportError, or a broken install
    HAVE_VLLM = False

if HAVE_VLLM:  # pragma: no cover
    from vllm.distributed import (ensure_model_parallel_initialized,  # noqa: F401
                                  init_distributed_environment)
    from vllm.model_executor import set_random_seed  # noqa: F401
    from vllm.platforms import Platform, PlatformEnum  # noqa: F401
    from vllm.sampling_params import SamplingParams  # noqa: F401
    from vllm.utils import make_tensor_with_pad  # noqa: F401
    from vllm.v1.core.sched.output import (CachedRequestData, NewRequestData,  # noqa: F401
                                           SchedulerOutput)
    from vllm.v1.core.sched.scheduler import Scheduler  # noqa: F401
    from vllm.v1.kv_cache_interface import (FullAttentionSpec, KVCacheConfig,  # noqa: F401
                                            KVCacheSpec)
    from vllm.v1.outputs import (EMPTY_MODEL_RUNNER_OUTPUT, DraftTokenIds,  # noqa: F401
                                 ModelRunnerOutput, SamplerOutput)
    from vllm.v1.request import Request, RequestStatus  # noqa: F401
    from vllm.v1.sample.sampler import Sampler  # noqa: F401
    from vllm.v1.worker.gpu_input_batch import CachedRequestState, InputBatch  # noqa: F401
    from vllm.v1.worker.worker_base import WorkerBase  # noqa: F401
else:
    # ------------------------------------------------------------------ platform / worker
    class PlatformEnum(enum.Enum):
        CUDA = enum.auto()
        ROCM = enum.auto()
        CPU = enum.auto()
        OOT = enum.auto()
        UNSPECIFIED = enum.auto()

    class Platform:
        _enum: PlatformEnum = PlatformEnum.UNSPECIFIED
        device_name: str = ""
        device_type: str = ""
        ray_device_key: str = ""
        supported_quantization: list = []
        device_control_env_var: str = ""

        def is_out_of_tree(self) -> bool:
            return self._enum == PlatformEnum.OOT

    class WorkerBase:
        def __init__(self, vllm_config, local_rank, rank, distributed_init_method,
                     is_driver_worker=False):
            self.vllm_config = vllm_config
            self.model_config = vllm_config.model_config
            self.cache_config = vllm_config.cache_config
            self.parallel_config = vllm_config.parallel_config
            self.scheduler_config = vllm_config.scheduler_config
            self.device_config = vllm_config.device_config
            self.lora_config = vllm_config.lora_config
            self.speculative_config = vllm_config.speculative_config
            self.local_rank, self.rank = local_rank, rank
            self.distributed_init_method = distributed_init_method
            self.is_driver_worker = is_driver_worker

    def init_distributed_environment(world_size=1, rank=0, local_rank=0,
                                     distributed_init_method=None, backend="gloo"):
        return None

    def ensure_model_parallel_initialized(tp=1, pp=1):
        return None

    def set_random_seed(seed):
        if seed is not None:
            torch.manual_seed(seed)

    def make_tensor_with_pad(x, pad, dtype, *, max_len=None, device=None, pin_memory=False):
        width = max_len if max_len is not None else max((len(r) for r in x), default=0)
        out = torch.full((len(x), width), pad, dtype=dtype)
        for i, row in enumerate(x):
            assert len(row) <= width
            out[i, :len(row)] = torch.as_tensor(row, dtype=dtype)
        return out

    # ------------------------------------------------------------------ configs (duck types)
    @dataclass
    class SamplingParams:
        temperature: float = 1.0
        top_k: int = 0            # 0 / -1 = disabled
        top_p: float = 1.0
        min_tokens: int = 0
        max_tokens: Optional[int] = 16
        ignore_eos: bool = False
        stop_token_ids: Optional[list] = None
        seed: Optional[int] = None

    # ------------------------------------------------------------------ scheduler I/O
    @dataclass
    class NewRequestData:
        req_id: str
        prompt_token_ids: list
        mm_features: list
        sampling_params: Any
        pooling_params: Any
        block_ids: tuple
        num_computed_tokens: int
        lora_request: Any = None

    @dataclass
    class CachedRequestData:
        req_ids: list = field(default_factory=list)
        resumed_from_preemption: list = field(default_factory=list)
        new_token_ids: list = field(default_factory=list)
        new_block_ids: list = field(default_factory=list)
        num_computed_tokens: list = field(default_factory=list)

    @dataclass
    class SchedulerOutput:
        scheduled_new_reqs: list
        scheduled_cached_reqs: CachedRequestData
        num_scheduled_tokens: dict
        total_num_scheduled_tokens: int
        scheduled_spec_decode_tokens: dict = field(default_factory=dict)
        finished_req_ids: set = field(default_factory=set)
        free_encoder_mm_hashes: list = field(default_factory=list)

    @dataclass
    class FullAttentionSpec:
        block_size: int
        num_kv_heads: int
        head_size: int
        dtype: torch.dtype
        sliding_window: Optional[int] = None

        @property
        def page_size_bytes(self) -> int:
            return 2 * self.block_size * self.num_kv_heads * self.head_size * \
                torch.empty((), dtype=self.dtype).element_size()

    KVCacheSpec = FullAttentionSpec

    @dataclass
    class KVCacheConfig:
        num_blocks: int
        kv_cache_tensors: list = field(default_factory=list)
        kv_cache_groups: list = field(default_factory=list)

    @dataclass
    class SamplerOutput:
        sampled_token_ids: torch.Tensor
        logprobs_tensors: Any = None

    @dataclass
    class ModelRunnerOutput:
        req_ids: list
        req_id_to_index: dict
        sampled_token_ids: list
        logprobs: Any = None
        prompt_logprobs_dict: dict = field(default_factory=dict)
        pooler_output: list = field(default_factory=list)

    @dataclass
    class DraftTokenIds:
        req_ids: list
        draft_token_ids: list

    EMPTY_MODEL_RUNNER_OUTPUT = ModelRunnerOutput(req_ids=[], req_id_to_index={},
                                                  sampled_token_ids=[])

    # ------------------------------------------------------------------ requests
    class RequestStatus(enum.IntEnum):
        WAITING = 0
        RUNNING = 1
        PREEMPTED = 2
        FINISHED_STOPPED = 3
        FINISHED_LENGTH_CAPPED = 4
        FINISHED_ABORTED = 5

        @staticmethod
        def is_finished(status) -> bool:
            return status >= RequestStatus.FINISHED_STOPPED

    class Request:
        def __init__(self, request_id, prompt_token_ids, sampling_params, eos_token_id=None,
                     pooling_params=None, arrival_time=0.0):
            self.request_id = request_id
            self.prompt_token_ids = list(prompt_token_ids)
            self.sampling_params = sampling_params
            self.pooling_params = pooling_params
            self.eos_token_id = eos_token_id
            self.arrival_time = arrival_time
            self.status = RequestStatus.WAITING
            self.stop_reason = None
            self.output_token_ids: list = []
            self.num_computed_tokens = 0
            self.num_cached_tokens = -1
            self.block_ids: list = []
            self.published = [0, None]         # prefix-cache chain: full blocks registered, hash of the last one
            self.max_tokens = sampling_params.max_tokens if sampling_params else 1

        @property
        def num_prompt_tokens(self):
            return len(self.prompt_token_ids)

        @property
        def num_tokens(self):
            return len(self.prompt_token_ids) + len(self.output_token_ids)

        @property
        def num_output_tokens(self):
            return len(self.output_token_ids)

        @property
        def all_token_ids(self):
            return self.prompt_token_ids + self.output_token_ids

        def append_output_token_ids(self, tok):
            self.output_token_ids.append(tok)

        def is_finished(self):
            return RequestStatus.is_finished(self.status)

    # ------------------------------------------------------------------ sampler + batch state
    @dataclass
    class SamplingMetadata:
        temperature: Optional[torch.Tensor]
        all_greedy: bool
        all_random: bool
        top_p: Optional[torch.Tensor]
        top_k: Optional[torch.Tensor]
        generators: dict
        max_num_logprobs: Optional[int] = None

    class Sampler:
        """CPU sampler with upstream's call shape: greedy rows are argmax; random rows apply
        temperature, top-k, top-p and draw from the per-request generator."""

        def __call__(self, logits: torch.Tensor, sampling_metadata: SamplingMetadata) -> SamplerOutput:
            logits = logits.to(torch.float32)
            greedy = logits.argmax(dim=-1)
            if sampling_metadata.all_greedy:
                return SamplerOutput(sampled_token_ids=greedy.to(torch.int32).unsqueeze(-1))
            temp = sampling_metadata.temperature
            scaled = logits / torch.where(temp < 1e-5, torch.ones_like(temp), temp)[:, None]
            if sampling_metadata.top_k is not None:
                k = sampling_metadata.top_k.clamp(min=1, max=logits.shape[-1]).long()
                kth = scaled.sort(dim=-1, descending=True).values.gather(1, (k - 1)[:, None])
                scaled = scaled.masked_fill(scaled < kth, float("-inf"))
            if sampling_metadata.top_p is not None:
                srt, idx = scaled.sort(dim=-1, descending=False)
                cum = srt.softmax(-1).cumsum(-1)
                drop = cum <= (1 - sampling_metadata.top_p)[:, None]
                drop[:, -1] = False
                scaled = scaled.masked_fill(drop.scatter(1, idx, drop), float("-inf"))
            probs = scaled.softmax(-1)
            out = torch.empty_like(greedy)
            for i in range(probs.shape[0]):
                g = sampling_metadata.generators.get(i)
                out[i] = torch.multinomial(probs[i], 1, generator=g)[0]
            out = torch.where(temp < 1e-5, greedy, out)
            return SamplerOutput(sampled_token_ids=out.to(torch.int32).unsqueeze(-1))

    @dataclass
    class CachedRequestState:
        req_id: str
        prompt_token_ids: list
        mm_features: list
        sampling_params: Any
        pooling_params: Any
        generator: Any
        block_ids: tuple
        num_computed_tokens: int
        output_token_ids: list
        lora_request: Any = None

        @property
        def num_tokens(self):
            return len(self.prompt_token_ids) + len(self.output_token_ids)

    class _BlockTable:
        def __init__(self, max_reqs):
            self.rows = [[] for _ in range(max_reqs)]

        def append_row(self, block_ids, row_idx):
            self.rows[row_idx].extend(block_ids[0] if block_ids and isinstance(block_ids[0], (list, tuple))
                                      else block_ids)

        def add_row(self, block_ids, row_idx):
            self.rows[row_idx] = []
            self.append_row(block_ids, row_idx)

        def move_row(self, src, dst):
            self.rows[dst] = self.rows[src]
            self.rows[src] = []

    class InputBatch:
        """Persistent batch: the subset of upstream's gpu_input_batch.InputBatch the runner
        touches (reference runner.py:111-119, 329-342, 404-510)."""

        def __init__(self, max_num_reqs, max_model_len, max_num_batched_tokens, device, pin_memory,
                     vocab_size, block_sizes, **_):
            self.max_num_reqs, self.max_model_len, self.vocab_size = max_num_reqs, max_model_len, vocab_size
            self._req_ids: list = [None] * max_num_reqs
            self.req_id_to_index: dict = {}
            self.token_ids_cpu = torch.zeros(max_num_reqs, max_model_len, dtype=torch.int32).numpy()
            self.num_tokens = [0] * max_num_reqs
            self.num_tokens_no_spec = [0] * max_num_reqs
            self.num_prompt_tokens = [0] * max_num_reqs
            self.num_computed_tokens_cpu = [0] * max_num_reqs
            self.block_table = _BlockTable(max_num_reqs)
            self._states: list = [None] * max_num_reqs
            self._dirty = True
            self._metadata = None

        @property
        def req_ids(self):
            return [r for r in self._req_ids[:self.num_reqs]]

        @property
        def num_reqs(self):
            return len(self.req_id_to_index)

        def add_request(self, request: CachedRequestState):
            idx = next(i for i, r in enumerate(self._req_ids) if r is None)
            self._req_ids[idx] = request.req_id
            self.req_id_to_index[request.req_id] = idx
            self._states[idx] = request
            n_p, n_o = len(request.prompt_token_ids), len(request.output_token_ids)
            self.token_ids_cpu[idx, :n_p] = request.prompt_token_ids
            self.token_ids_cpu[idx, n_p:n_p + n_o] = request.output_token_ids
            self.num_prompt_tokens[idx] = n_p
            self.num_tokens[idx] = self.num_tokens_no_spec[idx] = n_p + n_o
            self.num_computed_tokens_cpu[idx] = request.num_computed_tokens
            self.block_table.add_row(request.block_ids, idx)
            self._dirty = True
            return idx

        def remove_request(self, req_id):
            idx = self.req_id_to_index.pop(req_id, None)
            if idx is None:
                return None
            self._req_ids[idx] = None
            self._states[idx] = None
            self._dirty = True
            return idx

        def condense(self):
            """Close the gaps left by removed requests (move the last rows down)."""
            n = self.num_reqs
            for hole in range(n):
                if self._req_ids[hole] is not None:
                    continue
                last = max(i for i, r in enumerate(self._req_ids) if r is not None)
                if last < hole:
                    break
                rid = self._req_ids[last]
                self._req_ids[hole], self._req_ids[last] = rid, None
                self._states[hole], self._states[last] = self._states[last], None
                self.req_id_to_index[rid] = hole
                self.token_ids_cpu[hole] = self.token_ids_cpu[last]
                for arr in (self.num_tokens, self.num_tokens_no_spec, self.num_prompt_tokens,
                            self.num_computed_tokens_cpu):
                    arr[hole] = arr[last]
                self.block_table.move_row(last, hole)
            self._dirty = True

        def refresh_metadata(self):
            if not self._dirty:
                return
            n = self.num_reqs
            sps = [self._states[i].sampling_params for i in range(n)]
            temps = torch.tensor([float(sp.temperature) for sp in sps], dtype=torch.float32)
            greedy = [sp.temperature < 1e-5 for sp in sps]
            no_topk = all((sp.top_k or 0) <= 0 for sp in sps)
            no_topp = all(sp.top_p >= 1.0 for sp in sps)
            gens = {}
            for i, sp in enumerate(sps):
                st = self._states[i]
                if sp.seed is not None and st.generator is None:
                    st.generator = torch.Generator().manual_seed(sp.seed)
                if st.generator is not None:
                    gens[i] = st.generator
            self._metadata = SamplingMetadata(
                temperature=temps, all_greedy=all(greedy) if n else True, all_random=not any(greedy),
                top_p=None if no_topp else torch.tensor([sp.top_p for sp in sps], dtype=torch.float32),
                top_k=None if no_topk else torch.tensor(
                    [sp.top_k if (sp.top_k or 0) > 0 else self.vocab_size for sp in sps], dtype=torch.int32),
                generators=gens)
            self._dirty = False

        @property
        def sampling_metadata(self):
            if self._dirty or self._metadata is None:
                self.refresh_metadata()
            return self._metadata

    # ------------------------------------------------------------------ base scheduler
    class _BlockPool:
        """Ref-counted KV blocks 1..N-1 (0 = null block) with full-block prefix caching."""

        def __init__(self, num_blocks, block_size, enable_caching):
            self.block_size, self.enable_caching = block_size, enable_caching
            self.free = deque(range(1, num_blocks))
            self.ref = [0] * num_blocks
            self.hash_of: dict = {}            # block id -> prefix hash
            self.cached: dict = {}             # prefix hash -> block id
            self.evictable: OrderedDict = OrderedDict()   # cached blocks with ref 0 (LRU order)

        def num_free(self):
            return len(self.free) + len(self.evictable)

        def _take(self):
            if self.free:
                b = self.free.popleft()
            else:
                b, _ = self.evictable.popitem(last=False)
                self.cached.pop(self.hash_of.pop(b), None)
            self.ref[b] = 1
            return b

        def allocate(self, n):
            if n > self.num_free():
                return None
            return [self._take() for _ in range(n)]

        def lookup(self, token_ids):
            """Longest run of cached FULL blocks; never the whole prompt (>= 1 token is computed)."""
            hits, h = [], None
            if not self.enable_caching:
                return hits
            nfull = (len(token_ids) - 1) // self.block_size
            for i in range(nfull):
                h = hash((h, tuple(token_ids[i * self.block_size:(i + 1) * self.block_size])))
                b = self.cached.get(h)
                if b is None:
                    break
                hits.append(b)
            return hits

        def touch(self, blocks):
            for b in blocks:
                if self.ref[b] == 0:
                    self.evictable.pop(b, None)
                self.ref[b] += 1

        def publish(self, token_ids, blocks, state=None):
            """Register the full blocks of `token_ids` (whose KV is now written) for reuse.  `state` =
            [blocks already registered, hash of the last of them] of this request: only blocks that
            became full since the last call are hashed (vLLM keeps the chain per request likewise)."""
            if not self.enable_caching:
                return
            start, h = (state[0], state[1]) if state is not None else (0, None)
            nfull = len(token_ids) // self.block_size
            for i in range(start, nfull):
                h = hash((h, tuple(token_ids[i * self.block_size:(i + 1) * self.block_size])))
                if h not in self.cached and blocks[i] not in self.hash_of:
                    self.cached[h] = blocks[i]
                    self.hash_of[blocks[i]] = h
            if state is not None and nfull > start:
                state[0], state[1] = nfull, h

        def release(self, blocks):
            for b in reversed(blocks):
                self.ref[b] -= 1
                if self.ref[b] == 0:
                    if b in self.hash_of:
                        self.evictable[b] = None
                    else:
                        self.free.append(b)

    @dataclass
    class EngineCoreOutput:
        request_id: str
        new_token_ids: list
        finished: bool
        finish_reason: Any = None

    class Scheduler:
        """Minimal stand-in for vllm.v1.core.sched.scheduler.Scheduler: FCFS, running requests
        first (their next token, or the rest of a partly encoded prompt), then waiting requests
        (the prompt minus the cached prefix), bounded by max_num_seqs / max_num_batched_tokens /
        free KV blocks.  With scheduler_config.chunked_prefill_enabled a prompt that does not fit
        the step's token budget is encoded in chunks (vLLM's native behaviour); otherwise it waits."""

        def __init__(self, vllm_config, kv_cache_config=None, structured_output_manager=None,
                     include_finished_set=False, log_stats=False, **_):
            self.vllm_config = vllm_config
            self.scheduler_config = vllm_config.scheduler_config
            self.cache_config = vllm_config.cache_config
            self.max_num_running_reqs = self.scheduler_config.max_num_seqs
            self.max_num_scheduled_tokens = self.scheduler_config.max_num_batched_tokens
            self.max_model_len = self.scheduler_config.max_model_len
            self.block_size = self.cache_config.block_size
            self.chunked_prefill = bool(getattr(self.scheduler_config, "chunked_prefill_enabled", False))
            # speculative decoding: KV blocks are reserved that many positions ahead of a running request
            # (vLLM's num_lookahead_tokens); a step may then produce up to that many tokens per request
            spec = getattr(vllm_config, "speculative_config", None)
            self.num_lookahead_tokens = int(getattr(spec, "num_speculative_tokens", 0) or 0) if spec is not None else 0
            num_blocks = kv_cache_config.num_blocks if kv_cache_config else self.cache_config.num_gpu_blocks
            self.block_pool = _BlockPool(num_blocks, self.block_size,
                                         bool(self.cache_config.enable_prefix_caching))
            self.requests: dict = {}
            self.waiting: deque = deque()
            self.running: list = []
            self.finished_req_ids: set = set()

        def add_request(self, request: Request):
            self.requests[request.request_id] = request
            self.waiting.append(request)

        def has_unfinished_requests(self):
            return bool(self.waiting or self.running)

        has_requests = has_unfinished_requests

        def _blocks_needed(self, request, num_tokens_after):
            return -(-num_tokens_after // self.block_size) - len(request.block_ids)

        def schedule(self) -> SchedulerOutput:
            budget = self.max_num_scheduled_tokens
            new_reqs, num_sched = [], {}
            cached = CachedRequestData()
            for req in list(self.running):
                if budget <= 0:
                    break
                n_tok = req.num_tokens - req.num_computed_tokens
                if self.chunked_prefill:
                    n_tok = min(n_tok, budget)           # the next chunk of a long prompt
                after = min(req.num_computed_tokens + n_tok + self.num_lookahead_tokens, self.max_model_len)
                need = self._blocks_needed(req, max(after, req.num_computed_tokens + n_tok))
                new_blocks = self.block_pool.allocate(need) if need > 0 else []
                if new_blocks is None:
                    break                      # out of KV blocks: leave the rest unscheduled
                req.block_ids.extend(new_blocks)
                cached.req_ids.append(req.request_id)
                cached.resumed_from_preemption.append(False)
                cached.new_token_ids.append([])
                cached.new_block_ids.append((new_blocks,) if new_blocks else None)
                cached.num_computed_tokens.append(req.num_computed_tokens)
                num_sched[req.request_id] = n_tok
                budget -= n_tok
            while self.waiting and budget > 0 and len(self.running) < self.max_num_running_reqs:
                req = self.waiting[0]
                hits = self.block_pool.lookup(req.all_token_ids)
                n_cached = len(hits) * self.block_size
                n_new = req.num_tokens - n_cached
                if n_new > budget:
                    if not self.chunked_prefill:
                        break
                    n_new = budget                       # first chunk; the rest follows as a running request
                need = -(-(n_cached + n_new) // self.block_size) - len(hits)
                self.block_pool.touch(hits)
                fresh = self.block_pool.allocate(need)
                if fresh is None:
                    self.block_pool.release(hits)
                    break
                self.waiting.popleft()
                req.block_ids = hits + fresh
                req.num_computed_tokens = n_cached
                req.num_cached_tokens = n_cached
                req.status = RequestStatus.RUNNING
                self.running.append(req)
                new_reqs.append(NewRequestData(
                    req_id=req.request_id, prompt_token_ids=list(req.prompt_token_ids), mm_features=[],
                    sampling_params=req.sampling_params, pooling_params=req.pooling_params,
                    block_ids=(list(req.block_ids),), num_computed_tokens=n_cached))
                num_sched[req.request_id] = n_new
                budget -= n_new
            out = SchedulerOutput(
                scheduled_new_reqs=new_reqs, scheduled_cached_reqs=cached, num_scheduled_tokens=num_sched,
                total_num_scheduled_tokens=sum(num_sched.values()), finished_req_ids=self.finished_req_ids)
            self.finished_req_ids = set()
            return out

        def _update_request_with_output(self, request, new_token_ids):
            stopped = False
            for num_new, tok in enumerate(new_token_ids, 1):
                request.append_output_token_ids(tok)
                if request.num_tokens >= self.max_model_len or request.num_output_tokens >= request.max_tokens:
                    request.status = RequestStatus.FINISHED_LENGTH_CAPPED
                    stopped = True
                elif not request.sampling_params.ignore_eos and tok == request.eos_token_id:
                    request.status = RequestStatus.FINISHED_STOPPED
                    stopped = True
                if stopped:
                    del new_token_ids[num_new:]
                    break
            return new_token_ids, stopped

        def update_from_output(self, scheduler_output, model_runner_output):
            outputs = []
            for req_id, n_sched in scheduler_output.num_scheduled_tokens.items():
                req = self.requests.get(req_id)
                if req is None or req.is_finished():
                    continue
                req.num_computed_tokens += n_sched
                idx = model_runner_output.req_id_to_index.get(req_id)
                toks = list(model_runner_output.sampled_token_ids[idx]) if idx is not None else []
                if req.num_computed_tokens // self.block_size > req.published[0]:    # a block became full
                    self.block_pool.publish(req.all_token_ids[:req.num_computed_tokens], req.block_ids, req.published)
                stopped = False
                if toks:
                    toks, stopped = self._update_request_with_output(req, toks)
                if stopped:
                    self.running.remove(req)
                    self.block_pool.release(req.block_ids)
                    self.finished_req_ids.add(req_id)
                    del self.requests[req_id]
                if toks or stopped:
                    outputs.append(EngineCoreOutput(req_id, toks, stopped, req.status if stopped else None))
            return outputs


# ------------------------------------------------------------------ config duck types
# (used by the standalone engine and the tests; with vLLM installed the real VllmConfig is
# passed in and only the attributes read below matter)
@dataclass
class SimpleModelConfig:
    model: str
    hf_config: Any
    dtype: Any = torch.bfloat16
    max_model_len: int = 2048
    seed: Optional[int] = 0
    trust_remote_code: bool = False

    def get_vocab_size(self):
        return self.hf_config.vocab_size


@dataclass
class SimpleCacheConfig:
    block_size: Optional[int] = 32
    num_gpu_blocks_override: Optional[int] = None
    enable_prefix_caching: bool = True
    num_gpu_blocks: Optional[int] = None
    num_cpu_blocks: Optional[int] = None


@dataclass
class SimpleParallelConfig:
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    worker_cls: str = "auto"
    distributed_executor_backend: Optional[str] = None
    enable_expert_parallel: bool = False
    rank: int = 0   # TP rank of THIS process (one process per GPU)

    @property
    def world_size(self):
        return self.tensor_parallel_size * self.pipeline_parallel_size


@dataclass
class SimpleSchedulerConfig:
    max_num_seqs: Optional[int] = 4
    max_num_batched_tokens: int = 8192
    max_model_len: int = 2048
    chunked_prefill_enabled: bool = False
    scheduler_cls: Any = None


@dataclass
class SimpleSpeculativeConfig:
    """The attributes of vllm.config.SpeculativeConfig the plugin reads (reference loader.py:785-791,
    292-301; runner.py:828-830)."""
    num_speculative_tokens: int = 4
    method: Optional[str] = None            # "eagle" selects EAGLE drafts in the reference
    draft_model_config: Any = None          # a ModelConfig: .model (path), .hf_config


@dataclass
class SimpleDeviceConfig:
    device: Any = "cpu"


@dataclass
class SimpleVllmConfig:
    model_config: Optional[SimpleModelConfig]
    cache_config: SimpleCacheConfig = field(default_factory=SimpleCacheConfig)
    parallel_config: SimpleParallelConfig = field(default_factory=SimpleParallelConfig)
    scheduler_config: SimpleSchedulerConfig = field(default_factory=SimpleSchedulerConfig)
    device_config: SimpleDeviceConfig = field(default_factory=SimpleDeviceConfig)
    lora_config: Any = None
    load_config: Any = None
    speculative_config: Any = None
    observability_config: Any = None
    additional_config: dict = field(default_factory=dict)
