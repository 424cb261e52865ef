# SPDX-License-Identifier: Apache-2.0
"""Model runner: SchedulerOutput -> model inputs -> libmi355x_vllm -> sampled tokens.

Host-side mirror of the reference runner
(/root/reference/vllm_neuron/worker/neuronx_distributed_model_runner.py) for the text
continuous-batching path (with and without prefix caching): the persistent batch, the
per-step argument record (``ModelInputForMI355X`` has the reference's fields and pad
conventions: slot pad -1, block-table pad 0 / -1), the seq-id slot pool, CPU sampling through
vLLM's ``Sampler`` and the ``ModelRunnerOutput`` assembly.  Chunked prefill, multimodal
inputs, LoRA and speculative decoding raise ``NotImplementedError`` (out of scope).

Differences that are deliberate: slot mappings are built with tensor ops instead of an
O(max_model_len) Python list comprehension per prefill (reference runner.py:757-761), block
lists are not deep-copied per request per step (:744, :803), and nothing is formatted for
debug logging unless the logger is enabled (:258, :278, :1168-1196).
"""

from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch

from .._vllm_compat import (EMPTY_MODEL_RUNNER_OUTPUT, CachedRequestState, DraftTokenIds,
                            FullAttentionSpec, InputBatch, ModelRunnerOutput, Sampler,
                            SamplerOutput, make_tensor_with_pad)
from .mi355x_model_loader import get_mi355x_model

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class ModelInputForMI355X:
    """Per-step argument record (same fields as the reference's ModelInputForNeuron,
    runner.py:34-52).  All tensors are CPU int64 except sampling_params (float [B, 3])."""
    request_ids: list[str] | None = None
    input_tokens: torch.Tensor | None = None
    position_ids: torch.Tensor | None = None
    input_block_ids: torch.Tensor | None = None
    slot_mapping: torch.Tensor | None = None
    block_tables: torch.Tensor | None = None
    full_context_lens: torch.Tensor | None = None
    computed_context_lens: torch.Tensor | None = None
    sampling_params: torch.Tensor | None = None
    multi_modal_kwargs: dict | None = None
    adapter_ids: str | None = None
    prefill_completion_state: torch.Tensor | None = None


@dataclass
class IntermediateInputData:
    request_ids: list[str] = field(default_factory=list)
    input_tokens: list = field(default_factory=list)
    position_ids: list = field(default_factory=list)
    input_block_ids: list[int] = field(default_factory=list)
    full_context_lens: list[int] = field(default_factory=list)
    computed_context_lens: list[int] = field(default_factory=list)
    slot_mapping: list = field(default_factory=list)
    block_tables: list = field(default_factory=list)
    prefill_completion_state: list = field(default_factory=list)
    adapter_ids: list = field(default_factory=list)
    multi_modal_kwargs: dict | None = None


class MI355XModelRunner:
    # cap applied to per-request top_k when packing on-device sampling parameters
    _MAX_DEVICE_SAMPLING_TOP_K = 256
    # slot -1 = "do not write"; block-table pad 0 = vLLM's null block (a valid pool index that no
    # live position maps to)
    _SLOT_MAPPING_PAD = -1
    _BLOCK_TABLE_PAD = 0

    def __init__(self, vllm_config, device, device_id: int = 0, tp_device_ids=None):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.cache_config = vllm_config.cache_config
        self.lora_config = vllm_config.lora_config
        self.load_config = vllm_config.load_config
        self.parallel_config = vllm_config.parallel_config
        self.scheduler_config = vllm_config.scheduler_config
        self.speculative_config = vllm_config.speculative_config
        self.observability_config = vllm_config.observability_config
        self.device_config = vllm_config.device_config
        self.device = device
        self.device_id, self.tp_device_ids = device_id, list(tp_device_ids or [device_id])

        self.pin_memory = False
        self.block_size = self.cache_config.block_size
        self.max_num_reqs = self.scheduler_config.max_num_seqs
        self.max_model_len = self.model_config.max_model_len
        self.max_num_tokens = self.scheduler_config.max_num_batched_tokens

        self.input_batch = InputBatch(
            max_num_reqs=self.max_num_reqs, max_model_len=self.max_model_len,
            max_num_batched_tokens=self.max_num_tokens, device=self.device, pin_memory=self.pin_memory,
            vocab_size=self.model_config.get_vocab_size(), block_sizes=[self.block_size])
        self.requests: dict[str, CachedRequestState] = {}
        self.model = None
        self.is_block_kv_layout = False
        self.is_prefix_caching = False
        self.is_chunked_prefill = False
        # vLLM request id -> sequence slot ("batch line" of the contiguous-KV mode)
        self.use_custom_seq_id_mapping = True
        self.vllm_req_to_seq_id_mapping: Dict[str, int] = {}
        self.free_seq_ids = set(range(self.max_num_reqs))
        self._draft_token_ids = None
        self._decode_state = None        # persistent token-generation inputs (_prepare_decode_inputs_incremental)
        self.cpu_sampler = Sampler()
        self._kv_ready = False

    # the reference spells this mapping with "neuron" in its name; tests / tools may look for it
    @property
    def vllm_req_to_neuron_seq_id_mapping(self):
        return self.vllm_req_to_seq_id_mapping

    # ---- lifecycle ------------------------------------------------------------------------
    def load_model(self) -> None:
        if self.lora_config is not None:
            raise NotImplementedError("Multi-lora is not yet supported on the MI355X plugin")
        self.model = get_mi355x_model(
            self.model_config, cache_config=self.cache_config, parallel_config=self.parallel_config,
            scheduler_config=self.scheduler_config, lora_serving_config=None,
            speculative_config=self.speculative_config, additional_config=self.vllm_config.additional_config,
            device_id=self.device_id, tp_device_ids=self.tp_device_ids)
        cfg = self.model.mi355x_config
        self.is_block_kv_layout = cfg.is_block_kv_layout
        self.is_prefix_caching = cfg.is_prefix_caching
        self.is_chunked_prefill = cfg.chunked_prefill_config is not None
        self.use_custom_seq_id_mapping = not self.is_chunked_prefill     # reference runner.py:133
        self.model.is_reorder_needed = not (self.is_prefix_caching or self.is_chunked_prefill)
        self._validate_sampling_configuration()

    def initialize_kv_cache(self, kv_cache_config=None) -> None:
        """The library owns the KV pool (like NxDI in the reference, runner.py:142-150); this
        is where it is sized and allocated, once vLLM has decided the block count."""
        if self._kv_ready:
            return
        # (with fused speculation the draft keeps a pool of its own under the same block ids)
        for native in (self.model.model, getattr(self.model, "draft", None)):
            if native is None:
                continue
            if self.is_block_kv_layout and kv_cache_config is not None and getattr(kv_cache_config, "num_blocks", None):
                native.set_num_blocks(int(kv_cache_config.num_blocks))
            native.finalize()
        self._kv_ready = True

    def _validate_sampling_configuration(self) -> None:
        try:
            if self.model.mi355x_config.on_device_sampling_config is not None:
                # hardware sampling (reference runner.py:208-228): ids come back from the model call
                if not hasattr(self.model, "sample"):
                    raise RuntimeError("Model does not have required 'sample' method for hardware sampling")
                self.model._sample_seed = int(getattr(self.model_config, "seed", 0) or 0) & 0xFFFFFFFF
                logger.info("On-device sampling enabled: config=%s (top_k <= %d, per-request seeds and "
                            "logprobs are CPU-sampling features)", self.model.mi355x_config.on_device_sampling_config,
                            self._MAX_DEVICE_SAMPLING_TOP_K)
                return
            if self.cpu_sampler is None:
                raise RuntimeError("CPU sampling is required but cpu_sampler is not initialized")
            if not hasattr(self.model, "sample"):
                raise RuntimeError("Model does not have required 'sample' method for hardware sampling")
        except Exception as e:
            raise RuntimeError(f"Invalid sampling configuration: {str(e)}") from e
        logger.info("CPU sampling enabled: logits are sampled with vLLM's standard sampler.")

    def get_kv_cache_spec(self) -> dict:
        return {"layer": FullAttentionSpec(block_size=self.block_size,
                                           num_kv_heads=self.model.num_key_value_heads,
                                           head_size=self.model.head_dim, dtype=torch.bfloat16,
                                           sliding_window=None)}

    def _get_last_token_position(self, state: CachedRequestState) -> int:
        """0-based position of the token fed to the next decode step."""
        return len(state.prompt_token_ids) + len(state.output_token_ids) - 1

    # ---- one engine step -------------------------------------------------------------------
    @torch.inference_mode()
    def execute_model(self, scheduler_output, intermediate_tensors=None):
        if not self._kv_ready:
            self.initialize_kv_cache(None)
        if self.use_custom_seq_id_mapping:
            for req_id in scheduler_output.finished_req_ids:
                slot = self.vllm_req_to_seq_id_mapping.pop(req_id, None)
                if slot is not None:
                    self.free_seq_ids.add(slot)
        self._update_states(scheduler_output)
        if not scheduler_output.total_num_scheduled_tokens:
            return EMPTY_MODEL_RUNNER_OUTPUT
        model_input = self._prepare_model_input(scheduler_output)
        sampler_outputs = self._execute_model_for_text(model_input, intermediate_tensors)
        return self._generate_model_runner_output(sampler_outputs)

    def _generate_model_runner_output(self, sampler_outputs: SamplerOutput | None):
        if sampler_outputs is None:
            return EMPTY_MODEL_RUNNER_OUTPUT
        sampled = sampler_outputs.sampled_token_ids
        if self.speculative_config is not None and sampled.dim() == 3 and sampled.size(-1) == 1:
            sampled = sampled.squeeze(-1)       # fused speculation: [B, T, 1] -> [B, T] (reference runner.py:310-312)
        # -1 entries are pads (rows that produced no token, or the unused tail of a speculation
        # window); 0 is a real token id
        valid_sampled_token_ids = [[x for x in row if x != -1] for row in sampled.tolist()]
        if self.speculative_config is not None:
            # what vLLM calls the speculated tokens of the step: all but the last one generated
            # (reference runner.py:314-323)
            self.spec_token_ids = [kept[:-1] if kept else [] for kept in valid_sampled_token_ids]
        for req_idx, sampled_ids in enumerate(valid_sampled_token_ids):
            if not sampled_ids:
                continue
            start_idx = self.input_batch.num_tokens_no_spec[req_idx]
            end_idx = start_idx + len(sampled_ids)
            assert end_idx <= self.max_model_len, (
                "Sampled token IDs exceed the max model length. "
                f"Total number of tokens: {end_idx} > max_model_len: {self.max_model_len}")
            self.input_batch.token_ids_cpu[req_idx, start_idx:end_idx] = sampled_ids
            self.input_batch.num_tokens_no_spec[req_idx] = end_idx
            self.input_batch.num_tokens[req_idx] = end_idx
            self.requests[self.input_batch.req_ids[req_idx]].output_token_ids.extend(sampled_ids)
        logprobs = None
        if sampler_outputs.logprobs_tensors is not None:
            logprobs = sampler_outputs.logprobs_tensors.tolists()
        return ModelRunnerOutput(req_ids=self.input_batch.req_ids, req_id_to_index=self.input_batch.req_id_to_index,
                                 sampled_token_ids=valid_sampled_token_ids, logprobs=logprobs,
                                 prompt_logprobs_dict={}, pooler_output=[])

    def _update_states(self, scheduler_output) -> None:
        """Persistent-batch bookkeeping, same transitions as the reference (runner.py:381-510):
        drop finished and unscheduled requests, add new / resumed ones, append or (after a
        preemption) replace block ids, condense."""
        for req_id in scheduler_output.finished_req_ids:
            self.requests.pop(req_id, None)
        for req_id in scheduler_output.finished_req_ids:
            self.input_batch.remove_request(req_id)
        scheduled = scheduler_output.num_scheduled_tokens.keys()
        for req_id in self.input_batch.req_id_to_index.keys() - scheduled:
            self.input_batch.remove_request(req_id)

        reqs_to_add: list[CachedRequestState] = []
        for new_req in scheduler_output.scheduled_new_reqs:
            state = CachedRequestState(
                req_id=new_req.req_id, prompt_token_ids=new_req.prompt_token_ids,
                mm_features=new_req.mm_features or [], sampling_params=new_req.sampling_params,
                pooling_params=new_req.pooling_params, generator=None, block_ids=new_req.block_ids,
                num_computed_tokens=new_req.num_computed_tokens, output_token_ids=[],
                lora_request=new_req.lora_request)
            self.requests[new_req.req_id] = state
            reqs_to_add.append(state)

        cached = scheduler_output.scheduled_cached_reqs
        for i, req_id in enumerate(cached.req_ids):
            state = self.requests[req_id]
            new_block_ids = cached.new_block_ids[i]
            state.num_computed_tokens = self._get_last_token_position(state)
            if not cached.resumed_from_preemption[i]:
                if new_block_ids is not None:
                    for block_ids, new_ids in zip(state.block_ids, new_block_ids):
                        block_ids.extend(new_ids)
            else:
                assert new_block_ids is not None
                state.block_ids = new_block_ids
            req_index = self.input_batch.req_id_to_index.get(req_id)
            if req_index is None:
                reqs_to_add.append(state)   # was preempted / unscheduled: re-enters the batch
                continue
            self.input_batch.num_computed_tokens_cpu[req_index] = cached.num_computed_tokens[i]
            if new_block_ids is not None:
                self.input_batch.block_table.append_row(new_block_ids, req_index)
            spec_token_ids = scheduler_output.scheduled_spec_decode_tokens.get(req_id, ())
            if spec_token_ids:      # draft tokens vLLM scheduled for verification (reference runner.py:488-498)
                start = self.input_batch.num_tokens_no_spec[req_index]
                self.input_batch.token_ids_cpu[req_index, start:start + len(spec_token_ids)] = spec_token_ids
                self.input_batch.num_tokens[req_index] += len(spec_token_ids)

        for request in reqs_to_add:
            self.input_batch.add_request(request)
        self.input_batch.condense()
        self.input_batch.refresh_metadata()

    def _execute_model_for_text(self, model_input: ModelInputForMI355X, intermediate_tensors=None):
        logits = self.model(
            input_ids=model_input.input_tokens, position_ids=model_input.position_ids,
            input_block_ids=model_input.input_block_ids, slot_mapping=model_input.slot_mapping,
            block_tables=model_input.block_tables, full_context_lens=model_input.full_context_lens,
            computed_context_lens=model_input.computed_context_lens,
            sampling_params=model_input.sampling_params, adapter_ids=model_input.adapter_ids,
            prefill_completion_state=model_input.prefill_completion_state)
        return self._sample(logits, model_input)

    # ---- input preparation --------------------------------------------------------------------
    def _prepare_model_input(self, scheduler_output) -> ModelInputForMI355X:
        if self.is_chunked_prefill:
            return self._finalize_chunked_prefill_inputs(self._prepare_chunked_prefill_inputs(scheduler_output))
        if self.is_prefix_caching and not scheduler_output.scheduled_new_reqs:
            return self._prepare_decode_inputs_incremental(scheduler_output)
        data, is_prefill = self._prepare_continuous_batching_inputs(scheduler_output)
        return self._finalize_continuous_batching_inputs(data, is_prefill)

    def _prepare_decode_inputs_incremental(self, scheduler_output) -> ModelInputForMI355X:
        """Token-generation inputs without rebuilding what did not change.  Same
        record, same values as `_process_cached_request_*` + `_finalize_*` produce; the padded
        block-table rows persist across steps in one [max_num_seqs, MB] tensor -- a row is rewritten
        only when another request takes it or a block is appended (the reference pads, copies and
        stacks every table every step: runner.py:798-832, 887-917) -- and the library in turn keeps
        the tables on the device and re-sends only rows that changed."""
        cached = scheduler_output.scheduled_cached_reqs
        req_ids = cached.req_ids
        n = len(req_ids)
        st = self._decode_state
        mb = self.scheduler_config.max_model_len // self.cache_config.block_size
        bs = self.cache_config.block_size
        cfg = self.model.mi355x_config
        pad = -1 if (cfg.attn_tkg_nki_kernel_enabled or cfg.attn_block_tkg_nki_kernel_enabled) else self._BLOCK_TABLE_PAD
        if st is None or st["pad"] != pad:
            st = self._decode_state = {"pad": pad, "rows": [None] * self.max_num_reqs, "nblk": [0] * self.max_num_reqs,
                                       "tbl": [None] * self.max_num_reqs,
                                       "bt": torch.full((self.max_num_reqs, mb), pad, dtype=torch.long)}
        bt, rows, nblk, tbl = st["bt"], st["rows"], st["nblk"], st["tbl"]
        tokens, positions, seq_ids, slots = [], [], [], []
        for i, req_id in enumerate(req_ids):
            assert req_id in self.vllm_req_to_seq_id_mapping, (
                "The request ID for the current decode request is not found in request to sequence ID mapping")
            state = self.requests[req_id]
            block_table = state.block_ids[0]
            position = len(state.prompt_token_ids) + len(state.output_token_ids) - 1
            # `_update_states` extends a request's block list in place and REPLACES the list object
            # when the request resumes after a preemption (possibly with as many blocks as before):
            # the row is rebuilt whenever the request or the list object behind it changed.
            if rows[i] != req_id or tbl[i] is not block_table:
                bt[i].fill_(pad)
                bt[i, :len(block_table)] = torch.as_tensor(block_table, dtype=torch.long)
                rows[i], nblk[i], tbl[i] = req_id, len(block_table), block_table
            elif nblk[i] != len(block_table):        # blocks appended
                if len(block_table) < nblk[i]:
                    bt[i, len(block_table):nblk[i]] = pad
                bt[i, :len(block_table)] = torch.as_tensor(block_table, dtype=torch.long)
                nblk[i] = len(block_table)
            tokens.append(state.output_token_ids[-1])
            positions.append(position)
            seq_ids.append(self.vllm_req_to_seq_id_mapping[req_id])
            slots.append(self._decode_slots(block_table, position))
        for i in range(n, self.max_num_reqs):
            rows[i] = tbl[i] = None
        pos_t = torch.tensor(positions, dtype=torch.long).reshape(n, 1)
        input_tokens = torch.tensor(tokens, dtype=torch.long).reshape(n, 1)
        return ModelInputForMI355X(
            request_ids=list(req_ids), input_tokens=input_tokens, position_ids=pos_t,
            input_block_ids=torch.tensor(seq_ids, dtype=torch.long),
            slot_mapping=torch.tensor(slots, dtype=torch.long).reshape(n, -1), block_tables=bt[:n],
            full_context_lens=pos_t + 1, computed_context_lens=pos_t, prefill_completion_state=None,
            sampling_params=self._sampling_params_if_used(input_tokens), multi_modal_kwargs=None, adapter_ids=None)

    def _decode_slots(self, block_table, position) -> list:
        """K/V slot of the token fed to a token-generation step; with speculative decoding also the
        slots of the speculated positions behind it (the reference appends consecutive slot numbers,
        runner.py:825-830; here each position is looked up, so a window may cross a block boundary;
        positions past the blocks the request owns get the pad and are never written)."""
        bs = self.cache_config.block_size
        n = 1 if self.speculative_config is None else max(int(self.speculative_config.num_speculative_tokens), 1)
        out = []
        for p in range(position, position + n):
            blk = p // bs
            out.append(block_table[blk] * bs + p % bs if blk < len(block_table) and p < self.max_model_len
                       else self._SLOT_MAPPING_PAD)
        return out

    # ---- chunked prefill (vLLM's native scheduler; reference runner.py:654-680, 938-1051) ----------
    def _prepare_chunked_prefill_inputs(self, scheduler_output) -> IntermediateInputData:
        """A step is ONE ragged token batch: new requests are prompt chunks; cached requests are
        either the next chunk of a prompt or the single next token of a request that generates."""
        data = IntermediateInputData()
        num_scheduled = scheduler_output.num_scheduled_tokens
        for request_data in scheduler_output.scheduled_new_reqs:
            if request_data.mm_features:
                raise NotImplementedError("multimodal inputs are not supported on the MI355X plugin")
            assert len(request_data.block_ids) == 1
            self._append_chunk(data, request_data.req_id, request_data.prompt_token_ids, [],
                               request_data.block_ids[0], request_data.num_computed_tokens,
                               num_scheduled[request_data.req_id])
        cached = scheduler_output.scheduled_cached_reqs
        for i, req_id in enumerate(cached.req_ids):
            state = self.requests[req_id]
            self._append_chunk(data, req_id, state.prompt_token_ids, state.output_token_ids, state.block_ids[0],
                               cached.num_computed_tokens[i], num_scheduled[req_id])
        return data

    def _append_chunk(self, data: IntermediateInputData, req_id, prompt, outputs, block_table, start, n) -> None:
        end = start + n
        data.request_ids.append(req_id)
        # tokens start .. end-1 of the sequence (prompt, then what was generated so far)
        if end <= len(prompt):
            data.input_tokens.extend(prompt[start:end])
        else:
            seq = list(prompt) + list(outputs)
            data.input_tokens.extend(seq[start:end])
        data.position_ids.extend(range(start, end))
        data.input_block_ids.append(0)
        bs = self.cache_config.block_size
        data.slot_mapping.extend(block_table[i // bs] * bs + i % bs for i in range(start, end))
        data.block_tables.append(list(block_table))
        data.full_context_lens.append(end)
        data.computed_context_lens.append(start)
        data.prefill_completion_state.append(end >= len(prompt))

    def _finalize_chunked_prefill_inputs(self, data: IntermediateInputData) -> ModelInputForMI355X:
        input_tokens = torch.tensor(data.input_tokens, dtype=torch.long).reshape(1, -1)
        max_blocks = max(len(b) for b in data.block_tables)
        block_tables = torch.tensor([b + [self._BLOCK_TABLE_PAD] * (max_blocks - len(b)) for b in data.block_tables],
                                    dtype=torch.long)
        return ModelInputForMI355X(
            request_ids=data.request_ids, input_tokens=input_tokens,
            position_ids=torch.tensor(data.position_ids, dtype=torch.long).reshape(1, -1),
            input_block_ids=torch.tensor(data.input_block_ids[:1], dtype=torch.long),
            slot_mapping=torch.tensor(data.slot_mapping, dtype=torch.long), block_tables=block_tables,
            full_context_lens=torch.tensor(data.full_context_lens, dtype=torch.long),
            computed_context_lens=torch.tensor(data.computed_context_lens, dtype=torch.long),
            prefill_completion_state=torch.tensor(data.prefill_completion_state, dtype=torch.bool),
            sampling_params=self._chunked_sampling_params(data), multi_modal_kwargs=None, adapter_ids=None)

    def _chunked_sampling_params(self, data: IntermediateInputData):
        """(top_k, top_p, temperature) rows in the order of the ragged batch's requests."""
        if self.model.mi355x_config.on_device_sampling_config is None:
            return None
        max_topk = min(self.model_config.get_vocab_size(), self._MAX_DEVICE_SAMPLING_TOP_K)
        rows = []
        for rid in data.request_ids:
            sp = self.requests[rid].sampling_params
            top_k = sp.top_k if 0 < (sp.top_k or 0) < max_topk else max_topk
            temperature = sp.temperature
            if temperature == 0.0:
                top_k, temperature = 1, 1.0
            rows.append([float(top_k), float(sp.top_p), float(temperature)])
        return torch.tensor(rows, dtype=torch.float32)

    def _prepare_continuous_batching_inputs(self, scheduler_output) -> Tuple[IntermediateInputData, bool]:
        """New requests are prefills, cached requests are decodes; the scheduler never mixes
        them in one step."""
        data = IntermediateInputData()
        is_prefill = False
        for request_data in scheduler_output.scheduled_new_reqs:
            self._process_new_request_for_continuous_batching(request_data, data)
            is_prefill = True
        cached = scheduler_output.scheduled_cached_reqs
        for i, _ in enumerate(cached.req_ids):
            self._process_cached_request_for_continuous_batching(cached, i, data)
        return data, is_prefill

    def _process_new_request_for_continuous_batching(self, request_data, data: IntermediateInputData) -> None:
        assert request_data.req_id not in self.vllm_req_to_seq_id_mapping, (
            "Encountered an existing request ID while prefilling a new request")
        assert self.free_seq_ids, "No free sequence ID available!"
        if request_data.mm_features:
            raise NotImplementedError("multimodal inputs are not supported on the MI355X plugin")
        slot = self.free_seq_ids.pop()
        self.vllm_req_to_seq_id_mapping[request_data.req_id] = slot
        n = len(request_data.prompt_token_ids)
        data.request_ids.append(request_data.req_id)
        data.input_tokens.append(request_data.prompt_token_ids)      # the FULL prompt, even on a cache hit
        data.position_ids.append(list(range(n)))
        data.input_block_ids.append(slot)
        data.full_context_lens.append(n)
        data.prefill_completion_state.append(None)
        data.adapter_ids.append(None)
        if self.is_prefix_caching:
            self._process_new_request_for_continuous_batching_with_prefix_caching(request_data, data)

    def _padded_block_table(self, block_table, pad) -> torch.Tensor:
        max_blocks = self.scheduler_config.max_model_len // self.cache_config.block_size
        out = torch.full((max_blocks,), pad, dtype=torch.long)
        out[:len(block_table)] = torch.as_tensor(block_table, dtype=torch.long)
        return out

    def _process_new_request_for_continuous_batching_with_prefix_caching(self, request_data,
                                                                          data: IntermediateInputData) -> None:
        assert len(request_data.block_ids) == 1
        block_table = request_data.block_ids[0]
        bs = self.cache_config.block_size
        padded = self._padded_block_table(block_table, self._BLOCK_TABLE_PAD)
        data.block_tables.append(padded)
        n_cached = request_data.num_computed_tokens
        data.computed_context_lens.append(n_cached)
        # slot[i] = block(i) * bs + i % bs for the tokens that are NOT cached yet, i.e. the list
        # the reference builds over max_model_len and slices at num_computed_tokens (runner.py:756-763)
        pos = torch.arange(n_cached, len(request_data.prompt_token_ids), dtype=torch.long)
        data.slot_mapping.append(padded[pos // bs] * bs + pos % bs)

    def _process_cached_request_for_continuous_batching(self, request_data, index: int,
                                                        data: IntermediateInputData) -> None:
        req_id = request_data.req_ids[index]
        assert req_id in self.vllm_req_to_seq_id_mapping, (
            "The request ID for the current decode request is not found in request to sequence ID mapping")
        state = self.requests[req_id]
        position = self._get_last_token_position(state)
        data.request_ids.append(req_id)
        data.input_tokens.append([state.output_token_ids[-1]])
        data.position_ids.append([position])
        data.input_block_ids.append(self.vllm_req_to_seq_id_mapping[req_id])
        data.full_context_lens.append(position + 1)
        data.computed_context_lens.append(position)
        data.prefill_completion_state.append(None)
        data.adapter_ids.append(None)
        if self.is_prefix_caching:
            self._process_cached_request_for_continuous_batching_with_prefix_caching(request_data, index, data)

    def _process_cached_request_for_continuous_batching_with_prefix_caching(self, request_data, index: int,
                                                                             data: IntermediateInputData) -> None:
        state = self.requests[request_data.req_ids[index]]
        block_table = state.block_ids[0]
        cfg = self.model.mi355x_config
        # -1 padding lets a kernel skip pad entries; both conventions are accepted by the library,
        # which masks by context length and never reads pads
        pad = -1 if (cfg.attn_tkg_nki_kernel_enabled or cfg.attn_block_tkg_nki_kernel_enabled) \
            else self._BLOCK_TABLE_PAD
        data.block_tables.append(self._padded_block_table(block_table, pad))
        position = self._get_last_token_position(state)
        bs = self.cache_config.block_size
        data.slot_mapping.append(self._decode_slots(block_table, position))

    def _finalize_continuous_batching_inputs(self, data: IntermediateInputData, is_prefill: bool):
        max_model_len = self.scheduler_config.max_model_len
        if is_prefill:
            max_seq_len = max(data.full_context_lens)
            assert max_seq_len > 0
            input_tokens = make_tensor_with_pad(data.input_tokens, pad=0, max_len=max_seq_len, dtype=torch.long,
                                                device=self.device)
            position_ids = make_tensor_with_pad(data.position_ids, pad=0, max_len=max_seq_len, dtype=torch.long,
                                                device=self.device)
            slot_mapping = torch.full((len(data.slot_mapping), max_model_len), self._SLOT_MAPPING_PAD,
                                      dtype=torch.long)
            for i, s in enumerate(data.slot_mapping):
                slot_mapping[i, :len(s)] = torch.as_tensor(s, dtype=torch.long)
        else:
            input_tokens = make_tensor_with_pad(data.input_tokens, pad=0, max_len=1, dtype=torch.long,
                                                device=self.device)
            position_ids = make_tensor_with_pad(data.position_ids, pad=0, max_len=1, dtype=torch.long,
                                                device=self.device)
            slot_mapping = torch.tensor(data.slot_mapping, dtype=torch.long)
        block_tables = torch.stack(data.block_tables) if data.block_tables else torch.tensor([], dtype=torch.long)
        input_block_ids = torch.tensor(data.input_block_ids, dtype=torch.long)
        full_context_lens = torch.tensor(data.full_context_lens, dtype=torch.long).reshape(-1, 1)
        computed_context_lens = torch.tensor(data.computed_context_lens, dtype=torch.long).reshape(-1, 1)
        return ModelInputForMI355X(
            request_ids=data.request_ids, input_tokens=input_tokens, position_ids=position_ids,
            input_block_ids=input_block_ids, slot_mapping=slot_mapping, block_tables=block_tables,
            full_context_lens=full_context_lens, computed_context_lens=computed_context_lens,
            prefill_completion_state=None, sampling_params=self._sampling_params_if_used(input_tokens),
            multi_modal_kwargs=data.multi_modal_kwargs, adapter_ids=None)

    # ---- sampling ---------------------------------------------------------------------------
    def _sample(self, hidden_states: torch.Tensor, model_input: ModelInputForMI355X):
        # rows come back in model_input.request_ids order; the sampler wants input_batch order
        if list(model_input.request_ids) != list(self.input_batch.req_ids):
            order = {rid: i for i, rid in enumerate(model_input.request_ids)}
            reorder = torch.tensor([order[rid] for rid in self.input_batch.req_ids], dtype=torch.long)
            hidden_states = hidden_states[reorder]     # (a [B, V] copy: only when the orders differ)
        try:
            if self.model.mi355x_config.on_device_sampling_config is None:
                out = self._cpu_sample(hidden_states, model_input)
            else:
                out = self.model.sample(logits=hidden_states)
            if model_input.prefill_completion_state is not None:
                # chunked prefill: a request whose prompt is not fully encoded yet produces no token
                # (reference runner.py:1060-1063 marks those rows -1; -1 rows are stripped downstream)
                done = {rid: bool(f) for rid, f in zip(model_input.request_ids, model_input.prefill_completion_state.tolist())}
                ids = out.sampled_token_ids.clone()
                for row, rid in enumerate(self.input_batch.req_ids):
                    if not done[rid]:
                        ids[row] = -1
                out = SamplerOutput(sampled_token_ids=ids, logprobs_tensors=out.logprobs_tensors)
            return out
        except Exception as e:
            logger.error("Sampling failed for requests %s: %s", model_input.request_ids, e)
            raise RuntimeError(f"Sampling operation failed: {str(e)}") from e

    def _sampling_params_if_used(self, input_ids: torch.Tensor):
        """The (top_k, top_p, temperature) rows only feed the on-device sampler; the CPU-sampling
        path (where the reference passes them along unused) skips building them every step."""
        if self.model.mi355x_config.on_device_sampling_config is None:
            return None
        return self.get_mi355x_sampling_params(input_ids)

    def get_mi355x_sampling_params(self, input_ids: torch.Tensor) -> torch.Tensor:
        """Per-request (top_k, top_p, temperature) rows, packed like the reference's
        get_nxd_sampling_params (runner.py:1106-1140): greedy requests become (1, p, 1.0).
        Unused by the CPU-sampling path; consumed by mi_forward_tokens when
        on_device_sampling_config is set."""
        n = self.scheduler_config.max_num_seqs
        max_topk = min(self.model_config.get_vocab_size(), self._MAX_DEVICE_SAMPLING_TOP_K)
        params = torch.ones(n, 3, dtype=torch.float32)
        for i, request in enumerate(self.requests.values()):
            sp = request.sampling_params
            top_k = sp.top_k if 0 < (sp.top_k or 0) < max_topk else max_topk
            temperature = sp.temperature
            if temperature == 0.0:
                top_k, temperature = 1, 1.0
            params[i] = torch.tensor([float(top_k), float(sp.top_p), float(temperature)])
        if not self.is_chunked_prefill and input_ids.shape[0] != n:
            params = params[:input_ids.shape[0]]
        return params

    def _cpu_sample(self, logits: torch.Tensor, model_input: ModelInputForMI355X) -> SamplerOutput:
        try:
            if logits.dim() != 2:
                raise ValueError("Expected logits to be 2D tensor [batch_size, vocab_size], "
                                 f"got {logits.dim()}D tensor with shape {logits.shape}")
            vocab_size, expected = logits.shape[1], self.model_config.get_vocab_size()
            if vocab_size != expected:
                raise ValueError(f"Logits vocab size {vocab_size} does not match model vocab size {expected}")
            sampling_metadata = self.input_batch.sampling_metadata
            if sampling_metadata is None:
                raise RuntimeError("CPU sampling requires sampling metadata, but InputBatch.sampling_metadata "
                                   "is None. This indicates an issue with batch preparation.")
            sampler_output = self.cpu_sampler(logits, sampling_metadata)
            if sampler_output is None:
                raise RuntimeError("CPU sampler returned None output")
            if sampler_output.sampled_token_ids is None:
                raise RuntimeError("CPU sampler returned None sampled_token_ids")
            return sampler_output
        except Exception as e:
            logger.error("CPU sampling failed: %s (requests %s)", e, model_input.request_ids)
            raise RuntimeError(f"CPU sampling failed: {str(e)}") from e

    def take_draft_token_ids(self) -> DraftTokenIds | None:
        if self._draft_token_ids is None:
            return None
        req_ids, ids = self.input_batch.req_ids, self._draft_token_ids
        self._draft_token_ids = None
        return DraftTokenIds(req_ids, ids)
