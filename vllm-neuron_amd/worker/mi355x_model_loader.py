# SPDX-License-Identifier: Apache-2.0
"""Model adapter + config derivation for the MI355X backend.

Mirror of the reference's loader
(/root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py): the same config
dictionary (keys, defaults, override merge, ``pa_num_blocks`` <-> ``num_gpu_blocks_override``
reconciliation, validation errors) and the same adapter surface (``forward`` returning the
last-token logits for the CPU-sampling path, ``sample``, ``load_weights``), but the object
behind it is a ``NativeModel`` (libmi355x_vllm.so) instead of an NxDI model.

Override channel: ``vllm_config.additional_config["override_mi355x_config"]`` (the reference's
``"override_neuron_config"`` key is accepted as an alias so existing launch scripts keep
working).  Extra keys understood here:
  context_encoding_buckets   list[int]      workspace / graph buckets (README.md:80 semantics)
  synthetic_weights          {"seed","std"} random weights generated on the device (bench)
  state_dict                 dict[str, Tensor] in-memory HF-named weights (tests)
  use_graphs                 bool           hipGraph capture of token-generation steps
  tp_transport               "p2p" | "rccl" exchange between the rank shards of tensor_parallel_size > 1
  prefill_fp8_activations    bool           context-encoding GEMMs on FP8 weights quantize their bf16
                                            inputs per token to e4m3 and run on the MX-scaled MFMA
"""

from __future__ import annotations

import glob
import logging
import os
from contextlib import contextmanager
from math import ceil
from types import SimpleNamespace
from typing import Any

import torch
import torch.nn as nn

from .._vllm_compat import SamplerOutput
from .constants import (MI355X_MULTI_MODAL_MODELS, SUPPORTED_ARCHITECTURES,
                        TORCH_DTYPE_TO_MI355X_AMP)

logger = logging.getLogger(__name__)

OVERRIDE_KEYS = ("override_mi355x_config", "override_neuron_config")
_QUANT_DTYPES = {"int8": 2, "f8e4m3": 1, "fp8": 1, "float8_e4m3fn": 1}
_QUANT_TYPES = {"per_tensor_symmetric": 0, "per_channel_symmetric": 1}


class MI355XConfig(SimpleNamespace):
    """Attribute view of the merged config dict (plays the role of NxDI's NeuronConfig)."""

    def get(self, key, default=None):
        return getattr(self, key, default)


class MI355XModelBase(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.hf_config = config
        self.model = None                    # NativeModel, set by load_weights
        self.mi355x_config: MI355XConfig | None = None
        self.is_reorder_needed: bool = False
        self.architecture: str = ""
        self.num_key_value_heads: int = 0
        self.head_dim: int = 0
        # on-device sampling: the draw for call n is a pure function of (seed, n, row)
        self._sample_seed: int = 0
        self._sample_calls: int = 0

    # the reference exposes the merged config as `.neuron_config`; keep that spelling alive
    @property
    def neuron_config(self):
        return self.mi355x_config

    def forward(self, input_ids, positions, input_block_ids, sampling_params, **kwargs):
        raise NotImplementedError

    def sample(self, logits: torch.Tensor) -> SamplerOutput | None:
        raise NotImplementedError

    def load_weights(self, model_name_or_path: str, architecture: str, **kwargs):
        raise NotImplementedError

    @contextmanager
    def _reordered(self, input_block_ids: torch.Tensor, **inputs):
        """Yield (seq ids, inputs, restore).  With contiguous KV the reference must present
        sequences in seq-id order (loader.py:110-133); the block-table addressing used here
        does not depend on row order, but the contract (sort, run, restore) is kept."""
        if self.is_reorder_needed:
            sorted_ids, sorted_indices = torch.sort(input_block_ids)
            reordered = self._sort_inputs(inputs, sorted_indices)

            def restore(output: torch.Tensor) -> torch.Tensor:
                if sorted_ids.shape[0] != 1:
                    return torch.index_select(output, 0, torch.argsort(sorted_indices))
                return output

            yield sorted_ids, reordered, restore
        else:
            yield input_block_ids, inputs, lambda x: x

    @staticmethod
    def _sort_inputs(inputs: dict[str, Any], sorted_indices: torch.Tensor) -> dict[str, Any]:
        n = sorted_indices.shape[0]
        out = {}
        for key, val in inputs.items():
            if isinstance(val, torch.Tensor) and val.shape[0] > 0 and val.shape[0] == n:
                out[key] = torch.index_select(val, 0, sorted_indices)
            elif isinstance(val, list):
                out[key] = [val[i.item()] for i in sorted_indices]
            else:
                out[key] = val      # empty tensors, mismatched batch (prefill-only inputs), scalars
        return out


class MI355XCausalLM(MI355XModelBase):
    """`forward` = one call of libmi355x_vllm's mi_forward (reference loader.py:336-365)."""

    draft = None     # NativeModel of the draft model when fused speculation is on
    _draft_catchup: dict = {}    # sequence id -> (position of its next step, token the draft has not seen)

    def _remask_fused_spec_output(self, fused, inputs):
        """The fused speculation step hands back (reference loader.py:308-333)
            fused[0]  = the accepted tokens, [B, T], padded with 0
            fused[-1] = the position ids the sequences continue from.
        Token id 0 is a real token, so the padding is turned into -1 here -- everything behind the
        number of tokens generated in this step, next position minus the position of the token
        that was fed in -- and the runner strips the -1 entries."""
        accepted = fused[0]
        next_pos = fused[-1]
        next_pos = next_pos.squeeze(-1) if next_pos.dim() > 1 else next_pos
        fed = inputs["position_ids"][:, -1].to(next_pos.device)
        width = accepted.shape[1]
        counts = (next_pos - fed).to(torch.long).clamp_(0, width)
        columns = torch.arange(width, device=accepted.device)[None, :]
        return torch.where(columns < counts[:, None], accepted, torch.full_like(accepted, -1))

    def _forward_fused_speculation(self, ids, seq_ids, inputs, block_table, slot_mapping, computed):
        """Fused speculation (reference loader.py:349-355: the model call returns accepted tokens in
        place of logits).  Context encoding runs the target (its greedy token is the one generated)
        and the draft (to fill its K/V); token generation is ONE mi_forward_spec call."""
        cfg = self.mi355x_config
        k = int(cfg.speculation_length)
        sp = inputs.get("sampling_params")
        sampled = sp is not None and bool((sp[:, 0] != 1).any())
        positions = inputs["position_ids"]
        B = ids.shape[0]
        if sampled:
            # Acceptance here is by greedy agreement.  A step with a request that SAMPLES (top_k / top_p /
            # temperature; the reference's EAGLE test runs top_k = 50) generates one token per sequence with
            # the ordinary on-device sampler; the draft sees the same tokens so its K/V stays usable for the
            # greedy steps that follow.
            self._sample_calls += 1
            first = self.model.forward_tokens(ids, positions, seq_ids, block_table, slot_mapping[:, :ids.shape[1]]
                                              if ids.shape[1] == 1 else slot_mapping,
                                              inputs["full_context_lens"], computed, sampling_params=sp,
                                              seed=(self._sample_seed << 32) + self._sample_calls)
            self.draft.forward_tokens(ids, positions, seq_ids, block_table, slot_mapping[:, :ids.shape[1]]
                                      if ids.shape[1] == 1 else slot_mapping,
                                      inputs["full_context_lens"], computed, sampling_params=None, seed=0)
            accepted = torch.zeros(B, k, dtype=torch.long)
            accepted[:, 0] = first.reshape(B)
            next_pos = inputs["full_context_lens"].reshape(B).to(torch.long)
            for sid in seq_ids.reshape(-1).tolist():
                self._draft_catchup.pop(int(sid), None)
        elif ids.shape[1] > 1:      # context encoding
            first = self.model.forward_tokens(ids, positions, seq_ids, block_table, slot_mapping,
                                              inputs["full_context_lens"], computed, sampling_params=None, seed=0)
            self.draft.forward_tokens(ids, positions, seq_ids, block_table, slot_mapping,
                                      inputs["full_context_lens"], computed, sampling_params=None, seed=0)
            accepted = torch.zeros(B, k, dtype=torch.long)
            accepted[:, 0] = first.reshape(B)
            next_pos = inputs["full_context_lens"].reshape(B).to(torch.long)
            for sid in seq_ids.reshape(-1).tolist():
                self._draft_catchup.pop(int(sid), None)
        else:
            # the draft runs k - 1 steps: after a step that generated all k tokens, the one in front of the
            # last has not been through the draft -- it rides on the draft's first step of this call
            sids = [int(x) for x in seq_ids.reshape(-1).tolist()]
            pos_l = positions[:, 0].tolist()
            catch = []
            for sid, p in zip(sids, pos_l):
                st = self._draft_catchup.pop(sid, None)
                catch.append(st[1] if st is not None and st[0] == p else -1)
            accepted, next_pos = self.model.forward_spec(self.draft, ids[:, 0], positions[:, 0], block_table, k,
                                                         catchup_ids=torch.tensor(catch, dtype=torch.long))
            if k >= 2:
                for row, (sid, p) in enumerate(zip(sids, pos_l)):
                    if int(next_pos[row]) - p == k:
                        self._draft_catchup[sid] = (p + k, int(accepted[row, k - 2]))
        return self._remask_fused_spec_output([accepted, next_pos.reshape(B, 1)], inputs)

    def forward(self, input_ids, input_block_ids, **kwargs):
        cfg = self.mi355x_config
        if cfg.get("chunked_prefill_config") is not None:
            return self._forward_chunked(input_ids, **kwargs)
        with self._reordered(input_block_ids, input_ids=input_ids, **kwargs) as (seq_ids, inputs, restore):
            ids = inputs["input_ids"]
            if cfg.is_block_kv_layout:
                block_table, slot_mapping = inputs["block_tables"], inputs["slot_mapping"]
            else:
                block_table, slot_mapping = self._batch_line_addressing(seq_ids, inputs["position_ids"],
                                                                        inputs["full_context_lens"],
                                                                        inputs.get("computed_context_lens"),
                                                                        ids.shape[1])
            computed = inputs.get("computed_context_lens")
            if computed is None or computed.numel() == 0:
                # contiguous KV: prefill computes everything, decode everything but the new token
                full = inputs["full_context_lens"].reshape(-1)
                computed = torch.zeros_like(full) if ids.shape[1] > 1 else full - 1
            if cfg.get("enable_fused_speculation"):
                return restore(self._forward_fused_speculation(ids, seq_ids, inputs, block_table, slot_mapping, computed))
            if cfg.on_device_sampling_config:
                # the model returns sampled ids in place of logits (reference loader.py:350-356);
                # sampling_params rows = (top_k, top_p, temperature), greedy rewritten to top_k = 1
                self._sample_calls += 1
                tokens = self.model.forward_tokens(ids, inputs["position_ids"], seq_ids, block_table, slot_mapping,
                                                   inputs["full_context_lens"], computed,
                                                   sampling_params=inputs.get("sampling_params"),
                                                   seed=(self._sample_seed << 32) + self._sample_calls)
                return restore(tokens)
            # the runner samples from the logits before the next call: they may alias the library's
            # pinned buffer (no 2 MB host copy / allocation per step)
            logits = self.model.forward(ids, inputs["position_ids"], seq_ids, block_table, slot_mapping,
                                        inputs["full_context_lens"], computed, alias_ok=True)
            return restore(logits)

    def _forward_chunked(self, input_ids, **inputs):
        """One ragged token batch [1, sum S] (reference loader.py:339-361 with is_chunked_prefill):
        logits (or sampled ids) of EVERY request's last scheduled token, in request order.  The
        reference keeps only the rows whose prefill is complete (`prefill_completion_state`); here
        all rows come back and the runner drops the incomplete ones after sampling, so the row
        count always equals the number of scheduled requests."""
        assert inputs.get("prefill_completion_state") is not None
        cfg = self.mi355x_config
        args = (input_ids, inputs["position_ids"], inputs["slot_mapping"], inputs["block_tables"],
                inputs["full_context_lens"], inputs["computed_context_lens"])
        if cfg.on_device_sampling_config:
            self._sample_calls += 1
            return self.model.forward_chunked(*args, sampling_params=inputs.get("sampling_params"),
                                              seed=(self._sample_seed << 32) + self._sample_calls, tokens=True)
        return self.model.forward_chunked(*args)

    def _batch_line_addressing(self, seq_ids, position_ids, full_context_lens, computed, S):
        """Contiguous ('batch line') KV expressed through the block pool: sequence id s owns
        block s + 1, whose size is the (padded) max_model_len."""
        bs = self.native_block_size
        B = seq_ids.shape[0]
        block_table = (seq_ids.reshape(B, 1) + 1).to(torch.long)
        full = full_context_lens.reshape(-1)
        if S == 1:
            pos = position_ids.reshape(B, 1)
            slots = block_table * bs + pos
        else:
            ar = torch.arange(S, dtype=torch.long)[None, :].expand(B, S)
            slots = torch.where(ar < full[:, None], block_table * bs + ar, torch.full_like(ar, -1))
        return block_table, slots

    def sample(self, logits: torch.Tensor) -> SamplerOutput | None:
        if self.mi355x_config.on_device_sampling_config:
            return SamplerOutput(sampled_token_ids=logits.unsqueeze(-1), logprobs_tensors=None)
        raise RuntimeError("CPU sampling should be handled by the model runner, not the model. "
                           "This indicates a bug in the sampling path routing.")

    # ---- weights ------------------------------------------------------------------------
    def load_weights(self, model_name_or_path: str, architecture: str, **kwargs):
        from .._native import NativeModel
        cfg: dict = kwargs["mi355x_config"]
        hf = self.hf_config
        geo = _decoder_geometry(hf)
        max_model_len = cfg["seq_len"]
        if cfg["is_block_kv_layout"]:
            block_size, num_blocks = cfg["pa_block_size"], cfg["pa_num_blocks"]
        else:
            block_size = -(-max_model_len // 32) * 32
            num_blocks = cfg["batch_size"] + 1
        self.native_block_size = block_size
        quantized = bool(cfg.get("quantized"))
        qdtype = cfg.get("quantization_dtype", "int8")
        qtype = cfg.get("quantization_type", "per_tensor_symmetric")
        if quantized and (qdtype not in _QUANT_DTYPES or qtype not in _QUANT_TYPES):
            raise ValueError(f"unsupported quantization_dtype/type: {qdtype!r}/{qtype!r}")
        not_converted = cfg.get("modules_to_not_convert") or []
        tp_degree = int(cfg["tp_degree"])
        tp_devices = list(kwargs.get("tp_device_ids") or [])
        if tp_degree > 1 and len(tp_devices) != tp_degree:
            raise RuntimeError(f"tp_degree {tp_degree} needs {tp_degree} device ids, got {tp_devices}")
        from .._native import MI_TP_ALL_RANKS, MI_TP_TRANSPORT
        buckets = list(cfg.get("context_encoding_buckets") or [])
        max_num_seqs = int(cfg["batch_size"])
        chunked = cfg.get("chunked_prefill_config")
        if chunked:
            # the reference compiles a batch-1 model over max_num_batched_tokens (loader.py:734-736);
            # the ragged batch here holds up to chunked_prefill_config.max_num_seqs requests
            max_num_seqs = int(getattr(chunked, "max_num_seqs", None) or cfg.get("scheduler_max_num_seqs") or max_num_seqs)
            buckets = sorted(set(buckets + [int(cfg["max_context_length"])]))
        speculative_config = kwargs.get("speculative_config")
        fused_spec = bool(cfg.get("enable_fused_speculation"))
        spec_len = int(cfg.get("speculation_length") or 0) if fused_spec else 0
        if fused_spec:
            if spec_len < 1 or speculative_config is None or chunked or not cfg["is_block_kv_layout"]:
                raise NotImplementedError("fused speculation needs num_speculative_tokens >= 1, a draft model, the block "
                                          "KV layout (prefix caching on) and no chunked prefill")
        self.model = NativeModel(
            num_blocks=int(num_blocks), block_size=int(block_size),
            # the target scores every sequence's speculation window in one token-generation pass
            max_num_seqs=max_num_seqs * max(spec_len, 1), max_model_len=int(max_model_len),
            ctx_buckets=buckets,
            weight_dtype=_QUANT_DTYPES[qdtype] if quantized else 0,
            quant_type=_QUANT_TYPES[qtype] if quantized else 0,
            quantize_lm_head=int(quantized and not any("lm_head" in m for m in not_converted)),
            # every rank shard lives inside this one context (the reference's single-worker model,
            # platform.py:166-167): one GPU and one library thread per shard
            tp_degree=tp_degree, tp_rank=MI_TP_ALL_RANKS if tp_degree > 1 else 0,
            tp_device_ids=tp_devices if tp_degree > 1 else [],
            tp_transport=MI_TP_TRANSPORT[cfg.get("tp_transport", "p2p")],
            device_id=int(kwargs.get("device_id", 0)),
            use_graphs=int(cfg.get("use_graphs", 1)),
            prefill_fp8_activations=int(bool(cfg.get("prefill_fp8_activations", False))),
            **geo)
        if fused_spec:
            self._load_draft(speculative_config, cfg, num_blocks, block_size, max_num_seqs, max_model_len, buckets,
                             quantized, qdtype, qtype, not_converted, kwargs)
        synthetic = cfg.get("synthetic_weights")
        state_dict = cfg.get("state_dict")
        # Device-ready weight images on disk (quantized, tiled, sharded), the counterpart of the
        # reference's compiled-artifact directory (loader.py:160-226): try them first; on a miss take
        # the checkpoint the long way (read, quantize, tile) and leave the images for the next start.
        artifacts = self._artifact_dir(model_name_or_path, cfg, geo, quantized, qdtype, qtype, tp_degree)
        identity = self._checkpoint_identity(model_name_or_path, cfg)
        if artifacts is not None:
            try:
                self._check_artifact_identity(artifacts, identity)
                self.model.load_artifacts(artifacts)
                logger.info("Successfully loaded pre-built weight artifacts from %s", artifacts)
                self.compiled_artifacts_path, self.loaded_from_artifacts = artifacts, True
                return True, artifacts
            except (FileNotFoundError, ValueError) as e:
                logger.warning("Exception: %s", e)
                logger.warning("Unable to find pre-built weight artifacts under %s. Rebuilding...", artifacts)
        if synthetic is not None:
            self.model.init_synthetic_weights(int(synthetic.get("seed", 1)), float(synthetic.get("std", 0.02)))
        elif state_dict is not None:
            self.model.load_state_dict(state_dict)
        else:
            self._load_safetensors_dir(model_name_or_path)
        self.compiled_artifacts_path, self.loaded_from_artifacts = artifacts, False
        if artifacts is not None:
            # the weights are resident: a directory that cannot be written (read-only model cache, full disk) costs the
            # next start its shortcut, not this one its model
            try:
                self.model.save_artifacts(artifacts)
                self._write_artifact_identity(artifacts, identity)
                logger.info("Saved weight artifacts to %s", artifacts)
            except (OSError, ValueError, RuntimeError) as e:
                logger.warning("Could not save weight artifacts under %s (%s); continuing without them", artifacts, e)
                self.compiled_artifacts_path = None
                return False, None
        return False, artifacts

    _IDENTITY_FILE = "checkpoint.json"

    @staticmethod
    def _checkpoint_identity(model_name_or_path, cfg):
        """What tells one checkpoint of a geometry from another (base / instruct): the safetensors files' names, sizes and
        mtimes for a local directory; names, shapes and a fingerprint of the leading bytes for an in-memory state dict;
        the seed for synthetic weights.  None when there is no checkpoint to compare with (the artifacts are all there is)."""
        import hashlib
        synthetic, state_dict = cfg.get("synthetic_weights"), cfg.get("state_dict")
        if synthetic is not None:
            return {"synthetic": [int(synthetic.get("seed", 1)), float(synthetic.get("std", 0.02))]}
        if state_dict is not None:
            if not state_dict:
                return None
            h = hashlib.md5()
            for name in sorted(state_dict):
                t = state_dict[name]
                h.update(f"{name}:{tuple(t.shape)}:{t.dtype}".encode())
                h.update(t.detach().reshape(-1)[:64].to("cpu", torch.float32).numpy().tobytes())
            return {"state_dict": h.hexdigest()}
        if model_name_or_path and os.path.isdir(model_name_or_path):
            files = sorted(glob.glob(os.path.join(model_name_or_path, "*.safetensors")))
            if files:
                return {"files": [[os.path.basename(f), os.path.getsize(f), int(os.path.getmtime(f))] for f in files]}
        return None

    @classmethod
    def _check_artifact_identity(cls, artifacts, identity) -> None:
        """ValueError when the directory was built from another checkpoint than the one at hand.  The native header
        (mi_save_weights) holds geometry, quantization and sharding only."""
        import json
        path = os.path.join(artifacts, cls._IDENTITY_FILE)
        if identity is None or not os.path.isdir(artifacts):
            return                                # nothing to compare / nothing there (load_artifacts reports the latter)
        try:
            with open(path, "r", encoding="utf-8") as f:
                saved = json.load(f)
        except FileNotFoundError:
            raise ValueError(f"{artifacts} does not say which checkpoint it was built from ({cls._IDENTITY_FILE} missing)") from None
        if saved != identity:
            raise ValueError(f"{artifacts} was built from another checkpoint of this shape")

    @classmethod
    def _write_artifact_identity(cls, artifacts, identity) -> None:
        import json
        if identity is None:
            return
        with open(os.path.join(artifacts, cls._IDENTITY_FILE), "w", encoding="utf-8") as f:
            json.dump(identity, f)

    def _load_draft(self, speculative_config, cfg, num_blocks, block_size, max_num_seqs, max_model_len, buckets,
                    quantized, qdtype, qtype, not_converted, kwargs) -> None:
        """The draft model of fused speculation (reference loader.py:243-303: a clone of the target's
        configuration around the draft checkpoint, fused speculation switched off in the clone).  Same
        block ids as the target, a K/V pool of its own; quantized like the target unless
        draft_model_modules_to_not_convert says otherwise."""
        from .._native import NativeModel
        draft_cfg = speculative_config.draft_model_config
        geo = _decoder_geometry(draft_cfg.hf_config)
        if geo["vocab_size"] != self.hf_config.vocab_size:
            raise ValueError("fused speculation: draft and target must share the vocabulary")
        draft_skip = cfg.get("draft_model_modules_to_not_convert") or not_converted
        self._draft_catchup = {}
        self.draft = NativeModel(
            # rows: every sequence + one catch-up row each (mi_forward_spec)
            num_blocks=int(num_blocks), block_size=int(block_size), max_num_seqs=2 * max_num_seqs,
            max_model_len=int(max_model_len), ctx_buckets=buckets,
            weight_dtype=_QUANT_DTYPES[qdtype] if quantized else 0, quant_type=_QUANT_TYPES[qtype] if quantized else 0,
            quantize_lm_head=int(quantized and not any("lm_head" in m for m in draft_skip)),
            # the draft is not sharded: it runs on the GPU of the target's rank 0
            tp_degree=1, tp_rank=0, device_id=int((kwargs.get("tp_device_ids") or [kwargs.get("device_id", 0)])[0]),
            use_graphs=int(cfg.get("use_graphs", 1)),
            prefill_fp8_activations=int(bool(cfg.get("prefill_fp8_activations", False))), **geo)
        synthetic, state_dict = cfg.get("draft_synthetic_weights"), cfg.get("draft_state_dict")
        if synthetic is not None:
            self.draft.init_synthetic_weights(int(synthetic.get("seed", 2)), float(synthetic.get("std", 0.02)))
        elif state_dict is not None:
            self.draft.load_state_dict(state_dict)
        else:
            target = self.model
            try:
                self.model = self.draft            # _load_safetensors_dir fills self.model
                self._load_safetensors_dir(draft_cfg.model)
            finally:
                self.model = target

    @staticmethod
    def _artifact_dir(model_name_or_path, cfg, geo, quantized, qdtype, qtype, tp_degree):
        """Where this configuration's weight images live, or None for no caching.
          * quantized_checkpoints_path (loader.py:888-891): the directory the quantized weights are
            kept in -- used as given;
          * MI355X_COMPILED_ARTIFACTS (the reference's NEURON_COMPILED_ARTIFACTS, loader.py:198-199): as given;
          * a local checkpoint directory: <model>/mi355x-compiled-artifacts/<md5 of the configuration>
            (loader.py:186-210);
          * synthetic or in-memory weights without one of the two explicit paths: not cached."""
        import hashlib
        import json
        explicit = (cfg.get("quantized_checkpoints_path") if quantized else None) or os.getenv("MI355X_COMPILED_ARTIFACTS")
        if explicit:
            return str(explicit)
        if cfg.get("synthetic_weights") is not None or cfg.get("state_dict") is not None:
            return None
        if not (model_name_or_path and os.path.isdir(model_name_or_path)):
            return None
        files = sorted(glob.glob(os.path.join(model_name_or_path, "*.safetensors")))
        ident = [(os.path.basename(f), os.path.getsize(f), int(os.path.getmtime(f))) for f in files]
        key = json.dumps({"geo": geo, "quantized": quantized, "dtype": qdtype if quantized else "bf16",
                          "type": qtype if quantized else None, "tp": tp_degree,
                          "lm_head": not any("lm_head" in m for m in (cfg.get("modules_to_not_convert") or [])),
                          "checkpoint": ident}, sort_keys=True)
        hashed = hashlib.md5(key.encode("utf-8")).hexdigest()
        return os.path.join(model_name_or_path, "mi355x-compiled-artifacts", hashed)

    def _load_safetensors_dir(self, path: str) -> None:
        from safetensors import safe_open
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(
                f"no *.safetensors under {path!r}: the MI355X plugin loads local HF checkpoints "
                "(no hub access); use override_mi355x_config['synthetic_weights'] for benchmarks")
        for f in files:
            with safe_open(f, "pt") as sf:
                for name in sf.keys():
                    t = sf.get_tensor(name)
                    if t.dim() in (1, 2):
                        self.model.load_weight(name, t)


def _decoder_geometry(hf) -> dict:
    """HF config -> mi_model_config geometry fields."""
    head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
    rp = getattr(hf, "rope_parameters", None) or {}
    rs = getattr(hf, "rope_scaling", None) or rp
    rope_type = (rs or {}).get("rope_type", (rs or {}).get("type", "default"))
    if rope_type not in ("default", "llama3", None):
        raise NotImplementedError(f"rope scaling {rope_type!r} is not supported on MI355X yet")
    llama3 = rope_type == "llama3"
    return dict(
        num_layers=hf.num_hidden_layers, hidden_size=hf.hidden_size, num_heads=hf.num_attention_heads,
        num_kv_heads=hf.num_key_value_heads, head_dim=int(head_dim), intermediate_size=hf.intermediate_size,
        vocab_size=hf.vocab_size, rms_norm_eps=float(hf.rms_norm_eps),
        rope_theta=float(rp.get("rope_theta", getattr(hf, "rope_theta", 10000.0))),
        rope_type=int(llama3), rope_factor=float(rs["factor"]) if llama3 else 1.0,
        rope_low_freq_factor=float(rs["low_freq_factor"]) if llama3 else 1.0,
        rope_high_freq_factor=float(rs["high_freq_factor"]) if llama3 else 4.0,
        rope_original_max_position=int(rs["original_max_position_embeddings"]) if llama3 else 0,
        qkv_bias=int(getattr(hf, "model_type", "") == "qwen2" or bool(getattr(hf, "attention_bias", False))),
        tie_word_embeddings=int(bool(getattr(hf, "tie_word_embeddings", False))))


def _get_model_configs(config) -> tuple[str, int, int]:
    archs = getattr(config, "architectures", [])
    if not archs:
        raise ValueError("No architectures specified in the pretrained config.")
    architecture = archs[0]
    if architecture in MI355X_MULTI_MODAL_MODELS:
        config = getattr(config, "text_config", None)
    num_key_value_heads = getattr(config, "num_key_value_heads", None)
    head_dim = getattr(config, "head_dim", None)
    if not head_dim:
        num_attention_heads = getattr(config, "num_attention_heads", None)
        hidden_size = getattr(config, "hidden_size", None)
        if num_attention_heads and hidden_size:
            head_dim = hidden_size // num_attention_heads
    if not num_key_value_heads or not head_dim:
        raise ValueError("Missing required fields in the pretrained config.")
    return architecture, int(num_key_value_heads), int(head_dim)


def _check_architecture(architecture: str) -> None:
    if architecture not in SUPPORTED_ARCHITECTURES:
        raise ValueError(f"Model {architecture} is not supported on MI355X for now. "
                         f"Supported models: {list(SUPPORTED_ARCHITECTURES)}")


def get_override_config(additional_config) -> dict | None:
    for key in OVERRIDE_KEYS:
        val = (additional_config or {}).get(key)
        if val is not None:
            return val
    return None


def get_mi355x_model(model_config, cache_config, parallel_config, scheduler_config, lora_serving_config,
                     speculative_config=None, additional_config: Any | None = None, **native_kwargs) -> nn.Module:
    architecture, num_key_value_heads, head_dim = _get_model_configs(model_config.hf_config)
    if architecture in MI355X_MULTI_MODAL_MODELS:
        raise NotImplementedError(f"{architecture}: multimodal models are not supported on the MI355X plugin")
    _check_architecture(architecture)
    if lora_serving_config:
        raise NotImplementedError("Multi-lora is not yet supported on the MI355X plugin")
    if speculative_config is not None and getattr(speculative_config, "method", None) == "eagle":
        raise NotImplementedError("EAGLE drafts are not supported on the MI355X plugin (plain draft models are)")

    model = MI355XCausalLM(model_config.hf_config)
    default_args = _get_default_mi355x_config(model_config, cache_config, parallel_config, scheduler_config,
                                              lora_serving_config, speculative_config)
    override = get_override_config(additional_config)
    if override is not None:
        logger.info("override_mi355x_config keys: %s", sorted(override))
    override = dict(override) if override is not None else None
    cfg = _get_mi355x_config_after_override(default_args, override)
    if cfg.get("is_block_kv_layout"):
        cfg = _handle_pa_num_blocks(cache_config, cfg, override)
    cfg = _validate_mi355x_config(cache_config, scheduler_config, cfg)
    cfg["scheduler_max_num_seqs"] = scheduler_config.max_num_seqs

    if cfg.get("enable_fused_speculation") and not cfg.get("on_device_sampling_config"):
        # the fused step returns token ids, never logits (reference loader.py:349-355)
        logger.info("fused speculation: on-device sampling switched on (the step returns accepted token ids)")
        cfg["on_device_sampling_config"] = {"dynamic": True}
    model.load_weights(model_name_or_path=model_config.model, architecture=architecture, mi355x_config=cfg,
                       speculative_config=speculative_config, **native_kwargs)
    cfg.pop("state_dict", None)
    cfg.pop("draft_state_dict", None)
    model.mi355x_config = MI355XConfig(**{"attn_tkg_nki_kernel_enabled": False,
                                          "attn_block_tkg_nki_kernel_enabled": False,
                                          "chunked_prefill_config": None, **cfg})
    model.architecture = architecture
    model.num_key_value_heads = num_key_value_heads
    model.head_dim = head_dim
    return model.eval()


def _get_default_mi355x_config(model_config, cache_config, parallel_config, scheduler_config,
                               lora_serving_config, speculative_config) -> dict:
    """Same keys and defaults as the reference's `_get_default_neuron_config`
    (loader.py:725-793), except `on_device_sampling_config`, which defaults to None here (the
    reference defaults to OnDeviceSamplingConfig(dynamic=True, deterministic=False)): CPU sampling
    is the parity path of record; any truthy value (e.g. {"dynamic": True}) selects the on-device
    sampler (mi_forward_tokens)."""
    if scheduler_config.chunked_prefill_enabled:
        batch_size = 1
        max_context_length = scheduler_config.max_num_batched_tokens
    else:
        batch_size = scheduler_config.max_num_seqs
        max_context_length = scheduler_config.max_model_len

    default_num_blocks = ceil(scheduler_config.max_model_len // cache_config.block_size) * scheduler_config.max_num_seqs
    if cache_config.num_gpu_blocks_override is not None:
        default_num_blocks = cache_config.num_gpu_blocks_override

    speculation = {}
    if speculative_config is not None:      # reference loader.py:785-791
        speculation = {"enable_fused_speculation": True,
                       "speculation_length": getattr(speculative_config, "num_speculative_tokens", 0)}
        if getattr(speculative_config, "method", None) == "eagle":
            speculation["enable_eagle_speculation"] = True
    return {
        **speculation,
        "tp_degree": parallel_config.tensor_parallel_size,
        "ctx_batch_size": 1,
        "batch_size": batch_size,
        "max_context_length": max_context_length,
        "seq_len": scheduler_config.max_model_len,
        "enable_bucketing": True,
        "is_continuous_batching": (batch_size > 1),
        "quantized": False,
        "torch_dtype": TORCH_DTYPE_TO_MI355X_AMP[model_config.dtype],
        "padding_side": "right",
        "on_device_sampling_config": None,
        "lora_config": lora_serving_config,
        "pa_num_blocks": default_num_blocks,
        "pa_block_size": cache_config.block_size,
        "is_block_kv_layout": (scheduler_config.chunked_prefill_enabled or cache_config.enable_prefix_caching),
        "is_prefix_caching": cache_config.enable_prefix_caching,
    }


def _handle_pa_num_blocks(cache_config, mi355x_config: dict, override_config: dict | None) -> dict:
    """Keep vLLM's block count and the pool's in step (reference loader.py:796-831): vLLM saw
    N + 1 (null block); an explicit pa_num_blocks must equal the user's N and is bumped too."""
    explicit = bool(override_config) and "pa_num_blocks" in override_config
    if cache_config.num_gpu_blocks_override is not None:
        pa_num_blocks = mi355x_config.get("pa_num_blocks")
        user_blocks = cache_config.num_gpu_blocks_override - 1
        if explicit:
            if pa_num_blocks == user_blocks:
                mi355x_config["pa_num_blocks"] = cache_config.num_gpu_blocks_override
            else:
                raise ValueError(
                    f"pa_num_blocks ({pa_num_blocks}) must match your --num-gpu-blocks-override intent"
                    f"({user_blocks}) to ensure vLLM and the MI355X KV pool have consistent block counts. ")
    elif explicit:
        raise ValueError(
            f"When setting pa_num_blocks ({mi355x_config.get('pa_num_blocks')}) in override_mi355x_config, "
            "you must also set --num-gpu-blocks-override to the same value to ensure vLLM and the MI355X KV "
            "pool have consistent block counts.")
    return mi355x_config


def _validate_mi355x_config(cache_config, scheduler_config, mi355x_config: dict) -> dict:
    if cache_config.enable_prefix_caching:
        assert mi355x_config.get("is_prefix_caching", False)
        assert mi355x_config.get("is_block_kv_layout", False)
    if scheduler_config.chunked_prefill_enabled:
        assert mi355x_config.get("chunked_prefill_config")
        assert mi355x_config.get("is_block_kv_layout", False)
    if mi355x_config.get("is_block_kv_layout"):
        min_blocks_required = ceil(scheduler_config.max_model_len / cache_config.block_size) * scheduler_config.max_num_seqs
        if cache_config.num_gpu_blocks_override is not None:
            effective_blocks = cache_config.num_gpu_blocks_override - 1
        else:
            effective_blocks = mi355x_config.get("pa_num_blocks")
        assert effective_blocks >= min_blocks_required, (
            f"At least {min_blocks_required} blocks are required for max_model_len "
            f"{scheduler_config.max_model_len}, but only {effective_blocks} blocks are available "
            "(user-intended blocks, excluding the +1 for null block)")
    assert "text_neuron_config" not in mi355x_config and "vision_neuron_config" not in mi355x_config, (
        "text/vision sub-configs belong to ImageToText models, which this backend does not implement")
    return mi355x_config


def _get_mi355x_config_after_override(default_config: dict, overridden: dict | None) -> dict:
    """Shallow merge with the reference's special cases (loader.py:870-900)."""
    overridden = overridden or {}
    cfg = overridden.pop("chunked_prefill_config", None)
    if cfg:
        overridden["chunked_prefill_config"] = SimpleNamespace(**cfg)
    default_config.update(overridden)
    default_config.pop("text_neuron_config", None)
    default_config.pop("vision_neuron_config", None)
    if "quantized" in overridden:
        default_config.update({
            "quantized": overridden.pop("quantized", False),
            "quantized_checkpoints_path": overridden.pop("quantized_checkpoints_path", None),
            "quantization_type": overridden.pop("quantization_type", "per_tensor_symmetric"),
            "quantization_dtype": overridden.pop("quantization_dtype", "int8"),
        })
    return default_config
