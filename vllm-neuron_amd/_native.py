"""ctypes binding of libmi355x_vllm.so (include/mi355x_vllm.h).

There is NO CPU fallback: importing this module without the built library, or
creating a context without a HIP device, raises.  torch is imported first on
purpose: the wheel bundles its own libamdhip64.so.7 / librccl.so.1 and the
library's DT_NEEDED entries must resolve to those already-loaded copies so that
the process holds ONE HIP runtime (torch device pointers are then valid here).
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355X_VLLM_LIB points at another build of the same library (A/B measurements)
LIB_PATH = os.environ.get("MI355X_VLLM_LIB") or os.path.join(_HERE, "csrc", "libmi355x_vllm.so")

MI_F32, MI_BF16 = 0, 1
MI_W = {"bfloat16": 0, "bf16": 0, None: 0, "f8e4m3": 1, "fp8": 1, "int8": 2}
MI_Q = {"per_tensor_symmetric": 0, "per_channel_symmetric": 1}
K_CLASSES = ("gemv", "gemm", "attn_decode", "attn_prefill", "other", "comm")


class MiModelConfig(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32), ("hidden_size", C.c_int32), ("num_heads", C.c_int32),
        ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("intermediate_size", C.c_int32),
        ("vocab_size", C.c_int32), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
        ("rope_type", C.c_int32), ("rope_factor", C.c_float), ("rope_low_freq_factor", C.c_float),
        ("rope_high_freq_factor", C.c_float), ("rope_original_max_position", C.c_int32),
        ("qkv_bias", C.c_int32), ("tie_word_embeddings", C.c_int32),
        ("num_blocks", C.c_int32), ("block_size", C.c_int32), ("max_num_seqs", C.c_int32),
        ("max_model_len", C.c_int32), ("num_ctx_buckets", C.c_int32), ("ctx_buckets", C.c_int32 * 8),
        ("weight_dtype", C.c_int32), ("quant_type", C.c_int32), ("quantize_lm_head", C.c_int32),
        ("tp_degree", C.c_int32), ("tp_rank", C.c_int32), ("device_id", C.c_int32),
        ("use_graphs", C.c_int32), ("prefill_fp8_activations", C.c_int32),
        ("tp_device_ids", C.c_int32 * 16), ("tp_transport", C.c_int32),
    ]


MI_TP_ALL_RANKS = -1          # tp_rank: every rank shard inside this process (include/mi355x_vllm.h)
MI_TP_TRANSPORT = {"p2p": 0, "rccl": 1}


class MiKvStats(C.Structure):
    _fields_ = [("kv_bytes", C.c_int64), ("weight_bytes", C.c_int64), ("workspace_bytes", C.c_int64),
                ("device_free_bytes", C.c_int64), ("device_total_bytes", C.c_int64),
                ("num_blocks", C.c_int32), ("block_size", C.c_int32),
                ("num_kv_heads_local", C.c_int32), ("head_dim", C.c_int32), ("num_layers", C.c_int32),
                ("block_table_rows_sent", C.c_int64), ("block_table_rows_kept", C.c_int64)]


class MiTpPlan(C.Structure):
    _fields_ = [("q_head0", C.c_int32), ("q_heads_real", C.c_int32), ("q_heads_local", C.c_int32),
                ("kv_head0", C.c_int32), ("kv_heads_local", C.c_int32), ("inter0", C.c_int32), ("inter_local", C.c_int32),
                ("vocab0", C.c_int32), ("vocab_local", C.c_int32)]


class MiTpInfo(C.Structure):
    _fields_ = [("tp_degree", C.c_int32), ("transport", C.c_int32), ("selftest", C.c_int32), ("graphs", C.c_int32),
                ("mode", C.c_int32), ("timeout_ms", C.c_int32), ("device_ids", C.c_int32 * 16), ("peer_access", C.c_int32 * 16)]


# include/mi355x_vllm.h: mi_allreduce_fn / mi_allgather_fn
MI_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
MI_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_SIGS = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_int, []),
    "mi_ctx_create": (C.c_int, [C.POINTER(MiModelConfig), C.POINTER(C.c_void_p)]),
    "mi_ctx_destroy": (C.c_int, [C.c_void_p]),
    "mi_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32,
                                 C.POINTER(C.c_int64), C.c_int32]),
    "mi_init_synthetic_weights": (C.c_int, [C.c_void_p, C.c_uint64, C.c_float]),
    "mi_save_weights": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mi_load_weights_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mi_set_num_blocks": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi_finalize": (C.c_int, [C.c_void_p]),
    "mi_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "mi_forward_tokens": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.c_void_p]),
    "mi_forward_chunked": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mi_forward_spec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_replay_decode": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]),
    "mi_replay_decode_classes": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_float)]),
    "mi_kv_stats": (C.c_int, [C.c_void_p, C.POINTER(MiKvStats)]),
    "mi_kv_bytes_per_block": (C.c_int64, [C.c_void_p]),
    "mi_stream": (C.c_void_p, [C.c_void_p]),
    "mi_logits_buffer": (C.c_void_p, [C.c_void_p]),
    "mi_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                  C.POINTER(C.c_double)]),
    "mi_tp_unique_id": (C.c_int, [C.c_void_p]),
    "mi_tp_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_tp_init_transport": (C.c_int, [C.c_void_p, MI_ALLREDUCE_FN, MI_ALLGATHER_FN, C.c_void_p]),
    "mi_tp_plan": (C.c_int, [C.POINTER(MiModelConfig), C.c_int32, C.POINTER(MiTpPlan)]),
    "mi_tp_info": (C.c_int, [C.c_void_p, C.POINTER(MiTpInfo)]),
    "mi_op_tp_all_reduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "mi_op_sample": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mi_op_sample_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "mi_op_sample_ws": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mi_op_quantize_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_op_untile_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "mi_op_qlinear": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_op_qlinear_a8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_op_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                C.c_void_p]),
    "mi_op_kv_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mi_op_attn_scratch_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "mi_op_paged_attn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_op_paged_attn_prefill": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def load_library() -> C.CDLL:
    """dlopen the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class NativeError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().mi_last_error().decode("utf-8", "replace")
        # -1 = MI_EINVAL: argument/shape errors surface as ValueError like the reference's config checks
        if rc == -1:
            raise ValueError(f"mi355x_vllm: {msg}")
        raise NativeError(f"mi355x_vllm error {rc}: {msg}")


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


class NativeModel:
    """Owns one mi_ctx: device weights, paged KV pool, activations, stream."""

    def __init__(self, **cfg):
        self.lib = load_library()
        c = MiModelConfig()
        buckets = list(cfg.pop("ctx_buckets", None) or [])
        tp_devices = list(cfg.pop("tp_device_ids", None) or [])
        if len(tp_devices) > 16:
            raise ValueError("tp_device_ids: at most 16 ranks")
        for i, d in enumerate(tp_devices):
            c.tp_device_ids[i] = int(d)
        for k, v in cfg.items():
            if not hasattr(c, k):
                raise ValueError(f"unknown native config field {k!r}")
            setattr(c, k, v)
        c.num_ctx_buckets = min(len(buckets), 8)
        for i, b in enumerate(buckets[:8]):
            c.ctx_buckets[i] = int(b)
        self.cfg = c
        self.vocab_size = c.vocab_size
        self._logits_view = None
        self._ctx = C.c_void_p()
        check(self.lib.mi_ctx_create(C.byref(c), C.byref(self._ctx)))

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.mi_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weight(self, name: str, tensor: "torch.Tensor") -> None:
        t = tensor.detach().cpu().contiguous()
        if t.dtype == torch.bfloat16:
            dt = MI_BF16
        else:
            t, dt = t.to(torch.float32), MI_F32
        shape = (C.c_int64 * t.dim())(*t.shape)
        check(self.lib.mi_load_weight(self._ctx, name.encode(), t.data_ptr(), dt, shape, t.dim()))

    def load_state_dict(self, weights: dict) -> None:
        for name, t in weights.items():
            if t.dim() in (1, 2):
                self.load_weight(name, t)

    def save_artifacts(self, directory: str) -> None:
        """Write this context's device weight images (one file per rank) under `directory`."""
        os.makedirs(directory, exist_ok=True)
        check(self.lib.mi_save_weights(self._ctx, os.fsencode(directory)))

    def load_artifacts(self, directory: str) -> None:
        """Load the device weight images saved by `save_artifacts`; ValueError when they are
        missing or were made for another configuration."""
        check(self.lib.mi_load_weights_file(self._ctx, os.fsencode(directory)))

    def init_synthetic_weights(self, seed: int = 1, std: float = 0.02) -> None:
        check(self.lib.mi_init_synthetic_weights(self._ctx, seed, std))

    def tp_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        check(self.lib.mi_tp_unique_id(buf))
        return buf.raw

    def tp_init(self, uid: bytes) -> None:
        assert len(uid) == 128
        check(self.lib.mi_tp_init(self._ctx, C.create_string_buffer(uid, 128)))

    def tp_init_transport(self, all_reduce, all_gather) -> None:
        """Collectives supplied by the caller instead of RCCL (include/mi355x_vllm.h:
        mi_tp_init_transport).  all_reduce(buf_ptr, count, stream_ptr) -> 0 / nonzero,
        all_gather(send_ptr, recv_ptr, count, stream_ptr) -> 0 / nonzero."""
        self._xport = (MI_ALLREDUCE_FN(lambda user, buf, n, st: int(all_reduce(buf, n, st))),
                       MI_ALLGATHER_FN(lambda user, snd, rcv, n, st: int(all_gather(snd, rcv, n, st))))  # keep alive
        check(self.lib.mi_tp_init_transport(self._ctx, self._xport[0], self._xport[1], None))

    def tp_all_reduce(self, tensors) -> None:
        """In-process tensor-parallel context only: all-reduce the fp32 device tensors (one per
        rank, on that rank's GPU) in place through the library's exchange kernels."""
        n = tensors[0].numel()
        ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        check(self.lib.mi_op_tp_all_reduce(self._ctx, ptrs, n))

    def tp_info(self) -> dict:
        """What the in-process tensor-parallel group runs on (mi_tp_info): transport in use, self-test, devices, peer
        access, whether the step replays from hipGraphs, loopback mode."""
        info = MiTpInfo()
        check(self.lib.mi_tp_info(self._ctx, C.byref(info)))
        T = info.tp_degree
        return {"tp_degree": T, "transport_used": {0: "p2p", 1: "rccl"}[info.transport],
                "selftest": {1: "passed", 0: "not run", -1: "p2p failed, rccl passed", -2: "failed"}[info.selftest],
                "graphs": bool(info.graphs),
                "mode": {0: "one GPU per rank", 1: "single-GPU loopback, lockstep (one stream, host barriers)",
                         2: "single-GPU loopback, concurrent streams (device flag waits)"}[info.mode],
                "timeout_ms": info.timeout_ms, "device_ids": list(info.device_ids[:T]),
                "peer_access": [[bool(info.peer_access[r] >> p & 1) for p in range(T)] for r in range(T)]}

    def set_num_blocks(self, num_blocks: int) -> None:
        check(self.lib.mi_set_num_blocks(self._ctx, num_blocks))
        self.cfg.num_blocks = num_blocks

    def finalize(self) -> None:
        check(self.lib.mi_finalize(self._ctx))

    def _pinned_logits(self):
        """Zero-copy view [max_num_seqs, V] of the library's pinned logits buffer (None for
        tensor-parallel contexts)."""
        if self._logits_view is None:
            ptr = self.lib.mi_logits_buffer(self._ctx)
            if not ptr:
                self._logits_view = False
            else:
                n = self.cfg.max_num_seqs * self.vocab_size
                buf = (C.c_float * n).from_address(ptr)
                self._logits_view = torch.frombuffer(buf, dtype=torch.float32).reshape(self.cfg.max_num_seqs,
                                                                                       self.vocab_size)
        return self._logits_view if self._logits_view is not False else None

    def forward(self, input_ids, position_ids, seq_ids, block_table, slot_mapping,
                full_context_lens, computed_context_lens, alias_ok: bool = False) -> "torch.Tensor":
        """CPU int64 tensors in (the reference's ModelInputForNeuron fields), fp32 CPU
        last-token logits [B, V] out.  alias_ok: the result may be a view of the library's pinned
        buffer, overwritten by the next call (the runner samples from it at once: saves a 2 MB
        host copy and an allocation per step)."""
        def i64(t):
            return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()
        ids, pos = i64(input_ids), i64(position_ids)
        B, S = ids.shape
        bt, sm = i64(block_table).reshape(B, -1), i64(slot_mapping).reshape(B, -1)
        full, comp = i64(full_context_lens).reshape(-1), i64(computed_context_lens).reshape(-1)
        seq = i64(seq_ids).reshape(-1) if seq_ids is not None else torch.zeros(B, dtype=torch.int64)
        pinned = self._pinned_logits() if alias_ok else None
        out = pinned[:B] if pinned is not None else torch.empty(B, self.vocab_size, dtype=torch.float32)
        check(self.lib.mi_forward(self._ctx, B, S, ids.data_ptr(), pos.data_ptr(), seq.data_ptr(),
                                  bt.data_ptr(), bt.shape[1], sm.data_ptr(), sm.shape[1],
                                  full.data_ptr(), comp.data_ptr(), out.data_ptr()))
        return out

    def forward_tokens(self, input_ids, position_ids, seq_ids, block_table, slot_mapping,
                       full_context_lens, computed_context_lens, sampling_params=None, seed: int = 0) -> "torch.Tensor":
        """The same call with on-device sampling: `sampling_params` [B, 3] fp32 rows
        (top_k, top_p, temperature) or None for greedy; -> sampled ids [B] int64 (CPU)."""
        def i64(t):
            return t.to(torch.int64).contiguous()
        ids, pos = i64(input_ids), i64(position_ids)
        B, S = ids.shape
        bt, sm = i64(block_table).reshape(B, -1), i64(slot_mapping).reshape(B, -1)
        full, comp = i64(full_context_lens).reshape(-1), i64(computed_context_lens).reshape(-1)
        seq = i64(seq_ids).reshape(-1) if seq_ids is not None else torch.zeros(B, dtype=torch.int64)
        sp = None
        if sampling_params is not None:
            sp = sampling_params.to(torch.float32).contiguous()
            if sp.shape != (B, 3):
                raise ValueError(f"sampling_params must be [B, 3], got {tuple(sp.shape)}")
        out = torch.empty(B, dtype=torch.int64)
        check(self.lib.mi_forward_tokens(self._ctx, B, S, ids.data_ptr(), pos.data_ptr(), seq.data_ptr(),
                                         bt.data_ptr(), bt.shape[1], sm.data_ptr(), sm.shape[1],
                                         full.data_ptr(), comp.data_ptr(),
                                         sp.data_ptr() if sp is not None else None, seed & (2 ** 64 - 1),
                                         out.data_ptr()))
        return out

    def forward_chunked(self, input_ids, position_ids, slot_mapping, block_table, full_context_lens,
                        computed_context_lens, sampling_params=None, seed: int = 0, tokens: bool = False):
        """Chunked prefill: one ragged token batch (reference runner.py:1000-1051).  input_ids,
        position_ids, slot_mapping: flat [total]; block_table [n, MB]; the context lengths [n].
        -> fp32 logits [n, V] of every request's last scheduled token, or sampled ids [n] (tokens=True)."""
        def i64(t):
            return t.to(torch.int64).contiguous()
        ids, pos, sm = i64(input_ids).reshape(-1), i64(position_ids).reshape(-1), i64(slot_mapping).reshape(-1)
        full, comp = i64(full_context_lens).reshape(-1), i64(computed_context_lens).reshape(-1)
        n = full.numel()
        bt = i64(block_table).reshape(n, -1)
        sp = None
        if tokens and sampling_params is not None:
            sp = sampling_params.to(torch.float32).contiguous()
        out = torch.empty(n, dtype=torch.int64) if tokens else torch.empty(n, self.vocab_size, dtype=torch.float32)
        check(self.lib.mi_forward_chunked(self._ctx, n, ids.numel(), ids.data_ptr(), pos.data_ptr(), sm.data_ptr(),
                                          bt.data_ptr(), bt.shape[1], full.data_ptr(), comp.data_ptr(),
                                          None if tokens else out.data_ptr(),
                                          sp.data_ptr() if sp is not None else None, seed & (2 ** 64 - 1),
                                          out.data_ptr() if tokens else None))
        return out

    def forward_spec(self, draft: "NativeModel", input_ids, position_ids, block_table, k: int, catchup_ids=None):
        """Fused speculation step (reference: NxDI fused speculation behind loader.py:349-355): k - 1
        chained greedy steps of `draft`, one pass of this (target) model over the B * k candidates,
        greedy acceptance.  input_ids / position_ids [B]: the last token of every sequence and its
        position; block_table [B, MB]; catchup_ids [B] (-1 = none): the token at position - 1 where
        the draft has not seen it yet (the step before generated k tokens).  -> (accepted [B, k]
        int64, 0-padded like NxDI's accepted_tokens_with_padding; next_pos [B] int64)."""
        def i64(t):
            return t.to(torch.int64).contiguous()
        ids, pos = i64(input_ids).reshape(-1), i64(position_ids).reshape(-1)
        B = ids.numel()
        bt = i64(block_table).reshape(B, -1)
        acc = torch.empty(B, k, dtype=torch.int64)
        nxt = torch.empty(B, dtype=torch.int64)
        cu = i64(catchup_ids).reshape(-1) if catchup_ids is not None else None
        check(self.lib.mi_forward_spec(self._ctx, draft._ctx, B, k, ids.data_ptr(), pos.data_ptr(), bt.data_ptr(),
                                       bt.shape[1], cu.data_ptr() if cu is not None else None, acc.data_ptr(),
                                       nxt.data_ptr()))
        return acc, nxt

    def replay_decode(self, steps: int) -> float:
        """Replay the last token-generation step `steps` times, inputs resident; -> elapsed ms
        (HIP events on the library's stream)."""
        ms = C.c_float()
        check(self.lib.mi_replay_decode(self._ctx, steps, C.byref(ms)))
        return ms.value

    def replay_decode_classes(self, steps: int, classes) -> float:
        """Replay the last token-generation step with only the kernel classes named in `classes`
        (K_CLASSES names) launched; -> elapsed ms.  Timing only."""
        mask = 0
        for name in classes:
            mask |= 1 << K_CLASSES.index(name)
        ms = C.c_float()
        check(self.lib.mi_replay_decode_classes(self._ctx, steps, mask, C.byref(ms)))
        return ms.value

    def kv_stats(self) -> dict:
        s = MiKvStats()
        check(self.lib.mi_kv_stats(self._ctx, C.byref(s)))
        return {f: getattr(s, f) for f, _ in MiKvStats._fields_}

    def kv_bytes_per_block(self) -> int:
        return int(self.lib.mi_kv_bytes_per_block(self._ctx))

    def stream_handle(self) -> int:
        return int(self.lib.mi_stream(self._ctx) or 0)

    def profile_enable(self, on: bool) -> None:
        check(self.lib.mi_profile_enable(self._ctx, int(on)))

    def profile_read(self) -> dict:
        n = len(K_CLASSES)
        launches, ms, wbytes = (C.c_int32 * n)(), (C.c_float * n)(), C.c_double()
        check(self.lib.mi_profile_read(self._ctx, launches, ms, C.byref(wbytes)))
        return {"launches": dict(zip(K_CLASSES, launches)), "ms": dict(zip(K_CLASSES, ms)),
                "gemv_weight_bytes": wbytes.value}
