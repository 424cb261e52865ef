"""Tensor-parallel sharding on ONE GPU: every rank's context runs on device 0 in its own thread
and the two collectives go through a loopback transport (mi_tp_init_transport) instead of RCCL.

What this pins: the library's column / row sharding of the HF matrices (fused QKV by kv-head
group, gate|up, vocab-parallel lm_head), the per-rank KV pool, the fp32 partial + all-reduce +
next-norm-prologue fold, and the logits all-gather -- i.e. everything of the N-GPU path except
the transport itself (RCCL is exercised with a one-rank communicator in test_model_gpu.py; a real
multi-GPU run needs the driver's 8-GPU node).  Checked against the CPU oracle on the reference
call sequence, and rank 0 against rank 1 bit-for-bit.
"""

import ctypes
import threading

import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.synth import make_prompts, make_weights, zoo_config
from tests.test_model_gpu import BS, MAXLEN, NB, NSEQ, load_golden, scenario

pytestmark = pytest.mark.gpu

hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice = 1, 2, 3


class Loopback:
    """all-reduce / all-gather among `n` contexts of one process (one thread per rank)."""

    def __init__(self, n, lib):
        self.n, self.lib = n, lib
        self.barrier = threading.Barrier(n)
        self.ptrs = [None] * n
        self.calls = [0] * n
        lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.hipMemcpy.restype = ctypes.c_int

    def _copy(self, dst, src, nbytes, kind):
        assert self.lib.hipMemcpy(dst, src, nbytes, kind) == 0

    def all_reduce(self, rank, buf, count, stream):
        torch.cuda.synchronize()                      # everything queued on the rank's stream is done
        self.ptrs[rank] = buf
        self.calls[rank] += 1
        self.barrier.wait()
        if rank == 0:                                 # sum in rank order, hand the same bits to everyone
            host = [torch.empty(count, dtype=torch.float32) for _ in range(self.n)]
            for r in range(self.n):
                self._copy(host[r].data_ptr(), self.ptrs[r], count * 4, hipMemcpyDeviceToHost)
            total = host[0].clone()
            for r in range(1, self.n):
                total += host[r]
            for r in range(self.n):
                self._copy(self.ptrs[r], total.data_ptr(), count * 4, hipMemcpyHostToDevice)
        self.barrier.wait()
        return 0

    def all_gather(self, rank, send, recv, count, stream):
        torch.cuda.synchronize()
        self.ptrs[rank] = send
        self.barrier.wait()
        for r in range(self.n):
            self._copy(recv + r * count * 4, self.ptrs[r], count * 4, hipMemcpyDeviceToDevice)
        torch.cuda.synchronize()
        self.barrier.wait()                           # nobody reuses `send` before all have copied
        return 0


def _rank_model(cfg, weights, tp, rank, loop, weight_dtype, quant_type):
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
        num_blocks=NB, block_size=BS, max_num_seqs=NSEQ, max_model_len=MAXLEN,
        weight_dtype=MI_W[weight_dtype], quant_type=MI_Q[quant_type], quantize_lm_head=1,
        tp_degree=tp, tp_rank=rank, device_id=0, use_graphs=0)
    m.tp_init_transport(lambda buf, n, st: loop.all_reduce(rank, buf, n, st),
                        lambda snd, rcv, n, st: loop.all_gather(rank, snd, rcv, n, st))
    m.load_state_dict(weights)                        # FULL tensors: the library takes its shard
    m.finalize()
    return m


# llama31_like: 8 q / 2 kv heads -> 4 + 1 per rank at TP 2.  tinyllama_like: 8 q heads on ONE kv
# head -> at TP 4 every rank holds 2 q heads and a replica of the kv head (the reference skips
# vLLM's divisibility check for exactly this case, platform.py:58-64).
@pytest.mark.parametrize("name,tp,weight_dtype,quant_type", [
    ("llama31_like", 2, "bf16", "per_tensor_symmetric"),
    ("llama31_like", 2, "f8e4m3", "per_channel_symmetric"),
    ("tinyllama_like", 4, "f8e4m3", "per_channel_symmetric"),
    ("tinyllama_like", 2, "int8", "per_tensor_symmetric"),
])
def test_sharded_ranks_match_oracle(name, tp, weight_dtype, quant_type):
    from vllm_neuron_amd import _native
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    quant = None if weight_dtype == "bf16" else dict(quantized=True, quantization_dtype=weight_dtype,
                                                     quantization_type=quant_type)
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16", quant=quant)
    steps = list(scenario(prompts, gen))
    want = [oracle.forward(**inp) for _, inp, _ in steps]

    loop = Loopback(tp, _native.load_library())
    models = [_rank_model(cfg, w, tp, r, loop, weight_dtype, quant_type) for r in range(tp)]
    got = [[None] * len(steps) for _ in range(tp)]
    toks = [None] * tp
    errs = []

    def run(rank):
        try:
            for i, (_, inp, _) in enumerate(steps):
                got[rank][i] = models[rank].forward(**inp)
            # on-device sampling over the vocab-parallel logits (ids come from the gathered segments)
            toks[rank] = models[rank].forward_tokens(**steps[-1][1]).tolist()
        except Exception as e:                        # noqa: BLE001  (surface it in the main thread)
            errs.append((rank, e))
            loop.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(tp)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    assert not any(t.is_alive() for t in threads)
    assert all(n == (len(steps) + 1) * 2 * cfg.num_layers for n in loop.calls)
    assert all(t == got[0][-1].argmax(dim=1).tolist() for t in toks), toks
    for i, ref in enumerate(want):
        a = got[0][i]
        for r in range(1, tp):
            assert torch.equal(a, got[r][i]), f"step {i}: ranks 0 and {r} disagree"
        n = ref.shape[0]
        err = (a[:n] - ref).abs().max().item()
        assert err <= 0.06, (i, err)                  # same bound as the single-GPU parity tests
    for m in models:
        m.close()
