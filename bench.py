#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s + p50 TTFT, Llama-3.1-8B FP8 weights, block_size=32,
max_num_seqs=4 (BASELINE.json), on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python bench.py --gpus N ...                       (one process drives the N GPUs, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one token-generation pass of the hot path over one batch: B = max_num_seqs = 4
sequences at context 1024 (BASELINE.md roofline row), all layers + lm_head, through
libmi355x_vllm.  The timed region is exactly K steps with every input resident in HBM
(mi_replay_decode: hipGraph replays bracketed by HIP events on the library's stream, and by
barrier + torch.cuda.synchronize on both sides); `value` = B * K / max-over-ranks time.

What the line says, key by key:
  value / ms_per_step          the DEVICE step: the hot path replayed with resident inputs -- no
                               scheduler, no H2D, no logits D2H, no sampler.  Kernel quality.
  engine_decode_tokens_per_s   what a user of the plugin sees: scheduler -> runner -> library ->
                               CPU sampler (the reference's parity path), 4 requests x 128 tokens.
  engine_decode_tokens_per_s_on_device_sampling   the same loop with the on-device sampler.
  ttft_p50_ms                  p50 time to first token through the whole plugin path, per
                               context-encoding bucket, in the WEIGHT-ONLY mode (FP8 weights x bf16
                               activations: what the reference's quantized linears compute -- the
                               parity path).
  ttft_p50_ms_fp8_activations  the fast mode (per-token FP8 activations on the MX-scaled MFMA) and
  fp8_activation_first_token_agreement   how often its greedy first token equals the parity path's (all prompts);
  fp8_activation_agreement_gated         the same over the prompts whose weight-only top-2 logit gap exceeds the mode's own
                               logit noise (how many qualify, the rms between the modes' logits): what the MODE does,
                               not what near-tied random weights do.
  tp                           (--gpus N) what the tensor-parallel group ran on: transport in use, self-test, devices,
                               peer access, hipGraphs, loopback mode (mi_tp_info).
  roofline                     the weight-streaming GEMV (dominant kernel): algorithmic weight bytes
                               per launch / average launch duration.  The duration is measured live:
                               the GEMV launches of a step replayed back to back in a graph of their
                               own, HIP events around K such replays on the library's stream
                               (launch-to-launch, as rocprofv3's kernel trace counts it).
  cpu_baseline                 the CPU oracle (oracle/, test infrastructure) on the host cores.

N > 1: tensor parallelism inside ONE process, the reference's process model (a single worker
drives every core): the driver process owns one rank shard per GPU on threads inside the library.
Launched under torch.distributed.run, local rank 0 is that driver and the other ranks only take
part in the barriers and the max-over-ranks of the wall clock; launched directly with --gpus N it
needs no launcher.  Total work is fixed -> "scaling": "strong".

Weights are synthetic N(0, 0.02) (no checkpoints exist offline), generated on the device at the
real shapes and quantized per-channel.
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LLAMA3_ROPE = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
               "original_max_position_embeddings": 8192}
MODELS = {
    "llama31_8b": dict(
        architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=128256, hidden_size=4096,
        intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
        head_dim=128, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False, rope_scaling=LLAMA3_ROPE),
    "qwen25_7b": dict(
        architectures=["Qwen2ForCausalLM"], model_type="qwen2", vocab_size=152064, hidden_size=3584,
        intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
        head_dim=128, rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False, rope_scaling=None),
    "llama33_70b": dict(
        architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=128256, hidden_size=8192,
        intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64, num_key_value_heads=8,
        head_dim=128, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False, rope_scaling=LLAMA3_ROPE),
    # draft model of the speculation section: Llama-3.2-1B dims (same vocabulary as the 8B target)
    "llama32_1b": dict(
        architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=128256, hidden_size=2048,
        intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32, num_key_value_heads=8,
        head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=True,
        rope_scaling={**LLAMA3_ROPE, "factor": 32.0}),
    # BASELINE config 1 (CPU plumbing): TinyLlama-1.1B dims, used by cpu_baseline only
    "tinyllama_1b": dict(
        architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=32000, hidden_size=2048,
        intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32, num_key_value_heads=4,
        head_dim=64, rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False, rope_scaling=None),
}
MODEL_LABEL = {"llama31_8b": "Llama-3.1-8B", "qwen25_7b": "Qwen2.5-7B", "llama33_70b": "Llama-3.3-70B"}
BLOCK_SIZE, MAX_NUM_SEQS, MAX_MODEL_LEN, PA_NUM_BLOCKS = 32, 4, 2048, 4096
BUCKETS = [256, 512, 1024, 2048]
DECODE_CTX = 1024
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)


def weight_bytes_per_step(hf, tp, bytes_per_param=1):
    """Algorithmic weight bytes ONE GPU streams per token-generation step (fp8 / int8 = 1 B per
    parameter, bf16 = 2)."""
    H, I, V, L = hf["hidden_size"], hf["intermediate_size"], hf["vocab_size"], hf["num_hidden_layers"]
    qd = hf["num_attention_heads"] * hf["head_dim"]
    kvd = hf["num_key_value_heads"] * hf["head_dim"]
    per_layer = (qd + 2 * kvd) * H + H * qd + 2 * I * H + H * I
    return (L * per_layer + V * H) * bytes_per_param / tp


def kv_bytes_per_step(hf, tp, B, ctx):
    return B * ctx * 2 * max(hf["num_key_value_heads"] / tp, 1) * hf["head_dim"] * 2 * hf["num_hidden_layers"]


# ---- CPU baseline (the oracle: test infrastructure, only ever the thing measured BESIDE the product) ----
def _oracle_weights(torch, cfg, layers, g):
    H, hd = cfg.hidden_size, cfg.head_dim

    def mat(n, k):
        return torch.randn(n, k, generator=g) * 0.02
    w = {"model.embed_tokens.weight": torch.randn(1, H, generator=g).expand(cfg.vocab_size, H),
         "model.norm.weight": torch.ones(H), "lm_head.weight": mat(cfg.vocab_size, H)}
    for l in range(layers):
        p = f"model.layers.{l}."
        w[p + "self_attn.q_proj.weight"] = mat(cfg.num_heads * hd, H)
        w[p + "self_attn.k_proj.weight"] = mat(cfg.num_kv_heads * hd, H)
        w[p + "self_attn.v_proj.weight"] = mat(cfg.num_kv_heads * hd, H)
        w[p + "self_attn.o_proj.weight"] = mat(H, cfg.num_heads * hd)
        w[p + "mlp.gate_proj.weight"] = mat(cfg.intermediate_size, H)
        w[p + "mlp.up_proj.weight"] = mat(cfg.intermediate_size, H)
        w[p + "mlp.down_proj.weight"] = mat(H, cfg.intermediate_size)
        w[p + "input_layernorm.weight"] = torch.ones(H)
        w[p + "post_attention_layernorm.weight"] = torch.ones(H)
    return w


def _decoder_config(hf, layers):
    from oracle import DecoderConfig
    return DecoderConfig(num_layers=layers, hidden_size=hf["hidden_size"], num_heads=hf["num_attention_heads"],
                         num_kv_heads=hf["num_key_value_heads"], head_dim=hf["head_dim"],
                         intermediate_size=hf["intermediate_size"], vocab_size=hf["vocab_size"],
                         rms_norm_eps=hf["rms_norm_eps"], rope_theta=hf["rope_theta"], rope_scaling=hf["rope_scaling"])


def cpu_baseline(hf, B, ctx, layers=2, steps=3):
    """The CPU oracle timed on this box's host cores over a BOUNDED sample of the workload:
      * token generation at B x ctx and context encoding of the 256 bucket, over `layers` of the
        model's decoder layers + lm_head, extrapolated linearly in the layer count;
      * BASELINE config 1 in full: TinyLlama-1.1B dimensions (22 layers, fp32), 4 prompts of
        6 / 9 / 7 / 140 tokens, greedy, 16 new tokens, prefix caching on -- prompts encoded one at
        a time, then batched token generation, like the reference's scheduler."""
    import torch
    from oracle import PagedDecoderOracle
    from tests.helpers import decode_inputs, prefill_inputs
    # a 1-GPU box grants 16 host cores however many the machine reports
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("MI_BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(ncores)
    g = torch.Generator().manual_seed(1)
    cfg = _decoder_config(hf, layers)
    w = _oracle_weights(torch, cfg, layers, g)
    mb = MAX_MODEL_LEN // BLOCK_SIZE
    oracle = PagedDecoderOracle(cfg, w, 1 + B * mb, BLOCK_SIZE, compute="fp32")
    oracle.kv.normal_(generator=g)
    blocks = [[1 + b * mb + j for j in range(mb)] for b in range(B)]
    dec = decode_inputs([1] * B, [ctx - 1] * B, blocks, BLOCK_SIZE, MAX_MODEL_LEN)
    pre = prefill_inputs(torch.randint(0, cfg.vocab_size, (256 - 17,), generator=g).tolist(), blocks[0], BLOCK_SIZE,
                         MAX_MODEL_LEN, 0)

    def timed(o, inp, n):
        o.forward(**inp)
        t = time.perf_counter()
        for _ in range(n):
            o.forward(**inp)
        return (time.perf_counter() - t) / n
    from oracle import DecoderConfig
    cfg0 = DecoderConfig(**{**cfg.__dict__, "num_layers": 0})
    o0 = PagedDecoderOracle(cfg0, {k: v for k, v in w.items() if "layers" not in k}, 2, BLOCK_SIZE, compute="fp32")
    L = hf["num_hidden_layers"]
    t_full, t_head = timed(oracle, dec, steps), timed(o0, dec, steps)
    t_layer = max(t_full - t_head, 0.0) / layers
    t_step = t_layer * L + t_head
    p_full, p_head = timed(oracle, pre, 1), timed(o0, pre, 1)
    p_layer = max(p_full - p_head, 0.0) / layers
    ttft_256 = p_layer * L + p_head
    del oracle, o0, w

    # ---- config 1, in full ----
    tl = MODELS["tinyllama_1b"]
    tcfg = _decoder_config(tl, tl["num_hidden_layers"])
    tw = _oracle_weights(torch, tcfg, tl["num_hidden_layers"], g)
    tmb = 1024 // BLOCK_SIZE
    to = PagedDecoderOracle(tcfg, tw, 1 + 4 * tmb, BLOCK_SIZE, compute="fp32")
    prompts = [torch.randint(0, tcfg.vocab_size, (n,), generator=g).tolist() for n in (6, 9, 7, 140)]
    tblocks = [[1 + b * tmb + j for j in range(tmb)] for b in range(4)]
    t0 = time.perf_counter()
    seqs, ttfts = [], []
    for p, bl in zip(prompts, tblocks):
        t1 = time.perf_counter()
        lg = to.forward(**prefill_inputs(p, bl, BLOCK_SIZE, 1024, 0))
        seqs.append(p + [int(lg.argmax())])
        ttfts.append(time.perf_counter() - t1)
    t_dec0 = time.perf_counter()
    new_tokens = 16
    for _ in range(new_tokens - 1):
        lg = to.forward(**decode_inputs([s[-1] for s in seqs], [len(s) - 1 for s in seqs], tblocks, BLOCK_SIZE, 1024))
        for s, row in zip(seqs, lg):
            s.append(int(row.argmax()))
    t_end = time.perf_counter()
    config1 = {"workload": "TinyLlama-1.1B dims (22 layers, fp32), 4 prompts of 6/9/7/140 tokens, greedy, 16 new tokens, "
                           "block_size 32, max_num_seqs 4: run in full",
               "decode_tokens_per_s": round(4 * (new_tokens - 1) / (t_end - t_dec0), 2),
               "ttft_ms": [round(x * 1e3, 1) for x in ttfts], "total_s": round(t_end - t0, 2)}
    return {"value": round(B / t_step, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle (torch CPU fp32) token generation at B={B}, ctx={ctx}: {steps} steps over {layers} of "
                       f"{L} layers + lm_head, per-layer time extrapolated to {L} layers "
                       f"({t_layer * 1e3:.1f} ms/layer, {t_head * 1e3:.1f} ms head); TTFT: one 239-token prompt "
                       f"(bucket 256) over the same {layers} layers, extrapolated likewise"),
            "ttft_ms_bucket_256": round(ttft_256 * 1e3, 1), "config1_cpu": config1}


def speculation_section(torch, HF, wd, plain_ms, steps, k=4):
    """One fused speculation step (mi_forward_spec: k - 1 chained draft steps + ONE target pass over the
    B * k candidate rows + acceptance) timed next to its parts.  Weights are random, so NOTHING is
    said about acceptance: the section reports what a step costs and how many tokens per sequence
    it must yield on average to beat the plain step."""
    from tests.helpers import decode_inputs
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    from vllm_neuron_amd.worker.mi355x_model_loader import _decoder_geometry
    B, mb = MAX_NUM_SEQS, MAX_MODEL_LEN // BLOCK_SIZE
    nb = 1 + B * mb

    def build(hf, rows, seed):
        m = NativeModel(num_blocks=nb, block_size=BLOCK_SIZE, max_num_seqs=rows, max_model_len=MAX_MODEL_LEN,
                        ctx_buckets=BUCKETS, weight_dtype=MI_W[wd], quant_type=MI_Q["per_channel_symmetric"],
                        quantize_lm_head=1, tp_degree=1, tp_rank=0, device_id=0, use_graphs=1,
                        prefill_fp8_activations=0, **_decoder_geometry(SimpleNamespace(**hf)))
        m.init_synthetic_weights(seed, 0.02)
        m.finalize()
        return m
    target, draft = build(HF, B * k, 1), build(MODELS["llama32_1b"], 2 * B, 2)
    blocks = [[1 + b * mb + j for j in range(mb)] for b in range(B)]
    n = max(steps // 2, 8)

    def replay(model, rows, row_pos, row_blocks):
        model.forward(**decode_inputs([1] * rows, row_pos, row_blocks, BLOCK_SIZE, MAX_MODEL_LEN))
        model.replay_decode(4)
        return model.replay_decode(n) / n

    def case(nb_):
        bt = torch.tensor(blocks[:nb_], dtype=torch.long)
        last, pos = torch.arange(1, nb_ + 1), torch.full((nb_,), DECODE_CTX - 1, dtype=torch.long)
        for _ in range(3):
            target.forward_spec(draft, last, pos, bt, k)
        t = time.perf_counter()
        for _ in range(n):
            target.forward_spec(draft, last, pos, bt, k)
        spec_ms = (time.perf_counter() - t) / n * 1e3
        # the parts, device-resident: the target's pass over nb_ * k rows, one draft step, the plain step of nb_ rows
        rows = nb_ * k
        target_ms = replay(target, rows, [DECODE_CTX - 1 - (i % k) for i in range(rows)], [blocks[i // k] for i in range(rows)])
        draft_ms = replay(draft, nb_, [DECODE_CTX - 1] * nb_, blocks[:nb_])
        plain = plain_ms if nb_ == B else replay(target, nb_, [DECODE_CTX - 1] * nb_, blocks[:nb_])
        return {"B": nb_, "spec_step_ms": round(spec_ms, 4), "target_pass_ms_over_B_times_k_rows": round(target_ms, 4),
                "draft_step_ms": round(draft_ms, 4), "plain_step_ms": round(plain, 4),
                "tokens_per_sequence_and_step_to_break_even": round(spec_ms / plain, 3),
                "tokens_per_s_if_all_k_accepted": round(nb_ * k / (spec_ms * 1e-3), 1)}
    cases = [case(B), case(1)]
    draft.close()
    target.close()
    return {"k": k, "draft": f"Llama-3.2-1B dims, {wd}, synthetic", "cases": cases,
            "note": "spec_step_ms: host-timed calls (inputs H2D, k graph launches, accepted ids D2H; no catch-up row); the other times are "
                    "graph replays with resident inputs.  Random weights: no acceptance rate is claimed -- "
                    "tests/test_spec_decode_gpu.py checks that the text equals the target's greedy text"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama31_8b", choices=["llama31_8b", "qwen25_7b", "llama33_70b"])
    ap.add_argument("--ttft-prompts", type=int, default=32, help="prompts per context-encoding bucket (p50)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-speculation", action="store_true", help="skip the fused-speculation timing section")
    ap.add_argument("--weight-dtype", default=None, choices=["f8e4m3", "int8", "bf16"],
                    help="default: f8e4m3 (int8 for qwen25_7b)")
    ap.add_argument("--tp-transport", default="p2p", choices=["p2p", "rccl"])
    ap.add_argument("--tp-loopback", action="store_true",
                    help="dev: place every rank shard of --gpus N on GPU 0 (functional check of the TP path on one GPU)")
    args = ap.parse_args()
    wd = args.weight_dtype or ("int8" if args.model == "qwen25_7b" else "f8e4m3")
    HF = MODELS[args.model]

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world in (1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    tp = args.gpus
    if world > 1:
        # host-side control plane only (barriers, max-over-ranks of the wall clock): gloo
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    driver = rank == 0                      # the ONE process that owns the tensor-parallel group

    def host_barrier():
        if world > 1:
            dist.barrier()

    def finish(line):
        if driver:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()

    if not driver:
        # the driver's context drives every GPU; this rank only meets it at the barriers of the timed region
        for _ in range(2):
            host_barrier()
        tt = torch.zeros(1, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return finish(None)

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if args.tp_loopback:
        os.environ["MI355X_TP_LOOPBACK"] = "1"
    elif tp > 1:
        assert torch.cuda.device_count() >= tp, (
            f"--gpus {tp}: this process sees {torch.cuda.device_count()} GPUs; the tensor-parallel group lives in ONE "
            "process and needs all of them visible")
    torch.cuda.set_device(0)

    def barrier_sync():
        host_barrier()
        for d in range(1 if args.tp_loopback else tp):
            torch.cuda.synchronize(d)

    from vllm_neuron_amd._vllm_compat import SamplingParams
    from vllm_neuron_amd.engine import MI355XEngine
    hf = SimpleNamespace(**HF)
    override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": BUCKETS,
                "pa_num_blocks": PA_NUM_BLOCKS, "tp_transport": args.tp_transport}
    if wd != "bf16":
        override.update(quantized=True, quantization_dtype=wd, quantization_type="per_channel_symmetric")

    def make_engine(fp8_activations):
        o = dict(override)
        if fp8_activations:
            o["prefill_fp8_activations"] = True   # context-encoding GEMMs: per-token FP8 activations, MX-scaled MFMA
        return MI355XEngine(hf, max_model_len=MAX_MODEL_LEN, max_num_seqs=MAX_NUM_SEQS, block_size=BLOCK_SIZE,
                            num_gpu_blocks_override=PA_NUM_BLOCKS, enable_prefix_caching=True,
                            tensor_parallel_size=tp, override_mi355x_config=o)
    t0 = time.perf_counter()
    eng = make_engine(False)                # the parity path: weight-only quantization everywhere
    tp_note = None
    if tp > 1 and not args.tp_loopback and args.tp_transport == "p2p":
        # The peer-memory exchange has never run between two GPUs: if its first real steps fail -- the
        # library's own self-test passed, or it would have switched to RCCL by itself -- say so and measure the RCCL transport
        # instead of dying (flag waits are bounded: a broken exchange reports MI_ECOMM, it does not hang).
        try:
            eng.generate([list(range(1, 40))], SamplingParams(temperature=0.0, max_tokens=6))
        except Exception as e:                                  # noqa: BLE001 -- whatever the transport throws
            tp_note = f"p2p transport failed on its first steps ({type(e).__name__}: {str(e)[:200]}); RCCL transport measured instead"
            print(f"[bench] {tp_note}", file=sys.stderr, flush=True)
            try:
                eng.worker.model_runner.model.model.close()
            except Exception:                                   # noqa: BLE001
                pass
            override["tp_transport"] = "rccl"
            eng = make_engine(False)
    native = eng.worker.model_runner.model.model
    init_s = time.perf_counter() - t0

    # ---- p50 TTFT per bucket, through scheduler -> runner -> library -> CPU sampler ----------
    def ttft_sweep(engine, tag):
        g = torch.Generator().manual_seed(0)            # the same prompts for every mode
        res, first = {}, {}
        for bucket in BUCKETS:
            samples, toks = [], []
            for _ in range(args.ttft_prompts + 1):           # first one is warm-up
                prompt = torch.randint(0, hf.vocab_size, (bucket - 17,), generator=g).tolist()
                out = engine.generate([prompt], SamplingParams(temperature=0.0, max_tokens=1))[0]
                samples.append(out.ttft_s * 1e3)
                toks.append(out.token_ids[0])
            res[str(bucket)] = round(statistics.median(samples[1:]), 3)
            first[str(bucket)] = toks[1:]
            print(f"[bench] ttft {tag} bucket {bucket}: {[round(x, 2) for x in samples]} ms", file=sys.stderr, flush=True)
        return res, first
    ttft_a8, agreement, agreement_gated = None, None, None
    eng2 = make_engine(True) if wd == "f8e4m3" else None
    if eng2 is not None:
        # ---- does the FP8 x FP8 mode keep the greedy token?  Measured on the LOGITS of both modes for the same prompts, before
        #      any engine call (the pools are still unhashed, so writing blocks 1.. by hand disturbs nothing): a prompt only says
        #      something about the mode when the weight-only top-2 gap is larger than the mode's own logit noise on that prompt
        #      (on N(0, 0.02) weights the gap is usually far below it, and an ungated rate measures the weights).
        from tests.helpers import prefill_inputs
        native2 = eng2.worker.model_runner.model.model
        ga = torch.Generator().manual_seed(0)
        blocks_a = list(range(1, MAX_MODEL_LEN // BLOCK_SIZE + 1))
        agreement, agreement_gated = {}, {}
        for bucket in BUCKETS:
            same, qual, qual_same, rms_all, gap_all = 0, 0, 0, [], []
            for _ in range(args.ttft_prompts):
                prompt = torch.randint(0, hf.vocab_size, (bucket - 17,), generator=ga).tolist()
                inp = prefill_inputs(prompt, blocks_a, BLOCK_SIZE, MAX_MODEL_LEN, 0)
                lw = native.forward(**inp)[0].double()
                la = native2.forward(**inp)[0].double()
                top2 = torch.topk(lw, 2).values
                gap = float(top2[0] - top2[1])
                rms = float((la - lw).pow(2).mean().sqrt())
                eq = int(lw.argmax()) == int(la.argmax())
                same += eq
                if gap > 2.0 * rms:          # a 2-sigma rule on the difference of two noisy logits
                    qual += 1
                    qual_same += eq
                rms_all.append(rms)
                gap_all.append(gap)
            n = args.ttft_prompts
            agreement[str(bucket)] = round(same / n, 3)
            agreement_gated[str(bucket)] = {
                "prompts": n, "qualifying": qual, "agreement_over_qualifying": round(qual_same / qual, 3) if qual else None,
                "logit_rms_between_modes": round(statistics.median(rms_all), 4), "median_weight_only_top2_gap": round(statistics.median(gap_all), 4),
                "logit_std": round(float(lw.std()), 3)}
    ttft, first_wo = ttft_sweep(eng, "weight-only")
    if eng2 is not None:
        ttft_a8, first_a8 = ttft_sweep(eng2, "fp8xfp8")
        eng2.worker.model_runner.model.model.close()
        del eng2

    # ---- prefix caching (BASELINE config 4: sequences sharing a 512-token prefix) ----------------
    prefix_ttft = None
    if args.model == "qwen25_7b":
        g = torch.Generator().manual_seed(7)
        prefix_ttft = {}
        for suf in (64, 128, 256):
            miss, hit = [], []
            for rep in range(9):
                prefix = torch.randint(0, hf.vocab_size, (512,), generator=g).tolist()
                mk = lambda: prefix + torch.randint(0, hf.vocab_size, (suf,), generator=g).tolist()   # noqa: E731
                a = eng.generate([mk()], SamplingParams(temperature=0.0, max_tokens=1))[0]            # encodes the prefix
                b = eng.generate([mk()], SamplingParams(temperature=0.0, max_tokens=1))[0]            # 16 cached blocks
                assert a.num_cached_tokens == 0 and b.num_cached_tokens == 512, (a.num_cached_tokens, b.num_cached_tokens)
                miss.append(a.ttft_s * 1e3)
                hit.append(b.ttft_s * 1e3)
            prefix_ttft[str(suf)] = {"miss_ms": round(statistics.median(miss[1:]), 3),
                                     "hit_512_ms": round(statistics.median(hit[1:]), 3)}

    # ---- the whole serving loop (scheduler -> runner -> library -> sampler): 4 requests of 900
    #      prompt tokens decoding 128 tokens each, first with the CPU sampler (the parity path:
    #      [B, V] fp32 logits cross PCIe every step), then with on-device sampling (B ids cross).
    g = torch.Generator().manual_seed(3)

    def engine_decode_rate(sampling=None):
        prompts = [torch.randint(0, hf.vocab_size, (900,), generator=g).tolist() for _ in range(MAX_NUM_SEQS)]
        t_ = time.perf_counter()
        outs = eng.generate(prompts, sampling or SamplingParams(temperature=0.0, max_tokens=128))
        dt = time.perf_counter() - t_
        ntok = sum(len(o.token_ids) for o in outs)
        first = max(o.ttft_s for o in outs)
        return round((ntok - len(outs)) / (dt - first), 1)
    engine_cpu = engine_decode_rate()
    model_adapter = eng.worker.model_runner.model
    model_adapter.mi355x_config.on_device_sampling_config = {"dynamic": True, "deterministic": False}
    engine_dev = engine_decode_rate()
    # ... and with requests that SAMPLE (top_k 50, top_p 0.9, temperature 0.8): the radix-select sampler instead of the argmax
    engine_dev_sampled = engine_decode_rate(SamplingParams(temperature=0.8, top_k=50, top_p=0.9, max_tokens=128, ignore_eos=True))
    model_adapter.mi355x_config.on_device_sampling_config = None

    # ---- the same loop under vLLM's native scheduler (chunked prefill on): every step is a
    #      ragged record; once all prompts are encoded the library runs it as a token-generation step
    engine_chunked = None
    if tp == 1 and args.model == "llama31_8b":
        oc = dict(override)
        oc["chunked_prefill_config"] = {"max_num_seqs": MAX_NUM_SEQS}
        engc = MI355XEngine(hf, max_model_len=MAX_MODEL_LEN, max_num_seqs=MAX_NUM_SEQS, block_size=BLOCK_SIZE,
                            num_gpu_blocks_override=PA_NUM_BLOCKS, enable_prefix_caching=True, enable_chunked_prefill=True,
                            max_num_batched_tokens=2048, override_mi355x_config=oc)
        main_eng, eng = eng, engc
        try:
            engine_decode_rate()                      # warm-up: graph capture, allocator
            engine_chunked = engine_decode_rate()
        finally:
            eng = main_eng
        engc.worker.model_runner.model.model.close()
        del engc

    # ---- decode at other context lengths (ctx 256 / 1024 / 2040), device-resident ---
    from tests.helpers import decode_inputs
    mb = MAX_MODEL_LEN // BLOCK_SIZE
    perm = (torch.randperm(PA_NUM_BLOCKS, generator=torch.Generator().manual_seed(2)) + 1).tolist()
    blocks = [perm[b * mb:(b + 1) * mb] for b in range(MAX_NUM_SEQS)]
    toks = torch.randint(0, hf.vocab_size, (MAX_NUM_SEQS,), generator=g).tolist()
    by_ctx = {}
    for ctx_len in (256, 2040):
        inp_c = decode_inputs(toks, [ctx_len - 1] * MAX_NUM_SEQS, blocks, BLOCK_SIZE, MAX_MODEL_LEN)
        for _ in range(2):
            native.forward(**inp_c)
        native.replay_decode(8)
        n_c = max(args.steps // 2, 8)
        by_ctx[str(ctx_len)] = round(MAX_NUM_SEQS * n_c / (native.replay_decode(n_c) * 1e-3), 1)

    # ---- decode: B sequences at context DECODE_CTX: the timed region ----------------------------
    inp = decode_inputs(toks, [DECODE_CTX - 1] * MAX_NUM_SEQS, blocks, BLOCK_SIZE, MAX_MODEL_LEN)
    for _ in range(2):
        native.forward(**inp)                            # captures the graph, leaves inputs resident
    native.replay_decode(max(args.warmup, 1))            # W untimed warm-up steps
    barrier_sync()
    t = time.perf_counter()
    dev_ms = native.replay_decode(args.steps)            # exactly K timed steps
    for d in range(1 if args.tp_loopback else tp):
        torch.cuda.synchronize(d)
    wall = time.perf_counter() - t
    barrier_sync()
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt[0])
    ms_per_step = wall * 1e3 / args.steps
    value = MAX_NUM_SEQS * args.steps / wall

    # PCIe-inclusive rate: the same step through mi_forward (ids H2D, [B, V] fp32 logits D2H)
    t = time.perf_counter()
    for _ in range(args.steps):
        native.forward(**inp)
    e2e = (time.perf_counter() - t) / args.steps

    # ---- roofline of the dominant kernel (weight-streaming GEMV) ----------------------------------
    bpp = 2 if wd == "bf16" else 1
    step_bytes = weight_bytes_per_step(HF, tp, bpp) + kv_bytes_per_step(HF, tp, MAX_NUM_SEQS, DECODE_CTX)
    roofline = None
    if tp == 1:
        # launches and algorithmic bytes of the class from one eager, event-timed step ...
        native.profile_enable(True)
        native.forward(**inp)
        prof = native.profile_read()
        native.profile_enable(False)
        gemv_n, gemv_bytes = prof["launches"]["gemv"], prof["gemv_weight_bytes"]
        # ... their duration from the GEMV launches of a step replayed alone, back to back, in their own graph
        native.forward(**inp)
        gemv_ms = native.replay_decode_classes(args.steps, ["gemv"]) / args.steps
        native.forward(**inp)                               # the full graph again (and a meaningful state)
        achieved = gemv_bytes / (gemv_ms * 1e-3) / 1e9
        # HBM traffic per GEMV launch from the committed PMC pass of this same command (FETCH_SIZE,
        # gfx950-corrected; tools/pmc_summary.py) -- counters cannot be read from inside the process
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "r03_gemv_traffic.json")
        if args.model == "llama31_8b" and wd == "f8e4m3" and os.path.exists(tf):
            import hashlib
            with open(tf) as fh:
                doc = json.load(fh)
            traffic = doc["traffic_bytes_per_launch"]
            with open(os.path.join(ROOT, "vllm-neuron_amd", "csrc", "linear_kernels.hip"), "rb") as fh:
                fresh = hashlib.sha1(fh.read()).hexdigest() == doc.get("kernel_source_sha1")
            traffic_src = ("profiles/r03_gemv_traffic.json (rocprofv3 --pmc FETCH_SIZE pass of this command, x2 gfx950 "
                           "correction: the counter tallies 128-B requests at 64 B)"
                           + ("" if fresh else " -- STALE: the GEMV source has changed since that pass"))
        roofline = {"bound": "hbm", "kernel": "mi::gemv_kernel / gemv_priv_kernel (all projections + lm_head of a step)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(gemv_bytes / max(gemv_n, 1)),
                    "launches_per_step": gemv_n, "avg_launch_us": round(gemv_ms * 1e3 / max(gemv_n, 1), 2),
                    "gemv_only_ms_per_step": round(gemv_ms, 4),
                    "duration_source": "HIP events around graph replays of the step's GEMV launches alone (launch-to-launch)",
                    "step_algorithmic_GB": round(step_bytes / 1e9, 3),
                    "step_frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    else:
        roofline = {"bound": "hbm", "kernel": "per-GPU step (weights / TP + KV / TP)", "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": None, "step_algorithmic_GB": round(step_bytes / 1e9, 3)}

    # ---- fused speculation (SURVEY 8f-4 tail): what a speculation step costs next to a plain step --------
    speculation = None
    if tp == 1 and args.model == "llama31_8b" and wd != "bf16" and not args.no_speculation:
        speculation = speculation_section(torch, HF, wd, ms_per_step, args.steps)

    cpu = None
    if tp == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(MODELS["llama31_8b"] if args.model != "qwen25_7b" else HF, MAX_NUM_SEQS, DECODE_CTX)

    # ---- context encoding against the MFMA roofline: algorithmic flops = 2 W N + 4 L nh hd N^2 / 2 ----
    def prefill_flops(n):
        head = HF["vocab_size"] * HF["hidden_size"]                 # lm_head: one row (the last token) only
        body = weight_bytes_per_step(HF, 1, 1) - head
        return 2.0 * body * n + 2.0 * head + 4.0 * HF["num_hidden_layers"] * HF["num_attention_heads"] * HF["head_dim"] * n * n / 2.0
    ttft_frac = {"peak_PFLOPs": {"weight_only_bf16_mfma": 2.5 * tp, "fp8_activations_mx_fp8_mfma": 5.0 * tp},
                 "weight_only": {b: round(prefill_flops(int(b) - 17) / (t_ms * 1e-3) / (2.5e15 * tp), 4) for b, t_ms in ttft.items()},
                 "fp8_activations": ({b: round(prefill_flops(int(b) - 17) / (t_ms * 1e-3) / (5.0e15 * tp), 4) for b, t_ms in ttft_a8.items()}
                                     if ttft_a8 else None),
                 "note": "whole TTFT (scheduler, H2D, every kernel of the bucket, logits D2H, CPU sampler) against the dense MFMA "
                         "peak of the dtype; prompts are bucket - 17 tokens; flops = 2 (W - lm_head) N + 2 lm_head + 4 L nh hd N^2 / 2 (SURVEY 8d counts the lm_head for every token: 7 % more)"}

    label = MODEL_LABEL[args.model]
    wname = {"f8e4m3": "FP8", "int8": "INT8", "bf16": "bf16"}[wd]
    line = {
        "metric": f"decode tokens/sec ({label} {wname}, block_size=32, max_num_seqs=4): device step with resident inputs; "
                  "p50 TTFT in ttft_p50_ms; engine-level rate in engine_decode_tokens_per_s",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": tp, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": {"f8e4m3": "fp8_e4m3 weights x bf16 activations (weight-only quantization), f32 accumulate",
                  "int8": "int8 weights x bf16 activations, f32 accumulate",
                  "bf16": "bf16, f32 accumulate"}[wd],
        "data": "synthetic (seeded N(0,0.02) weights at real shapes; random token ids)",
        "config": {"workload": f"{label} {wd} TP={tp}: token generation B={MAX_NUM_SEQS} "
                               f"ctx={DECODE_CTX}, block_size={BLOCK_SIZE}, pa_num_blocks={PA_NUM_BLOCKS}, "
                               f"max_model_len={MAX_MODEL_LEN}, buckets={BUCKETS}",
                   "parallelism": f"tp{tp}" + (" (all shards on GPU 0: functional check)" if args.tp_loopback else ""),
                   "global_batch": MAX_NUM_SEQS, "ctx": DECODE_CTX},
        "engine_decode_tokens_per_s": engine_cpu,
        "engine_decode_tokens_per_s_on_device_sampling": engine_dev,
        "engine_decode_tokens_per_s_on_device_sampling_top_k50_top_p09": engine_dev_sampled,
        "engine_decode_tokens_per_s_chunked_prefill_scheduler": engine_chunked,
        "ttft_p50_ms": ttft, "ttft_mode": "weight-only quantization (the parity path)",
        "ttft_p50_ms_fp8_activations": ttft_a8, "fp8_activation_first_token_agreement": agreement,
        "fp8_activation_agreement_gated": ({"rule": "a prompt qualifies when the weight-only top-2 logit gap exceeds 2 x the rms of "
                                            "(FP8 x FP8 logits - weight-only logits) over its vocabulary row", "per_bucket": agreement_gated}
                                           if agreement_gated else None),
        "ttft_frac_of_mfma_peak": ttft_frac,
        "prefix_cache_ttft_ms": prefix_ttft,
        "device_ms_per_step": round(dev_ms / args.steps, 4),
        "decode_tokens_per_s_by_ctx": {**by_ctx, str(DECODE_CTX): round(value, 1)},
        "pcie_inclusive_tokens_per_s": round(MAX_NUM_SEQS / e2e, 2),
        "speculation": speculation,
        "init_s": round(init_s, 2), "roofline": roofline, "cpu_baseline": cpu,
    }
    if tp > 1:
        # what the tensor-parallel group actually ran on: "N ranks over peer memory, in hipGraphs" and
        # "fell back to RCCL, eager" must be told apart in a SCALE record
        try:
            line["tp"] = native.tp_info()
        except Exception as ex:                                   # noqa: BLE001 -- the bench line must still be printed
            line["tp"] = {"error": str(ex)}
        if tp_note:
            line["tp"]["note"] = tp_note
    finish(line)


if __name__ == "__main__":
    main()
