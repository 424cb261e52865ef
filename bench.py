#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s + p50 TTFT, Llama-3.1-8B FP8 weights, block_size=32,
max_num_seqs=4 (BASELINE.json), on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one token-generation pass of the hot path over one batch: B = max_num_seqs = 4
sequences at context 1024 (BASELINE.md roofline row), all 32 layers + lm_head, through
libmi355x_vllm.  The timed region is exactly K steps with every input resident in HBM
(mi_replay_decode: hipGraph replays bracketed by HIP events on the library's stream, and by
barrier + torch.cuda.synchronize on both sides); `value` = B * K / max-over-ranks time.  With
N > 1 the model is tensor-parallel over the N GPUs (one process per GPU, RCCL all-reduce of the
row-parallel partials): total work is fixed -> "scaling": "strong".

Also reported (same JSON line): p50 TTFT through the whole plugin path (scheduler -> runner ->
library -> CPU sampler) per context-encoding bucket, the PCIe-inclusive decode rate through
mi_forward, the roofline of the dominant kernel (the weight-streaming GEMV) and, at N = 1, the
CPU oracle timed on the host cores over a bounded sample of the same workload.

Weights are synthetic N(0, 0.02) (no checkpoints exist offline), generated on the device at the
real Llama-3.1-8B shapes and quantized per-channel to OCP e4m3.
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

LLAMA31_8B = dict(
    architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=128256, hidden_size=4096,
    intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
    head_dim=128, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False,
    rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                  "original_max_position_embeddings": 8192})
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
    return B * ctx * 2 * (hf["num_key_value_heads"] / tp) * hf["head_dim"] * 2 * hf["num_hidden_layers"]


def cpu_baseline(hf, B, ctx, layers=2, steps=4):
    """The CPU oracle (test infrastructure, `oracle/`) timed on this box's host cores over a
    bounded sample: `layers` of the 32 decoder layers + the lm_head, `steps` decode steps at
    B x ctx; extrapolated linearly in the layer count."""
    import torch
    from oracle import DecoderConfig, PagedDecoderOracle
    # a 1-GPU box grants 16 host cores however many the machine reports
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("MI_BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(ncores)
    cfg = DecoderConfig(num_layers=layers, hidden_size=hf["hidden_size"], num_heads=hf["num_attention_heads"],
                        num_kv_heads=hf["num_key_value_heads"], head_dim=hf["head_dim"],
                        intermediate_size=hf["intermediate_size"], vocab_size=hf["vocab_size"],
                        rms_norm_eps=hf["rms_norm_eps"], rope_theta=hf["rope_theta"],
                        rope_scaling=hf["rope_scaling"])
    g = torch.Generator().manual_seed(1)
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
    mb = MAX_MODEL_LEN // BLOCK_SIZE
    oracle = PagedDecoderOracle(cfg, w, 1 + B * mb, BLOCK_SIZE, compute="fp32")
    oracle.kv.normal_(generator=g)
    from tests.helpers import decode_inputs
    blocks = [[1 + b * mb + j for j in range(mb)] for b in range(B)]
    inp = decode_inputs([1] * B, [ctx - 1] * B, blocks, BLOCK_SIZE, MAX_MODEL_LEN)

    def timed(o, n):
        o.forward(**inp)
        t = time.perf_counter()
        for _ in range(n):
            o.forward(**inp)
        return (time.perf_counter() - t) / n
    t_full = timed(oracle, steps)
    # lm_head + embedding alone: the same oracle with zero layers
    cfg0 = DecoderConfig(**{**cfg.__dict__, "num_layers": 0})
    o0 = PagedDecoderOracle(cfg0, {k: v for k, v in w.items() if "layers" not in k}, 2, BLOCK_SIZE, compute="fp32")
    t_head = timed(o0, steps)
    t_layer = max(t_full - t_head, 0.0) / layers
    t_step = t_layer * hf["num_hidden_layers"] + t_head
    return {"value": round(B / t_step, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle (torch CPU fp32) decode at B={B}, ctx={ctx}: {steps} steps over {layers} of "
                       f"{hf['num_hidden_layers']} layers + lm_head, per-layer time extrapolated to 32 layers "
                       f"({t_layer * 1e3:.1f} ms/layer, {t_head * 1e3:.1f} ms head)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ttft-prompts", type=int, default=32, help="prompts per context-encoding bucket (p50)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weight-dtype", default="f8e4m3", choices=["f8e4m3", "int8", "bf16"])
    ap.add_argument("--bf16-prefill-activations", action="store_true",
                    help="weight-only quantization in the context-encoding GEMMs too (default: FP8 x FP8)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        # host-side control plane only (unique-id broadcast, barriers, max-over-ranks of the wall
        # clock): gloo.  The data-path collectives are RCCL calls inside the library, on its own
        # communicator and stream; a second (torch) RCCL communicator per process would add nothing.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def barrier_sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from vllm_neuron_amd._vllm_compat import SamplingParams
    from vllm_neuron_amd.engine import MI355XEngine
    hf = SimpleNamespace(**LLAMA31_8B)
    override = {"synthetic_weights": {"seed": 1, "std": 0.02}, "context_encoding_buckets": BUCKETS,
                "pa_num_blocks": PA_NUM_BLOCKS}
    if args.weight_dtype != "bf16":
        override.update(quantized=True, quantization_dtype=args.weight_dtype,
                        quantization_type="per_channel_symmetric")
    if args.weight_dtype == "f8e4m3" and not args.bf16_prefill_activations:
        # context-encoding GEMMs: per-token FP8 activations on the MX-scaled MFMA (FP8 x FP8)
        override["prefill_fp8_activations"] = True
    t0 = time.perf_counter()
    eng = MI355XEngine(hf, max_model_len=MAX_MODEL_LEN, max_num_seqs=MAX_NUM_SEQS, block_size=BLOCK_SIZE,
                       num_gpu_blocks_override=PA_NUM_BLOCKS, enable_prefix_caching=True,
                       tensor_parallel_size=world, override_mi355x_config=override, rank=rank,
                       local_rank=local_rank)
    native = eng.worker.model_runner.model.model
    init_s = time.perf_counter() - t0

    # ---- p50 TTFT per bucket, through scheduler -> runner -> library -> CPU sampler ----------
    g = torch.Generator().manual_seed(0)

    def ttft_sweep(engine, tag):
        res = {}
        for bucket in BUCKETS:
            samples = []
            for _ in range(args.ttft_prompts + 1):           # first one is warm-up
                prompt = torch.randint(0, hf.vocab_size, (bucket - 17,), generator=g).tolist()
                out = engine.generate([prompt], SamplingParams(temperature=0.0, max_tokens=1))[0]
                samples.append(out.ttft_s * 1e3)
            res[str(bucket)] = round(statistics.median(samples[1:]), 3)
            if rank == 0:
                print(f"[bench] ttft {tag} bucket {bucket}: {[round(x, 2) for x in samples]} ms", file=sys.stderr, flush=True)
        return res
    ttft = ttft_sweep(eng, "fp8xfp8" if override.get("prefill_fp8_activations") else "weight-only")
    # the other context-encoding numerics mode, for the record: FP8 weights x bf16 activations on the
    # bf16 MFMA (what the reference's weight-only quantized linears compute); one more engine
    ttft_wo = None
    if override.get("prefill_fp8_activations") and world == 1:
        o2 = dict(override)
        o2["prefill_fp8_activations"] = False
        eng2 = MI355XEngine(hf, max_model_len=MAX_MODEL_LEN, max_num_seqs=MAX_NUM_SEQS, block_size=BLOCK_SIZE,
                            num_gpu_blocks_override=PA_NUM_BLOCKS, enable_prefix_caching=True,
                            tensor_parallel_size=world, override_mi355x_config=o2, rank=rank, local_rank=local_rank)
        ttft_wo = ttft_sweep(eng2, "weight-only")
        eng2.worker.model_runner.model.model.close()
        del eng2

    # ---- the whole serving loop (scheduler -> runner -> library -> sampler), for the record: 4
    #      requests of 900 prompt tokens decoding 128 tokens each, first with the CPU sampler (the
    #      parity path: [B, V] fp32 logits cross PCIe every step), then with on-device sampling
    #      (B ids cross).  Rates are of the decode phase; not the headline `value`.
    def engine_decode_rate():
        prompts = [torch.randint(0, hf.vocab_size, (900,), generator=g).tolist() for _ in range(MAX_NUM_SEQS)]
        t_ = time.perf_counter()
        outs = eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=128))
        dt = time.perf_counter() - t_
        ntok = sum(len(o.token_ids) for o in outs)
        first = max(o.ttft_s for o in outs)
        return round((ntok - len(outs)) / (dt - first), 1)
    engine_rates = {"cpu_sampling": engine_decode_rate()}
    model_adapter = eng.worker.model_runner.model
    model_adapter.mi355x_config.on_device_sampling_config = {"dynamic": True, "deterministic": False}
    engine_rates["on_device_sampling"] = engine_decode_rate()
    model_adapter.mi355x_config.on_device_sampling_config = None

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

    # ---- decode: B sequences at context DECODE_CTX ------------------------------------------
    inp = decode_inputs(toks, [DECODE_CTX - 1] * MAX_NUM_SEQS, blocks, BLOCK_SIZE, MAX_MODEL_LEN)
    for _ in range(2):
        native.forward(**inp)                            # captures the graph, leaves inputs resident
    native.replay_decode(max(args.warmup, 1))            # W untimed warm-up steps
    barrier_sync()
    t = time.perf_counter()
    dev_ms = native.replay_decode(args.steps)            # exactly K timed steps
    torch.cuda.synchronize()
    wall = time.perf_counter() - t
    barrier_sync()
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt[0])
    ms_per_step = wall * 1e3 / args.steps
    value = MAX_NUM_SEQS * args.steps / wall

    # PCIe-inclusive rate: the same step through mi_forward (ids H2D, [B, V] fp32 logits D2H)
    barrier_sync()
    t = time.perf_counter()
    for _ in range(args.steps):
        native.forward(**inp)
    e2e = (time.perf_counter() - t) / args.steps

    # ---- roofline of the dominant kernel (weight-streaming GEMV), HIP events per launch --------
    native.profile_enable(True)
    for _ in range(4):
        native.forward(**inp)
    prof = native.profile_read()
    native.profile_enable(False)
    gemv_ms, gemv_n = prof["ms"]["gemv"], prof["launches"]["gemv"]
    achieved = prof["gemv_weight_bytes"] / (gemv_ms * 1e-3) / 1e9 if gemv_ms > 0 else 0.0
    step_bytes = weight_bytes_per_step(LLAMA31_8B, world, 2 if args.weight_dtype == "bf16" else 1) + kv_bytes_per_step(LLAMA31_8B, world, MAX_NUM_SEQS, DECODE_CTX)
    # HBM traffic per GEMV launch from the committed PMC pass of this same command (FETCH_SIZE,
    # gfx950-corrected; tests/pmc_summary.py) -- counters cannot be read from inside the process
    traffic, traffic_src = None, None
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gemv_traffic.json")
    if world == 1 and args.weight_dtype == "f8e4m3" and os.path.exists(tf):
        with open(tf) as fh:
            traffic = json.load(fh)["traffic_bytes_per_launch"]
        traffic_src = "profiles/r01_gemv_traffic.json (rocprofv3 --pmc FETCH_SIZE pass, x2 gfx950 correction)"
    roofline = {"bound": "hbm", "kernel": "mi::gemv_kernel (all projections + lm_head of a step)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(prof["gemv_weight_bytes"] / max(gemv_n, 1)),
                "launches_per_step": gemv_n // 4, "avg_launch_us": round(gemv_ms * 1e3 / max(gemv_n, 1), 2),
                "step_algorithmic_GB": round(step_bytes / 1e9, 3),
                "step_frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(LLAMA31_8B, MAX_NUM_SEQS, DECODE_CTX)

    if rank == 0:
        line = {
            "metric": "decode tokens/sec (Llama-3.1-8B FP8, block_size=32, max_num_seqs=4); p50 TTFT in ttft_p50_ms",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"f8e4m3": ("fp8_e4m3 weights; token generation x bf16 activations, context encoding x "
                                 + ("bf16" if args.bf16_prefill_activations else "per-token fp8_e4m3")
                                 + " activations; f32 accumulate"),
                      "int8": "int8 weights x bf16 activations, f32 accumulate",
                      "bf16": "bf16, f32 accumulate"}[args.weight_dtype],
            "data": "synthetic (seeded N(0,0.02) weights at real shapes; random token ids)",
            "config": {"workload": f"Llama-3.1-8B {args.weight_dtype} TP={world}: token generation B={MAX_NUM_SEQS} "
                                   f"ctx={DECODE_CTX}, block_size={BLOCK_SIZE}, pa_num_blocks={PA_NUM_BLOCKS}, "
                                   f"max_model_len={MAX_MODEL_LEN}, buckets={BUCKETS}",
                       "parallelism": f"tp{world}", "global_batch": MAX_NUM_SEQS, "ctx": DECODE_CTX},
            "ttft_p50_ms": ttft, "ttft_p50_ms_bf16_activations": ttft_wo, "device_ms_per_step": round(dev_ms / args.steps, 4),
            "decode_tokens_per_s_by_ctx": {**by_ctx, str(DECODE_CTX): round(value, 1)},
            "engine_decode_tokens_per_s": engine_rates,
            "pcie_inclusive_tokens_per_s": round(MAX_NUM_SEQS / e2e, 2),
            "init_s": round(init_s, 2), "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
