"""Chunked prefill (reference runner.py:938-1051, loader.py:357-361, platform.py:146-175):
ONE ragged token batch per step through mi_forward_chunked -- prompt chunks and single generation
tokens of several requests concatenated, every request attending to its own blocks.

  * the C entry against the CPU oracle: a hand-built schedule (a long prompt encoded in three
    chunks while two other requests generate), every returned row compared with the oracle's
    logits for the same request state (the oracle sees each request on its own: full token list +
    computed_context_lens, the reference's prefix-caching contract);
  * the whole plugin path: MI355XEngine(enable_chunked_prefill=True) = vLLM's native scheduler
    (stand-in) -> runner `_prepare_chunked_prefill_inputs` -> adapter -> library -> CPU sampler,
    with a token budget smaller than the prompts, against the committed HF greedy goldens; and the
    same with on-device sampling.
"""
import pytest
import torch

from oracle import PagedDecoderOracle
from oracle.synth import ZOO, make_prompts, make_weights, zoo_config
from tests.helpers import prefill_inputs
from tests.test_engine_gpu import check_against_golden, hf_like
from tests.test_model_gpu import native_model

pytestmark = pytest.mark.gpu
BS, MAXLEN, NSEQ = 32, 256, 4
MB = MAXLEN // BS
NB = 1 + NSEQ * MB


@pytest.mark.parametrize("name,wd,qt", [("llama31_like", "bf16", "per_tensor_symmetric"),
                                        ("llama31_like", "f8e4m3", "per_channel_symmetric"),
                                        ("qwen25_like", "int8", "per_channel_symmetric")])
def test_ragged_batches_match_oracle(name, wd, qt):
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    quant = None if wd == "bf16" else dict(quantized=True, quantization_dtype=wd, quantization_type=qt)
    oracle = PagedDecoderOracle(cfg, w, NB, BS, compute="bf16", quant=quant)
    from vllm_neuron_amd._native import MI_Q, MI_W, NativeModel
    rs = cfg.rope_scaling or {}
    model = NativeModel(
        num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads,
        head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, rope_type=1 if rs else 0,
        rope_factor=rs.get("factor", 1.0), rope_low_freq_factor=rs.get("low_freq_factor", 1.0),
        rope_high_freq_factor=rs.get("high_freq_factor", 4.0),
        rope_original_max_position=rs.get("original_max_position_embeddings", 0), qkv_bias=int(cfg.qkv_bias),
        tie_word_embeddings=int(cfg.tie_word_embeddings), num_blocks=NB, block_size=BS, max_num_seqs=NSEQ,
        max_model_len=MAXLEN, weight_dtype=MI_W[wd], quant_type=MI_Q[qt], quantize_lm_head=1, tp_degree=1, tp_rank=0,
        device_id=0, use_graphs=1, ctx_buckets=[MAXLEN])
    model.load_state_dict(w)
    model.finalize()
    g = torch.Generator().manual_seed(11)
    seqs = {r: torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for r, n in (("a", 150), ("b", 20), ("c", 45))}
    prompt_len = {r: len(t) for r, t in seqs.items()}
    blocks = {r: [1 + i * MB + j for j in range(MB)] for i, r in enumerate(seqs)}
    done = {r: 0 for r in seqs}                                   # tokens whose K/V are in the pool
    # (request, tokens scheduled) per step: a's prompt in three chunks while b and c encode and then generate
    schedule = [[("a", 64), ("b", 20)], [("a", 64), ("b", 1), ("c", 45)], [("a", 22), ("b", 1), ("c", 1)],
                [("a", 1), ("b", 1), ("c", 1)], [("c", 1), ("a", 1)]]
    worst = 0.0
    for step in schedule:
        ids, pos, slots, bts, full, comp = [], [], [], [], [], []
        for r, n in step:
            start, end = done[r], done[r] + n
            assert end <= len(seqs[r])
            ids += seqs[r][start:end]
            pos += list(range(start, end))
            slots += [blocks[r][i // BS] * BS + i % BS for i in range(start, end)]
            bts.append(blocks[r])
            full.append(end)
            comp.append(start)
        logits = model.forward_chunked(torch.tensor(ids), torch.tensor(pos), torch.tensor(slots), torch.tensor(bts),
                                       torch.tensor(full), torch.tensor(comp))
        toks = model.forward_chunked(torch.tensor(ids), torch.tensor(pos), torch.tensor(slots), torch.tensor(bts),
                                     torch.tensor(full), torch.tensor(comp), tokens=True)   # same state, idempotent
        assert toks.tolist() == logits.argmax(dim=1).tolist()
        for row, (r, n) in enumerate(step):
            start, end = done[r], done[r] + n
            ref = oracle.forward(**prefill_inputs(seqs[r][:end], blocks[r], BS, MAXLEN, start))
            worst = max(worst, (logits[row] - ref[0]).abs().max().item())
            done[r] = end
            if end >= prompt_len[r] and end == len(seqs[r]):       # generating: feed the greedy token back
                seqs[r].append(int(ref.argmax()))
    assert worst <= 0.06, worst
    model.close()


@pytest.mark.parametrize("name", list(ZOO))
@pytest.mark.parametrize("device_sampling", [False, True])
def test_engine_with_chunked_prefill_matches_hf_golden(name, device_sampling):
    from vllm_neuron_amd._vllm_compat import SamplingParams
    from vllm_neuron_amd.engine import MI355XEngine
    cfg = zoo_config(name)
    override = {"state_dict": make_weights(cfg, 1), "is_block_kv_layout": True,
                "chunked_prefill_config": {"max_num_seqs": 4}}
    if device_sampling:
        override["on_device_sampling_config"] = {"dynamic": True}
    eng = MI355XEngine(hf_like(name), max_model_len=256, max_num_seqs=4, block_size=32, enable_prefix_caching=False,
                       enable_chunked_prefill=True, max_num_batched_tokens=48, override_mi355x_config=override)
    assert eng.worker.model_runner.is_chunked_prefill
    prompts = make_prompts(cfg.vocab_size, 0)                     # 140-token prompt: three chunks of <= 48
    outs = eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=12))
    check_against_golden(name, outs)
    eng.worker.model_runner.model.model.close()


def test_engine_with_chunked_prefill_and_tensor_parallelism(monkeypatch):
    """vLLM's native scheduler on top of the in-process tensor-parallel group (both rank shards on
    GPU 0): ragged records fan out to the shards, all-decode records run as token-generation steps
    inside every shard, mixed records batch their 1-token requests into one decode-attention launch."""
    from vllm_neuron_amd._vllm_compat import SamplingParams
    from vllm_neuron_amd.engine import MI355XEngine
    monkeypatch.setenv("MI355X_TP_LOOPBACK", "1")
    name = "llama31_like"
    cfg = zoo_config(name)
    override = {"state_dict": make_weights(cfg, 1), "is_block_kv_layout": True,
                "chunked_prefill_config": {"max_num_seqs": 4}}
    eng = MI355XEngine(hf_like(name), max_model_len=256, max_num_seqs=4, block_size=32, enable_prefix_caching=False,
                       tensor_parallel_size=2, enable_chunked_prefill=True, max_num_batched_tokens=48,
                       override_mi355x_config=override)
    prompts = make_prompts(cfg.vocab_size, 0)
    outs = eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=12))
    check_against_golden(name, outs)
    eng.worker.model_runner.model.model.close()
