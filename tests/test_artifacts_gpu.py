"""Weight artifacts (reference loader.py:160-226 compiled-artifact cache, :888-891
quantized_checkpoints_path): the device images of the weights -- quantized, tiled, sharded -- saved
once and streamed back at the next start instead of reading, quantizing and tiling the checkpoint.

  * library level: save -> fresh context -> load gives bit-identical logits; a context of another
    quantization / sharding refuses the file (ValueError: the caller rebuilds), a truncated file too;
  * in-process tensor parallelism: one file per rank;
  * plugin level: a local safetensors checkpoint directory gets <dir>/mi355x-compiled-artifacts/<md5>
    on the first start and is served from it on the second (no tensor of the checkpoint is read:
    the safetensors files are emptied in between); `quantized_checkpoints_path` names the directory
    explicitly; a changed configuration hashes to another directory.
"""
import os

import pytest
import torch

from oracle.synth import make_prompts, make_weights, zoo_config
from tests.helpers import decode_inputs, prefill_inputs
from tests.test_engine_gpu import hf_like
from tests.test_model_gpu import BS, MAXLEN, MB, native_model
from tests.test_tp_group_gpu import _group_model

pytestmark = pytest.mark.gpu


def _probe(model, cfg):
    p = make_prompts(cfg.vocab_size, 0)[3]
    blocks = [1 + j for j in range(MB)]
    a = model.forward(**prefill_inputs(p, blocks, BS, MAXLEN, 0))
    b = model.forward(**decode_inputs([int(a.argmax())], [len(p)], [blocks], BS, MAXLEN))
    return torch.cat([a, b])


@pytest.mark.parametrize("wd,qt", [("f8e4m3", "per_channel_symmetric"), ("int8", "per_tensor_symmetric"),
                                   ("bf16", "per_tensor_symmetric")])
def test_saved_images_reload_bit_identically(tmp_path, wd, qt):
    cfg = zoo_config("qwen25_like")                       # qkv bias included
    w = make_weights(cfg, seed=1)
    m = native_model(cfg, w, wd, qt)
    want = _probe(m, cfg)
    m.save_artifacts(str(tmp_path))
    m.close()
    assert os.path.exists(tmp_path / "rank0_of1.miw")
    m2 = native_model(cfg, None, wd, qt, artifacts=str(tmp_path))
    assert torch.equal(_probe(m2, cfg), want)
    m2.close()
    # another quantization must not accept these images
    other = ("int8", "per_tensor_symmetric") if wd != "int8" else ("f8e4m3", "per_channel_symmetric")
    with pytest.raises(ValueError, match="another model"):
        native_model(cfg, None, *other, artifacts=str(tmp_path))
    # a truncated file neither
    f = tmp_path / "rank0_of1.miw"
    data = f.read_bytes()
    f.write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError, match="truncated"):
        native_model(cfg, None, wd, qt, artifacts=str(tmp_path))
    with pytest.raises(ValueError, match="not found"):
        native_model(cfg, None, wd, qt, artifacts=str(tmp_path / "nowhere"))


def test_tensor_parallel_group_saves_one_file_per_rank(tmp_path):
    cfg = zoo_config("llama31_like")
    w = make_weights(cfg, seed=1)
    m = _group_model(cfg, 2, "f8e4m3", "per_channel_symmetric", weights=w)
    want = _probe(m, cfg)
    m.save_artifacts(str(tmp_path))
    m.close()
    assert sorted(os.listdir(tmp_path)) == ["rank0_of2.miw", "rank1_of2.miw"]
    from vllm_neuron_amd._native import MI_Q, MI_TP_ALL_RANKS, MI_W, NativeModel
    rs = cfg.rope_scaling or {}
    m2 = NativeModel(
        num_layers=cfg.num_layers, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads,
        head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, rope_type=1, rope_factor=rs["factor"],
        rope_low_freq_factor=rs["low_freq_factor"], rope_high_freq_factor=rs["high_freq_factor"],
        rope_original_max_position=rs["original_max_position_embeddings"], qkv_bias=0, tie_word_embeddings=0,
        num_blocks=m.cfg.num_blocks, block_size=BS, max_num_seqs=m.cfg.max_num_seqs, max_model_len=MAXLEN,
        weight_dtype=MI_W["f8e4m3"], quant_type=MI_Q["per_channel_symmetric"], quantize_lm_head=1, tp_degree=2,
        tp_rank=MI_TP_ALL_RANKS, tp_device_ids=[0, 0], device_id=0, use_graphs=1)
    m2.load_artifacts(str(tmp_path))
    m2.finalize()
    assert torch.equal(_probe(m2, cfg), want)
    m2.close()


def test_plugin_builds_and_reuses_the_artifact_directory(tmp_path):
    from safetensors.torch import save_file
    from vllm_neuron_amd._vllm_compat import SamplingParams
    from vllm_neuron_amd.engine import MI355XEngine
    name = "llama31_like"
    cfg = zoo_config(name)
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in make_weights(cfg, 1).items()},
              str(ckpt / "model.safetensors"))
    prompts = make_prompts(cfg.vocab_size, 0)
    q = dict(quantized=True, quantization_dtype="f8e4m3", quantization_type="per_channel_symmetric")

    def run(override, model=str(ckpt)):
        eng = MI355XEngine(hf_like(name), model=model, max_model_len=256, max_num_seqs=4, block_size=32,
                           override_mi355x_config=dict(override))
        adapter = eng.worker.model_runner.model
        toks = [o.token_ids for o in eng.generate(prompts, SamplingParams(temperature=0.0, max_tokens=8))]
        info = (adapter.loaded_from_artifacts, adapter.compiled_artifacts_path)
        adapter.model.close()
        return toks, info

    toks1, (hit1, dir1) = run(q)
    assert not hit1 and dir1.startswith(str(ckpt / "mi355x-compiled-artifacts")) and sorted(os.listdir(dir1)) == ["checkpoint.json", "rank0_of1.miw"]
    # second start: the checkpoint's tensors are not needed any more
    st = os.stat(ckpt / "model.safetensors")
    (ckpt / "model.safetensors").write_bytes(b"")
    os.truncate(ckpt / "model.safetensors", st.st_size)
    os.utime(ckpt / "model.safetensors", (st.st_atime, st.st_mtime))   # same name / size / mtime -> same hash (set LAST: truncate touches it)
    toks2, (hit2, dir2) = run(q)
    assert hit2 and dir2 == dir1 and toks2 == toks1
    # another configuration -> another directory (and the emptied checkpoint can no longer serve it)
    with pytest.raises(Exception):
        run(dict(q, quantization_dtype="int8"))
    # quantized_checkpoints_path names the directory explicitly (reference loader.py:888-891)
    explicit = tmp_path / "quantized"
    toks3, (hit3, dir3) = run(dict(q, quantized_checkpoints_path=str(explicit), state_dict=make_weights(cfg, 1)), model="")
    assert not hit3 and dir3 == str(explicit) and toks3 == toks1
    toks4, (hit4, _) = run(dict(q, quantized_checkpoints_path=str(explicit), state_dict={}), model="")
    assert hit4 and toks4 == toks1
    # ... but another checkpoint of the same shape must not pick it up silently: rebuilt from the weights at hand
    other = make_weights(cfg, 2)
    toks5, (hit5, dir5) = run(dict(q, quantized_checkpoints_path=str(explicit), state_dict=other), model="")
    assert not hit5 and dir5 == str(explicit) and toks5 != toks1
    toks6, (hit6, _) = run(dict(q, quantized_checkpoints_path=str(explicit), state_dict=other), model="")
    assert hit6 and toks6 == toks5
    # a directory that cannot be written does not fail the start (the weights are resident already)
    blocked = tmp_path / "blocked"
    blocked.write_text("a file where the directory should go")
    toks7, (hit7, dir7) = run(dict(q, quantized_checkpoints_path=str(blocked), state_dict=make_weights(cfg, 1)), model="")
    assert not hit7 and dir7 is None and toks7 == toks1
