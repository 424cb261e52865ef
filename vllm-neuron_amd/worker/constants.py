# SPDX-License-Identifier: Apache-2.0
"""Dtype-name resolution and architecture lists of the MI355X backend."""
import torch

_F32, _F16, _BF16 = "float32", "float16", "bfloat16"


def _amp_table() -> dict:
    """What a vLLM `ModelConfig.dtype` (string or torch dtype) means for the weights the library
    keeps unquantized.  Semantics as the reference resolves them
    (vllm_neuron/worker/constants.py:9-19) -- notably "auto" means fp32, not "follow the
    checkpoint"."""
    table = {name: name for name in (_F32, _F16, _BF16)}
    table.update(auto=_F32, float=_F32, half=_F16)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        table[dt] = str(dt).rsplit(".", 1)[-1]
    return table


TORCH_DTYPE_TO_MI355X_AMP = _amp_table()

# served through multimodal model classes by the reference: out of this backend's scope
MI355X_MULTI_MODAL_MODELS = [f"{family}ForConditionalGeneration" for family in ("Mllama", "Llava", "Llama4")]

SUPPORTED_ARCHITECTURES = ("LlamaForCausalLM", "Qwen2ForCausalLM", "MistralForCausalLM")
