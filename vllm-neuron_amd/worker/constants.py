# SPDX-License-Identifier: Apache-2.0
import torch

# same table as the reference (vllm_neuron/worker/constants.py:9-19), including "auto" -> float32
TORCH_DTYPE_TO_MI355X_AMP = {
    "auto": "float32",
    "half": "float16",
    "float16": "float16",
    "bfloat16": "bfloat16",
    "float": "float32",
    "float32": "float32",
    torch.float16: "float16",
    torch.bfloat16: "bfloat16",
    torch.float32: "float32",
}

# architectures handled by the reference through multimodal NxDI classes: out of scope here
MI355X_MULTI_MODAL_MODELS = [
    "MllamaForConditionalGeneration", "LlavaForConditionalGeneration",
    "Llama4ForConditionalGeneration",
]

SUPPORTED_ARCHITECTURES = ("LlamaForCausalLM", "Qwen2ForCausalLM", "MistralForCausalLM")
