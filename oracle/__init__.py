"""CPU oracle for the MI355X paged-KV decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``vllm_neuron_amd``) never imports this package and fails loudly when the HIP
library is missing.

What it restates (reference = vllm-project/vllm-neuron @ 2025-11-21):

* the per-step *input contract* the reference runner hands to the model
  (``vllm_neuron/worker/neuronx_distributed_model_runner.py:704-936``) and the
  model call + last-token logits slice
  (``vllm_neuron/worker/neuronx_distributed_model_loader.py:336-365``);
* the decoder arithmetic that the reference outsources to the third-party
  package ``neuronx_distributed_inference`` (AWS Neuron SDK 2.26.1,
  ``README.md:11``; absent from ``/root/reference`` and from this image).  The
  reference's own oracle of record for that arithmetic is HF-transformers
  greedy decoding (``test/e2e/online/online_server_runner.py:95-146``), so the
  restatement follows the published Llama / Qwen2 decoder equations and is
  PINNED against ``transformers`` ``LlamaForCausalLM`` / ``Qwen2ForCausalLM``
  outputs generated in the build container (``oracle/gen_golden.py`` →
  ``tests/golden/*.safetensors``);
* the CPU-sampling tail (greedy == argmax,
  ``neuronx_distributed_model_runner.py:1142-1239``).

Pinning status
  bf16 / fp32 decoder path ........ pinned (HF transformers fixtures)
  INT8 / FP8 weight-quantized path . PARITY UNPINNED: no reference test checks a
      quantized result; the quantizer semantics are this
      oracle's own statement of ``per_tensor_symmetric`` /
      ``per_channel_symmetric`` (loader.py:886-898).
"""

from .quant import dequantize_weight, quantize_weight  # noqa: F401
from .paged_decoder import DecoderConfig, PagedDecoderOracle  # noqa: F401
