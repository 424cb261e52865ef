# SPDX-License-Identifier: Apache-2.0
"""Out-of-tree vLLM platform for AMD Instinct MI355X (gfx950).

Same hook set and the same config rewrites as the reference platform
(/root/reference/vllm_neuron/platform.py:25-262): OOT enum, worker / scheduler class
injection, "uni" executor, +1 null block on ``num_gpu_blocks_override``,
``block_size = max_model_len`` unless prefix caching is on, scheduler defaults.

``device_type`` stays "cpu" for the same reason as in the reference: every tensor vLLM's
engine core touches (token ids, block tables, logits for the CPU sampler) lives on the host;
device memory is owned by libmi355x_vllm.so behind the model runner.
"""

import enum
import logging
import os
from typing import TYPE_CHECKING

from ._vllm_compat import Platform, PlatformEnum

if TYPE_CHECKING:  # pragma: no cover
    from vllm.config import ModelConfig, ParallelConfig, VllmConfig

logger = logging.getLogger(__name__)

WORKER_CLS = "vllm_neuron_amd.worker.mi355x_worker.MI355XWorker"
SCHEDULER_CLS = "vllm_neuron_amd.core.scheduler.ContinuousBatchingMI355XScheduler"
_NULL_BLOCK_MARK = "_mi355x_null_block_adjusted"


class MI355XFramework(enum.Enum):
    HIP_NATIVE = "libmi355x_vllm"


class MI355XPlatform(Platform):
    _enum = PlatformEnum.OOT
    device_name: str = "cpu"
    device_type: str = "cpu"
    ray_device_key: str = "GPU"
    # weight-only int8 / fp8 are applied at load time by the HIP library ("quantized",
    # "quantization_dtype", "quantization_type" override keys), like the reference's neuron_quant
    supported_quantization: list[str] = ["mi355x_quant", "fbgemm_fp8"]
    device_control_env_var: str = "HIP_VISIBLE_DEVICES"

    # ModelConfig monkeypatches are applied once per process
    _config_overrides_applied = False

    def __init__(self):
        super().__init__()
        self._ensure_config_overrides_applied()

    @classmethod
    def _ensure_config_overrides_applied(cls) -> None:
        """Relax the upstream ModelConfig verifiers exactly where the reference does
        (platform.py:42-113): TP degree need not divide the head count (the library replicates
        KV heads), quantization / cuda-graph verification do not apply, and the user's
        max_model_len is trusted.  Re-applied in every (spawned) process; idempotent."""
        if cls._config_overrides_applied:
            return
        try:
            from vllm.config import ModelConfig
        except ImportError as e:
            logger.warning("vLLM config module not importable, skipping ModelConfig overrides: %s", e)
            return

        def verify_with_parallel_config(self, parallel_config: "ParallelConfig") -> None:
            if parallel_config.distributed_executor_backend == "external_launcher":
                assert self.seed is not None, (
                    "Seed must be set when using external launcher backend to "
                    "make sure sampling results are the same across workers.")
            if parallel_config.enable_expert_parallel:
                self._verify_with_expert_parallelism()
            if parallel_config.pipeline_parallel_size > 1:
                if not self.registry.is_pp_supported_model(self.architectures):
                    raise NotImplementedError(
                        "Pipeline parallelism is not supported for this model. "
                        "Supported models implement the `SupportsPP` interface.")
                if self.use_async_output_proc:
                    self.use_async_output_proc = False

        def get_and_verify_max_len(self, max_model_len: int):
            if self.spec_target_max_model_len is not None:
                return self.spec_target_max_model_len
            return max_model_len

        ModelConfig.verify_with_parallel_config = verify_with_parallel_config
        ModelConfig._verify_quantization = lambda self: None
        ModelConfig._verify_cuda_graph = lambda self: None
        ModelConfig.get_and_verify_max_len = get_and_verify_max_len
        cls._config_overrides_applied = True
        logger.info("MI355X ModelConfig overrides applied")

    @classmethod
    def get_device_name(cls, device_id: int = 0) -> str:
        return "mi355x"

    @classmethod
    def is_async_output_supported(cls, enforce_eager: bool | None) -> bool:
        return False

    @classmethod
    def pre_register_and_update(cls, parser=None) -> None:
        cls._ensure_config_overrides_applied()

    @classmethod
    def check_and_update_config(cls, vllm_config: "VllmConfig") -> None:
        cls._ensure_config_overrides_applied()
        # vLLM validates every VllmConfig it builds, including empty default ones
        if vllm_config.model_config is None:
            return

        native_scheduler = bool(int(os.getenv("DISABLE_MI355X_CUSTOM_SCHEDULER", "0")))

        # vLLM allocates block 0 lazily as the null block: hand it one more than the user asked
        # for, once per CacheConfig instance
        cache_config = vllm_config.cache_config
        if cache_config and cache_config.num_gpu_blocks_override is not None \
                and _NULL_BLOCK_MARK not in cache_config.__dict__:
            logger.info("num_gpu_blocks_override %d -> %d (null block)",
                        cache_config.num_gpu_blocks_override, cache_config.num_gpu_blocks_override + 1)
            cache_config.num_gpu_blocks_override += 1
            setattr(cache_config, _NULL_BLOCK_MARK, True)

        parallel_config = vllm_config.parallel_config
        if parallel_config.worker_cls == "auto":
            parallel_config.worker_cls = WORKER_CLS
        if parallel_config.world_size > 1:
            # one engine-side worker; the library context behind it drives every GPU of the
            # tensor-parallel group (rank shards on threads of libmi355x_vllm), as NxDI does in the reference
            parallel_config.distributed_executor_backend = "uni"

        if native_scheduler:
            logger.warning("The vLLM V1 native scheduler will be used with chunked prefill enabled: ragged "
                           "token batches through mi_forward_chunked.")
            assert vllm_config.cache_config.block_size is not None, (
                "When vLLM V1 native scheduler is enabled, block_size must be set.")
            return

        sched = vllm_config.scheduler_config
        sched.scheduler_cls = SCHEDULER_CLS
        sched.chunked_prefill_enabled = False
        sched.max_num_batched_tokens = 131072
        if not sched.max_num_seqs:
            sched.max_num_seqs = 32
        if not vllm_config.cache_config.enable_prefix_caching:
            # contiguous ("batch line") KV: one block per sequence
            vllm_config.cache_config.block_size = vllm_config.model_config.max_model_len
        else:
            assert vllm_config.cache_config.block_size is not None, (
                "When prefix caching is enabled, block_size must be set.")

    @classmethod
    def is_pin_memory_available(cls) -> bool:
        return False

    @classmethod
    def use_all_gather(cls) -> bool:
        return True

    @classmethod
    def supports_v1(cls, model_config: "ModelConfig") -> bool:
        return True

    @classmethod
    def is_hip_native_available(cls) -> bool:
        from ._native import LIB_PATH
        return os.path.exists(LIB_PATH)

    def get_framework_to_use(self):
        if not self.is_hip_native_available():
            raise AssertionError("libmi355x_vllm.so is not built; run `python __graft_entry__.py`.")
        return MI355XFramework.HIP_NATIVE
