# SPDX-License-Identifier: Apache-2.0
"""vLLM worker for MI355X: lifecycle shim around the model runner, same method set as the
reference worker (/root/reference/vllm_neuron/worker/neuron_worker.py:21-146).

Tensor parallelism: the reference hides TP inside ONE worker process (NxDI drives all
NeuronCores).  The idiomatic MI355X layout is one process per GPU, so a TP group is
`tensor_parallel_size` copies of this worker, launched by torchrun / the bench harness with
RANK / LOCAL_RANK / WORLD_SIZE set; they rendezvous through torch.distributed only to share the
128-byte RCCL unique id, after which every collective (all-reduce of the row-parallel partials,
all-gather of the vocab-sharded logits) runs inside libmi355x_vllm on its own stream.  Every
rank executes the same SchedulerOutput; rank 0 is the driver whose output vLLM sees.
"""

import logging
import os
from typing import Set

import torch

from .._vllm_compat import (WorkerBase, ensure_model_parallel_initialized,
                            init_distributed_environment, set_random_seed)

logger = logging.getLogger(__name__)


class MI355XWorker(WorkerBase):
    def __init__(self, vllm_config, local_rank: int, rank: int, distributed_init_method: str,
                 is_driver_worker: bool = False) -> None:
        super().__init__(vllm_config=vllm_config, local_rank=local_rank, rank=rank,
                         distributed_init_method=distributed_init_method, is_driver_worker=is_driver_worker)
        if getattr(self.model_config, "trust_remote_code", False):
            try:
                from vllm.utils import init_cached_hf_modules
                init_cached_hf_modules()
            except ImportError:
                pass
        self.device = self.device_config.device
        self.tp_size = self.parallel_config.tensor_parallel_size
        self.tp_rank = int(os.environ.get("RANK", rank)) % max(self.tp_size, 1) if self.tp_size > 1 else 0
        self.device_id = int(os.environ.get("LOCAL_RANK", local_rank)) if self.tp_size > 1 else max(local_rank, 0)
        self.model_runner = None

    def get_mi355x_model_runner(self, vllm_config, device, tp_unique_id=None):
        from .mi355x_model_runner import MI355XModelRunner
        return MI355XModelRunner(vllm_config=vllm_config, device=device, tp_rank=self.tp_rank,
                                 device_id=self.device_id, tp_unique_id=tp_unique_id)

    def init_device(self) -> None:
        self.init_distributed_environment()
        self._bound_host_threads()
        set_random_seed(self.model_config.seed)
        uid = self._exchange_tp_unique_id() if self.tp_size > 1 else None
        self.model_runner = self.get_mi355x_model_runner(self.vllm_config, self.device, uid)

    def _bound_host_threads(self) -> None:
        """Keep torch's CPU pool (the sampler's argmax / top-k over [B, vocab]) inside this
        process's CPU share.  torch sizes its pool from the machine's core count; inside a
        container with a CFS quota (16 of 256 cores on the benchmark boxes) the idle workers'
        spin-waits exhaust the quota and the kernel parks the WHOLE process for the rest of
        the 100 ms period -- measured as 40-80 ms added to every third TTFT."""
        import os
        import torch
        share = len(os.sched_getaffinity(0))
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                quota, period = f.read().split()[:2]
            if quota != "max":
                share = min(share, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        share = max(1, share // max(1, self.tp_size))
        want = min(share, 8)
        if torch.get_num_threads() > want:
            logger.info("bounding torch CPU threads %d -> %d (cpu share %d)", torch.get_num_threads(),
                        want, share)
            torch.set_num_threads(want)

    def _exchange_tp_unique_id(self) -> bytes:
        """Rank 0 asks RCCL for a unique id; torch.distributed carries the 128 bytes."""
        import torch.distributed as dist
        from .._native import check, load_library
        import ctypes
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=self.tp_rank, world_size=self.tp_size)
        buf = ctypes.create_string_buffer(128)
        if self.tp_rank == 0:
            check(load_library().mi_tp_unique_id(buf))
        obj = [buf.raw]
        dist.broadcast_object_list(obj, src=0)
        return obj[0]

    def determine_available_memory(self):
        """Free device memory after the weights are resident (the reference asks the Neuron
        runtime and falls back to 20 GiB, neuron_worker.py:51-63)."""
        try:
            return int(self.model_runner.model.model.kv_stats()["device_free_bytes"])
        except Exception as e:
            logger.debug("Failed to get memory stats: %s", e)
            return 1024 * 1024 * 1024 * 20

    def execute_model(self, scheduler_output):
        output = self.model_runner.execute_model(scheduler_output)
        return output if self.is_driver_worker else None

    def profile(self, is_start: bool = True):
        """Per-kernel-class HIP-event timing of the following steps (eager launches)."""
        self.model_runner.model.model.profile_enable(bool(is_start))
        return None if is_start else self.model_runner.model.model.profile_read()

    def initialize_cache(self, num_gpu_blocks: int, num_cpu_blocks: int) -> None:
        self.cache_config.num_gpu_blocks = num_gpu_blocks
        self.cache_config.num_cpu_blocks = num_cpu_blocks

    def load_model(self):
        self.model_runner.load_model()

    def compile_or_warm_up_model(self) -> None:
        return None

    def get_model(self) -> torch.nn.Module:
        raise NotImplementedError

    def get_kv_cache_spec(self) -> dict:
        return self.model_runner.get_kv_cache_spec()

    def initialize_from_config(self, kv_cache_config) -> None:
        self.model_runner.initialize_kv_cache(kv_cache_config)

    def check_health(self) -> None:
        return

    def init_distributed_environment(self):
        """vLLM wants a (dummy, 1-rank, gloo) distributed environment even though the engine
        sees a single worker (reference neuron_worker.py:106-121)."""
        init_distributed_environment(world_size=1, rank=self.rank, local_rank=self.local_rank,
                                     distributed_init_method=self.distributed_init_method, backend="gloo")
        ensure_model_parallel_initialized(1, 1)

    def add_adapter(self, lora_request) -> bool:
        return

    def add_lora(self, lora_request) -> bool:
        raise NotImplementedError("Multi-lora is not yet supported on the MI355X plugin")

    def remove_lora(self, lora_id: int) -> bool:
        raise NotImplementedError("Multi-lora is not yet supported on the MI355X plugin")

    def pin_lora(self, lora_id: int) -> bool:
        raise NotImplementedError("Multi-lora is not yet supported on the MI355X plugin")

    def list_loras(self) -> Set[int]:
        return set()

    def get_supported_tasks(self):
        return ["generate"]

    def take_draft_token_ids(self):
        return self.model_runner.take_draft_token_ids()
