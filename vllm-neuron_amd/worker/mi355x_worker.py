# SPDX-License-Identifier: Apache-2.0
"""vLLM worker for MI355X: lifecycle shim around the model runner, same method set as the
reference worker (/root/reference/vllm_neuron/worker/neuron_worker.py:21-146).

Tensor parallelism: exactly the reference's process model.  The platform forces vLLM's "uni"
executor (platform.py), so there is ONE worker whatever `tensor_parallel_size` says, and the model
object behind it drives every device (reference: NxDI inside the single NeuronWorker,
neuron_worker.py:106-121, loader.py:752-753).  Here the library context is created with
`tp_rank = MI_TP_ALL_RANKS`: it owns one rank shard per GPU, each on a host thread of its own inside
libmi355x_vllm, exchanging through the library's all-reduce over peer-mapped memory (xGMI).  The
worker hands the same SchedulerOutput to that one context; nothing here knows about ranks.

Which GPUs: `tensor_parallel_size` devices starting at `local_rank` (normally 0 .. T-1), or the
comma-separated list in MI355X_TP_DEVICES.  When fewer GPUs are visible than ranks and
MI355X_TP_LOOPBACK=1, every shard is placed on the first device (tests on a one-GPU box).
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
        self.device_id = max(local_rank, 0)
        self.tp_device_ids = self._pick_tp_devices() if self.tp_size > 1 else [self.device_id]
        self.model_runner = None

    def _pick_tp_devices(self) -> list:
        """One GPU per rank shard (see the module docstring)."""
        env = os.environ.get("MI355X_TP_DEVICES")
        if env:
            ids = [int(x) for x in env.split(",") if x.strip() != ""]
            if len(ids) != self.tp_size:
                raise ValueError(f"MI355X_TP_DEVICES lists {len(ids)} devices, tensor_parallel_size is {self.tp_size}")
            return ids
        visible = torch.cuda.device_count()
        if visible >= self.device_id + self.tp_size:
            return list(range(self.device_id, self.device_id + self.tp_size))
        if os.environ.get("MI355X_TP_LOOPBACK") == "1":
            return [self.device_id] * self.tp_size
        raise RuntimeError(
            f"tensor_parallel_size={self.tp_size} needs {self.tp_size} GPUs from device {self.device_id}, "
            f"{visible} visible (set MI355X_TP_DEVICES, or MI355X_TP_LOOPBACK=1 to place every shard on one GPU)")

    def get_mi355x_model_runner(self, vllm_config, device):
        from .mi355x_model_runner import MI355XModelRunner
        return MI355XModelRunner(vllm_config=vllm_config, device=device, device_id=self.device_id,
                                 tp_device_ids=self.tp_device_ids)

    def init_device(self) -> None:
        self.init_distributed_environment()
        self._bound_host_threads()
        set_random_seed(self.model_config.seed)
        self.model_runner = self.get_mi355x_model_runner(self.vllm_config, self.device)

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
        want = min(share, 8)
        if torch.get_num_threads() > want:
            logger.info("bounding torch CPU threads %d -> %d (cpu share %d)", torch.get_num_threads(),
                        want, share)
            torch.set_num_threads(want)

    def determine_available_memory(self):
        """Memory vLLM may turn into KV blocks (the reference asks the Neuron runtime and falls
        back to 20 GiB, neuron_worker.py:51-63).  vLLM divides this by the page size of the ONE
        layer `get_kv_cache_spec` advertises, while the pool holds every layer of the model and the
        library still has its activations / exchange buffers to allocate at `mi_finalize`: so
        report (free - that workspace - a margin) scaled by one-layer-page / all-layers-block, i.e.
        the block count vLLM derives is one the pool can really hold."""
        try:
            native = self.model_runner.model.model
            st = native.kv_stats()
            usable = int(st["device_free_bytes"] * 0.95) - int(st["workspace_bytes"])
            per_block_all_layers = native.kv_bytes_per_block()
            draft = getattr(self.model_runner.model, "draft", None)
            if draft is not None:   # fused speculation: a second pool under the same block ids
                per_block_all_layers += draft.kv_bytes_per_block()
                usable -= int(draft.kv_stats()["workspace_bytes"])
            spec = self.model_runner.get_kv_cache_spec()["layer"]
            # the spec counts all kv heads of one layer; a shard holds its share of every layer
            blocks = max(usable, 0) // max(per_block_all_layers, 1)
            return int(blocks * spec.page_size_bytes)
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
