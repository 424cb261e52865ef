# SPDX-License-Identifier: Apache-2.0
"""Continuous-batching scheduler with the batch shapes the MI355X kernels are built for:
a step is EITHER one context-encoding request (B = 1) OR token generation for every running
request (B <= max_num_seqs).  Same policy and the same min_tokens-aware stop rule as the
reference (/root/reference/vllm_neuron/core/scheduler.py:19-166)."""

import logging
from collections import deque

from .._vllm_compat import Request, RequestStatus, Scheduler

logger = logging.getLogger(__name__)

MAX_PROMPT_BATCH_SIZE = 1


class MI355XScheduler(Scheduler):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        # requests parked here are invisible to the base scheduler for one step
        self.holdback_queue: deque[Request] = deque()

    def _update_request_with_output(self, request: Request, new_token_ids: list[int]):
        """Append the sampled tokens; stop checks honour min_tokens (the upstream checker
        ignores it, reference scheduler.py:32-56)."""
        stopped = False
        for num_new, token_id in enumerate(new_token_ids, 1):
            request.append_output_token_ids(token_id)
            stopped = check_stop_with_min_tokens(request, self.max_model_len)
            if stopped:
                del new_token_ids[num_new:]
                break
        return new_token_ids, stopped


class ContinuousBatchingMI355XScheduler(MI355XScheduler):
    running: list[Request]

    def schedule(self):
        # park everything that is waiting, then re-admit what this step may take
        while self.waiting:
            self.holdback_queue.append(self.waiting.popleft())
        while self.holdback_queue and self.can_schedule(self.holdback_queue[0]):
            self.waiting.append(self.holdback_queue.popleft())

        if len(self.waiting) > 0:
            # context-encoding step: hide the running decodes from the base scheduler
            running_holdback, self.running = self.running, []
        else:
            running_holdback = []

        outputs = super(MI355XScheduler, self).schedule()

        self.running = self.running + running_holdback
        while self.holdback_queue:
            self.waiting.append(self.holdback_queue.popleft())
        return outputs

    def can_schedule(self, request) -> bool:
        in_flight = len(self.running) + len(self.waiting)
        if in_flight == 0:
            return True
        return in_flight < self.max_num_running_reqs and len(self.waiting) < MAX_PROMPT_BATCH_SIZE


def check_stop_with_min_tokens(request: Request, max_model_len: int, pooler_output=None) -> bool:
    if request.num_tokens >= max_model_len or request.num_output_tokens >= request.max_tokens:
        request.status = RequestStatus.FINISHED_LENGTH_CAPPED
        return True
    if request.pooling_params:
        if pooler_output is not None:
            request.status = RequestStatus.FINISHED_STOPPED
            return True
        return False
    sampling_params = request.sampling_params
    assert sampling_params is not None
    if sampling_params.min_tokens > 0 and request.num_output_tokens < sampling_params.min_tokens:
        return False
    last_token_id = request.output_token_ids[-1]
    if not sampling_params.ignore_eos and last_token_id == request.eos_token_id:
        request.status = RequestStatus.FINISHED_STOPPED
        return True
    if last_token_id in (sampling_params.stop_token_ids or ()):
        request.status = RequestStatus.FINISHED_STOPPED
        request.stop_reason = last_token_id
        return True
    return False
