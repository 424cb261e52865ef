"""TEST INFRASTRUCTURE -- CPU restatement of the reference's fused speculation step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

What it restates (parity status: **unpinned** beyond the integer contract -- the arithmetic of the
fused draft + target graph lives in NxDI, which is absent from /root/reference, and no reference
test holds a speculative golden output; what IS pinned is the output contract, by the reference's
own known-answer test of ``_remask_fused_spec_output``, tests/test_boundary_cpu.py):

* one step = the draft model proposes tokens greedily, one at a time, the target scores all of them
  in one pass, the longest prefix on which both agree is kept plus the target's own next token
  (NxDI fused speculation, reached through
  /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:349-355);
* output contract: ``accepted_tokens_with_padding`` [B, k], 0-padded, and ``next_pos_ids`` [B]
  (loader.py:308-333: the loader turns the padding into -1 from ``next_pos - position``);
* K/V slots of the speculated positions follow the block table (the reference appends consecutive
  slots, neuronx_distributed_model_runner.py:825-830, which is the same thing inside one block).

The property that makes the step checkable without NxDI: with greedy acceptance the tokens are exactly
the ones the target alone generates greedily.  ``fused_speculation_step`` is written with plain
oracle decode calls (one row at a time for the target as well), so that property holds bit for bit
for this restatement and tests assert it.
"""
from __future__ import annotations

import torch

from .paged_decoder import PagedDecoderOracle, greedy_sample


def _decode_inputs(tokens, positions, block_rows, block_size):
    B = len(tokens)
    slots = [[int(block_rows[b][positions[b] // block_size]) * block_size + positions[b] % block_size] for b in range(B)]
    return dict(
        input_ids=torch.tensor(tokens, dtype=torch.long).reshape(B, 1),
        position_ids=torch.tensor(positions, dtype=torch.long).reshape(B, 1),
        seq_ids=torch.arange(B, dtype=torch.long),
        block_table=torch.tensor([list(map(int, r)) for r in block_rows], dtype=torch.long),
        slot_mapping=torch.tensor(slots, dtype=torch.long),
        full_context_lens=torch.tensor([p + 1 for p in positions], dtype=torch.long).reshape(B, 1),
        computed_context_lens=torch.tensor(positions, dtype=torch.long).reshape(B, 1),
    )


def fused_speculation_step(target: PagedDecoderOracle, draft: PagedDecoderOracle, last_tokens, positions,
                           block_table, k: int, block_size: int, max_model_len: int):
    """-> (accepted [B, k] int64 0-padded, next_pos [B] int64).  ``last_tokens[b]`` sits at
    ``positions[b]``; both oracles hold the K/V of everything before it."""
    B = len(last_tokens)
    accepted = torch.zeros(B, k, dtype=torch.long)
    next_pos = torch.zeros(B, dtype=torch.long)
    for b in range(B):
        pos, row = int(positions[b]), block_table[b]
        lim = max(1, min(k, max_model_len - pos - 1))          # candidate rows: the sequence stays within max_model_len tokens
        cand = [int(last_tokens[b])]
        # the draft sees every candidate here; the HIP path runs k - 1 draft steps and feeds the last candidate
        # to the draft at the start of the NEXT step when (and only when) it was accepted -- the same K/V
        # wherever it is ever read
        for i in range(lim):
            logits = draft.forward(**_decode_inputs([cand[i]], [pos + i], [row], block_size))
            cand.append(int(greedy_sample(logits)[0]))
        target_tokens = []
        for i in range(lim):                                    # the target's pass over the candidates (one row at a time here)
            logits = target.forward(**_decode_inputs([cand[i]], [pos + i], [row], block_size))
            target_tokens.append(int(greedy_sample(logits)[0]))
        n = 0
        while n + 1 < lim and cand[n + 1] == target_tokens[n]:
            n += 1
        accepted[b, :n + 1] = torch.tensor(target_tokens[:n + 1])
        next_pos[b] = pos + n + 1
    return accepted, next_pos


def remask(accepted: torch.Tensor, next_pos: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """loader.py:308-333: 0-padding -> -1 beyond the number of tokens generated this step."""
    counts = (next_pos - positions).clamp(0, accepted.shape[1])
    out = accepted.clone()
    for b in range(accepted.shape[0]):
        out[b, int(counts[b]):] = -1
    return out
