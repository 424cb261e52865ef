"""Test-side helpers: reference-contract input builders and torch references."""

from __future__ import annotations

import math

import torch

PAD_SLOT, PAD_BLOCK = -1, 0


def prefill_inputs(prompt, blocks, block_size, max_model_len, num_computed=0):
    """The tensors the reference runner builds for one new request with prefix caching
    (/root/reference/vllm_neuron/worker/neuronx_distributed_model_runner.py:739-763, 853-885)."""
    L = len(prompt)
    mb = max_model_len // block_size
    bt = [PAD_BLOCK] * mb
    bt[:len(blocks)] = blocks
    slots = [(blocks[i // block_size] * block_size + i % block_size) if i < L else PAD_SLOT
             for i in range(max_model_len)][num_computed:]
    slots = slots + [PAD_SLOT] * (max_model_len - len(slots))
    return dict(
        input_ids=torch.tensor([prompt], dtype=torch.long),
        position_ids=torch.arange(L, dtype=torch.long)[None],
        seq_ids=torch.tensor([0], dtype=torch.long),
        block_table=torch.tensor([bt], dtype=torch.long),
        slot_mapping=torch.tensor([slots], dtype=torch.long),
        full_context_lens=torch.tensor([[L]], dtype=torch.long),
        computed_context_lens=torch.tensor([[num_computed]], dtype=torch.long),
    )


def decode_inputs(last_tokens, positions, block_lists, block_size, max_model_len, pad_block=PAD_BLOCK):
    """Token-generation inputs (runner.py:765-832, 887-917)."""
    mb = max_model_len // block_size
    bts, slots = [], []
    for pos, blocks in zip(positions, block_lists):
        bt = [pad_block] * mb
        bt[:len(blocks)] = blocks
        bts.append(bt)
        slots.append([blocks[pos // block_size] * block_size + pos % block_size])
    B = len(last_tokens)
    return dict(
        input_ids=torch.tensor(last_tokens, dtype=torch.long).reshape(B, 1),
        position_ids=torch.tensor(positions, dtype=torch.long).reshape(B, 1),
        seq_ids=torch.arange(B, dtype=torch.long),
        block_table=torch.tensor(bts, dtype=torch.long),
        slot_mapping=torch.tensor(slots, dtype=torch.long),
        full_context_lens=torch.tensor([p + 1 for p in positions], dtype=torch.long).reshape(B, 1),
        computed_context_lens=torch.tensor(positions, dtype=torch.long).reshape(B, 1),
    )


def ref_attention(q, K, V, q_pos):
    """q [T, nh, hd], K/V [S, nkv, hd] (fp32), q_pos [T] absolute positions -> [T, nh*hd]."""
    T, nh, hd = q.shape
    group = nh // K.shape[1]
    Kx, Vx = K.repeat_interleave(group, 1), V.repeat_interleave(group, 1)
    s = torch.einsum("thd,jhd->htj", q, Kx) / math.sqrt(hd)
    mask = torch.arange(K.shape[0])[None, :] > q_pos[:, None]
    p = torch.softmax(s.masked_fill(mask[None], float("-inf")), -1)
    return torch.einsum("htj,jhd->thd", p, Vx).reshape(T, nh * hd)


def pool_to_native(k_tokens, v_tokens, num_blocks, block_size):
    """Logical [NB*bs, nkv, hd] token rows -> library pool [2][NB][nkv][bs][hd]."""
    nkv, hd = k_tokens.shape[1:]
    def one(t):
        return t.reshape(num_blocks, block_size, nkv, hd).permute(0, 2, 1, 3).contiguous()
    return torch.stack([one(k_tokens), one(v_tokens)])
