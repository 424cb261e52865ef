// Block-layout paged-KV attention (reference K5/K6/K7 of SURVEY.md §2.2: the NxDI
// context-encoding and token-generation attention reached through
// /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:339-348 with the
// block_table / slot_mapping / *_context_lens contract of
// /root/reference/vllm_neuron/worker/neuronx_distributed_model_runner.py:739-832).
//
// Pool layout (library-owned, one per layer and per K/V):
//     [num_blocks][nkv][block_size][hd] bf16
// i.e. one (block, kv-head) tile is a dense block_size*hd*2-byte range (8 KiB at 32 x 128),
// which a wave streams as 1 KiB coalesced reads.  Block 0 is vLLM's null block
// (reference platform.py:150-159): valid memory, never referenced by a live position.
#pragma once
#include "mi_common.h"

namespace mi {

constexpr int kAttnMaxSplits = 16;

int attn_decode_splits(int B, int nkv);
size_t attn_scratch_bytes(int B, int nh, int hd);

// q [B, nh, hd] bf16 -> out [B, nh*hd] bf16.  ctx_lens[b] = number of keys (incl. the new one).
int launch_attn_decode(const uint16_t* q, const uint16_t* kpool, const uint16_t* vpool, int block_size,
                       const int32_t* block_table, int MB, const int32_t* ctx_lens, int B, int nh,
                       int nkv, int hd, uint16_t* out, void* scratch, hipStream_t s, bool tickets_zeroed = false,
                       int rows_per_seq = 1, int num_blocks = 0);
// num_blocks > 0 (blocks of the pool): a wave's first K/V request does not wait for the context length (ids clamped into the pool)
// rows_per_seq R > 1: batch rows b R .. b R + R - 1 are one sequence at consecutive positions (block_table row b R
// serves them all); their heads share the MFMA columns and the sequence's K/V is read once
// tickets_zeroed: the first 4 KiB of `scratch` (the merge tickets) were zeroed once by the owner and
// only ever touched by these launches (which leave them at zero): enables the in-launch merge

// One sequence: q [T, nh, hd] at absolute positions q_pos0.., keys 0..q_pos0+T-1 from the pool.
// The sequence's block table is staged in LDS once: at most kPrefillMaxBlocks blocks (131072 tokens at block_size 32).
constexpr int kPrefillMaxBlocks = 4096;
int launch_attn_prefill(const uint16_t* q, int T, int q_pos0, const uint16_t* kpool,
                        const uint16_t* vpool, int block_size, const int32_t* block_table, int nh,
                        int nkv, int hd, uint16_t* out, hipStream_t s);

// k, v [T, nkv, hd] bf16 -> pool at slots [T] (-1 skips)
int launch_kv_write(const uint16_t* k, const uint16_t* v, const int64_t* slots, int T, int nkv, int hd,
                    uint16_t* kpool, uint16_t* vpool, int block_size, hipStream_t s);

}  // namespace mi
