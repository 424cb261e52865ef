#pragma once
#include "mi_common.h"

namespace mi {
int launch_embed(const int32_t* ids, const uint16_t* table, int T, int H, float* resid, hipStream_t s);
// slabs: the K-split sum of the projection that produced this residual add, not summed yet (SlabSum): the row is
// resid_in + (sum of slabs) * scale + bias, written to resid_out and normalised -- same arithmetic, same order as
// splitk_reduce + residual epilogue followed by this kernel
int launch_norm_rows(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                     int H, float eps, uint16_t* y, hipStream_t s, const SlabSum* slabs = nullptr);
// the same row, emitted as the FP8 GEMM's input: bf16-rounded, then e4m3 with scale amax / 448,
// K-step-major image [H / 128][T][128 B] + row_scale[T]
int launch_norm_rows_fp8(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                         int H, float eps, uint8_t* x8, float* row_scale, hipStream_t s, const SlabSum* slabs = nullptr);
// dst[i, :] = src[rows[i], :]   (fp32 rows of H elements, H % 4 == 0)
// ---- fused speculation (model.hip, mi_forward_spec) ----
// After draft step `step` (0-based): the draft's sampled tokens become (a) candidate step + 1 of every
// sequence -- the target's input ids of row b * k + step + 1 and the record the acceptance compares
// against -- and (b) the draft's own next input: position and context + 1, slot from the resident
// block table.  A sequence whose candidate window is exhausted (step + 1 >= limit[b]) stops advancing
// and gets slot -1, so it writes no K/V beyond what it may own.
int launch_spec_advance(int B, int k, int step, const int32_t* draft_tokens, int32_t* draft_dec, int draft_rows,
                        const int32_t* draft_bt, int MB, int block_size, const int32_t* limit, int32_t* target_ids,
                        int32_t* cand, hipStream_t s);
// Greedy acceptance: n = number of leading candidates cand[b][1..] that equal the target's own choice one
// row earlier; out[b] = the target's tokens of rows 0..n (n + 1 of them, at most limit[b]), 0-padded
// to k like NxDI's accepted_tokens_with_padding; next_pos[b] = pos0[b] + n + 1.
int launch_spec_accept(int B, int k, const int32_t* target_tokens, const int32_t* cand, const int32_t* limit,
                       const int32_t* pos0, int32_t* out_tokens, int32_t* next_pos, hipStream_t s);
int launch_gather_rows(const float* src, const int32_t* rows, int n, int H, float* dst, hipStream_t s);
int launch_f32_to_bf16(const float* in, uint16_t* out, size_t n, hipStream_t s);
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t tensor_id, float std, hipStream_t s);
int launch_fill_f32(float* out, size_t n, float v, hipStream_t s);
// On-device sampling over fp32 logits rows (vocab-parallel segments [T][row_stride][V_l]); params
// [B,3] = (top_k, top_p, temperature) per row, null = greedy; tokens[B] int32 (device).
int launch_sample_rows(const float* logits, int T, int row_stride, int V_l, int B, const float* params,
                       unsigned long long seed, int row0, int32_t* tokens, hipStream_t s, void* scratch = nullptr);
// scratch: sample_scratch_bytes(rows) of device memory; with it and params == nullptr (every row greedy) the argmax is
// split over the chip (two small launches instead of one work-group per row)
size_t sample_scratch_bytes(int max_rows);
}  // namespace mi
