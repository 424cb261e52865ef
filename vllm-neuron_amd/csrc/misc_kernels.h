#pragma once
#include "mi_common.h"

namespace mi {
int launch_embed(const int32_t* ids, const uint16_t* table, int T, int H, float* resid, hipStream_t s);
int launch_norm_rows(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                     int H, float eps, uint16_t* y, hipStream_t s);
// the same row, emitted as the FP8 GEMM's input: bf16-rounded, then e4m3 with scale amax / 448,
// K-step-major image [H / 128][T][128 B] + row_scale[T]
int launch_norm_rows_fp8(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                         int H, float eps, uint8_t* x8, float* row_scale, hipStream_t s);
// dst[i, :] = src[rows[i], :]   (fp32 rows of H elements, H % 4 == 0)
int launch_gather_rows(const float* src, const int32_t* rows, int n, int H, float* dst, hipStream_t s);
int launch_f32_to_bf16(const float* in, uint16_t* out, size_t n, hipStream_t s);
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t tensor_id, float std, hipStream_t s);
int launch_fill_f32(float* out, size_t n, float v, hipStream_t s);
// On-device sampling over fp32 logits rows (vocab-parallel segments [T][row_stride][V_l]); params
// [B,3] = (top_k, top_p, temperature) per row, null = greedy; tokens[B] int32 (device).
int launch_sample_rows(const float* logits, int T, int row_stride, int V_l, int B, const float* params,
                       unsigned long long seed, int row0, int32_t* tokens, hipStream_t s);
}  // namespace mi
