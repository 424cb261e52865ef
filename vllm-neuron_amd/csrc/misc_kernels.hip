// Small memory-bound kernels around the hot path: embedding gather, row-wise RMSNorm for the
// context-encoding pass, synthetic weight generation, fp32 -> bf16 copies.
#include "misc_kernels.h"

namespace mi {

// resid[t, :] = fp32(E[ids[t], :])            (reference K1)
__global__ void embed_kernel(const int32_t* __restrict__ ids, const uint16_t* __restrict__ table, int H,
                             float* __restrict__ resid) {
  const int t = blockIdx.x;
  const uint16_t* row = table + (size_t)ids[t] * H;
  for (int c8 = threadIdx.x; c8 < H / 8; c8 += blockDim.x) {
    const uint4 v = *reinterpret_cast<const uint4*>(row + c8 * 8);
    float* o = resid + (size_t)t * H + c8 * 8;
    *reinterpret_cast<float4*>(o) = make_float4(bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y));
    *reinterpret_cast<float4*>(o + 4) = make_float4(bf16lo_to_f32(v.z), bf16hi_to_f32(v.z), bf16lo_to_f32(v.w), bf16hi_to_f32(v.w));
  }
}
int launch_embed(const int32_t* ids, const uint16_t* table, int T, int H, float* resid, hipStream_t s) {
  MI_CHECK(H % 8 == 0, "embed: H % 8 == 0 required");
  hipLaunchKernelGGL(embed_kernel, dim3(T), dim3(256), 0, s, ids, table, H, resid);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// h = resid_in (+ partial) -> resid_out (optional); y = bf16(rmsnorm(h) * gain)   (reference K2)
// FP8: instead of the bf16 row, emit what rowquant_fp8_kernel would make of it -- the row rounded
// to bf16, then e4m3 with the per-token scale amax / 448 -- as the K-step-major image
// [H / 128][T][128 B] the FP8 GEMM reads (same arithmetic, one launch and one bf16 round trip less).
template <bool FP8>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* __restrict__ resid_in, const float* __restrict__ partial,
                                                        float* __restrict__ resid_out, const float* __restrict__ gain,
                                                        int H, float eps, uint16_t* __restrict__ y,
                                                        uint8_t* __restrict__ x8, float* __restrict__ row_scale) {
  extern __shared__ __attribute__((aligned(16))) float hrow[];  // H floats + 8
  const int t = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)t * H;
  float ss = 0.f;
  for (int c4 = tid; c4 < H / 4; c4 += 256) {
    float4 a = *reinterpret_cast<const float4*>(resid_in + o + c4 * 4);
    if (partial) {
      const float4 p = *reinterpret_cast<const float4*>(partial + o + c4 * 4);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    if (resid_out) *reinterpret_cast<float4*>(resid_out + o + c4 * 4) = a;
    *reinterpret_cast<float4*>(hrow + c4 * 4) = a;
    ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  ss = wave_sum(ss);
  float* red = hrow + H;
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float rinv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  float amax = 0.f;
  for (int c4 = tid; c4 < H / 4; c4 += 256) {
    const float4 a = *reinterpret_cast<const float4*>(hrow + c4 * 4);
    const float4 g = *reinterpret_cast<const float4*>(gain + c4 * 4);
    const uint2 pk = make_uint2(pack_bf16x2(a.x * rinv * g.x, a.y * rinv * g.y), pack_bf16x2(a.z * rinv * g.z, a.w * rinv * g.w));
    if constexpr (FP8) {
      *reinterpret_cast<uint2*>(hrow + c4 * 4) = pk;   // own 16-byte slot: the bf16 values, for pass 3
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16lo_to_f32(pk.x)), fabsf(bf16hi_to_f32(pk.x))),
                               fmaxf(fabsf(bf16lo_to_f32(pk.y)), fabsf(bf16hi_to_f32(pk.y)))));
    } else {
      *reinterpret_cast<uint2*>(y + o + c4 * 4) = pk;
    }
  }
  if constexpr (FP8) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const float scale = amax > 0.f ? amax / 448.f : 1.f;
    if (tid == 0) row_scale[t] = scale;
    for (int c4 = tid; c4 < H / 4; c4 += 256) {
      const uint2 pk = *reinterpret_cast<const uint2*>(hrow + c4 * 4);
      const float v0 = fminf(fmaxf(bf16lo_to_f32(pk.x) / scale, -448.f), 448.f);
      const float v1 = fminf(fmaxf(bf16hi_to_f32(pk.x) / scale, -448.f), 448.f);
      const float v2 = fminf(fmaxf(bf16lo_to_f32(pk.y) / scale, -448.f), 448.f);
      const float v3 = fminf(fmaxf(bf16hi_to_f32(pk.y) / scale, -448.f), 448.f);
      const uint32_t q = f32_to_e4m3fn(v0) | (f32_to_e4m3fn(v1) << 8) | (f32_to_e4m3fn(v2) << 16) | (f32_to_e4m3fn(v3) << 24);
      const int k0 = c4 * 4;
      *reinterpret_cast<uint32_t*>(x8 + ((size_t)(k0 >> 7) * gridDim.x + t) * 128 + (k0 & 127)) = q;
    }
  }
}
int launch_norm_rows(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                     int H, float eps, uint16_t* y, hipStream_t s) {
  MI_CHECK(H % 4 == 0 && H <= 32768, "rmsnorm: H % 4 == 0 and H <= 32768 required");
  hipLaunchKernelGGL(norm_rows_kernel<false>, dim3(T), dim3(256), (H + 8) * sizeof(float), s, resid_in, partial, resid_out,
                     gain, H, eps, y, nullptr, nullptr);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
int launch_norm_rows_fp8(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                         int H, float eps, uint8_t* x8, float* row_scale, hipStream_t s) {
  MI_CHECK(H % 128 == 0 && H <= 32768, "rmsnorm + fp8: H % 128 == 0 and H <= 32768 required");
  hipLaunchKernelGGL(norm_rows_kernel<true>, dim3(T), dim3(256), (H + 8) * sizeof(float), s, resid_in, partial, resid_out,
                     gain, H, eps, nullptr, x8, row_scale);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = f32_to_bf16(in[i]);
}
int launch_f32_to_bf16(const float* in, uint16_t* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(2048), dim3(256), 0, s, in, out, n);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// counter-based N(0, std): value depends only on (seed, tensor id, logical index)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed, uint64_t tensor_id, float std) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix64(seed ^ splitmix64(tensor_id * 0x100000001B3ull + i));
    const float u1 = ((uint32_t)(r >> 40) + 1u) * (1.0f / 16777217.0f);  // (0, 1)
    const float u2 = (uint32_t)(r & 0xFFFFFFu) * (1.0f / 16777216.0f);
    out[i] = std * sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
  }
}
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t tensor_id, float std, hipStream_t s) {
  hipLaunchKernelGGL(randn_kernel, dim3(4096), dim3(256), 0, s, out, n, seed, tensor_id, std);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

__global__ void fill_f32_kernel(float* __restrict__ out, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
int launch_fill_f32(float* out, size_t n, float v, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(256), dim3(256), 0, s, out, n, v);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
