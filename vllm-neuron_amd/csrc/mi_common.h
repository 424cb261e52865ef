// Shared device/host helpers for libmi355x_vllm (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/mi355x_vllm.h"

namespace mi {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one MFMA 16x16x32 operand
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int kWave = 64;

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string& msg);
#define MI_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      mi::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return MI_EHIP;                                                                 \
    }                                                                                 \
  } while (0)
#define MI_CHECK(cond, msg)                                                           \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      mi::set_error(std::string(msg) + " [" #cond "]");                               \
      return MI_EINVAL;                                                               \
    }                                                                                 \
  } while (0)

// ---- scalar conversions ---------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) {
  return __builtin_bit_cast(float, packed << 16);
}
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) {
  return __builtin_bit_cast(float, packed & 0xffff0000u);
}
// round-to-nearest-even; plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// ---- weight-fragment decode: 8 stored weights -> 8 bf16 (one MFMA operand) --------------
// fp8 e4m3fn and int8 values are exactly representable in bf16.
template <int WD>
__device__ __forceinline__ bf16x8_t decode8(uint32_t lo, uint32_t hi);

template <>
__device__ __forceinline__ bf16x8_t decode8<MI_W_F8E4M3>(uint32_t lo, uint32_t hi) {
  union { bf16x2_t p[4]; bf16x8_t v; } u;
  u.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
  u.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  u.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
  u.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  return u.v;
}

__device__ __forceinline__ uint32_t i8x2_to_bf16x2(uint32_t w, int byte0, int byte1) {
  float a = (float)(int)(int8_t)(w >> (8 * byte0));
  float b = (float)(int)(int8_t)(w >> (8 * byte1));
  return pack_bf16x2(a, b);
}
template <>
__device__ __forceinline__ bf16x8_t decode8<MI_W_INT8>(uint32_t lo, uint32_t hi) {
  union { uint32_t p[4]; bf16x8_t v; } u;
  u.p[0] = i8x2_to_bf16x2(lo, 0, 1);
  u.p[1] = i8x2_to_bf16x2(lo, 2, 3);
  u.p[2] = i8x2_to_bf16x2(hi, 0, 1);
  u.p[3] = i8x2_to_bf16x2(hi, 2, 3);
  return u.v;
}

// ---- cross-lane sums inside an aligned group of 8 or 16 lanes (DPP, no LDS) ------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  float t = __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
  return v + t;
}
// after the call every lane of the group holds the group's sum
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LANES == 8 || LANES == 16, "group of 8 or 16 lanes");
  if (LANES == 16) v = dpp_add<0x140>(v);  // row_mirror       i <-> 15-i
  v = dpp_add<0x141>(v);                   // row_half_mirror  i <-> 7-i (within 8)
  v = dpp_add<0xB1>(v);                    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                    // quad_perm [2,3,0,1]
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum<16>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// streamed-once 16-byte load (global_load_dwordx4 ... nt): keeps weight bytes out of the way
// of the L2 lines that ARE reused (activations, scales)
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// fp32 -> OCP e4m3fn, round-nearest-even, saturating at +-448: the hardware convert of gfx950 (v_cvt_pk_fp8_f32; OCP
// formats on this chip).  The clamp in front makes the result independent of the instruction's own out-of-range rule; a NaN
// input becomes +-448 like any other out-of-range value.  (Until round 3 this was ~20 integer / select instructions per value:
// the per-token quantization passes of the FP8-activation path were VALU-bound on it.)
__device__ __forceinline__ uint32_t f32_to_e4m3fn(float f) {
  const float c = __builtin_fminf(__builtin_fmaxf(f, -448.f), 448.f);
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.f, 0, false) & 0xffu;
}


// ---- weight stream primitives -----------------------------------------------------------
// MI_STREAM_ASM = 0 (default): plain non-temporal loads; hipcc owns the s_waitcnt placement.
//   It drains to vmcnt(0) once per pair of batches, so latency is hidden by wave-level
//   parallelism (16 waves per CU), not by depth per wave.
// MI_STREAM_ASM = 1: loads issued from inline asm and retired by hand-counted waits
//   (cdna_hip_programming.md 5.7).  Measured hazard on hipcc 7.2: whenever an in-flight buffer is
//   live across a control-flow merge the register allocator resolves the phi with v_mov copies
//   of registers whose loads have not landed.  csrc/audit_stream.py detects exactly that in the
//   emitted ISA; the build refuses such a library.  Kept for experiments only.
#ifndef MI_STREAM_ASM
#define MI_STREAM_ASM 0
#endif
#if MI_STREAM_ASM
__device__ __forceinline__ void stream_load16(u32x4_t& dst, const uint4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt ; stream_load" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void stream_wait(u32x4_t& r) {
  asm volatile("s_waitcnt vmcnt(%1) ; stream_release %0" : "+v"(r) : "n"(N) : "memory");
}
#else
__device__ __forceinline__ void stream_load16(u32x4_t& dst, const uint4* p) {
#ifdef MI_STREAM_PLAIN   // probe builds: default cache policy instead of the streaming hint
  dst = *reinterpret_cast<const u32x4_t*>(p);
#else
  dst = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#endif
}
template <int N>
__device__ __forceinline__ void stream_wait(u32x4_t&) {}
#endif

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- per-device launch state ------------------------------------------------------------
// One process may drive several GPUs (the in-process tensor-parallel group): kernel attributes
// and the CU count belong to the CURRENT device, never to the process.
// A K-split GEMM whose slabs have not been summed yet: y[m, n] = (sum_z slab[z][m][n]) * scale[n] (* row_scale[m]) + bias[n]
// (the arithmetic of splitk_reduce_kernel + epilogue, in that order).  Consumed by the row-norm kernel that reads the
// residual stream next (one launch and one round trip of the residual less), or flushed by launch_splitk_flush.
struct SlabSum {
  const float* slab = nullptr;      // [KS][T][N] raw fp32 accumulators
  int KS = 0, T = 0, N = 0;
  const float* scale = nullptr;     // [N]
  const float* bias = nullptr;      // [N] or null
  const float* row_scale = nullptr; // [T] or null (FP8 activations)
};
int device_num_cu(int* out);                                  // CUs of the current device
int ensure_dynamic_lds(const void* kernel, int bytes);        // hipFuncSetAttribute once per (device, kernel)

// weight tile geometry: a tile is 16 rows x TILE_K(wd) columns = 1 KiB, lane l owns
// row (l & 15), k-chunk (l >> 4) of 16 bytes.
__host__ __device__ __forceinline__ int tile_k(int wd) { return wd == MI_W_BF16 ? 32 : 64; }
__host__ __device__ __forceinline__ int elem_bytes(int wd) { return wd == MI_W_BF16 ? 2 : 1; }

}  // namespace mi
