// Small memory-bound kernels around the hot path: embedding gather, row-wise RMSNorm for the
// context-encoding pass, synthetic weight generation, fp32 -> bf16 copies.
#include "misc_kernels.h"

#include <mutex>
#include <set>
#include <utility>

namespace mi {

// ---- per-device launch state (mi_common.h) ------------------------------------------------
int device_num_cu(int* out) {
  static int cu[64] = {0};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  MI_CHECK(dev >= 0 && dev < 64, "device id out of range");
  int v = __atomic_load_n(&cu[dev], __ATOMIC_RELAXED);
  if (v == 0) {
    MI_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    __atomic_store_n(&cu[dev], v, __ATOMIC_RELAXED);
  }
  *out = v;
  return MI_OK;
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, kernel})) return MI_OK;
  MI_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kernel});
  return MI_OK;
}

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

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows, int H,
                                   float* __restrict__ dst) {
  const float4* from = reinterpret_cast<const float4*>(src + (size_t)rows[blockIdx.x] * H);
  float4* to = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) to[i] = from[i];
}
int launch_gather_rows(const float* src, const int32_t* rows, int n, int H, float* dst, hipStream_t s) {
  MI_CHECK(H % 4 == 0 && n >= 1, "gather_rows: H % 4 == 0 required");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, src, rows, H, dst);
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
                                                        uint8_t* __restrict__ x8, float* __restrict__ row_scale, SlabSum sl) {
  extern __shared__ __attribute__((aligned(16))) float hrow[];  // H floats + 8
  const int t = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)t * H;
  float ss = 0.f;
  const float sl_rs = (sl.slab && sl.row_scale) ? sl.row_scale[t] : 1.f;   // read before this launch may overwrite it (FP8: same array)
  for (int c4 = tid; c4 < H / 4; c4 += 256) {
    float4 a = *reinterpret_cast<const float4*>(resid_in + o + c4 * 4);
    if (partial) {
      const float4 p = *reinterpret_cast<const float4*>(partial + o + c4 * 4);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    if (sl.slab) {   // the residual add of a K-split projection: slabs in slice order, scale, bias, + residual
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < sl.KS; ++z) {
        const float4 q = *reinterpret_cast<const float4*>(sl.slab + ((size_t)z * sl.T + t) * sl.N + c4 * 4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      float4 sc = *reinterpret_cast<const float4*>(sl.scale + c4 * 4);
      if (sl.row_scale) { sc.x *= sl_rs; sc.y *= sl_rs; sc.z *= sl_rs; sc.w *= sl_rs; }
      v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
      if (sl.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(sl.bias + c4 * 4);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
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
                     int H, float eps, uint16_t* y, hipStream_t s, const SlabSum* slabs) {
  MI_CHECK(H % 4 == 0 && H <= 32768, "rmsnorm: H % 4 == 0 and H <= 32768 required");
  MI_CHECK(!slabs || (slabs->T == T && slabs->N == H && resid_out), "rmsnorm: pending K-split sum of another shape");
  hipLaunchKernelGGL(norm_rows_kernel<false>, dim3(T), dim3(256), (H + 8) * sizeof(float), s, resid_in, partial, resid_out,
                     gain, H, eps, y, nullptr, nullptr, slabs ? *slabs : SlabSum());
  MI_HIP(hipGetLastError());
  return MI_OK;
}
int launch_norm_rows_fp8(const float* resid_in, const float* partial, float* resid_out, const float* gain, int T,
                         int H, float eps, uint8_t* x8, float* row_scale, hipStream_t s, const SlabSum* slabs) {
  MI_CHECK(H % 128 == 0 && H <= 32768, "rmsnorm + fp8: H % 128 == 0 and H <= 32768 required");
  MI_CHECK(!slabs || (slabs->T == T && slabs->N == H && resid_out), "rmsnorm: pending K-split sum of another shape");
  hipLaunchKernelGGL(norm_rows_kernel<true>, dim3(T), dim3(256), (H + 8) * sizeof(float), s, resid_in, partial, resid_out,
                     gain, H, eps, nullptr, x8, row_scale, slabs ? *slabs : SlabSum());
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

// =====================================================================================
// On-device sampling (reference K10: NxDI on-device sampling behind on_device_sampling_config,
// /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:352-356, 367-375; the
// per-request (top_k, top_p, temperature) rows of runner.py:1106-1140)
// =====================================================================================
// One work-group (1024 threads) per sequence over its fp32 logits row.  The row may be split
// vocab-parallel over T rank segments (TP): element v lives at seg[v / V_l][row][v % V_l].
//   top_k == 1 : argmax, lowest index on ties (== torch.argmax).
//   otherwise  : keep the top_k (<= kSampleMaxTopK) logits -- 4-pass radix select on the order-
//                preserving integer image of the floats, candidates sorted (value desc, index asc)
//                by a bitonic network in LDS -- p_i = exp((l_i - l_max) / temperature); nucleus:
//                keep i while the mass BEFORE i is < top_p * total (the first is always kept);
//                draw u from splitmix64(seed, row) and take the first i whose cumulative mass
//                exceeds u * kept.  Same arithmetic order as oracle/sampling.py (fp32, serial).
constexpr int kSampleThreads = 1024;
constexpr int kSampleMaxTopK = 256;
constexpr int kSampleCand = 1024;   // candidates kept for the sort (top_k plus ties at the threshold)
constexpr int kSampleList = 4096;   // logits sharing the threshold's top key byte, set aside after the first pass

__device__ __forceinline__ uint32_t f32_order_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kSampleThreads) void sample_rows_kernel(
    const float* __restrict__ logits, int T, int row_stride, int V_l, const float* __restrict__ params /*[B,3] or null*/,
    unsigned long long seed, int row0, int32_t* __restrict__ tokens, const int32_t* __restrict__ idx_map) {
  // idx_map != null: the row is not the vocabulary but its pre-selected candidates (sample_slice_kernel: T == 1, V_l values,
  // element j being vocabulary word idx_map[b * V_l + j], in ascending word order wherever values are equal)
  __shared__ unsigned int hist[256];
  __shared__ float cand_v[kSampleCand];
  __shared__ int cand_i[kSampleCand];
  __shared__ float red_v[kSampleThreads / 64];
  __shared__ int red_i[kSampleThreads / 64];
  __shared__ unsigned int s_prefix, s_remaining, s_count, s_neq;
  __shared__ unsigned int wave_tot[4], wave_above[4], wave_cnt2[4];
  __shared__ int wave_hi[4];
  // After a pass only the logits inside the selected key range can still be the threshold (the ones above it are
  // candidates for certain).  As soon as that range holds few enough -- after the second pass at the latest for any
  // realistic row: a top byte covers two binades, two bytes 1/64 of one -- the next pass copies them aside and the
  // remaining passes walk that list instead of the vocabulary.
  __shared__ float list_v[kSampleList];
  __shared__ int list_i[kSampleList];
  __shared__ unsigned int s_list_n;
  __shared__ int s_compact;
  __shared__ unsigned int s_ties;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = T * V_l;
  auto word = [&](int v) -> int { return idx_map ? idx_map[(size_t)b * V_l + v] : v; };
  auto at = [&](int v) -> float {
    if (T == 1) return logits[(size_t)b * V_l + v];          // one part: no division per element
    return logits[((size_t)(v / V_l) * row_stride + b) * V_l + (v % V_l)];
  };
  // one pass over the row: 8 independent loads in flight per thread (a single dependent load per iteration made
  // every pass latency-bound: ~45 us per pass of a 128k vocabulary), elements visited in ascending index per thread
  auto for_each = [&](auto&& fn) {
    for (int part = 0; part < T; ++part) {   // vocabulary-parallel logits: one slice per rank, walked in rank order (no division per element)
      const float* row = logits + ((size_t)part * row_stride + b) * V_l;
      const int vbase = part * V_l;
      for (int j0 = tid; j0 < V_l; j0 += 8 * kSampleThreads) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * kSampleThreads;
          x[u] = j < V_l ? row[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * kSampleThreads;
          if (j < V_l) fn(vbase + j, x[u]);
        }
      }
    }
  };
  int top_k = 1;
  float top_p = 1.f, temperature = 1.f;
  if (params) {
    top_k = (int)params[b * 3];
    top_p = params[b * 3 + 1];
    temperature = params[b * 3 + 2];
  }
  top_k = min(max(top_k, 1), min(kSampleMaxTopK, V));

  if (top_k == 1) {   // ---- greedy ----------------------------------------------------------
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for_each([&](int v, float x) {
      if (x > bv) { bv = x; bi = v; }      // ascending v per thread: strict > keeps the lowest index
    });
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float ov = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kSampleThreads / 64; ++w)
        if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
      tokens[b] = bi == 0x7fffffff ? 0 : word(bi);   // all-NaN row: token 0
    }
    return;
  }

  // ---- radix select: the key of the top_k-th largest logit --------------------------------
  if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)top_k; s_count = 0; s_ties = 0; s_list_n = 0; s_compact = 0; }
  for (int i = tid; i < kSampleCand; i += kSampleThreads) { cand_v[i] = -INFINITY; cand_i[i] = 0x7fffffff; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned int prefix = s_prefix;
    const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    // a lane's consecutive elements mostly fall into the same bin on the first passes (sign / exponent bits):
    // runs are counted in registers and added once, which takes the 64-way contention off the LDS atomics
    unsigned int run_bin = 0xffffffffu, run_cnt = 0;
    auto count = [&](uint32_t key) {
      if ((key & himask) == prefix) {
        const unsigned int bin = (key >> shift) & 255u;
        if (bin == run_bin) {
          ++run_cnt;
        } else {
          if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
          run_bin = bin;
          run_cnt = 1;
        }
      }
    };
    // s_compact: 0 = walking the vocabulary, 1 = THIS pass walks it for the last time and sets the survivors aside
    // (the bin selected so far holds few enough), 2 = walking the list
    const int compact = s_compact;
    if (compact == 2) {
      const unsigned int n = s_list_n;
      for (unsigned int i = tid; i < n; i += kSampleThreads) count(f32_order_key(list_v[i]));
    } else if (compact == 1) {   // certain candidates (above the selected range) out, the selected range's logits aside
      for_each([&](int v, float x) {
        const uint32_t key = f32_order_key(x);
        const uint32_t hi = key & himask;
        if (hi > prefix) {
          const unsigned int slot = atomicAdd(&s_count, 1u);   // fewer than top_k of them by construction
          cand_v[slot] = x;
          cand_i[slot] = word(v);
        } else if (hi == prefix) {
          const unsigned int slot = atomicAdd(&s_list_n, 1u);
          list_v[slot] = x;
          list_i[slot] = word(v);
          count(key);
        }
      });
    } else {
      for_each([&](int, float x) { count(f32_order_key(x)); });
    }
    if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
    __syncthreads();
    // the bin the remaining rank falls into: the highest bin whose suffix count (this bin and all above) reaches
    // it.  Four waves hold the 256 bins: suffix sums inside a wave by shuffles, the waves' totals through LDS.
    {
      const int lane = tid & 63, wv = tid >> 6;
      unsigned int cnt = 0, suf = 0;
      if (tid < 256) {
        cnt = hist[tid];
        suf = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned int up = __shfl_down(suf, off);
          if (lane + off < 64) suf += up;
        }
        if (lane == 0) wave_tot[wv] = suf;
      }
      __syncthreads();
      if (tid < 256) {
        for (int w = wv + 1; w < 4; ++w) suf += wave_tot[w];
        const unsigned int rem = s_remaining;
        const unsigned long long ok = __ballot(suf >= rem);
        if (lane == 0) wave_hi[wv] = ok ? 63 - __builtin_clzll(ok) : -1;
        if (ok && lane == 63 - __builtin_clzll(ok)) { wave_above[wv] = suf - cnt; wave_cnt2[wv] = cnt; }
      }
      __syncthreads();
      if (tid == 0) {
        int bin = 0;
        unsigned int above = 0, here = hist[0];
        bool found = false;
        for (int w = 3; w >= 0 && !found; --w)
          if (wave_hi[w] >= 0) { bin = w * 64 + wave_hi[w]; above = wave_above[w]; here = wave_cnt2[w]; found = true; }
        if (!found) {   // (cannot happen while the counts are consistent: the serial walk ended at bin 0 with what was left)
          above = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3] - hist[0];
        }
        s_prefix = prefix | ((unsigned)bin << shift);
        s_remaining = s_remaining - above;
        if (s_compact == 1) s_compact = 2;                                        // this pass has built the list
        else if (s_compact == 0 && pass < 3 && here <= (unsigned)kSampleList) s_compact = 1;   // the next pass builds it
        if (pass == 3) s_neq = here;   // all 32 bits fixed: the logits EQUAL to the threshold
      }
      __syncthreads();
    }
  }
  const uint32_t kth = s_prefix;          // key of the top_k-th largest element
  // ---- candidates: exactly top_k of them ---------------------------------------------------
  // 1. every logit strictly above the threshold key (fewer than top_k by construction: any order,
  //    the sort below fixes it);  2. of the logits AT the threshold -- there may be many exact ties --
  //    the `s_remaining` with the LOWEST vocabulary indices, found by walking the vocabulary in index
  //    order.  (Collecting ">= threshold" in arrival order, as a first version did, could fill the
  //    candidate buffer with ties and drop larger logits, and depended on thread timing.)
  __shared__ unsigned int wave_cnt[kSampleThreads / 64];
  const unsigned int need = s_remaining;                   // ties to take (>= 1: the threshold element itself)
  const unsigned int n_gt = (unsigned)top_k - need;        // logits strictly above the threshold
  const bool all_ties = s_neq == need;                     // every logit at the threshold is taken: no order to respect
  auto collect = [&](int v, float x) {
    const uint32_t key = f32_order_key(x);
    if (key > kth) {
      const unsigned int slot = atomicAdd(&s_count, 1u);
      if (slot < n_gt) { cand_v[slot] = x; cand_i[slot] = v; }
    } else if (all_ties && key == kth) {
      const unsigned int slot = n_gt + atomicAdd(&s_ties, 1u);
      if (slot < (unsigned)top_k) { cand_v[slot] = x; cand_i[slot] = v; }
    }
  };
  if (s_compact == 2) {   // the logits above the selected range are in already (list-building pass); the rest come from the list
    const unsigned int n = s_list_n;
    for (unsigned int i = tid; i < n; i += kSampleThreads) collect(list_i[i], list_v[i]);
  } else {
    for_each([&](int v, float x) { collect(word(v), x); });
  }
  __syncthreads();
  for (int base = 0; base < V && !all_ties; base += kSampleThreads) {   // more ties than places: the lowest indices, in order
    if (s_ties >= need) break;                              // uniform: written behind the barrier below
    const int v = base + tid;
    const float x = v < V ? at(v) : 0.f;
    const bool tie = v < V && f32_order_key(x) == kth;
    const unsigned long long bal = __ballot(tie);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned int before = 0, total = 0;
    for (int w = 0; w < kSampleThreads / 64; ++w) {
      if (w < wv) before += wave_cnt[w];
      total += wave_cnt[w];
    }
    const unsigned int t = s_ties + before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (tie && t < need) { cand_v[n_gt + t] = x; cand_i[n_gt + t] = word(v); }
    __syncthreads();
    if (tid == 0) s_ties += total;
    __syncthreads();
  }
  // bitonic sort of the first P slots (P = the power of two that holds top_k; the slots behind top_k are -inf
  // fillers): value descending, index ascending (deterministic)
  int P = 2;
  while (P < top_k) P <<= 1;
  __syncthreads();
  if (tid >= kSampleMaxTopK) return;   // P <= 256 slots are sorted by the first four waves alone: their barriers are the cheap ones
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = tid, j = i ^ stride;
      if (j > i && j < P) {
        const bool desc = (i & size) == 0;
        const float vi = cand_v[i], vj = cand_v[j];
        const int ii = cand_i[i], ij = cand_i[j];
        const bool i_first = vi > vj || (vi == vj && ii < ij);   // i belongs before j in the final order
        if (desc ? !i_first : i_first) {
          cand_v[i] = vj; cand_v[j] = vi;
          cand_i[i] = ij; cand_i[j] = ii;
        }
      }
      __syncthreads();
    }
  }
  // the exponentials in parallel (one per sorted candidate), the sums serially in a fixed order
  float* cand_e = reinterpret_cast<float*>(hist);   // the histogram is dead: room for kSampleMaxTopK floats
  {
    const float lmax = cand_v[0];
    const float inv_t = 1.f / temperature;
    if (tid < top_k) cand_e[tid] = expf((cand_v[tid] - lmax) * inv_t);
  }
  __syncthreads();
  // The sums are serial fp32 additions in candidate order (oracle/sampling.py); one lane makes the chain of running sums
  // once, the three searches of the rule over them are counts (the sums never decrease: the masses are >= 0):
  //   kept candidates: i = 0, and every i whose mass BEFORE it is < top_p * total;   pick: the first i with cum_i > u * kept
  float* cum = cand_v;                      // the sorted values are dead once the exponentials exist
  __shared__ int s_nkeep, s_pick;
  if (tid == 0) {
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < top_k; ++i) { acc += cand_e[i]; cum[i] = acc; }
    s_nkeep = 0; s_pick = 0;
  }
  __syncthreads();
  const float total = cum[top_k - 1];
  const float limit = top_p * total;
  {
    const bool keep = tid < top_k && (tid == 0 || cum[tid - 1] < limit);
    const unsigned long long m = __ballot(keep);
    if ((tid & 63) == 0 && m) atomicAdd(&s_nkeep, (int)__popcll(m));
  }
  __syncthreads();
  const int nkeep = s_nkeep;               // a prefix of the candidates (the sums are monotone)
  const float kept = cum[nkeep - 1];
  const unsigned long long r = splitmix64(seed ^ splitmix64(0x5EEDull + (unsigned long long)(row0 + b)));
  const float u = (float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f);   // [0, 1)
  const float target = u * kept;
  {
    const bool before = tid < nkeep && !(cum[tid] > target);   // candidates in front of the pick
    const unsigned long long m = __ballot(before);
    if ((tid & 63) == 0 && m) atomicAdd(&s_pick, (int)__popcll(m));
  }
  __syncthreads();
  if (tid == 0) tokens[b] = cand_i[min(s_pick, nkeep - 1)];
}

// ---- top-k / top-p rows: the vocabulary pre-selected by many work-groups ----------------------------
// One work-group per row walks a 128k vocabulary twice and takes 87 us for four rows (top_k 50; 112 at top_k 256).  Here
// kSampleSplits work-groups per row each take a slice of the row and write the slice's top_k -- every logit above the slice's
// top_k-th largest key and, of the ones equal to it, the lowest-indexed -- in ascending word order to the row's candidate
// array (kSampleMaxTopK slots per slice, the unused ones filled with a negative NaN: its key is below -inf's, so it is never
// selected).  The row's top_k are among the slices' top_k, and an equal-valued candidate the one-work-group rule would take is
// among its slice's: whatever ranks before it in the slice ranks before it in the row.  sample_rows_kernel then runs on the
// candidates (idx_map) and picks the same word.  (First version: the slice in LDS, a bitwise bisection with one barrier per
// round over four waves: 20.8 us; before that a 4-pass radix select in LDS whose 256-bin histogram collided on a handful of
// exponent bins: 17-23 us.  This one with the full 32-round search, measured by early exits: loads 4.8 us, first stage 6.4 us,
// second stage 9.0 us = 20.1 us -- a round is 16 compare -> ballot -> scalar popcount chains, each paying the VALU -> SALU
// hand-off; with the search stopping once the candidates fit the slots: 11.8 us at top_k 50.)
constexpr int kSampleSplits = 32;
constexpr int kSliceMax = 8192;          // words of a slice (four waves x 32 registers x 64 lanes)
constexpr int kSliceThreads = 256;

// Wave-level selection, no LDS and no barrier: the wave holds up to KP x 64 elements in registers, element e = 64 i + lane of
// its list (ascending word order in e), keys[i] its order key or 0 for "no element"; n elements in all.  Writes a SUPERSET of the
// kk largest (kk <= n) of at most `cap` elements to dst in ascending e and returns its size.  The threshold is found bit by bit
// from the top (t keeps a bit if at least kk keys are >= t with it set; a round is KP compares + ballots + scalar popcounts) and
// the search stops as soon as no more than `cap` keys are >= t: all of those are written -- typically after a few rounds, the
// exact selection being the business of whoever reads the list.  If the search runs to the last bit (more than cap keys >= the
// kk-th largest: a mass of equal values), exactly kk are written: every key above the threshold and, of the equal ones, those
// with the lowest e.
template <int KP, typename WordFn>
__device__ __forceinline__ int wave_topk(const uint32_t (&keys)[KP], const float (&vals)[KP], WordFn word_of, int n, int kk, int cap,
                                         float* dst_v, int32_t* dst_i) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t t = 0;
  int n_ge = n;                            // keys >= t (every element's key is >= 1 > 0)
  for (int bit = 31; bit >= 0 && n_ge > cap; --bit) {
    const uint32_t c = t | (1u << bit);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < KP; ++i) cnt += (int)__popcll(__ballot(keys[i] >= c));
    if (cnt >= kk) { t = c; n_ge = cnt; }
  }
  int n_gt = 0;
#pragma unroll
  for (int i = 0; i < KP; ++i) n_gt += (int)__popcll(__ballot(keys[i] > t));
  // everything >= t when that fits; else (all 32 bits fixed, t = the kk-th largest key) kk - n_gt of the elements AT t, lowest e first
  const int need = n_ge <= cap ? n_ge - n_gt : kk - n_gt;
  const uint32_t t_cmp = t == 0u ? 1u : t;   // t = 0: nothing was searched (n <= cap): every element (key >= 1) goes out via "above"
  int ties = 0, outp = 0;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const bool tie = keys[i] == t_cmp && t != 0u;
    const unsigned long long tb = __ballot(tie);
    const bool emit = (t == 0u ? keys[i] != 0u : keys[i] > t) || (tie && ties + (int)__popcll(tb & below) < need);
    const unsigned long long eb = __ballot(emit);
    if (emit) {
      const int pos = outp + (int)__popcll(eb & below);
      dst_v[pos] = vals[i];
      dst_i[pos] = word_of(i);
    }
    ties += (int)__popcll(tb);
    outp += (int)__popcll(eb);
  }
  return outp;
}

// One work-group per (slice, row): each of its four waves selects the top_k of a quarter of the slice on its own, wave 0 then
// selects the slice's top_k from the four lists (<= 1024 candidates, still in ascending word order) and writes them.
template <int KP>   // 64-element registers per lane and wave in the first stage: 16 up to a 131072-word vocabulary, else 32
__global__ __launch_bounds__(kSliceThreads) void sample_slice_kernel(const float* __restrict__ logits, int T, int row_stride, int V_l,
                                                                     const float* __restrict__ params, float* __restrict__ cand_v,
                                                                     int32_t* __restrict__ cand_i) {
  __shared__ float s_v[4][kSampleMaxTopK];
  __shared__ int32_t s_i[4][kSampleMaxTopK];
  __shared__ int s_cnt[4];
  const int sp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int V = T * V_l, per = ceil_div(V, kSampleSplits), sub = ceil_div(per, 4);
  const int v1 = min(sp * per + per, V);
  const int w0 = min(sp * per + wv * sub, v1), nw = min(w0 + sub, v1) - w0;   // this wave's words [w0, w0 + nw)
  int top_k = params ? (int)params[b * 3] : 1;
  top_k = min(max(top_k, 1), min(kSampleMaxTopK, V));
  float* out_v = cand_v + ((size_t)b * kSampleSplits + sp) * kSampleMaxTopK;
  int32_t* out_i = cand_i + ((size_t)b * kSampleSplits + sp) * kSampleMaxTopK;
  const float kFill = __builtin_bit_cast(float, 0xffc00000u);
  {
    float vals[KP];
    uint32_t keys[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {          // every load of the lane in flight at once
      const int j = i * 64 + lane, v = w0 + j;
      vals[i] = j >= nw ? 0.f : T == 1 ? logits[(size_t)b * V_l + v] : logits[((size_t)(v / V_l) * row_stride + b) * V_l + (v % V_l)];
    }
#pragma unroll
    for (int i = 0; i < KP; ++i) keys[i] = i * 64 + lane < nw ? f32_order_key(vals[i]) : 0u;   // 0: below every float's key (but one NaN's)
    const int kk = min(top_k, nw);
    int wrote = 0;
    if (kk > 0) wrote = wave_topk<KP>(keys, vals, [&](int i) { return w0 + i * 64 + lane; }, nw, kk, kSampleMaxTopK, s_v[wv], s_i[wv]);
    if (lane == 0) s_cnt[wv] = wrote;
  }
  __syncthreads();
  if (wv != 0) return;
  {
    constexpr int KP2 = 4 * kSampleMaxTopK / 64;   // 16
    float vals[KP2];
    uint32_t keys[KP2];
    int32_t words[KP2];
    int total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) total += s_cnt[w];
#pragma unroll
    for (int i = 0; i < KP2; ++i) {
      const int e = i * 64 + lane, w = e / kSampleMaxTopK, j = e % kSampleMaxTopK;
      const bool real = j < s_cnt[w];
      vals[i] = real ? s_v[w][j] : 0.f;
      words[i] = real ? s_i[w][j] : 0;
      keys[i] = real ? f32_order_key(vals[i]) : 0u;
    }
    const int kk = min(top_k, total);
    int wrote = 0;
    if (kk > 0) wrote = wave_topk<KP2>(keys, vals, [&](int i) { return words[i]; }, total, kk, kSampleMaxTopK, out_v, out_i);
    for (int j = wrote + lane; j < kSampleMaxTopK; j += 64) { out_v[j] = kFill; out_i[j] = 0x7fffffff; }
  }
}

// ---- all rows greedy: argmax split over the chip -------------------------------------------------
// One work-group per row walks a 128k vocabulary in ~50 us; kArgmaxSplits work-groups per row take ~3 us,
// a second launch merges their (value, index) pairs -- same rule as the one-work-group form: the
// largest logit, ties to the lowest vocabulary index, an all-NaN row gives token 0.
constexpr int kArgmaxSplits = 64;
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, int T, int row_stride, int V_l,
                                                             float* __restrict__ part_v, int32_t* __restrict__ part_i) {
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  const int b = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const int V = T * V_l, per = ceil_div(V, kArgmaxSplits);
  const int v0 = sp * per, v1 = min(v0 + per, V);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = v0 + tid; v < v1; v += 256) {
    const float x = logits[((size_t)(v / V_l) * row_stride + b) * V_l + (v % V_l)];
    if (x > bv) { bv = x; bi = v; }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
    part_v[b * kArgmaxSplits + sp] = bv;
    part_i[b * kArgmaxSplits + sp] = bi;
  }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ part_v, const int32_t* __restrict__ part_i,
                                                          int32_t* __restrict__ tokens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float bv = part_v[b * kArgmaxSplits + lane];
  int bi = part_i[b * kArgmaxSplits + lane];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) tokens[b] = bi == 0x7fffffff ? 0 : bi;
}
static_assert(kArgmaxSplits == 64, "argmax_final_kernel merges one partial per lane of one wave");
static_assert(kSampleSplits * kSampleMaxTopK >= kArgmaxSplits, "the candidate arrays also hold the argmax partials");
size_t sample_scratch_bytes(int max_rows) { return (size_t)max_rows * kSampleSplits * kSampleMaxTopK * 8; }

int launch_sample_rows(const float* logits, int T, int row_stride, int V_l, int B, const float* params,
                       unsigned long long seed, int row0, int32_t* tokens, hipStream_t s, void* scratch) {
  MI_CHECK(B >= 1 && T >= 1 && V_l >= 1, "sample: bad shape");
  if (!params && scratch) {   // every row greedy and the caller has room for the partials
    float* pv = reinterpret_cast<float*>(scratch);
    int32_t* pi = reinterpret_cast<int32_t*>(pv + (size_t)B * kArgmaxSplits);
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(kArgmaxSplits, B), dim3(256), 0, s, logits, T, row_stride, V_l, pv, pi);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, s, pv, pi, tokens);
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  const int V = T * V_l;
  if (scratch && V >= 8192 && ceil_div(V, kSampleSplits) <= kSliceMax) {   // the vocabulary pre-selected over the chip
    const int NC = kSampleSplits * kSampleMaxTopK;
    float* cv = reinterpret_cast<float*>(scratch);
    int32_t* ci = reinterpret_cast<int32_t*>(cv + (size_t)B * NC);
    if (ceil_div(ceil_div(V, kSampleSplits), 4) <= 16 * 64)
      hipLaunchKernelGGL(sample_slice_kernel<16>, dim3(kSampleSplits, B), dim3(kSliceThreads), 0, s, logits, T, row_stride, V_l, params, cv, ci);
    else
      hipLaunchKernelGGL(sample_slice_kernel<32>, dim3(kSampleSplits, B), dim3(kSliceThreads), 0, s, logits, T, row_stride, V_l, params, cv, ci);
    hipLaunchKernelGGL(sample_rows_kernel, dim3(B), dim3(kSampleThreads), 0, s, cv, 1, B, NC, params, seed, row0, tokens, ci);
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  hipLaunchKernelGGL(sample_rows_kernel, dim3(B), dim3(kSampleThreads), 0, s, logits, T, row_stride, V_l, params,
                     seed, row0, tokens, nullptr);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// =====================================================================================
// fused speculation: the glue between the draft's chained steps and the target's pass
// =====================================================================================
__global__ void spec_advance_kernel(int B, int k, int step, const int32_t* __restrict__ draft_tokens,
                                    int32_t* __restrict__ draft_dec, int draft_rows, const int32_t* __restrict__ draft_bt,
                                    int MB, int bs, const int32_t* __restrict__ limit, int32_t* __restrict__ target_ids,
                                    int32_t* __restrict__ cand) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int tok = draft_tokens[b];
  if (step + 1 < k) {
    cand[b * k + step + 1] = tok;
    target_ids[b * k + step + 1] = tok;
  }
  int32_t* d_ctx = draft_dec;
  int32_t* d_ids = draft_dec + draft_rows;
  int32_t* d_pos = draft_dec + 2 * draft_rows;
  int32_t* d_slots = draft_dec + 3 * draft_rows;
  if (step + 1 < limit[b]) {
    const int pos = d_pos[b] + 1;
    d_ids[b] = tok;
    d_pos[b] = pos;
    d_ctx[b] = pos + 1;
    d_slots[b] = draft_bt[(size_t)b * MB + pos / bs] * bs + pos % bs;
  } else {
    d_slots[b] = -1;   // same row again, nothing written
  }
}

int launch_spec_advance(int B, int k, int step, const int32_t* draft_tokens, int32_t* draft_dec, int draft_rows,
                        const int32_t* draft_bt, int MB, int block_size, const int32_t* limit, int32_t* target_ids,
                        int32_t* cand, hipStream_t s) {
  hipLaunchKernelGGL(spec_advance_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, B, k, step, draft_tokens, draft_dec,
                     draft_rows, draft_bt, MB, block_size, limit, target_ids, cand);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

__global__ void spec_accept_kernel(int B, int k, const int32_t* __restrict__ target_tokens, const int32_t* __restrict__ cand,
                                   const int32_t* __restrict__ limit, const int32_t* __restrict__ pos0,
                                   int32_t* __restrict__ out_tokens, int32_t* __restrict__ next_pos) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int lim = limit[b];   // rows of this sequence that were computed: 1..k
  int n = 0;
  while (n + 1 < lim && cand[b * k + n + 1] == target_tokens[b * k + n]) ++n;
  for (int i = 0; i < k; ++i) out_tokens[b * k + i] = i <= n ? target_tokens[b * k + i] : 0;
  next_pos[b] = pos0[b] + n + 1;
}

int launch_spec_accept(int B, int k, const int32_t* target_tokens, const int32_t* cand, const int32_t* limit,
                       const int32_t* pos0, int32_t* out_tokens, int32_t* next_pos, hipStream_t s) {
  hipLaunchKernelGGL(spec_accept_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, B, k, target_tokens, cand, limit, pos0,
                     out_tokens, next_pos);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
