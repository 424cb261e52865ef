// Weight-quantized linear layers for gfx950 (MI355X).
//
//   y[m, n] = (sum_k x[m, k] * q[n, k]) * scale[n] (+ bias[n])      q in {e4m3fn, int8, bf16}
//
// Replaces the NxDI quantized Column/RowParallel linears the reference reaches through
// /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:339-348, with the
// quantization keys of loader.py:886-898.
//
// Storage: the library owns the weights, so they are kept in the shape the matrix cores
// want.  A tile is 16 output rows x 64 bytes of K (fp8/int8; 32 elements for bf16) = 1 KiB
// laid out lane-linear: lane l holds the 16 bytes of row (l & 15), k-chunk (l >> 4).  One
// 16-byte load per lane is then (a) a perfectly coalesced 1 KiB wave read and (b) exactly
// the A operand of v_mfma_f32_16x16x32_bf16 after an in-register fp8->bf16 decode (two
// k-steps per load for the 1-byte types).  Tiles of one 16-row group are contiguous along K,
// so a wave streams a dense byte range.
//
// Decode (M <= 16): gemv_kernel — pure weight streaming, HBM-bound.  x lives in LDS in
//   fragment order; every wave streams a K-slice of one row-tile with 2 x 8 KiB in flight;
//   the MFMA does the dot products AND the cross-lane reduction; waves combine through LDS
//   in a fixed order (deterministic, no atomics).
// Prefill (M > 16): gemm_kernel — 128 x 128 tile, x through swizzled LDS, W straight
//   from global in fragment order.
//
// Both share one epilogue on D[n = 4g + r][m = c] (lane = 16 g + c): dequant scale, bias,
// then fp32 store | RoPE + q/KV-pool scatter | SwiGLU.

#include "linear_kernels.h"

namespace mi {

#define MI_TRY_(expr)              \
  do {                             \
    int _rc = (expr);              \
    if (_rc != MI_OK) return _rc;  \
  } while (0)

// =====================================================================================
// Epilogue
// =====================================================================================
template <int EPI>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int m, int n0, f32x4_t v);

template <int EPI>
__device__ __forceinline__ void epilogue(const EpiArgs& e, int m, int n0, f32x4_t v) {
  float4 sc = *reinterpret_cast<const float4*>(e.scale + n0);
  if (e.row_scale) {   // FP8 activations: the token's scale multiplies the channel's
    const float rs = e.row_scale[m];
    sc.x *= rs; sc.y *= rs; sc.z *= rs; sc.w *= rs;
  }
  v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
  if (e.bias) {
    const float4 b = *reinterpret_cast<const float4*>(e.bias + n0);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  epilogue_store<EPI>(e, m, n0, v);
}

// GEMV form: the scales / biases of the row-tiles a work-group owns were copied to LDS at
// kernel start, so the epilogue never queues a global load behind the weight stream (vmcnt
// retires in order: a late scale load would drain the whole prefetch).
template <int EPI>
__device__ __forceinline__ void epilogue_lds(const EpiArgs& e, int m, int n0, const float* sc4, const float* b4,
                                             f32x4_t v, float rmul = 1.f) {
  const float4 sc = *reinterpret_cast<const float4*>(sc4);
  v[0] *= sc.x * rmul; v[1] *= sc.y * rmul; v[2] *= sc.z * rmul; v[3] *= sc.w * rmul;
  if (b4) {
    const float4 b = *reinterpret_cast<const float4*>(b4);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  epilogue_store<EPI>(e, m, n0, v);
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int m, int n0, f32x4_t v) {
  if constexpr (EPI == EPI_F32) {
    *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ld_out + n0) =
        make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_RESID) {  // residual stream: out = resid_in + y (fp32)
    const float4 r = *reinterpret_cast<const float4*>(e.resid_in + (size_t)m * e.ld_out + n0);
    *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ld_out + n0) =
        make_float4(r.x + v[0], r.y + v[1], r.z + v[2], r.w + v[3]);
  } else if constexpr (EPI == EPI_SWIGLU) {
    const float a0 = v[0] / (1.f + __expf(-v[0])) * v[1];
    const float a1 = v[2] / (1.f + __expf(-v[2])) * v[3];
    *reinterpret_cast<uint32_t*>(e.act_out + (size_t)m * e.ld_act + (n0 >> 1)) =
        pack_bf16x2(a0, a1);
  } else {  // EPI_QKV
    const int qk_end = e.q_dim + e.kv_dim;
    if (n0 < qk_end) {  // rotary: (v0,v1) and (v2,v3) are (d, d + hd/2) pairs
      const int j = (n0 < e.q_dim ? n0 : n0 - e.q_dim) % e.hd;
      const int half = e.hd >> 1;
      const float2 cs = *reinterpret_cast<const float2*>(e.rope_cos + (size_t)e.pos[m] * half + (j >> 1));
      const float2 sn = *reinterpret_cast<const float2*>(e.rope_sin + (size_t)e.pos[m] * half + (j >> 1));
      const float a0 = v[0] * cs.x - v[1] * sn.x, a1 = v[1] * cs.x + v[0] * sn.x;
      const float b0 = v[2] * cs.y - v[3] * sn.y, b1 = v[3] * cs.y + v[2] * sn.y;
      v[0] = a0; v[1] = a1; v[2] = b0; v[3] = b1;
    }
    const uint2 packed = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    if (n0 < e.q_dim) {
      *reinterpret_cast<uint2*>(e.q_out + (size_t)m * e.q_dim + n0) = packed;
    } else {
      const int slot = e.slots[m];
      if (slot >= 0) {
        const bool is_k = n0 < qk_end;
        const int rel = is_k ? n0 - e.q_dim : n0 - qk_end;
        const int head = rel / e.hd, d = rel % e.hd;
        const int blk = slot / e.block_size, off = slot % e.block_size;
        uint16_t* pool = is_k ? e.kpool : e.vpool;
        *reinterpret_cast<uint2*>(pool + (((size_t)blk * e.nkv + head) * e.block_size + off) * e.hd + d) = packed;
      }
    }
  }
}

// Operands the epilogue needs from memory besides the accumulator (the residual it adds to; the
// rotary factors and the KV slot of the token).  The GEMV requests them for the work-group's
// first row-tile in its prologue, so that the end of the kernel is not a chain of dependent
// L2 round trips (position -> cos/sin -> store) behind the last weight batch.
template <int EPI> struct EpiPre {};
template <> struct EpiPre<EPI_RESID> { float4 r; };
template <> struct EpiPre<EPI_QKV> { float2 cs, sn; int pos, slot; };

// independent loads (requested with the kernel's first operands) ...
template <int EPI>
__device__ __forceinline__ void epi_prefetch(const EpiArgs& e, int m, int n0, EpiPre<EPI>& p) {
  if constexpr (EPI == EPI_RESID) {
    p.r = *reinterpret_cast<const float4*>(e.resid_in + (size_t)m * e.ld_out + n0);
  } else if constexpr (EPI == EPI_QKV) {
    p.pos = e.pos[m];
    p.slot = e.slots[m];
  }
}
// ... and the ones that hang off them (requested once the activations are staged: the position
// has long arrived, and nothing waits for these before the last weight batch)
template <int EPI>
__device__ __forceinline__ void epi_prefetch_dependent(const EpiArgs& e, int n0, EpiPre<EPI>& p) {
  if constexpr (EPI == EPI_QKV) {
    const int qk_end = e.q_dim + e.kv_dim;
    const int nn = n0 < qk_end ? n0 : 0;                      // V columns: any valid table entry
    const int j = (nn < e.q_dim ? nn : nn - e.q_dim) % e.hd;
    const size_t at = (size_t)p.pos * (e.hd >> 1) + (j >> 1);
    p.cs = *reinterpret_cast<const float2*>(e.rope_cos + at);
    p.sn = *reinterpret_cast<const float2*>(e.rope_sin + at);
  }
}

// epilogue_lds with the prefetched operands
template <int EPI>
__device__ __forceinline__ void epilogue_pre(const EpiArgs& e, int m, int n0, const float* sc4, const float* b4,
                                             f32x4_t v, const EpiPre<EPI>& p, float rmul = 1.f) {
  if constexpr (EPI == EPI_RESID || EPI == EPI_QKV) {
    const float4 sc = *reinterpret_cast<const float4*>(sc4);
    v[0] *= sc.x * rmul; v[1] *= sc.y * rmul; v[2] *= sc.z * rmul; v[3] *= sc.w * rmul;
    if (b4) {
      const float4 b = *reinterpret_cast<const float4*>(b4);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if constexpr (EPI == EPI_RESID) {
      *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ld_out + n0) =
          make_float4(p.r.x + v[0], p.r.y + v[1], p.r.z + v[2], p.r.w + v[3]);
    } else {
      const int qk_end = e.q_dim + e.kv_dim;
      if (n0 < qk_end) {
        const float a0 = v[0] * p.cs.x - v[1] * p.sn.x, a1 = v[1] * p.cs.x + v[0] * p.sn.x;
        const float b0 = v[2] * p.cs.y - v[3] * p.sn.y, b1 = v[3] * p.cs.y + v[2] * p.sn.y;
        v[0] = a0; v[1] = a1; v[2] = b0; v[3] = b1;
      }
      const uint2 packed = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      if (n0 < e.q_dim) {
        *reinterpret_cast<uint2*>(e.q_out + (size_t)m * e.q_dim + n0) = packed;
      } else if (p.slot >= 0) {
        const bool is_k = n0 < qk_end;
        const int rel = is_k ? n0 - e.q_dim : n0 - qk_end;
        const int head = rel / e.hd, d = rel % e.hd;
        const int blk = p.slot / e.block_size, off = p.slot % e.block_size;
        uint16_t* pool = is_k ? e.kpool : e.vpool;
        *reinterpret_cast<uint2*>(pool + (((size_t)blk * e.nkv + head) * e.block_size + off) * e.hd + d) = packed;
      }
    }
  } else {
    epilogue_lds<EPI>(e, m, n0, sc4, b4, v, rmul);
  }
}

// =====================================================================================
// GEMV (token generation)
// =====================================================================================
// LDS: [x fragments: K/8 chunks * M * 16 B][zero slot 16 B][red: 2 * WAVES * 64 lanes * 16 B][scales][biases][sums of squares]
constexpr int kGemvWaves = 8;
constexpr int kGemvU = 8;  // 1 KiB loads per batch; two batches in flight per wave

constexpr int kGemvMaxTilesPerWg = 64;  // row-tiles one work-group may own (scale/bias cache)
size_t gemv_lds_bytes(int M, int K) {
  // x image, zero slot, reduction buffers, scale + bias cache, per-wave sums of squares [16 rows][16]
  return (size_t)M * K * 2 + 16 + 2 * kGemvWaves * 64 * 16 + 2 * kGemvMaxTilesPerWg * 16 * 4 + 16 * 16 * 4;
}
// every M <= 16 streams the weights: images that do not fit in LDS whole are staged in K-chunks
// (gemv_bigk_kernel)
bool gemv_fits(int M, int K) { (void)K; return M <= 32; }
// the whole-image kernels multiply one 16-column MFMA group
static bool gemv_fits_whole(int M, int K) { return M <= 16 && gemv_lds_bytes(M, K) <= 160 * 1024; }

// position (in 16-byte units) of the 8-element chunk c8 of row m in the fragment image
template <int WD>
__device__ __forceinline__ int xfrag_slot(int c8, int m, int M) {
  if constexpr (WD == MI_W_BF16) {
    return c8 * M + m;  // kt = c8 / 4, g = c8 % 4
  } else {
    const int kt = c8 >> 3, c = c8 & 7;  // k = kt*64 + g*16 + s*8 + j
    return (((kt * 2 + (c & 1)) * 4 + (c >> 1)) * M) + m;
  }
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ---- staging of the activations into LDS (MFMA-fragment order) ----------------------------
// A wave's vector loads retire IN ORDER.  Operands requested after the first weight batch only
// arrive once those weights have crossed the chip (microseconds when every CU asks at once), so
// the kernel requests its own L2-resident operands FIRST (gemv_pro_load), then the first weight
// batch, and only then consumes the operands (gemv_pro_finish): the prologue arithmetic runs
// under the weight fetch.  What does not fit that scheme (K > 8 * threads, M > 4 with a norm
// prologue, > kXMax chunks per thread) goes through gemv_stage_late after the weights.
constexpr int kXMax = 16;  // 16-byte activation chunks one thread may hold ahead of the weights

// register-resident prologue operands (plain local arrays: hipcc keeps them in VGPRs)
template <int PRO> struct ProShape {
  static constexpr int NX = PRO == PRO_BF16 ? kXMax : 1;       // 16-byte activation chunks
  static constexpr int NH = PRO == PRO_BF16 ? 1 : 4;           // residual rows
  static constexpr int NP = PRO == PRO_NORM_PARTIAL ? 4 : 1;   // partial rows
};
template <int PRO>
__device__ __forceinline__ bool gemv_pro_is_early(int M, int K) {
  if constexpr (PRO == PRO_BF16) return true;                    // (first kXMax chunks per thread)
  else return (K >> 3) <= (int)blockDim.x && M <= 4;
}

template <int PRO>
__device__ __forceinline__ void gemv_pro_load(const ProArgs& p, int M, int K, u32x4_t (&xv)[ProShape<PRO>::NX],
                                              float (&h)[ProShape<PRO>::NH][8], float (&pv)[ProShape<PRO>::NP][8],
                                              float (&g)[8]) {
  const int tid = threadIdx.x, nthr = blockDim.x, nchunk = K >> 3;
  if constexpr (PRO == PRO_BF16) {
    // As many 16-byte chunks per thread as the image needs, in groups of 4 (work-group-uniform
    // branches; inside a group the index is clamped, never predicated).  Loads nobody needs would
    // still take slots of the CU's ~64-deep load queue ahead of the first weight batch.
    const int total = M * nchunk;
#pragma unroll
    for (int q0 = 0; q0 < kXMax; q0 += 4) {
      if (q0 * nthr < total) {
#pragma unroll
        for (int q = q0; q < q0 + 4; ++q) {
          const int ic = min(tid + q * nthr, total - 1);
          const int m = ic / nchunk, c8 = ic - m * nchunk;
          xv[q] = *reinterpret_cast<const u32x4_t*>(p.x + (size_t)m * p.ldx + (size_t)c8 * 8);
        }
      }
    }
  } else {
    const int c0 = min(tid, nchunk - 1);
    load8(p.gain + c0 * 8, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) load8(p.resid_in + (size_t)min(r, M - 1) * K + c0 * 8, h[r]);
    if constexpr (PRO == PRO_NORM_PARTIAL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) load8(p.partial + (size_t)min(r, M - 1) * K + c0 * 8, pv[r]);
    }
  }
}

// bf16(h * rinv * g) -> fragment slot; optionally h -> resid_out (block 0 only)
template <int WD>
__device__ __forceinline__ void gemv_emit_row(const ProArgs& p, int M, int K, int m, int c8, const float (&h)[8],
                                              const float (&g)[8], float rinv, uint4* xf) {
  if (p.resid_out && blockIdx.x == 0) {
    float* o = p.resid_out + (size_t)m * K + c8 * 8;
    *reinterpret_cast<float4*>(o) = make_float4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(h[4], h[5], h[6], h[7]);
  }
  uint4 o4;
  o4.x = pack_bf16x2(h[0] * rinv * g[0], h[1] * rinv * g[1]);
  o4.y = pack_bf16x2(h[2] * rinv * g[2], h[3] * rinv * g[3]);
  o4.z = pack_bf16x2(h[4] * rinv * g[4], h[5] * rinv * g[5]);
  o4.w = pack_bf16x2(h[6] * rinv * g[6], h[7] * rinv * g[7]);
  xf[xfrag_slot<WD>(c8, m, M)] = o4;
}

template <int WD, int PRO>
__device__ __forceinline__ void gemv_pro_finish(const ProArgs& p, int M, int K, u32x4_t (&xv)[ProShape<PRO>::NX],
                                                float (&h)[ProShape<PRO>::NH][8], float (&pv)[ProShape<PRO>::NP][8],
                                                float (&g)[8], uint4* xf, float* red_f) {
  const int tid = threadIdx.x, nthr = blockDim.x, nchunk = K >> 3;
  if constexpr (PRO == PRO_BF16) {
    const int total = M * nchunk;
#pragma unroll
    for (int q = 0; q < kXMax; ++q) {
      const int i = tid + q * nthr;
      if (i < total) {
        const int m = i / nchunk, c8 = i - m * nchunk;
        reinterpret_cast<u32x4_t*>(xf)[xfrag_slot<WD>(c8, m, M)] = xv[q];
      }
    }
    for (int i = tid + kXMax * nthr; i < total; i += nthr) {   // rare: more than kXMax chunks per thread
      const int m = i / nchunk, c8 = i - m * nchunk;
      xf[xfrag_slot<WD>(c8, m, M)] = *reinterpret_cast<const uint4*>(p.x + (size_t)m * p.ldx + (size_t)c8 * 8);
    }
  } else {
    const int wave = tid >> 6, lane = tid & 63;
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (PRO == PRO_NORM_PARTIAL) h[r][e] += pv[r][e];
        ss[r] = __builtin_fmaf(h[r][e], h[r][e], ss[r]);   // one fixed fma chain per row: identical rows -> identical bits
      }
      if (tid >= nchunk) ss[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = wave_sum(ss[r]);
      if (lane == 0) red_f[r * 16 + wave] = t;     // read after the staging barrier (gemv_row_rinv)
    }
    if (tid < nchunk) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < M) gemv_emit_row<WD>(p, M, K, r, tid, h[r], g, 1.0f, xf);
    }
  }
}

// General norm prologue (any M <= 16, any K), requested after the first weight batch: one pass
// over the rows -- stage bf16(h * g), leave the per-wave sums of squares in LDS.
template <int WD, int PRO>
__device__ __forceinline__ void gemv_stage_late(const ProArgs& p, int M, int K, uint4* xf, float* red_f) {
  const int tid = threadIdx.x, nthr = blockDim.x, nchunk = K >> 3;
  const int wave = tid >> 6, lane = tid & 63;
  for (int m0 = 0; m0 < M; m0 += 4) {
    int row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) row[r] = min(m0 + r, M - 1);
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c8 = tid; c8 < nchunk; c8 += nthr) {
      float h[4][8], g[8];
      load8(p.gain + c8 * 8, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) load8(p.resid_in + (size_t)row[r] * K + c8 * 8, h[r]);
      if constexpr (PRO == PRO_NORM_PARTIAL) {
        float pv[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) load8(p.partial + (size_t)row[r] * K + c8 * 8, pv[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) h[r][e] += pv[r][e];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss[r] = __builtin_fmaf(h[r][e], h[r][e], ss[r]);
        if (m0 + r < M) gemv_emit_row<WD>(p, M, K, m0 + r, c8, h[r], g, 1.0f, xf);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = wave_sum(ss[r]);
      if (lane == 0 && m0 + r < M) red_f[(m0 + r) * 16 + wave] = t;
    }
  }
}

// 1 / rms of activation row m, from the per-wave sums the staging left in LDS (after its barrier).
// The normalisation is applied to the projection's OUTPUT, y = (W . (h * g)) / rms(h): the same
// value as W . (h * g / rms) with the row reduction off the staging path (one barrier and a chain
// of LDS round trips less before the first MFMA; -0.07 ms per step measured).  The GEMV input is
// therefore bf16(h * g) where the context-encoding path rounds bf16(h * g / rms): one rounding of
// an fp32 product in both, at a different scale.
__device__ __forceinline__ float gemv_row_rinv(const float* ssq, int m, int K, float eps) {
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < kGemvWaves; ++w) tot += ssq[m * 16 + w];
  return rsqrtf(tot / (float)K + eps);
}

// Phase timeline of the decode GEMVs (dev builds only: -DMI_TRACE, see tools/trace_gemv.py).
// Wave 0 of every work-group stamps the 100 MHz wall clock at fixed points into
// g_trace_buf[launch % kTraceLaunches][block][stamp]; launches are numbered by a device counter.
#ifdef MI_TRACE
constexpr int kTraceLaunches = 160, kTraceBlocks = 512, kTraceStamps = 8;
__device__ unsigned long long* g_trace_buf = nullptr;
__device__ unsigned int g_trace_seq = 0;
extern "C" int mi_debug_trace(void* buf) {
  unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_seq), &zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
#define MI_TRACE_BEGIN()                                                                            \
  unsigned long long* tb_ = nullptr;                                                                \
  if (g_trace_buf && blockIdx.x < kTraceBlocks)                                                     \
    tb_ = g_trace_buf + ((size_t)(__hip_atomic_load(&g_trace_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % kTraceLaunches) * kTraceBlocks + blockIdx.x) * kTraceStamps;
#define MI_STAMP(k) do { if (tb_ && threadIdx.x == 0) tb_[k] = wall_clock64(); } while (0)
#define MI_TRACE_END() do { if (tb_ && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_trace_seq, 1u); } while (0)
#else
#define MI_TRACE_BEGIN()
#define MI_STAMP(k)
#define MI_TRACE_END()
#endif

// Work decomposition.  A work-group is 8 waves.  KS of them (KS in {1,2,4,8}) split the K
// dimension of ONE row-tile and a work-group covers TP <= 8/KS row-tiles per pass, both chosen on
// the host so that every CU streams the same number of bytes (e.g. gate|up of Llama-8B: 1792
// row-tiles = 256 CUs x 7 -> KS 1, TP 7, the eighth wave only helps staging).  KS = 1: waves never
// meet (no barrier, no LDS reduction).
// The hot loop is branch-free: weight loads are unconditional and never predicated (a batch that
// does not exist is requested as ONE 16-byte line, every lane the same address; k-tiles past the
// wave's slice re-request its last tile, an L1 hit, and multiply a zeroed LDS slot), so hipcc keeps
// counted vmcnt waits.  A CU holds at most ~64 wave-loads in flight whatever its waves ask for
// (measured: ~24 GB/s per CU from HBM), so the queue is kept full rather than deep: batch A before
// the activations are staged, batch B before the work-group meets, then A/B alternate.
template <int WD, int PRO, int EPI, int KS>
__global__ __launch_bounds__(kGemvWaves * 64) void gemv_kernel(const uint4* __restrict__ W, int NT, int KT,
                                                               int M, int K, int TP, ProArgs p, EpiArgs e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* xf = reinterpret_cast<uint4*>(smem);
  const int zero_slot = (M * K) >> 3;  // one extra 16-byte slot of zeros behind the image
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + (size_t)M * K * 2 + 16);
  float* sc_lds = reinterpret_cast<float*>(smem + (size_t)M * K * 2 + 16 + 2 * kGemvWaves * 64 * 16);
  float* bi_lds = sc_lds + kGemvMaxTilesPerWg * 16;
  float* ssq_lds = bi_lds + kGemvMaxTilesPerWg * 16;   // [16 rows][16 waves] sums of squares (norm prologues)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int tsub = wave / KS, kslice = wave % KS;
  const bool wave_on = tsub < TP;      // waves beyond KS * TP stage activations and wait
  const int ktw = ceil_div(KT, KS);
  const int kbeg = min(kslice * ktw, KT), kend = min(kbeg + ktw, KT);
  const int klast = max(kend - 1, 0);
  const int nb = ceil_div(ktw, kGemvU);  // batches per row-tile, same for every wave
  const int units = ceil_div(NT, TP);
  const int my_units = blockIdx.x < units ? ceil_div(units - blockIdx.x, gridDim.x) : 0;
  const int total = my_units * nb;
  const bool col_ok = c < M;
  MI_TRACE_BEGIN();
  MI_STAMP(0);

  u32x4_t bufA[kGemvU], bufB[kGemvU];
  auto issue = [&](u32x4_t (&buf)[kGemvU], int i) {
    const bool live = wave_on && i < total;
    i = min(i, total - 1);
    const int tile = min((int)(blockIdx.x + (i / nb) * gridDim.x) * TP + tsub, NT - 1);
    const int kt0 = kbeg + (i % nb) * kGemvU;
    const uint4* base = live ? W + (size_t)tile * KT * 64 + lane : W;
    const size_t kstep = live ? 64 : 0;
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) stream_load16(buf[u], base + (size_t)min(kt0 + u, klast) * kstep);
  };

  // ---- prologue.  A wave's vector loads RETIRE IN ORDER, so the request order is the wait order:
  //   1. the activations (L2-resident: the previous launch has just written them) -- all the
  //      staging arithmetic waits for;
  //   2. the scales / biases / epilogue operands of this work-group -- first touched in this step,
  //      so they come from HBM; requested behind the activations they no longer hold them up
  //      (requested first they cost every launch an HBM round trip before staging: -2.5 % of the
  //      decode step, measured), and they have landed by the time staging is done;
  //   3. the first weight batch.
  const int nsc = my_units * TP * 16;   // scale / bias of local tile slot j = pass * TP + sub-tile
  u32x4_t pro_x[ProShape<PRO>::NX];
  float pro_h[ProShape<PRO>::NH][8], pro_p[ProShape<PRO>::NP][8], pro_g[8];
  const bool early = gemv_pro_is_early<PRO>(M, K);
  gemv_pro_load<PRO>(p, M, K, pro_x, pro_h, pro_p, pro_g);   // unconditional (clamped); ignored on the late path
  __builtin_amdgcn_sched_barrier(0);
  // only the threads that own an entry ask for it (a work-group owns 16..512 scales)
  float scv[2] = {0.f, 0.f}, biv[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = tid + q * (int)blockDim.x;
    if (j < nsc) {
      const int slot = j >> 4, r = j & 15;
      const int tile = min((int)(blockIdx.x + (slot / TP) * gridDim.x) * TP + (slot % TP), NT - 1);
      scv[q] = e.scale[tile * 16 + r];
      biv[q] = e.bias ? e.bias[tile * 16 + r] : 0.f;
    }
  }
  // epilogue operands of this work-group's first row-tile (used by the finishing lanes)
  const int tile0 = min((int)blockIdx.x * TP + min(tsub, TP - 1), NT - 1);
  EpiPre<EPI> epre{};
  const bool finisher = wave_on && kslice == 0;      // the waves that run this work-group's epilogues
  if (finisher) epi_prefetch<EPI>(e, min(c, M - 1), tile0 * 16 + g * 4, epre);
  __builtin_amdgcn_sched_barrier(0);
  issue(bufA, 0);
  __builtin_amdgcn_sched_barrier(0);
  MI_STAMP(1);
  if (tid == 0) xf[zero_slot] = make_uint4(0, 0, 0, 0);
  if (early) gemv_pro_finish<WD, PRO>(p, M, K, pro_x, pro_h, pro_p, pro_g, xf, ssq_lds);
  else gemv_stage_late<WD, PRO>(p, M, K, xf, ssq_lds);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = tid + q * (int)blockDim.x;
    if (j < nsc) {
      sc_lds[j] = scv[q];
      bi_lds[j] = biv[q];
    }
  }
  MI_STAMP(2);
  if (finisher) epi_prefetch_dependent<EPI>(e, tile0 * 16 + g * 4, epre);
  issue(bufB, 1);   // a full queue only blocks a wave that would wait at the barrier anyway
  __syncthreads();
  MI_STAMP(3);
  float rmul = 1.f;     // 1 / rms of this lane's activation row (norm prologues), applied in the epilogue
  if constexpr (PRO != PRO_BF16) {
    if (col_ok) rmul = gemv_row_rinv(ssq_lds, c, K, p.eps);
  }
#ifdef MI_TRACE
  if (tb_ && tid < 4) reinterpret_cast<float*>(tb_ + 6)[tid] = rmul;
#endif

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  int parity = 0;
  auto step = [&](u32x4_t w, int kt) {
    const bool ok = col_ok && kt < kend;
    if constexpr (WD == MI_W_BF16) {
      const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xf[ok ? (kt * 4 + g) * M + c : zero_slot]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), b, acc, 0, 0, 0);
    } else {
      const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 0) * 4 + g) * M + c : zero_slot]);
      const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 1) * 4 + g) * M + c : zero_slot]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(decode8<WD>(w[0], w[1]), b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(decode8<WD>(w[2], w[3]), b1, acc, 0, 0, 0);
    }
  };
  auto finish_tile = [&](int i) {
    const int pass = i / nb;
    const int tile = (int)(blockIdx.x + pass * gridDim.x) * TP + tsub;
    const int lslot = (pass * TP + min(tsub, TP - 1)) * 16 + g * 4;
    const float* sc4 = sc_lds + lslot;
    const float* b4 = e.bias ? bi_lds + lslot : nullptr;
    const bool mine = wave_on && col_ok && tile < NT;
    f32x4_t s = acc;
    if constexpr (KS > 1) {  // combine the K-slices in wave order (deterministic)
      red[(parity * kGemvWaves + wave) * 64 + lane] = acc;
      __syncthreads();
      if (kslice == 0) {
        s = red[(parity * kGemvWaves + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) {
          const f32x4_t t = red[(parity * kGemvWaves + wave + w) * 64 + lane];
          s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
      }
      parity ^= 1;
    }
    if (kslice == 0 && mine) {
      if (pass == 0) epilogue_pre<EPI>(e, c, tile * 16 + g * 4, sc4, b4, s, epre, rmul);
      else epilogue_lds<EPI>(e, c, tile * 16 + g * 4, sc4, b4, s, rmul);
    }
    acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  static_assert(kGemvU == 8, "the wait ladders below are written for 8 loads per batch");
  auto process = [&](u32x4_t (&buf)[kGemvU], int i) {
    const int kt0 = kbeg + (i % nb) * kGemvU;
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) step(buf[u], kt0 + u);
    if ((i % nb) == nb - 1) finish_tile(i);
  };

  for (int i = 0; i < total; i += 2) {
    process(bufA, i);
    if (i == 0) MI_STAMP(4);
    issue(bufA, i + 2);
    if (i + 1 < total) process(bufB, i + 1);
    issue(bufB, i + 3);
  }
  MI_STAMP(5);
  MI_TRACE_END();
}


// Kernel arguments are fetched on demand: hipcc places an s_load + s_waitcnt lgkmcnt(0) in front of
// the first use of each piece, i.e. one scalar-cache miss (the kernarg block of a launch is cold)
// per piece, back to back, on the way to the first weight request.  Naming every field as an SGPR
// input of an empty asm at kernel entry makes the compiler fetch the whole block at once.
template <typename T>
__device__ __forceinline__ void pin_sgpr(const T& v) { asm volatile("" ::"s"(v)); }
template <int PRO, int EPI>
__device__ __forceinline__ void pin_gemv_args(const uint4* W, int NT, int KT, int M, int K, const ProArgs& p, const EpiArgs& e) {
  pin_sgpr(W); pin_sgpr(NT); pin_sgpr(KT); pin_sgpr(M); pin_sgpr(K);
  if constexpr (PRO == PRO_BF16) { pin_sgpr(p.x); pin_sgpr(p.ldx); }
  else { pin_sgpr(p.resid_in); pin_sgpr(p.resid_out); pin_sgpr(p.gain); pin_sgpr(p.eps); }
  if constexpr (PRO == PRO_NORM_PARTIAL) pin_sgpr(p.partial);
  pin_sgpr(e.scale); pin_sgpr(e.bias);
  if constexpr (EPI == EPI_F32 || EPI == EPI_RESID) { pin_sgpr(e.out_f32); pin_sgpr(e.ld_out); }
  if constexpr (EPI == EPI_RESID) pin_sgpr(e.resid_in);
  if constexpr (EPI == EPI_SWIGLU) { pin_sgpr(e.act_out); pin_sgpr(e.ld_act); }
  if constexpr (EPI == EPI_QKV) {
    pin_sgpr(e.q_out); pin_sgpr(e.q_dim); pin_sgpr(e.kv_dim); pin_sgpr(e.hd); pin_sgpr(e.nkv); pin_sgpr(e.pos);
    pin_sgpr(e.slots); pin_sgpr(e.rope_cos); pin_sgpr(e.rope_sin); pin_sgpr(e.kpool); pin_sgpr(e.vpool);
    pin_sgpr(e.block_size);
  }
}

// =====================================================================================
// GEMV, wave-private staging (the short K-split projections: QKV, O, down at M <= 4)
// =====================================================================================
// In-kernel timeline of gemv_kernel on the Llama-8B decode shapes (tools/trace_gemv.py, us, median
// work-group): issue 1.2-1.7 | stage 2.0-2.9 | barrier 0.7-1.6 | first batch 1.0-1.5 | stream 1.0 (O)
// ... 13.4 (gate|up).  For O-proj the 64 KiB a CU streams are 1 us of a 7 us kernel: the rest is the
// serial chain kernel entry -> index arithmetic (runtime divisions) -> x from L2 -> LDS -> work-group
// barrier -> first MFMA.  With K split over the waves (KS > 1) a wave multiplies only ITS K-slice
// of the activations, so it stages exactly that slice, for itself, and never meets the other
// waves before the K-slice reduction at the end of the row-tile:
//   * no prologue barrier, a wave starts its MFMAs as soon as its own 1/KS of x has landed;
//   * TP = 8 / KS and the batch counters are compile-time / incremental: no integer division on
//     the way to the first weight request;
//   * sums of squares of the norm prologues are per-wave partials that meet behind the reduction
//     barrier, where the epilogue (which applies 1 / rms) runs anyway.
// Same image layout, same K-slices, same MFMA order as gemv_kernel: bit-identical products.
constexpr int kPrivJ = 4;       // 16-byte chunks per lane per activation row (slice <= 2048 elements)
constexpr int kPrivJNorm = 2;   // norm prologues hold fp32 rows: slice <= 1024 elements

template <int WD, int PRO, int EPI, int KS>
__global__ __launch_bounds__(kGemvWaves * 64) void gemv_priv_kernel(const uint4* __restrict__ W, int NT, int KT,
                                                                    int M, int K, int G, ProArgs p, EpiArgs e) {
  // G = the grid size as an argument: the builtin reads the dispatch packet (two more dependent scalar loads)
  static_assert(KS == 2 || KS == 4 || KS == 8, "wave-private staging needs a K split");
  pin_gemv_args<PRO, EPI>(W, NT, KT, M, K, p, e);
  pin_sgpr(G);
  constexpr int TP = kGemvWaves / KS;
  constexpr int TK8 = (WD == MI_W_BF16 ? 32 : 64) / 8;   // 16-byte activation chunks per k-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* xf = reinterpret_cast<uint4*>(smem);
  const int zero_slot = (M * K) >> 3;
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + (size_t)M * K * 2 + 16);
  float* sc_lds = reinterpret_cast<float*>(smem + (size_t)M * K * 2 + 16 + 2 * kGemvWaves * 64 * 16);
  float* bi_lds = sc_lds + kGemvMaxTilesPerWg * 16;
  float* ssq_lds = bi_lds + kGemvMaxTilesPerWg * 16;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int tsub = wave / KS, kslice = wave % KS;
  const int ktw = ceil_div(KT, KS);
  const int kbeg = min(kslice * ktw, KT), kend = min(kbeg + ktw, KT);
  const int klast = max(kend - 1, 0);
  const int nb = ceil_div(ktw, kGemvU);
  const int units = ceil_div(NT, TP);
  const int my_units = (int)blockIdx.x < units ? ceil_div(units - (int)blockIdx.x, G) : 0;
  const bool col_ok = c < M;
  MI_TRACE_BEGIN();
  MI_STAMP(0);

  u32x4_t bufA[kGemvU], bufB[kGemvU];
  int is_pass = 0, is_b = 0;     // the next batch to request
  auto issue = [&](u32x4_t (&buf)[kGemvU]) {
    const bool live = is_pass < my_units;
    const int tile = min((int)(blockIdx.x + is_pass * G) * TP + tsub, NT - 1);
    const int kt0 = kbeg + is_b * kGemvU;
    const uint4* base = live ? W + (size_t)tile * KT * 64 + lane : W;
    const size_t kstep = live ? 64 : 0;
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) stream_load16(buf[u], base + (size_t)min(kt0 + u, klast) * kstep);
    if (++is_b == nb) { is_b = 0; ++is_pass; }
  };

  // ---- 1. this wave's slice of the activations (requested first: vector loads retire in order) ----
  const int c0 = kbeg * TK8, nch = (kend - kbeg) * TK8;     // 16-byte chunks [c0, c0 + nch) of every row
  u32x4_t xv[PRO == PRO_BF16 ? 4 : 1][PRO == PRO_BF16 ? kPrivJ : 1];
  float hv[PRO == PRO_BF16 ? 1 : kPrivJNorm][PRO == PRO_BF16 ? 1 : 4][8];
  float pvv[PRO == PRO_NORM_PARTIAL ? kPrivJNorm : 1][PRO == PRO_NORM_PARTIAL ? 4 : 1][8];
  float gv[PRO == PRO_BF16 ? 1 : kPrivJNorm][8];
  if constexpr (PRO == PRO_BF16) {
#pragma unroll
    for (int j = 0; j < kPrivJ; ++j) {
      if (j * 64 < nch) {                                   // wave-uniform
        const int cc = c0 + min(lane + 64 * j, nch - 1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < M) xv[m][j] = *reinterpret_cast<const u32x4_t*>(p.x + (size_t)m * p.ldx + (size_t)cc * 8);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPrivJNorm; ++j) {
      if (j * 64 < nch) {
        const int cc = c0 + min(lane + 64 * j, nch - 1);
        load8(p.gain + (size_t)cc * 8, gv[j]);
#pragma unroll
        for (int m = 0; m < 4; ++m) load8(p.resid_in + (size_t)min(m, M - 1) * K + (size_t)cc * 8, hv[j][m]);
        if constexpr (PRO == PRO_NORM_PARTIAL) {
#pragma unroll
          for (int m = 0; m < 4; ++m) load8(p.partial + (size_t)min(m, M - 1) * K + (size_t)cc * 8, pvv[j][m]);
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. scales / biases of this work-group's row-tiles, epilogue operands of its first one ----
  const int nsc = my_units * TP * 16;
  float scv[2] = {0.f, 0.f}, biv[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = tid + q * kGemvWaves * 64;
    if (j < nsc) {
      const int slot = j >> 4, r = j & 15;
      const int tile = min((int)(blockIdx.x + (slot / TP) * G) * TP + (slot % TP), NT - 1);
      scv[q] = e.scale[tile * 16 + r];
      biv[q] = e.bias ? e.bias[tile * 16 + r] : 0.f;
    }
  }
  const int tile0 = min((int)blockIdx.x * TP + tsub, NT - 1);
  EpiPre<EPI> epre{};
  // unconditional (every wave asks, the finishers use it): under a branch hipcc waits for the load
  // at the merge point -- vmcnt(0) in front of the first weight request
  epi_prefetch<EPI>(e, min(c, M - 1), tile0 * 16 + g * 4, epre);
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. first weight batch, then consume the activations under its flight ----
  issue(bufA);
  __builtin_amdgcn_sched_barrier(0);
  MI_STAMP(1);
  if (lane == 0) xf[zero_slot] = make_uint4(0, 0, 0, 0);   // every wave writes the same zeros: no barrier needed
  if constexpr (PRO == PRO_BF16) {
#pragma unroll
    for (int j = 0; j < kPrivJ; ++j) {
      const int cl = lane + 64 * j;
      if (cl < nch) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < M) reinterpret_cast<u32x4_t*>(xf)[xfrag_slot<WD>(c0 + cl, m, M)] = xv[m][j];
      }
    }
  } else {
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kPrivJNorm; ++j) {
      const int cl = lane + 64 * j;
      if (j * 64 < nch) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if constexpr (PRO == PRO_NORM_PARTIAL) hv[j][m][q] += pvv[j][m][q];
            if (cl < nch) ss[m] = __builtin_fmaf(hv[j][m][q], hv[j][m][q], ss[m]);
          }
          if (cl < nch && m < M) {
            if (p.resid_out && blockIdx.x == 0 && tsub == 0) {
              float* o = p.resid_out + (size_t)m * K + (size_t)(c0 + cl) * 8;
              *reinterpret_cast<float4*>(o) = make_float4(hv[j][m][0], hv[j][m][1], hv[j][m][2], hv[j][m][3]);
              *reinterpret_cast<float4*>(o + 4) = make_float4(hv[j][m][4], hv[j][m][5], hv[j][m][6], hv[j][m][7]);
            }
            uint4 o4;
            o4.x = pack_bf16x2(hv[j][m][0] * gv[j][0], hv[j][m][1] * gv[j][1]);
            o4.y = pack_bf16x2(hv[j][m][2] * gv[j][2], hv[j][m][3] * gv[j][3]);
            o4.z = pack_bf16x2(hv[j][m][4] * gv[j][4], hv[j][m][5] * gv[j][5]);
            o4.w = pack_bf16x2(hv[j][m][6] * gv[j][6], hv[j][m][7] * gv[j][7]);
            xf[xfrag_slot<WD>(c0 + cl, m, M)] = o4;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float t = wave_sum(ss[m]);
      if (lane == 0) ssq_lds[m * 16 + wave] = t;     // read behind the K-slice reduction barrier
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = tid + q * kGemvWaves * 64;
    if (j < nsc) {
      sc_lds[j] = scv[q];
      bi_lds[j] = biv[q];
    }
  }
  MI_STAMP(2);
  epi_prefetch_dependent<EPI>(e, tile0 * 16 + g * 4, epre);
  issue(bufB);
  MI_STAMP(3);

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  int parity = 0;
  int pr_pass = 0, pr_b = 0;     // the batch being multiplied
  auto process = [&](u32x4_t (&buf)[kGemvU]) {
    const int kt0 = kbeg + pr_b * kGemvU;
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) {
      const int kt = kt0 + u;
      const bool ok = col_ok && kt < kend;
      const u32x4_t w = buf[u];
      if constexpr (WD == MI_W_BF16) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xf[ok ? (kt * 4 + g) * M + c : zero_slot]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), b, acc, 0, 0, 0);
      } else {
        const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 0) * 4 + g) * M + c : zero_slot]);
        const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 1) * 4 + g) * M + c : zero_slot]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(decode8<WD>(w[0], w[1]), b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(decode8<WD>(w[2], w[3]), b1, acc, 0, 0, 0);
      }
    }
    if (++pr_b == nb) {          // row-tile done: combine the K-slices in wave order (deterministic)
      pr_b = 0;
      const int pass = pr_pass++;
      const int tile = (int)(blockIdx.x + pass * G) * TP + tsub;
      red[(parity * kGemvWaves + wave) * 64 + lane] = acc;
      __syncthreads();
      if (kslice == 0 && col_ok && tile < NT) {
        f32x4_t sum = red[(parity * kGemvWaves + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < KS; ++w) {
          const f32x4_t t = red[(parity * kGemvWaves + wave + w) * 64 + lane];
          sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
        }
        float rmul = 1.f;
        if constexpr (PRO != PRO_BF16) {   // the KS waves of row-tile 0 hold one partial per K-slice
          float tot = 0.f;
#pragma unroll
          for (int w = 0; w < KS; ++w) tot += ssq_lds[c * 16 + w];
          rmul = rsqrtf(tot / (float)K + p.eps);
        }
        const int lslot = (pass * TP + tsub) * 16 + g * 4;
        const float* sc4 = sc_lds + lslot;
        const float* b4 = e.bias ? bi_lds + lslot : nullptr;
        if (pass == 0) epilogue_pre<EPI>(e, c, tile * 16 + g * 4, sc4, b4, sum, epre, rmul);
        else epilogue_lds<EPI>(e, c, tile * 16 + g * 4, sc4, b4, sum, rmul);
      }
      parity ^= 1;
      acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  };
  const int total = my_units * nb;
  for (int i = 0; i < total; i += 2) {
    process(bufA);
    if (i == 0) MI_STAMP(4);
    issue(bufA);
    if (i + 1 < total) process(bufB);
    issue(bufB);
  }
  MI_STAMP(5);
  MI_TRACE_END();
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses into 1 KiB of LDS starting at `l` (wave-uniform)
__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
static bool gemv_kstream_enabled() {
  static const bool on = [] { const char* v = getenv("MI355X_GEMV_KSTREAM"); return !(v && v[0] == '0'); }();
  return on;
}
static bool gemv_priv_enabled() {
  static const bool on = [] { const char* v = getenv("MI355X_GEMV_PRIV"); return !(v && v[0] == '0'); }();
  return on;
}
// wave-private staging applies when all 8 waves stream (TP = 8 / KS), at most 4 rows, and a wave's
// K-slice fits the per-lane register budget of the prologue
template <int WD, int PRO>
static bool gemv_priv_ok(int M, int KT, int ks, int tp) {
  if (!gemv_priv_enabled() || ks < 2 || ks * tp != kGemvWaves || M > 4) return false;
  const int nch = ceil_div(KT, ks) * (tile_k(WD) / 8);
  return nch <= 64 * (PRO == PRO_BF16 ? kPrivJ : kPrivJNorm);
}

template <int WD, int PRO, int EPI, int KS>
static int launch_gemv_priv(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s, int num_cu) {
  constexpr int TP = kGemvWaves / KS;
  const int NT = w.N / 16, KT = w.K / tile_k(WD);
  const size_t lds = gemv_lds_bytes(M, w.K);
  auto kern = gemv_priv_kernel<WD, PRO, EPI, KS>;
  MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
  const int units = ceil_div(NT, TP);
  int grid = min(units, num_cu);
  grid = max(grid, ceil_div(units * TP, kGemvMaxTilesPerWg));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemvWaves * 64), lds, s, reinterpret_cast<const uint4*>(w.w), NT, KT, M,
                     w.K, grid, p, e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// (KS, TP) for NT row-tiles of KT k-tiles on `slots` resident work-groups: fewest k-tiles on the
// busiest CU (passes x TP x KT/KS x KS), then the fewest passes, then the most streaming waves,
// then the smallest K split.
struct GemvShape { int ks, tp; };
static GemvShape gemv_pick_shape(int NT, int KT, int slots) {
  GemvShape best{8, 1};
  long best_load = 1L << 60;
  int best_pass = 1 << 30, best_waves = 0;
  for (int ks = 1; ks <= 8; ks *= 2) {
    if (KT < ks) break;
    for (int tp = 1; tp * ks <= kGemvWaves; ++tp) {
      const int units = ceil_div(NT, tp);
      const int passes = ceil_div(units, slots);
      const long load = (long)passes * tp * ceil_div(KT, ks) * ks;
      const int waves = tp * ks;
      const bool better = load < best_load || (load == best_load && (passes < best_pass ||
                          (passes == best_pass && (waves > best_waves))));
      if (better) { best = {ks, tp}; best_load = load; best_pass = passes; best_waves = waves; }
    }
  }
  return best;
}

template <int WD, int PRO, int EPI, int KS>
static int launch_gemv_ks(const LinearW& w, int M, int TP, const ProArgs& p, const EpiArgs& e, hipStream_t s, int num_cu) {
  const int NT = w.N / 16, KT = w.K / tile_k(WD);
  const size_t lds = gemv_lds_bytes(M, w.K);
  auto kern = gemv_kernel<WD, PRO, EPI, KS>;
  MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
  // the grid is one resident wave of work-groups (one per CU: ~200 VGPRs x 8 waves), each walking
  // its share of the row-tiles
  const int units = ceil_div(NT, TP);
  int grid = min(units, num_cu);
  // a work-group caches the scales of at most kGemvMaxTilesPerWg row-tiles
  grid = max(grid, ceil_div(units * TP, kGemvMaxTilesPerWg));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemvWaves * 64), lds, s,
                     reinterpret_cast<const uint4*>(w.w), NT, KT, M, w.K, TP, p, e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

template <int WD, int PRO, int EPI>
static int launch_gemv_bigk_t(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s);
template <int WD, int EPI>
static int launch_gemv_kstream(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s);

template <int WD, int PRO, int EPI>
static int launch_gemv_t(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s) {
  if (!gemv_fits_whole(M, w.K)) {
    if constexpr (PRO == PRO_BF16 && (EPI == EPI_F32 || EPI == EPI_RESID)) {
      static const int min_m = [] { const char* v = getenv("MI355X_GEMV_KSTREAM_MIN_M"); return v ? atoi(v) : 1; }();
      if (gemv_kstream_enabled() && M >= min_m) return launch_gemv_kstream<WD, EPI>(w, M, p, e, s);
    }
    return launch_gemv_bigk_t<WD, PRO, EPI>(w, M, p, e, s);
  }
  // (measured: replacing the whole-image kernels by the streamed form wherever the activations are bf16 gains
  //  nothing at 16 rows and loses at 8 and 4 -- 2.72 vs 2.67 ms and 2.29 vs 2.14 ms per Llama-8B pass)
  int num_cu = 0;
  MI_TRY_(device_num_cu(&num_cu));
  const GemvShape sh = gemv_pick_shape(w.N / 16, w.K / tile_k(WD), num_cu);
  if (gemv_priv_ok<WD, PRO>(M, w.K / tile_k(WD), sh.ks, sh.tp)) {
    switch (sh.ks) {
      case 2: return launch_gemv_priv<WD, PRO, EPI, 2>(w, M, p, e, s, num_cu);
      case 4: return launch_gemv_priv<WD, PRO, EPI, 4>(w, M, p, e, s, num_cu);
      default: return launch_gemv_priv<WD, PRO, EPI, 8>(w, M, p, e, s, num_cu);
    }
  }
  switch (sh.ks) {
    case 1: return launch_gemv_ks<WD, PRO, EPI, 1>(w, M, sh.tp, p, e, s, num_cu);
    case 2: return launch_gemv_ks<WD, PRO, EPI, 2>(w, M, sh.tp, p, e, s, num_cu);
    case 4: return launch_gemv_ks<WD, PRO, EPI, 4>(w, M, sh.tp, p, e, s, num_cu);
    default: return launch_gemv_ks<WD, PRO, EPI, 8>(w, M, sh.tp, p, e, s, num_cu);
  }
}


// =====================================================================================
// GEMV with the activations staged in K-chunks (images larger than the LDS)
// =====================================================================================
// down_proj of Qwen2.5-7B (K = 18944) and of Llama-3.3-70B (K = 28672 at TP 1, 14336 at TP 2), or
// any projection at M > 8 rows of an 8192-wide model: M x K bf16 does not fit beside the reduction
// buffers.  Same arithmetic as gemv_kernel (bf16 activations in MFMA-fragment order, fp32
// accumulation, dequant / norm scale in the epilogue), other schedule: one work-group = 8 waves
// owns ONE row-tile at a time, its waves split every K-chunk eight ways, the chunk's activations
// are staged between two barriers while the next chunk's first weight batches are already in
// flight.  The K-slices are combined in wave order (deterministic).
// LDS: [x chunk: CKT k-tiles][zero slot][red: 8 waves x 64 lanes x 16 B][scale 16][bias 16][ssq 16 x 16]
// (17 .. 32 rows: two 16-column MFMA groups per wave share one weight stream -- MG = 2: twice the K-slice reduction buffer, 32 rows of sums of squares)
static size_t gemv_bigk_fixed_lds(int mg) { return 16 + (size_t)mg * kGemvWaves * 64 * 16 + 2 * kGemvWaves * 16 * 4 + (size_t)mg * 16 * 16 * 4; }
static int gemv_bigk_chunk_kt(int M, int K, int wd) {
  const int KT = K / tile_k(wd);
  const size_t per_kt = (size_t)M * tile_k(wd) * 2;
  const int cap = (int)((160 * 1024 - gemv_bigk_fixed_lds(M > 16 ? 2 : 1)) / per_kt);
  const int NC = ceil_div(KT, cap);
  return min(cap, ceil_div(ceil_div(KT, NC), kGemvWaves) * kGemvWaves);   // equal slices for the 8 waves
}

// stage x[:, k0 .. k0 + ck) (ck elements, a multiple of the k-tile) into the fragment image; norm
// prologues leave / accumulate the per-wave sums of squares in ssq[row * 16 + wave]
template <int WD, int PRO>
__device__ __forceinline__ void gemv_stage_chunk(const ProArgs& p, int M, int K, int k0, int ck, uint4* xf, float* ssq,
                                                 bool first_chunk, bool take_ssq, bool store_resid) {
  const int tid = threadIdx.x, nthr = blockDim.x, nch = ck >> 3, c0 = k0 >> 3;
  if constexpr (PRO == PRO_BF16) {
    const int total = M * nch;
    for (int i = tid; i < total; i += nthr) {
      const int m = i / nch, c8 = i - m * nch;
      xf[xfrag_slot<WD>(c8, m, M)] = *reinterpret_cast<const uint4*>(p.x + (size_t)m * p.ldx + (size_t)(c0 + c8) * 8);
    }
  } else {
    const int wave = tid >> 6, lane = tid & 63;
    for (int m0 = 0; m0 < M; m0 += 4) {
      int row[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) row[r] = min(m0 + r, M - 1);
      float ss[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c8 = tid; c8 < nch; c8 += nthr) {
        float h[4][8], g[8];
        load8(p.gain + (size_t)(c0 + c8) * 8, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) load8(p.resid_in + (size_t)row[r] * K + (size_t)(c0 + c8) * 8, h[r]);
        if constexpr (PRO == PRO_NORM_PARTIAL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv[8];
            load8(p.partial + (size_t)row[r] * K + (size_t)(c0 + c8) * 8, pv);
#pragma unroll
            for (int e = 0; e < 8; ++e) h[r][e] += pv[e];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ss[r] = __builtin_fmaf(h[r][e], h[r][e], ss[r]);
          if (m0 + r < M) {
            if (store_resid && p.resid_out) {
              float* o = p.resid_out + (size_t)(m0 + r) * K + (size_t)(c0 + c8) * 8;
              *reinterpret_cast<float4*>(o) = make_float4(h[r][0], h[r][1], h[r][2], h[r][3]);
              *reinterpret_cast<float4*>(o + 4) = make_float4(h[r][4], h[r][5], h[r][6], h[r][7]);
            }
            uint4 o4;
            o4.x = pack_bf16x2(h[r][0] * g[0], h[r][1] * g[1]);
            o4.y = pack_bf16x2(h[r][2] * g[2], h[r][3] * g[3]);
            o4.z = pack_bf16x2(h[r][4] * g[4], h[r][5] * g[5]);
            o4.w = pack_bf16x2(h[r][6] * g[6], h[r][7] * g[7]);
            xf[xfrag_slot<WD>(c8, m0 + r, M)] = o4;
          }
        }
      }
      if (take_ssq) {   // the sums belong to the activations, not to the row-tile: taken once
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = wave_sum(ss[r]);
          if (lane == 0 && m0 + r < M) ssq[(m0 + r) * 16 + wave] = first_chunk ? t : ssq[(m0 + r) * 16 + wave] + t;
        }
      }
    }
  }
}

// KS waves split the K dimension of ONE row-tile and a work-group walks TP = 8 / KS ... row-tiles side by side, all of them
// against the SAME staged activation chunk (gemv_kernel's decomposition): with 17 .. 32 rows the activations are the larger
// operand -- 32 rows x 4096 fp32 residuals = 512 KiB per work-group and pass against 64 KiB of weights per row-tile -- so
// they must cross the chip once per work-group, not once per row-tile.  MG = 2: two 16-column MFMA groups (rows 0..15,
// 16..31) per wave over one weight stream, the weight fragment decoded once.  KS = 8, TP = 1, MG = 1 is the original form.
// The same staging for MANY rows (17 .. 32).  gemv_stage_chunk walks the rows four at a time with one dependent trip to
// L2 per step -- 8 steps x 2 chunks at 32 rows, ~25 us of a 35 us QKV launch.  Here a wave owns rows wave, wave + 8, ...
// whole (its lanes cover the chunk's columns, kRowJ 16-byte groups each), two rows in flight, and leaves the row's sum of
// squares in its own slot of ssq (the other waves' slots stay zero: gemv_row_rinv adds all eight).
constexpr int kRowJ = 4;   // 8-element column groups per lane and row: chunks of up to 64 * 4 * 8 = 2048 columns
template <int WD, int PRO>
__device__ __forceinline__ void gemv_stage_chunk_rows(const ProArgs& p, int M, int K, int k0, int ck, uint4* xf, float* ssq,
                                                      bool first_chunk, bool take_ssq, bool store_resid) {
  static_assert(PRO != PRO_BF16, "norm prologues");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nch = ck >> 3, c0 = k0 >> 3;
  float g[kRowJ][8];
#pragma unroll
  for (int j = 0; j < kRowJ; ++j) load8(p.gain + (size_t)(c0 + min(lane + 64 * j, nch - 1)) * 8, g[j]);
  float hA[kRowJ][8], hB[kRowJ][8];
  auto load_row = [&](float (&h)[kRowJ][8], int m) {
    const int mm = min(m, M - 1);
#pragma unroll
    for (int j = 0; j < kRowJ; ++j) {
      const size_t at = (size_t)mm * K + (size_t)(c0 + min(lane + 64 * j, nch - 1)) * 8;
      load8(p.resid_in + at, h[j]);
      if constexpr (PRO == PRO_NORM_PARTIAL) {
        float pv[8];
        load8(p.partial + at, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[j][e] += pv[e];
      }
    }
  };
  auto emit_row = [&](float (&h)[kRowJ][8], int m) {
    if (m >= M) return;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < kRowJ; ++j) {
      const int c8 = lane + 64 * j;
      if (c8 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(h[j][e], h[j][e], ss);
        if (store_resid && p.resid_out) {
          float* o = p.resid_out + (size_t)m * K + (size_t)(c0 + c8) * 8;
          *reinterpret_cast<float4*>(o) = make_float4(h[j][0], h[j][1], h[j][2], h[j][3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(h[j][4], h[j][5], h[j][6], h[j][7]);
        }
        uint4 o4;
        o4.x = pack_bf16x2(h[j][0] * g[j][0], h[j][1] * g[j][1]);
        o4.y = pack_bf16x2(h[j][2] * g[j][2], h[j][3] * g[j][3]);
        o4.z = pack_bf16x2(h[j][4] * g[j][4], h[j][5] * g[j][5]);
        o4.w = pack_bf16x2(h[j][6] * g[j][6], h[j][7] * g[j][7]);
        xf[xfrag_slot<WD>(c8, m, M)] = o4;
      }
    }
    if (take_ssq) {
      const float t = wave_sum(ss);
      if (lane == 0) ssq[m * 16 + wave] = first_chunk ? t : ssq[m * 16 + wave] + t;
    }
  };
  load_row(hA, wave);
  for (int m = wave; m < M; m += 2 * kGemvWaves) {
    load_row(hB, m + kGemvWaves);
    emit_row(hA, m);
    load_row(hA, m + 2 * kGemvWaves);
    emit_row(hB, m + kGemvWaves);
  }
}

template <int WD, int PRO, int EPI, int MG = 1, int KS = kGemvWaves>
__global__ __launch_bounds__(kGemvWaves * 64) void gemv_bigk_kernel(const uint4* __restrict__ W, int NT, int KT, int M,
                                                                    int K, int CKT, int TP, ProArgs p, EpiArgs e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TK = (WD == MI_W_BF16) ? 32 : 64;
  uint4* xf = reinterpret_cast<uint4*>(smem);
  const int zero_slot = (M * CKT * TK) >> 3;
  unsigned char* tail = smem + (size_t)M * CKT * TK * 2 + 16;
  f32x4_t* red = reinterpret_cast<f32x4_t*>(tail);                     // [MG][8 waves][64 lanes]
  float* sc_lds = reinterpret_cast<float*>(tail + MG * kGemvWaves * 64 * 16);   // [8 row-tiles][16]
  float* bi_lds = sc_lds + kGemvWaves * 16;
  float* ssq_lds = bi_lds + kGemvWaves * 16;                           // [16 MG rows][16 waves]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int tsub = wave / KS, kslice = wave % KS;
  const bool wave_on = tsub < TP;                                      // waves beyond KS * TP only stage activations
  bool col_ok[MG];                                                     // column c of group q = activation row 16 q + c
#pragma unroll
  for (int q = 0; q < MG; ++q) col_ok[q] = 16 * q + c < M;
  const int NC = ceil_div(KT, CKT);
  const int ktw = ceil_div(CKT, KS);                  // k-tiles of a chunk per wave
  const int nb = ceil_div(ktw, kGemvU);               // batches per (pass, chunk), same for every wave
  const int units = ceil_div(NT, TP);                 // a pass = TP row-tiles
  const int my_units = (int)blockIdx.x < units ? ceil_div(units - (int)blockIdx.x, (int)gridDim.x) : 0;
  const int total = my_units * NC * nb;

  // sequence index -> (pass, chunk, batch); the slice of K-slice ks inside chunk ch is
  // [ch * CKT + ks * ktw, ... + ktw) clipped to the chunk and to KT
  auto slice = [&](int ch, int& kbeg, int& kend) {
    const int cend = min((ch + 1) * CKT, KT);
    kbeg = min(ch * CKT + kslice * ktw, cend);
    kend = min(kbeg + ktw, cend);
  };
  u32x4_t bufA[kGemvU], bufB[kGemvU];
  auto issue = [&](u32x4_t (&buf)[kGemvU], int i) {
    const bool live = i < total && wave_on;
    i = min(i, max(total - 1, 0));
    const int it = i / (NC * nb), rem = i - it * NC * nb, ch = rem / nb, b = rem - ch * nb;
    int kbeg, kend;
    slice(ch, kbeg, kend);
    const int tile_raw = ((int)blockIdx.x + it * (int)gridDim.x) * TP + tsub;
    const int tile = min(tile_raw, NT - 1);
    const int klast = max(kend - 1, kbeg);
    const bool any = live && kbeg < kend && tile_raw < NT;
    const uint4* base = any ? W + (size_t)tile * KT * 64 + lane : W;
    const size_t kstep = any ? 64 : 0;
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) stream_load16(buf[u], base + (size_t)min(kbeg + b * kGemvU + u, min(klast, KT - 1)) * kstep);
  };

  if (tid == 0) xf[zero_slot] = make_uint4(0, 0, 0, 0);
  if constexpr (MG > 1 && PRO != PRO_BF16) {   // (row-wise staging fills one wave's slot per row; the first staging barrier orders this)
    for (int i = tid; i < MG * 16 * 16; i += blockDim.x) ssq_lds[i] = 0.f;
  }
  issue(bufA, 0);
  issue(bufB, 1);

  f32x4_t acc[MG];
#pragma unroll
  for (int q = 0; q < MG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto process = [&](u32x4_t (&buf)[kGemvU], int i) {
    const int it = i / (NC * nb), rem = i - it * NC * nb, ch = rem / nb, b = rem - ch * nb;
    int kbeg, kend;
    slice(ch, kbeg, kend);
    const int tile0 = ((int)blockIdx.x + it * (int)gridDim.x) * TP;    // first row-tile of the pass
    if (b == 0) {   // a new chunk: every wave is done with the previous image
      __syncthreads();
      if (ch == 0 && tid < 16 * TP) {
        const int tile = min(tile0 + (tid >> 4), NT - 1);
        sc_lds[tid] = e.scale[tile * 16 + (tid & 15)];
        bi_lds[tid] = e.bias ? e.bias[tile * 16 + (tid & 15)] : 0.f;
      }
      const int k0 = ch * CKT * TK, ck = (min((ch + 1) * CKT, KT) - ch * CKT) * TK;
      if constexpr (MG > 1 && PRO != PRO_BF16) {
        if (ck <= 64 * kRowJ * 8) gemv_stage_chunk_rows<WD, PRO>(p, M, K, k0, ck, xf, ssq_lds, ch == 0, it == 0, it == 0 && blockIdx.x == 0);
        else gemv_stage_chunk<WD, PRO>(p, M, K, k0, ck, xf, ssq_lds, ch == 0, it == 0, it == 0 && blockIdx.x == 0);
      } else {
        gemv_stage_chunk<WD, PRO>(p, M, K, k0, ck, xf, ssq_lds, ch == 0, it == 0, it == 0 && blockIdx.x == 0);
      }
      __syncthreads();
    }
    const int kl0 = kbeg - ch * CKT + b * kGemvU;     // k-tile index inside the chunk image
#pragma unroll
    for (int u = 0; u < kGemvU; ++u) {
      const int kt = kl0 + u;
      const bool live = (ch * CKT + kt) < kend;
      const u32x4_t w = buf[u];
      if constexpr (WD == MI_W_BF16) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w);
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, xf[live && col_ok[q] ? (kt * 4 + g) * M + 16 * q + c : zero_slot]);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[q], 0, 0, 0);
        }
      } else {
        const bf16x8_t a0 = decode8<WD>(w[0], w[1]), a1 = decode8<WD>(w[2], w[3]);   // decoded once for all column groups
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          const bool ok = live && col_ok[q];
          const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 0) * 4 + g) * M + 16 * q + c : zero_slot]);
          const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, xf[ok ? ((kt * 2 + 1) * 4 + g) * M + 16 * q + c : zero_slot]);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[q], 0, 0, 0);
        }
      }
    }
    if (ch == NC - 1 && b == nb - 1) {   // the pass's row-tiles are done: combine the K-slices in slice order
      if constexpr (KS > 1) {
#pragma unroll
        for (int q = 0; q < MG; ++q) red[(q * kGemvWaves + wave) * 64 + lane] = acc[q];
        __syncthreads();                 // (the next write to `red` comes behind the staging barriers of the next pass)
      }
      const int tile = tile0 + tsub;
      if (kslice == 0 && wave_on && tile < NT) {   // the first wave of a row-tile finishes all its column groups
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          if (!col_ok[q]) continue;
          f32x4_t sum = acc[q];
          if constexpr (KS > 1) {
            sum = red[(q * kGemvWaves + wave) * 64 + lane];
#pragma unroll
            for (int w = 1; w < KS; ++w) {
              const f32x4_t t = red[(q * kGemvWaves + wave + w) * 64 + lane];
              sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
            }
          }
          float rmul = 1.f;
          if constexpr (PRO != PRO_BF16) rmul = gemv_row_rinv(ssq_lds, 16 * q + c, K, p.eps);
          epilogue_lds<EPI>(e, 16 * q + c, tile * 16 + g * 4, sc_lds + tsub * 16 + g * 4, e.bias ? bi_lds + tsub * 16 + g * 4 : nullptr,
                            sum, rmul);
        }
      }
#pragma unroll
      for (int q = 0; q < MG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  };
  for (int i = 0; i < total; i += 2) {
    process(bufA, i);
    issue(bufA, i + 2);
    if (i + 1 < total) process(bufB, i + 1);
    issue(bufB, i + 3);
  }
}

// =====================================================================================
// GEMV, activations streamed beside the weights (bf16 activations that do not fit the LDS whole)
// =====================================================================================
// gemv_bigk_kernel stages a K-chunk of the activations for all waves, meets, multiplies, meets again: the
// weight stream stalls at every chunk (Qwen2.5-7B down_proj, K = 18944 at 4 rows: 22.8 us for 68 MB;
// Llama-8B down_proj at 16 rows, the target's pass of a speculation step: 35 us for 59 MB).  The K
// dimension of a row-tile is split over the 8 waves as everywhere, so a wave only ever multiplies ITS
// K-slice: here it also stages that slice for itself, a sub-chunk of U k-tiles at a time, by LDS-DMA
// (global_load_lds: no registers, no arithmetic -- hence bf16 activations only) into a double buffer of
// its own, two sub-chunks ahead together with the weight batch of the same k-tiles.  No work-group
// barrier before the K-slice reduction of a row-tile.  The lane -> source-address map of the DMA does
// the transpose into the MFMA B-fragment image (xfrag_slot: [k-chunk][row] 16-byte slots); a sub-chunk is always
// 4 or 8 DMA instructions (U = 8 / 8 / 4 k-tiles for up to 4 / 8 / 16 rows), so the waits are
// counted at compile time.  Same K-slices and MFMA order as gemv_bigk_kernel's chunks of one wave:
// the products differ from it only in the association of the per-wave partial sums (none: each wave
// still adds its k-tiles in order, the waves are combined in wave order).
constexpr int kKsBuf = 8 * 1024;                   // bytes of one wave-private activation buffer (<= 8 DMA instructions)
// reduction buffer: [2 parities][MG column groups][waves 1..7][64 lanes] (wave 0 finishes every group from its own registers)
static size_t gemv_kstream_lds(int mg) { return (size_t)kGemvWaves * 2 * kKsBuf + (size_t)2 * mg * (kGemvWaves - 1) * 64 * 16 + 64 * 4 + 16; }

template <int WD, int EPI, int U, int kKsDma, int MG = 1>   // U k-tiles and kKsDma DMA instructions (64 slots each) per sub-chunk; MG 16-column groups
__global__ __launch_bounds__(kGemvWaves * 64) void gemv_kstream_kernel(const uint4* __restrict__ W, int NT, int KT, int M,
                                                                       int K, int G, ProArgs p, EpiArgs e) {
  constexpr int TK = (WD == MI_W_BF16) ? 32 : 64;
  constexpr int QPT = TK / 8;                      // 16-byte activation slots per k-tile and row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  bool col_ok[MG];                                                       // column c of group q = activation row 16 q + c
#pragma unroll
  for (int q = 0; q < MG; ++q) col_ok[q] = 16 * q + c < M;
  unsigned char* xb = smem + (size_t)wave * 2 * kKsBuf;
  unsigned char* tail = smem + (size_t)kGemvWaves * 2 * kKsBuf;
  constexpr int RW = kGemvWaves - 1;                                     // waves that hand their partial sums over
  f32x4_t* red = reinterpret_cast<f32x4_t*>(tail);                       // [2 parities][MG][waves 1..7][64 lanes]
  float* sc_lds = reinterpret_cast<float*>(tail + 2 * MG * RW * 64 * 16);   // [2 parities][scale 16 | bias 16]
  uint4* zero = reinterpret_cast<uint4*>(sc_lds + 64);
  pin_sgpr(W); pin_sgpr(NT); pin_sgpr(KT); pin_sgpr(M); pin_sgpr(K); pin_sgpr(G); pin_sgpr(p.x); pin_sgpr(p.ldx);
  pin_sgpr(e.scale); pin_sgpr(e.bias); pin_sgpr(e.out_f32); pin_sgpr(e.ld_out);
  if constexpr (EPI == EPI_RESID) pin_sgpr(e.resid_in);

  const int ktw = ceil_div(KT, kGemvWaves);            // k-tiles per wave
  const int kbeg = min(wave * ktw, KT), kend = min(kbeg + ktw, KT);
  const int nsc = ceil_div(ktw, U);                    // sub-chunks per row-tile, the same for every wave
  const int my_tiles = (int)blockIdx.x < NT ? ceil_div(NT - (int)blockIdx.x, G) : 0;
  const int total = my_tiles * nsc;
  const int slots = U * QPT * M;                       // live 16-byte slots of a sub-chunk (<= 64 kKsDma)

  // lane -> (k offset, row) of its slot in each of the 8 DMA instructions of a sub-chunk (fixed)
  int src_off[kKsDma], src_k[kKsDma];                  // element offset inside the sub-chunk's [rows][U * TK] window (-1: no slot), its k part
  constexpr int CPR = U * QPT;                         // 16-byte chunks of a row in a sub-chunk
#pragma unroll
  for (int d = 0; d < kKsDma; ++d) {
    const int sl = d * 64 + lane;
    if constexpr (MG == 1) {
      const int q = sl / M, r = sl - q * M;
      // slot q of the image holds the 8-element chunk xfrag_slot maps there: inside a 64-wide k-tile the
      // chunks are interleaved (slot (h, g) <- chunk 2 g + h) to match the byte order of the 1-byte weight tiles
      int ch = q;
      if constexpr (WD != MI_W_BF16) ch = (q & ~7) | ((q & 3) << 1) | ((q >> 2) & 1);
      src_off[d] = sl < slots ? r * p.ldx + ch * 8 : -1;
      src_k[d] = ch * 8;
    } else {
      // 17 .. 32 rows: the [chunk][row] image above makes a DMA instruction fetch 64 pieces of 16 bytes from up to 64 different
      // rows (64 cache lines for 1 KiB).  Row-major image instead: 16 consecutive lanes bring one row's 256 bytes (CPR = 16
      // chunks), the chunks XOR-swizzled by the row so that the fragment reads (16 rows, one chunk position) are
      // conflict-free; the swizzle is applied to the SOURCE address, the LDS write stays lane-linear.
      static_assert(CPR == 16, "row-major image: 16 chunks per row and sub-chunk");
      const int r = sl / CPR, sslot = sl % CPR;
      const int ch = sslot ^ (r & 15);                 // natural 8-element chunk of the sub-chunk's window held by this slot
      src_off[d] = r < M ? r * p.ldx + ch * 8 : -1;
      src_k[d] = ch * 8;
    }
  }
  // fragment reads of the row-major image (MG == 2): 16-byte slot of (row, natural chunk)
  auto rm_slot = [&](int row, int ch) { return row * CPR + (ch ^ (row & 15)); };
  auto issue_x = [&](int sel, int i) {
    const bool live = i < total;
    const int j = live ? i % nsc : 0;
    const int k0 = (kbeg + j * U) * TK;                // first element of the sub-chunk
    const int klim = kend * TK;
#pragma unroll
    for (int d = 0; d < kKsDma; ++d) {
      // a slot past the wave's slice (or no slot at all) fetches the first line of x: never multiplied
      const int off = src_off[d];
      const bool ok = live && off >= 0 && k0 + src_k[d] < klim;
      glds16(p.x + (ok ? (size_t)k0 + off : 0), xb + sel * kKsBuf + d * 1024);
    }
  };
  u32x4_t bufA[U], bufB[U];
  auto issue_w = [&](u32x4_t (&buf)[U], int i) {
    const bool live = i < total && kbeg < kend;
    const int ii = min(i, max(total - 1, 0));
    const int it = ii / nsc, j = ii - it * nsc;
    const int tile = min((int)blockIdx.x + it * G, NT - 1);
    const int klast = max(kend - 1, kbeg);
    const uint4* base = live ? W + (size_t)tile * KT * 64 + lane : W;
    const size_t kstep = live ? 64 : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) stream_load16(buf[u], base + (size_t)min(kbeg + j * U + u, min(klast, KT - 1)) * kstep);
  };

  if (tid == 0) *zero = make_uint4(0, 0, 0, 0);
  issue_x(0, 0);
  issue_w(bufA, 0);
  issue_x(1, 1);
  issue_w(bufB, 1);
  float scv = 0.f, biv = 0.f;                          // scale / bias of the row-tile in flight (threads 0..15)
  __syncthreads();                                     // the zero slot

  f32x4_t acc[MG];
#pragma unroll
  for (int q = 0; q < MG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int parity = 0;
  auto process = [&](u32x4_t (&buf)[U], int sel, int i) {
    const int it = i / nsc, j = i - it * nsc;
    const int tile = (int)blockIdx.x + it * G;
    if (j == 0 && tid < 16) {
      scv = e.scale[tile * 16 + tid];
      biv = e.bias ? e.bias[tile * 16 + tid] : 0.f;
    }
    // everything but the DMAs and weight loads of the NEXT sub-chunk has landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKsDma + U) : "memory");
    const uint4* xf = reinterpret_cast<const uint4*>(xb + sel * kKsBuf);
    const int kt0 = kbeg + j * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = kt0 + u < kend;
      const u32x4_t w = buf[u];
      if constexpr (WD == MI_W_BF16) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w);
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          const int at = MG == 1 ? (u * 4 + g) * M + 16 * q + c : rm_slot(16 * q + c, u * 4 + g);
          const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, live && col_ok[q] ? xf[at] : *zero);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[q], 0, 0, 0);
        }
      } else {
        const bf16x8_t a0 = decode8<WD>(w[0], w[1]), a1 = decode8<WD>(w[2], w[3]);   // decoded once for all column groups
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          const bool ok = live && col_ok[q];
          // k-tile u, MFMA h: lane group g multiplies the elements 16 g + 8 h .. of the k-tile = natural chunk 2 g + h
          const int at0 = MG == 1 ? ((u * 2 + 0) * 4 + g) * M + 16 * q + c : rm_slot(16 * q + c, u * 8 + 2 * g);
          const int at1 = MG == 1 ? ((u * 2 + 1) * 4 + g) * M + 16 * q + c : rm_slot(16 * q + c, u * 8 + 2 * g + 1);
          const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, ok ? xf[at0] : *zero);
          const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, ok ? xf[at1] : *zero);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[q], 0, 0, 0);
        }
      }
    }
    if (j == nsc - 1) {   // row-tile done: combine the K-slices in wave order (wave 0's own partial first, then waves 1..7)
      if (wave > 0) {
#pragma unroll
        for (int q = 0; q < MG; ++q) red[((parity * MG + q) * RW + wave - 1) * 64 + lane] = acc[q];
      }
      if (tid < 16) { sc_lds[parity * 32 + tid] = scv; sc_lds[parity * 32 + 16 + tid] = biv; }
      __syncthreads();
      if (wave == 0 && tile < NT) {
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          if (!col_ok[q]) continue;
          f32x4_t sacc = acc[q];
#pragma unroll
          for (int w2 = 0; w2 < RW; ++w2) {
            const f32x4_t t = red[((parity * MG + q) * RW + w2) * 64 + lane];
            sacc[0] += t[0]; sacc[1] += t[1]; sacc[2] += t[2]; sacc[3] += t[3];
          }
          epilogue_lds<EPI>(e, 16 * q + c, tile * 16 + g * 4, sc_lds + parity * 32 + g * 4,
                            e.bias ? sc_lds + parity * 32 + 16 + g * 4 : nullptr, sacc, 1.f);
        }
      }
      parity ^= 1;
#pragma unroll
      for (int q = 0; q < MG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // the buffer is re-filled next: its fragment reads must have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i < total; i += 2) {
    process(bufA, 0, i);
    issue_x(0, i + 2);
    issue_w(bufA, i + 2);
    if (i + 1 < total) {
      process(bufB, 1, i + 1);
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKsDma + U) : "memory");   // keep the DMA / wait pairing of the loop
    }
    issue_x(1, i + 3);
    issue_w(bufB, i + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may land after the work-group has gone
}

template <int WD, int EPI, int U, int ND, int MG = 1>
static int launch_gemv_kstream_u(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s) {
  const int NT = w.N / 16, KT = w.K / tile_k(WD);
  MI_CHECK(U * (tile_k(WD) / 8) * M <= ND * 64 && M <= 16 * MG, "gemv_kstream: sub-chunk larger than its DMA budget");
  auto kern = gemv_kstream_kernel<WD, EPI, U, ND, MG>;
  MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
  int num_cu = 0;
  MI_TRY_(device_num_cu(&num_cu));
  int grid = NT;
  if (NT > num_cu) grid = ceil_div(NT, ceil_div(NT, num_cu));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemvWaves * 64), gemv_kstream_lds(MG), s, reinterpret_cast<const uint4*>(w.w), NT,
                     KT, M, w.K, grid, p, e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
template <int WD, int EPI>
static int launch_gemv_kstream(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s) {
  MI_CHECK(p.ldx % 8 == 0, "gemv: activation row stride must be a multiple of 8 elements");
  if (M <= 4) return launch_gemv_kstream_u<WD, EPI, 8, 4>(w, M, p, e, s);
  if (M <= 8) return launch_gemv_kstream_u<WD, EPI, 8, 8>(w, M, p, e, s);
  if (M <= 16) return launch_gemv_kstream_u<WD, EPI, 4, 8>(w, M, p, e, s);
  // 17 .. 32 rows: two column groups over one weight stream; a sub-chunk is 2 k-tiles x 32 rows (8 KiB of activations)
  if constexpr (WD == MI_W_BF16) return launch_gemv_kstream_u<WD, EPI, 4, 8, 2>(w, M, p, e, s);
  else return launch_gemv_kstream_u<WD, EPI, 2, 8, 2>(w, M, p, e, s);
}

template <int WD, int PRO, int EPI, int MG, int KS>
static int launch_gemv_bigk_ks(const LinearW& w, int M, int TP, const ProArgs& p, const EpiArgs& e, hipStream_t s, int num_cu) {
  const int NT = w.N / 16, KT = w.K / tile_k(WD);
  const int CKT = gemv_bigk_chunk_kt(M, w.K, WD);
  const size_t lds = (size_t)M * CKT * tile_k(WD) * 2 + gemv_bigk_fixed_lds(MG);
  MI_CHECK(CKT >= kGemvWaves && lds <= 160 * 1024, "gemv: activation chunk does not fit in LDS");
  auto kern = gemv_bigk_kernel<WD, PRO, EPI, MG, KS>;
  MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024));
  // one pass (TP row-tiles) per work-group at a time; more passes than CUs -> every work-group walks an equal share
  const int units = ceil_div(NT, TP);
  int grid = units;
  if (units > num_cu) grid = ceil_div(units, ceil_div(units, num_cu));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemvWaves * 64), lds, s, reinterpret_cast<const uint4*>(w.w), NT, KT, M, w.K,
                     CKT, TP, p, e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
template <int WD, int PRO, int EPI>
static int launch_gemv_bigk_t(const LinearW& w, int M, const ProArgs& p, const EpiArgs& e, hipStream_t s) {
  int num_cu = 0;
  MI_TRY_(device_num_cu(&num_cu));
  // up to 16 rows: the K dimension of one row-tile over all 8 waves (the summation order the whole-image kernels' tests pin)
  if (M <= 16) return launch_gemv_bigk_ks<WD, PRO, EPI, 1, kGemvWaves>(w, M, 1, p, e, s, num_cu);
  // 17 .. 32 rows: as many row-tiles side by side as keep every wave streaming (gate|up of Llama-8B: 7 per work-group, one pass)
  const GemvShape sh = gemv_pick_shape(w.N / 16, w.K / tile_k(WD), num_cu);
  switch (sh.ks) {
    case 1: return launch_gemv_bigk_ks<WD, PRO, EPI, 2, 1>(w, M, sh.tp, p, e, s, num_cu);
    case 2: return launch_gemv_bigk_ks<WD, PRO, EPI, 2, 2>(w, M, sh.tp, p, e, s, num_cu);
    case 4: return launch_gemv_bigk_ks<WD, PRO, EPI, 2, 4>(w, M, sh.tp, p, e, s, num_cu);
    default: return launch_gemv_bigk_ks<WD, PRO, EPI, 2, 8>(w, M, sh.tp, p, e, s, num_cu);
  }
}

// valid (prologue, epilogue) pairs: bf16 activations feed the row-parallel projections
// (fp32 / residual out); the norm prologues feed QKV, gate|up and lm_head
template <int WD>
static int launch_gemv_wd(const LinearW& w, int M, int pro, const ProArgs& p, int epi, const EpiArgs& e, hipStream_t s) {
  if (pro == PRO_BF16) {
    if (epi == EPI_RESID) return launch_gemv_t<WD, PRO_BF16, EPI_RESID>(w, M, p, e, s);
    if (epi == EPI_F32) return launch_gemv_t<WD, PRO_BF16, EPI_F32>(w, M, p, e, s);
  } else if (p.partial) {
    if (epi == EPI_QKV) return launch_gemv_t<WD, PRO_NORM_PARTIAL, EPI_QKV>(w, M, p, e, s);
    if (epi == EPI_SWIGLU) return launch_gemv_t<WD, PRO_NORM_PARTIAL, EPI_SWIGLU>(w, M, p, e, s);
    if (epi == EPI_F32) return launch_gemv_t<WD, PRO_NORM_PARTIAL, EPI_F32>(w, M, p, e, s);
  } else {
    if (epi == EPI_QKV) return launch_gemv_t<WD, PRO_NORM, EPI_QKV>(w, M, p, e, s);
    if (epi == EPI_SWIGLU) return launch_gemv_t<WD, PRO_NORM, EPI_SWIGLU>(w, M, p, e, s);
    if (epi == EPI_F32) return launch_gemv_t<WD, PRO_NORM, EPI_F32>(w, M, p, e, s);
  }
  set_error("gemv: unsupported prologue / epilogue combination");
  return MI_EINVAL;
}

int launch_gemv(const LinearW& w, int M, int pro, const ProArgs& p, int epi, const EpiArgs& e, hipStream_t s) {
  MI_CHECK(M >= 1 && M <= 32, "gemv: M must be 1..32");
  MI_CHECK(w.N % 16 == 0 && w.K % 64 == 0, "gemv: N % 16 == 0 and K % 64 == 0 required");
  switch (w.wd) {
    case MI_W_BF16: return launch_gemv_wd<MI_W_BF16>(w, M, pro, p, epi, e, s);
    case MI_W_F8E4M3: return launch_gemv_wd<MI_W_F8E4M3>(w, M, pro, p, epi, e, s);
    case MI_W_INT8: return launch_gemv_wd<MI_W_INT8>(w, M, pro, p, epi, e, s);
  }
  set_error("gemv: bad weight dtype");
  return MI_EINVAL;
}

// =====================================================================================
// GEMM (context encoding)
// =====================================================================================
// Work-group tile 128 (n) x 128 (m); 4 waves as 2 (n) x 2 (m), each 64 x 64 = 4 x 4 MFMA tiles.
// x tile [128 m][64 k] bf16 through LDS (16-byte chunks XOR-swizzled by row); W fragments
// straight from the pre-tiled global image.
constexpr int kBM = 128, kBN = 128, kBK = 64;

// XOR swizzle of the 16-byte chunks of a 128-byte activation row in LDS.  A ds_read_b128 is
// serviced in four groups of 16 lanes ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS); with the
// fragment addressing of the MFMA B operand (lane (g, c): row c, chunk 2g / 2g+1 for 1-byte weights,
// g / 4+g for bf16 weights) the plain `chunk ^ (row & 7)` image is conflict-free for the bf16-weight
// chunk order but puts every group of the 1-byte order two-deep on its banks; folding row bit 3 into
// the key makes that order conflict-free too (enumerated over the lane groups).
template <int WD>
__device__ __forceinline__ int xs_swz(int r) { return WD == MI_W_BF16 ? (r & 7) : ((r & 7) ^ ((r >> 3) & 1)); }

// SPLIT: grid.z K-slices write raw fp32 accumulators to slab[z][m][n] (see gemm_a8_kernel).
// Registers are kept under 170 per lane so that three work-groups share a CU (the same lesson as
// the FP8 GEMM): the weight tiles of a K-step are decoded once into their A fragments, the raw
// registers are refilled with the next K-step's tiles straight away, and the activation
// fragments are read per 16-token tile instead of all at once.
template <int WD, int EPI, bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WD == MI_W_BF16 ? 2 : 3, WD == MI_W_BF16 ? 2 : 3))) void gemm_kernel(const uint4* __restrict__ W, int NT, int KT, int T, int K,
                                                   const uint16_t* __restrict__ x, int ldx, EpiArgs e,
                                                   float* __restrict__ slab) {
  __shared__ __attribute__((aligned(16))) uint4 xs[kBM * 8];  // 16 KiB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int wn = wave >> 1, wm = wave & 1;
  // XCD-aware tile order (see gemm_a8_kernel): XCD = linear id % 8 owns every eighth weight slab
  // and walks all token blocks of a slab back to back
  const int mtiles = ceil_div(T, kBM), ntiles = ceil_div(NT, kBN / 16);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mblk = seq % mtiles, nblk = (seq / mtiles) * 8 + xcd;
  if (nblk >= ntiles) return;
  const int m0 = mblk * kBM, nt0 = nblk * (kBN / 16) + wn * 4;
  const int nks_all = K / kBK;
  const int ks_per = SPLIT ? ceil_div(nks_all, (int)gridDim.z) : nks_all;
  const int ks_beg = SPLIT ? (int)blockIdx.z * ks_per : 0;
  const int nks = min(ks_beg + ks_per, nks_all);
  constexpr int WPK = (WD == MI_W_BF16) ? 2 : 1;  // weight tiles per K-step

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // x staging: thread -> rows (tid/8 + 32 i), 16-byte chunk tid%8
  const int srow = tid >> 3, sch = tid & 7;
  u32x4_t xr[4], wr[4][WPK];
  auto load_x = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + srow + 32 * i, T - 1);   // rows past T: a copy of the last row, never stored
      xr[i] = *reinterpret_cast<const u32x4_t*>(x + (size_t)m * ldx + ks * kBK + sch * 8);
    }
  };
  auto load_w = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < WPK; ++t)
        wr[i][t] = *reinterpret_cast<const u32x4_t*>(W + ((size_t)min(nt0 + i, NT - 1) * KT + ks * WPK + t) * 64 + lane);
  };

  if (ks_beg < nks) {
    load_x(ks_beg);
    load_w(ks_beg);
  }
  for (int ks = ks_beg; ks < nks; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 32 * i;
      *reinterpret_cast<u32x4_t*>(&xs[r * 8 + (sch ^ xs_swz<WD>(r))]) = xr[i];
    }
    bf16x8_t a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (WD == MI_W_BF16) {
        a0[i] = __builtin_bit_cast(bf16x8_t, wr[i][0]);
        a1[i] = __builtin_bit_cast(bf16x8_t, wr[i][1]);
      } else {
        const u32x4_t w4 = wr[i][0];
        a0[i] = decode8<WD>(w4[0], w4[1]);
        a1[i] = decode8<WD>(w4[2], w4[3]);
      }
    }
    __syncthreads();
    if (ks + 1 < nks) {      // the raw registers are free again: next K-step's operands
      load_x(ks + 1);
      load_w(ks + 1);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int r = wm * 64 + mt * 16 + c;
      const int ch0 = (WD == MI_W_BF16) ? g : 2 * g, ch1 = (WD == MI_W_BF16) ? 4 + g : 2 * g + 1;
      const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, xs[r * 8 + (ch0 ^ xs_swz<WD>(r))]);
      const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, xs[r * 8 + (ch1 ^ xs_swz<WD>(r))]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0, acc[i][mt], 0, 0, 0);
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1, acc[i][mt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (nt0 + i >= NT) continue;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + c;
      if (m >= T) continue;
      const int n0 = (nt0 + i) * 16 + g * 4;
      if constexpr (SPLIT) {
        *reinterpret_cast<float4*>(slab + ((size_t)blockIdx.z * T + m) * (NT * 16) + n0) =
            make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
      } else {
        epilogue<EPI>(e, m, n0, acc[i][mt]);
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int KS, int T, int N, EpiArgs e);

// K-split for grids too small to occupy the chip (768 work-groups are resident at once): below 256
// tiles aim for 512 work-groups, below 384 for 768 (rule measured on the FP8 GEMM).
static int gemm_pick_splitk(int tiles, int nks, int T, int N, size_t ws_bytes) {
  if (!ws_bytes || tiles >= 384) return 1;
  int KS = min(min(8, nks), ceil_div(tiles < 256 ? 512 : 768, tiles));
  while (KS > 1 && (size_t)KS * T * N * sizeof(float) > ws_bytes) --KS;
  return KS;
}

template <int WD>
static int launch_gemm_wd(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                          float* splitk_ws, size_t splitk_ws_bytes, SlabSum* defer) {
  const int NT = w.N / 16, KT = w.K / tile_k(WD);
  const int mtiles = ceil_div(T, kBM), ntiles = ceil_div(w.N, kBN);
  const int KS = gemm_pick_splitk(mtiles * ntiles, w.K / kBK, T, w.N, splitk_ws ? splitk_ws_bytes : 0);
  dim3 grid(8 * mtiles * ceil_div(ntiles, 8), 1, KS);
  const uint4* W = reinterpret_cast<const uint4*>(w.w);
#define MI_G(EPI_) \
  do { \
    if (KS == 1) { \
      hipLaunchKernelGGL((gemm_kernel<WD, EPI_, false>), grid, dim3(256), 0, s, W, NT, KT, T, w.K, x, ldx, e, nullptr); \
    } else { \
      hipLaunchKernelGGL((gemm_kernel<WD, EPI_, true>), grid, dim3(256), 0, s, W, NT, KT, T, w.K, x, ldx, e, splitk_ws); \
      if (defer && EPI_ == EPI_RESID) { *defer = SlabSum{splitk_ws, KS, T, w.N, e.scale, e.bias, e.row_scale}; } \
      else hipLaunchKernelGGL((splitk_reduce_kernel<EPI_>), dim3(ceil_div(T * (w.N / 4), 256)), dim3(256), 0, s, splitk_ws, KS, T, w.N, e); \
    } \
  } while (0)
  if (epi == EPI_QKV) MI_G(EPI_QKV);
  else if (epi == EPI_SWIGLU) MI_G(EPI_SWIGLU);
  else if (epi == EPI_RESID) MI_G(EPI_RESID);
  else MI_G(EPI_F32);
#undef MI_G
  MI_HIP(hipGetLastError());
  return MI_OK;
}


// =====================================================================================
// GEMM, wide-N tile with an LDS-DMA ring and two wave groups half a K-step apart (context encoding, 1-byte weights)
// =====================================================================================
// The 128 x 128 kernel above is L1-fill bound: per 64-deep K-step a work-group pulls
// 16 KiB of bf16 activations + 8 KiB of fp8 weights for 2.1 MFLOP.  Weights are half the bytes per
// element of the activations, so the tile grows along N: 128 tokens x 256 weight rows = 16 + 16 KiB
// for 4.2 MFLOP (1.5 x the flops per byte), 256 x 256 = 32 + 16 KiB for 8.4.  Eight waves as 2 (tokens) x 4 (weight
// rows): activation fragments are shared by four waves, weight fragments by two.  Both operands arrive by LDS-DMA
// (global_load_lds, 1 KiB per wave-instruction: the weight tiles land lane-linear = fragment
// order; the activation rows land in the XOR-swizzled image of the kernel above, the swizzle being
// applied to the per-lane SOURCE address) into a ring of three stages: two K-steps are in
// flight while one is multiplied, retired by counted vmcnt waits in front of raw barriers -- no ordinary global
// load in the loop, so hipcc has nothing to drain.  Round 3: a wave's K-step is a READ phase (fragments LDS -> registers,
// weight codes decoded on the way) and a MULTIPLY phase (MFMAs only), and the two waves of a SIMD are in opposite phases
// (see "Schedule" in the kernel): gate|up at 2048 tokens 436.6 -> 373.4 us (1.10 -> 1.29 PF/s), down 220 -> 198 us.
constexpr int kWideBN = 256, kWideWBytes = 16 * 1024, kWideStages = 3;


// BM = 128: each wave 64 x 64 (64 accumulator registers).  BM = 256: each 128 x 64 (128 accumulator registers, 246 VGPRs
// with the decoded fragments of one K-step).
// SPLIT: grid.z K-slices write raw fp32 accumulators to slab[z][m][n] (see gemm_a8_kernel); K / 64 is a multiple of grid.z.
template <int WD, int EPI, int BM, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const uint4* __restrict__ W, int NT, int KT, int T, int K,
                                                        const uint16_t* __restrict__ x, int ldx, EpiArgs e,
                                                        float* __restrict__ slab) {
  static_assert(WD != MI_W_BF16, "1-byte weight tiles (64 k per tile)");
  static_assert(BM == 128 || BM == 256, "token block");
  // waves as 2 (tokens) x 4 (weight rows), each MT 16-token tiles x 4 16-row weight tiles.  (4 x 2 waves of 64 x 128 read
  // 16 instead of 20 KiB of fragments per K-step but decode every weight code four times instead of twice: measured slower,
  // 401 against 387 us on the gate-up shape -- the decode, not the LDS read, is what the schedule has to hide.)
  constexpr int MT = BM / 32;                 // 16-token tiles per wave
  constexpr int WM = 2, NTW = 4;
  constexpr int XP = BM / 64;                 // 8-row activation pieces per wave and stage
  constexpr int kStage = kWideWBytes + BM * 128;
  constexpr int kDma = 2 + XP;                // DMAs per wave and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int wm = wave % WM, wn = wave / WM;
  // XCD-aware tile order (see gemm_a8_kernel): an XCD walks all token blocks of a weight slab back to back
  const int mtiles = ceil_div(T, BM), ntiles = ceil_div(NT, kWideBN / 16);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mblk = seq % mtiles, nblk = (seq / mtiles) * 8 + xcd;
  if (nblk >= ntiles) return;
  const int m0 = mblk * BM, ntb = nblk * (kWideBN / 16);
  const int nks = SPLIT ? K / 64 / (int)gridDim.z : K / 64;     // this slice's K-steps, from K-step ks0 on
  const int ks0 = SPLIT ? (int)blockIdx.z * nks : 0;

  // this lane's DMA sources: wave w brings weight tiles 2w, 2w + 1 of the work-group's 16 (fragment
  // order) and XP 8-row pieces of the activation tile
  const uint4* wsrc[2];
  const uint16_t* xsrc[XP];
#pragma unroll
  for (int i = 0; i < 2; ++i) wsrc[i] = W + ((size_t)min(ntb + 2 * wave + i, NT - 1) * KT + ks0) * 64 + lane;
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int r = (XP * wave + i) * 8 + (lane >> 3), sl = lane & 7;
    xsrc[i] = x + (size_t)min(m0 + r, T - 1) * ldx + (size_t)ks0 * 64 + ((sl ^ xs_swz<WD>(r)) * 8);   // LDS slot sl of row r holds chunk sl ^ swz(r)
  }
  auto issue = [&](int ks) {
    unsigned char* st = smem + (ks % kWideStages) * kStage;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(wsrc[i] + (size_t)ks * 64, st + (2 * wave + i) * 1024);
#pragma unroll
    for (int i = 0; i < XP; ++i) glds16(xsrc[i] + (size_t)ks * 64, st + kWideWBytes + (XP * wave + i) * 1024);
  };

  f32x4_t acc[NTW][MT];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Schedule.  A K-step of a wave is two phases: READ (its fragments, 20 KiB at BM = 256, from LDS into registers, the weight codes decoded to bf16 on the way) and MULTIPLY
  // (2 NTW MT MFMAs from registers).  With one barrier per K-step every wave is in the same phase at the same time, the two waves
  // of a SIMD included, and LDS-read time and MFMA time add up: 1.7 us per K-step at BM = 256 against 1.0 us of MFMAs.  Here
  // the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run half a K-step apart -- group 1 executes one
  // barrier more at the start -- so one group multiplies while the other reads:
  //     physical barrier:      P0        P1        P2        P3        P4  ...
  //     group 0 (waves 0-3):   | READ 0  | MUL 0   | READ 1  | MUL 1   | ...
  //     group 1 (waves 4-7):   |         | READ 0  | MUL 0   | READ 1  | MUL 1 ...
  // Stage k is read in [P2k, P2k+2]; K-step k + 2 goes into the stage of K-step k - 1 (read until P2k), requested by every wave
  // behind its top barrier of iteration k: two K-steps ahead of its first read, as before.  A wave waits for ITS pieces of a
  // stage before the barrier in front of the stage's first read by anyone (group 0: its top barrier; group 1: its second one).
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  issue(0);
  if (nks > 1) issue(1);
  if (grp == 1) {   // its pieces of stage 0 are in before P0
    if (nks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // P0
  }
  for (int ks = 0; ks < nks; ++ks) {
    if (grp == 0) {   // own pieces of stage ks (at most those of stage ks + 1 are younger)
      if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();           // group 0: P2ks | group 1: P2ks+1 -- either way nobody reads stage ks - 1 any more
    if (ks + 2 < nks) issue(ks + 2);
    const unsigned char* st = smem + (ks % kWideStages) * kStage;
    const uint4* xs = reinterpret_cast<const uint4*>(st + kWideWBytes);
    u32x4_t b0[MT], b1[MT];
    bf16x8_t a0[NTW], a1[NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = wm * (BM / 2) + mt * 16 + c;
      b0[mt] = *reinterpret_cast<const u32x4_t*>(&xs[r * 8 + ((2 * g) ^ xs_swz<WD>(r))]);
      b1[mt] = *reinterpret_cast<const u32x4_t*>(&xs[r * 8 + ((2 * g + 1) ^ xs_swz<WD>(r))]);
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {   // decoded here, in the READ phase, while the SIMD's other wave multiplies: the MULTIPLY phase is MFMAs only
      const u32x4_t w4 = *reinterpret_cast<const u32x4_t*>(st + (wn * NTW + i) * 1024 + lane * 16);
      a0[i] = decode8<WD>(w4[0], w4[1]);
      a1[i] = decode8<WD>(w4[2], w4[3]);
    }
    // the fragments are in registers (the empty asm takes them as operands: the compiler's wait for the reads sits in front of it)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { asm volatile("" : "+v"(b0[mt])); asm volatile("" : "+v"(b1[mt])); }
#pragma unroll
    for (int i = 0; i < NTW; ++i) { asm volatile("" : "+v"(a0[i])); asm volatile("" : "+v"(a1[i])); }
    if (grp == 1 && ks + 1 < nks) {   // own pieces of stage ks + 1: group 0 reads it behind the next barrier
      if (ks + 2 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();           // group 0: P2ks+1 | group 1: P2ks+2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], __builtin_bit_cast(bf16x8_t, b0[mt]), acc[i][mt], 0, 0, 0);
        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], __builtin_bit_cast(bf16x8_t, b1[mt]), acc[i][mt], 0, 0, 0);
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // the barrier group 1 executed first
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int nt = ntb + wn * NTW + i;
    if (nt >= NT) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + wm * (BM / 2) + mt * 16 + c;
      if (m >= T) continue;
      if constexpr (SPLIT) {
        *reinterpret_cast<float4*>(slab + ((size_t)blockIdx.z * T + m) * (NT * 16) + nt * 16 + g * 4) =
            make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
      } else {
        epilogue<EPI>(e, m, nt * 16 + g * 4, acc[i][mt]);
      }
    }
  }
}

template <int WD, int BM>
static int launch_gemm_wide_bm(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                               int KS, float* splitk_ws, SlabSum* defer) {
  const int NT = w.N / 16, KT = w.K / 64;
  const int mtiles = ceil_div(T, BM), ntiles = ceil_div(w.N, kWideBN);
  dim3 grid(8 * mtiles * ceil_div(ntiles, 8), 1, KS);
  const uint4* W = reinterpret_cast<const uint4*>(w.w);
  constexpr int lds = kWideStages * (kWideWBytes + BM * 128);
#define MI_GW(EPI_) \
  do { \
    if (KS == 1) { \
      MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_wide_kernel<WD, EPI_, BM, false>), lds)); \
      hipLaunchKernelGGL((gemm_wide_kernel<WD, EPI_, BM, false>), grid, dim3(512), lds, s, W, NT, KT, T, w.K, x, ldx, e, nullptr); \
    } else { \
      MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_wide_kernel<WD, EPI_, BM, true>), lds)); \
      hipLaunchKernelGGL((gemm_wide_kernel<WD, EPI_, BM, true>), grid, dim3(512), lds, s, W, NT, KT, T, w.K, x, ldx, e, splitk_ws); \
      if (defer && EPI_ == EPI_RESID) { *defer = SlabSum{splitk_ws, KS, T, w.N, e.scale, e.bias, e.row_scale}; } \
      else hipLaunchKernelGGL((splitk_reduce_kernel<EPI_>), dim3(ceil_div(T * (w.N / 4), 256)), dim3(256), 0, s, splitk_ws, KS, T, w.N, e); \
    } \
  } while (0)
  if (epi == EPI_QKV) MI_GW(EPI_QKV);
  else if (epi == EPI_SWIGLU) MI_GW(EPI_SWIGLU);
  else if (epi == EPI_RESID) MI_GW(EPI_RESID);
  else MI_GW(EPI_F32);
#undef MI_GW
  MI_HIP(hipGetLastError());
  return MI_OK;
}
// The wide kernel runs ONE work-group per CU (96 / 144 KiB of LDS), so what decides is how full its rounds of the chip are,
// and between the two token blocks, rounds x time per K-step (measured on full grids: 0.90 us at 128 tokens, 1.45 us at 256 --
// twice the flops).  Measured on the Llama-8B shapes (round 3, staggered schedule; wide against the 128 x 128 kernel, us):
//     gate|up  256 tokens: 224 work-groups of 128 tokens (one round, 88 %)   62.0 vs  73.4      512: 448 (two rounds)  115.3 vs 141.6
//     gate|up 1024 tokens: 896 (3.5 rounds)                                 224.5 vs 227.6      QKV 1024: 192 (one round, 75 %) 66.6 vs 84.2
//     QKV  512: 96 work-groups   64.6 vs 40.1      O / down 1024: 128   57.1 vs 40.7 / 166.6 vs 116.2 (K-split on the other side)
//     QKV 2048: 384 (1.5 rounds, 75 %)  134.5 vs 112.4;  gate|up 2048: 403.6 at 128 tokens (7 rounds), 373.4 at 256 (3.5)
// -> a shape is eligible with one partial round from 70 % of the CUs on or several rounds at 80 % fill; the cheapest eligible
// (token block, K-split) runs: rounds x K-steps x time per K-step, plus, for a K-split, the slabs written and read once more
// (~4 bytes/ns) and the launch that sums them.
struct WidePlan { int bm, ks; };   // bm 0: not wanted
static WidePlan gemm_wide_plan(int T, int N, int K, size_t ws_bytes, int wd) {
  int cus = 256;
  if (device_num_cu(&cus) != MI_OK || cus < 1) cus = 256;
  const int nks = K / 64;
  WidePlan best{0, 1};
  double best_cost = 0.;
  for (int bm = 128; bm <= 256; bm *= 2) {
    for (int ks = 1; ks <= 8; ks *= 2) {
      if (nks % ks != 0 || nks / ks < 8) continue;
      if (ks > 1 && (size_t)ks * T * N * sizeof(float) > ws_bytes) continue;
      const int wgs = ceil_div(T, bm) * ceil_div(N, kWideBN) * ks, rounds = ceil_div(wgs, cus);
      const bool ok = wgs < cus ? wgs * 10 >= cus * 7 : wgs * 5 >= rounds * cus * 4;
      if (!ok) continue;
      // INT8 codes cost the READ phase ~6 x the VALU work of the fp8 hardware convert: 1.38 / 1.73 us per K-step measured on
      // the Qwen2.5-7B shapes (gate|up at 1024 tokens 387.8 us at 128-token blocks against 303.5 on the 128 x 128 kernel;
      // at 2048 tokens 485 at 256-token blocks against 592) -- there the wide tile must also beat what the 128 x 128 kernel
      // holds: 0.9 PF/s on grids that keep three work-groups per CU resident, 0.65 on its K-split grids
      const bool i8 = wd == MI_W_INT8;
      double cost = (double)rounds * (nks / ks) * (bm == 128 ? (i8 ? 1.38 : 0.90) : (i8 ? 1.73 : 1.45));
      if (ks > 1) cost += 4.0 + (double)(ks + 1) * T * N * 4.0 / 4.0e6 * 1.0;   // slabs out and in, the sum out
      const double alt_pf = ceil_div(T, 128) * ceil_div(N, 128) >= 3 * cus ? 0.9e9 : 0.65e9;   // flops per us: full residency / a K-split grid
      if (i8 && cost > 2.0 * T * N * (double)K / alt_pf) continue;
      if (best.bm == 0 || cost < best_cost) { best = WidePlan{bm, ks}; best_cost = cost; }
    }
  }
  return best;
}
static int gemm_wide_mode() {   // MI355X_GEMM_WIDE: 0 never, 1 whenever the shape allows, unset: by size
  static const int m = [] { const char* v = getenv("MI355X_GEMM_WIDE"); return v ? atoi(v) : -1; }();
  return m;
}
static int gemm_wide_bm() {      // MI355X_GEMM_WIDE_BM: 128 / 256 forces the token block (default: by grid size)
  static const int m = [] { const char* v = getenv("MI355X_GEMM_WIDE_BM"); return v ? atoi(v) : 0; }();
  return m;
}
static int gemm_wide_ks() {      // MI355X_GEMM_WIDE_KS: 1 = never split K in the wide kernel
  static const int m = [] { const char* v = getenv("MI355X_GEMM_WIDE_KS"); return v ? atoi(v) : 0; }();
  return m;
}
template <int WD>
static int launch_gemm_wide_wd(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                               float* splitk_ws, size_t splitk_ws_bytes, SlabSum* defer, int force_bm, int force_ks) {
  WidePlan pl = gemm_wide_plan(T, w.N, w.K, gemm_wide_ks() == 1 || !splitk_ws ? 0 : splitk_ws_bytes, w.wd);
  const int forced_bm = force_bm ? force_bm : gemm_wide_bm();
  if (forced_bm == 128 || forced_bm == 256) pl = WidePlan{forced_bm, 1};
  if (pl.bm == 0) pl = WidePlan{ceil_div(T, 256) * ceil_div(w.N, kWideBN) >= 512 ? 256 : 128, 1};   // forced onto a shape the plan would not take
  if (force_ks) {
    MI_CHECK(force_ks >= 1 && (w.K / 64) % force_ks == 0 && (force_ks == 1 || (splitk_ws && (size_t)force_ks * T * w.N * 4 <= splitk_ws_bytes)),
             "gemm_wide: forced K-split must divide K / 64 and fit the workspace");
    pl.ks = force_ks;
  }
  if (pl.bm == 256) return launch_gemm_wide_bm<WD, 256>(w, T, x, ldx, epi, e, s, pl.ks, splitk_ws, defer);
  return launch_gemm_wide_bm<WD, 128>(w, T, x, ldx, epi, e, s, pl.ks, splitk_ws, defer);
}
// the wide tile needs 1-byte weights
static bool gemm_wide_wanted(const LinearW& w, int T, size_t ws_bytes) {
  if (w.wd == MI_W_BF16 || w.K % 64 != 0 || (w.N / 16) < 2) return false;
  const int mode = gemm_wide_mode();
  if (mode == 1) return true;
  if (mode == 0) return false;
  if (T <= 128) return false;   // up to one 128-token block: the 128 x 128 kernel and its K-split (also the 33 .. 128-row decode batches)
  return gemm_wide_plan(T, w.N, w.K, gemm_wide_ks() == 1 ? 0 : ws_bytes, w.wd).bm != 0;
}
int launch_gemm_wide(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                     float* splitk_ws, size_t splitk_ws_bytes, SlabSum* defer, int force_bm, int force_ks) {
  if (defer) *defer = SlabSum();
  MI_CHECK(w.wd != MI_W_BF16 && w.K % 64 == 0 && ldx % 8 == 0, "gemm_wide: 1-byte weights, K % 64 == 0");
  if (w.wd == MI_W_F8E4M3)
    return launch_gemm_wide_wd<MI_W_F8E4M3>(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer, force_bm, force_ks);
  return launch_gemm_wide_wd<MI_W_INT8>(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer, force_bm, force_ks);
}

int launch_gemm(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                float* splitk_ws, size_t splitk_ws_bytes, SlabSum* defer) {
  if (defer) *defer = SlabSum();
  MI_CHECK(T >= 1, "gemm: T must be >= 1");
  if (gemm_wide_wanted(w, T, splitk_ws ? splitk_ws_bytes : 0) && ldx % 8 == 0)
    return launch_gemm_wide(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer);
  MI_CHECK(w.N % 16 == 0 && w.K % 64 == 0, "gemm: N % 16 == 0 and K % 64 == 0 required");
  MI_CHECK(ldx % 8 == 0, "gemm: x row stride must be a multiple of 8 elements");
  switch (w.wd) {
    case MI_W_BF16: return launch_gemm_wd<MI_W_BF16>(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer);
    case MI_W_F8E4M3: return launch_gemm_wd<MI_W_F8E4M3>(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer);
    case MI_W_INT8: return launch_gemm_wd<MI_W_INT8>(w, T, x, ldx, epi, e, s, splitk_ws, splitk_ws_bytes, defer);
  }
  set_error("gemm: bad weight dtype");
  return MI_EINVAL;
}

// =====================================================================================
// GEMM with FP8 activations (context encoding, the MFMA-bound contraction)
// =====================================================================================
// y[m, n] = (sum_k q_x[m, k] * q_w[n, k]) * s_w[n] * s_x[m]   both operands OCP e4m3.
// v_mfma_scale_f32_16x16x128_f8f6f4 with every block scale = 2^0 runs at twice the rate of
// the bf16 (and of the unscaled fp8) MFMA.  Its lane (g = lane >> 4, r = lane & 15) holds 32
// bytes of row r: they are taken as the lane's 16 bytes of weight tile 2s and of tile 2s + 1
// (the contraction order inside an MFMA is free as long as both operands agree), so the
// weight image is the same one the GEMV streams.  Work-group = 4 waves, tile 256 (n) x 128 (m):
// wave w owns n-tiles 4w..4w+3 against all 8 m-tiles (128 fp32 accumulators per lane, one wave
// per SIMD); W fragments come straight from global (every wave reads distinct rows), x8 goes
// through LDS; the next K-step's operands are fetched into registers under the MFMAs.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
constexpr int kA8BM = 128, kA8BK = 128;   // tile: (64 NTW) weight rows x 128 tokens x 128 K bytes
constexpr int kA8Pitch = kA8BK + 32;   // LDS row pitch in bytes: 160 makes the fragment reads (row c, 16 B at g * 16 / + 64) conflict-free
                                       // over the ds_read_b128 lane groups; 144 left every group two-deep on its banks

// SPLIT: grid.z K-slices each write their raw fp32 accumulators to slab[z][m][n]; splitk_reduce
// sums the slabs in slice order (deterministic) and applies the epilogue.  Used when a short
// prompt leaves the (m-tile x n-tile) grid too small to pull the weights at HBM rate.
template <int EPI, bool SPLIT, int NTW>   // NTW 16-row weight tiles per wave: the tile is (64 NTW) x 128
__global__ __launch_bounds__(256) void gemm_a8_kernel(const uint4* __restrict__ W, int NT, int KT, int T, int K,
                                                      const uint8_t* __restrict__ x8, int ldx, EpiArgs e,
                                                      float* __restrict__ slab) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[kA8BM * kA8Pitch];   // 20 KiB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  // XCD-aware tile order.  Work-groups go round-robin over the 8 XCDs (own 4 MiB L2 each), so
  // linear id L runs on XCD L % 8.  An XCD takes every eighth 256-row weight slab and walks ALL
  // token blocks of a slab back to back: the slab (and the K-step the co-resident work-groups
  // are at) is fetched over the fabric once per XCD instead of once per XCD per token block.
  constexpr int BN = 64 * NTW;
  const int mtiles = ceil_div(T, kA8BM), ntiles = ceil_div(NT, BN / 16);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mblk = seq % mtiles, nblk = (seq / mtiles) * 8 + xcd;
  if (nblk >= ntiles) return;
  const int m0 = mblk * kA8BM, nt0 = nblk * (BN / 16) + wave * NTW;
  const int nks_all = K / kA8BK;
  const int ks_per = SPLIT ? ceil_div(nks_all, (int)gridDim.z) : nks_all;
  const int ks_beg = SPLIT ? (int)blockIdx.z * ks_per : 0;
  const int nks = min(ks_beg + ks_per, nks_all);   // exclusive end of this slice
  constexpr int kUnit = 0x7f7f7f7f;   // E8M0 block scales: 2^0

  f32x4_t acc[NTW][8];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // x staging: 128 rows x 128 B per K-step = 1024 chunks of 16 B, 4 per thread.  The fp8 image
  // is K-step-major, [K / 128][ldx = T rows][128 B]: a work-group's K-step is 16 KiB contiguous.
  u32x4_t xr[4], wr[NTW][2], wn[NTW][2];
  auto load_x = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
      const int m = min(m0 + row, T - 1);   // rows past T are computed on a copy of the last row, never stored
      xr[i] = *reinterpret_cast<const u32x4_t*>(x8 + ((size_t)ks * ldx + m) * kA8BK + ch * 16);
    }
  };
  auto load_w = [&](u32x4_t (&dst)[NTW][2], int ks) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int nt = min(nt0 + i, NT - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        dst[i][t] = *reinterpret_cast<const u32x4_t*>(W + ((size_t)nt * KT + ks * 2 + t) * 64 + lane);
    }
  };
  MI_TRACE_BEGIN();
  MI_STAMP(0);
  if (ks_beg < nks) {
    load_x(ks_beg);
    load_w(wr, ks_beg);
  }
  for (int ks = ks_beg; ks < nks; ++ks) {
    if (ks == ks_beg + 4) MI_STAMP(1);     // tools/trace_gemm.py: phases of the fifth K-step
    if (ks == ks_beg + 5) MI_STAMP(6);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<u32x4_t*>(&xs[row * kA8Pitch + ch * 16]) = xr[i];
    }
    __syncthreads();
    if (ks == ks_beg + 4) MI_STAMP(2);
    if (ks + 1 < nks) {
      load_x(ks + 1);
      load_w(wn, ks + 1);
    }
    if (ks == ks_beg + 4) MI_STAMP(3);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const unsigned char* xrow = &xs[(mt * 16 + c) * kA8Pitch + g * 16];
      const u32x4_t b0 = *reinterpret_cast<const u32x4_t*>(xrow);        // k-tile 2ks,   chunk g
      const u32x4_t b1 = *reinterpret_cast<const u32x4_t*>(xrow + 64);   // k-tile 2ks+1, chunk g
      const i32x8_t b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const i32x8_t a = {(int)wr[i][0][0], (int)wr[i][0][1], (int)wr[i][0][2], (int)wr[i][0][3],
                           (int)wr[i][1][0], (int)wr[i][1][1], (int)wr[i][1][2], (int)wr[i][1][3]};
        acc[i][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i][mt], 0, 0, 0, kUnit, 0, kUnit);
      }
    }
    if (ks == ks_beg + 4) MI_STAMP(4);
    __syncthreads();
    if (ks == ks_beg + 4) MI_STAMP(5);
    if (ks + 1 < nks) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        wr[i][0] = wn[i][0];
        wr[i][1] = wn[i][1];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    if (nt0 + i >= NT) continue;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + mt * 16 + c;
      if (m >= T) continue;
      const int n0 = (nt0 + i) * 16 + g * 4;
      if constexpr (SPLIT) {
        *reinterpret_cast<float4*>(slab + ((size_t)blockIdx.z * T + m) * (NT * 16) + n0) =
            make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
      } else {
        epilogue<EPI>(e, m, n0, acc[i][mt]);
      }
    }
  }
  MI_STAMP(7);
  MI_TRACE_END();
}

// ---- FP8 activations on the wide tile ------------------------------------------------------------
// The kernel above keeps three work-groups of 128 x 128 on a CU and takes in 32 KiB per 4.2 MFLOP; this one is
// gemm_wide_kernel's schedule (LDS-DMA ring of three stages, the two wave groups half a K-step apart) on 128 tokens x 256
// weight rows x 128 K bytes: 16 + 32 KiB per 8.4 MFLOP, half the bytes per flop.  (256 x 256 would halve them again but is
// 64 KiB a stage: two stages in 160 KiB, and two stages cannot carry the stagger -- profiles/r03_a8_wide_experiment.txt.)
// A stage: [16 n-tiles][2 k-tiles][64 lanes x 16 B] weight fragments as stored, then 128 activation rows x 128 B with
// 16-byte chunk ch of row r at slot ch ^ (r & 7) (the lane reads chunks g and 4 + g: conflict-free with that key).
constexpr int kA8wStage = 32 * 1024 + 128 * 128, kA8wDma = 6;
// (A ninth wave touching the lines of the K-step four ahead, to make the ring's DMAs L2 hits, cost 27 %: 260 -> 330 us on
// gate|up at 2048 tokens -- profiles/r03_a8_wide_experiment.txt.)
template <int EPI, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_a8_wide_kernel(const uint4* __restrict__ W, int NT, int KT, int T, int K,
                                                           const uint8_t* __restrict__ x8, int ldx, EpiArgs e,
                                                           float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int wm = wave & 1, wn = wave >> 1;      // waves 2 (tokens) x 4 (weight rows), each 64 x 64
  const int mtiles = ceil_div(T, 128), ntiles = ceil_div(NT, 16);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mblk = seq % mtiles, nblk = (seq / mtiles) * 8 + xcd;
  if (nblk >= ntiles) return;
  const int m0 = mblk * 128, ntb = nblk * 16;
  const int nks = SPLIT ? K / 128 / (int)gridDim.z : K / 128;
  const int ks0 = SPLIT ? (int)blockIdx.z * nks : 0;
  constexpr int kUnit = 0x7f7f7f7f;   // E8M0 block scales: 2^0

  // DMA sources of this lane: weight n-tiles 2w, 2w + 1 (two k-tiles each, adjacent in the image), activation pieces 2w, 2w + 1
  const uint4* wsrc[2];
  const uint8_t* xsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) wsrc[i] = W + ((size_t)min(ntb + 2 * wave + i, NT - 1) * KT + 2 * ks0) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3), sl = lane & 7;
    xsrc[i] = x8 + ((size_t)ks0 * ldx + min(m0 + r, T - 1)) * 128 + ((sl ^ (r & 7)) * 16);
  }
  auto issue = [&](int ks) {
    unsigned char* st = smem + (ks % kWideStages) * kA8wStage;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) glds16(wsrc[i] + ((size_t)ks * 2 + t) * 64, st + ((2 * wave + i) * 2 + t) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(xsrc[i] + (size_t)ks * ldx * 128, st + 32 * 1024 + (2 * wave + i) * 1024);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // the schedule of gemm_wide_kernel: READ (fragments to registers) / MULTIPLY (MFMAs only), the groups in opposite phases
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  issue(0);
  if (nks > 1) issue(1);
  if (grp == 1) {
    if (nks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kA8wDma) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int ks = 0; ks < nks; ++ks) {
    if (grp == 0) {
      if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kA8wDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ks + 2 < nks) issue(ks + 2);
    const unsigned char* st = smem + (ks % kWideStages) * kA8wStage;
    const uint4* xs = reinterpret_cast<const uint4*>(st + 32 * 1024);
    u32x4_t b0[4], b1[4], a0[4], a1[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int r = wm * 64 + mt * 16 + c;
      b0[mt] = *reinterpret_cast<const u32x4_t*>(&xs[r * 8 + (g ^ (r & 7))]);
      b1[mt] = *reinterpret_cast<const u32x4_t*>(&xs[r * 8 + ((4 + g) ^ (r & 7))]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = *reinterpret_cast<const u32x4_t*>(st + ((wn * 4 + i) * 2) * 1024 + lane * 16);
      a1[i] = *reinterpret_cast<const u32x4_t*>(st + ((wn * 4 + i) * 2 + 1) * 1024 + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(b0[j])); asm volatile("" : "+v"(b1[j]));
      asm volatile("" : "+v"(a0[j])); asm volatile("" : "+v"(a1[j]));
    }
    if (grp == 1 && ks + 1 < nks) {
      if (ks + 2 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kA8wDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const i32x8_t b = {(int)b0[mt][0], (int)b0[mt][1], (int)b0[mt][2], (int)b0[mt][3], (int)b1[mt][0], (int)b1[mt][1], (int)b1[mt][2], (int)b1[mt][3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x8_t a = {(int)a0[i][0], (int)a0[i][1], (int)a0[i][2], (int)a0[i][3], (int)a1[i][0], (int)a1[i][1], (int)a1[i][2], (int)a1[i][3]};
        acc[i][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i][mt], 0, 0, 0, kUnit, 0, kUnit);
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nt = ntb + wn * 4 + i;
    if (nt >= NT) continue;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + c;
      if (m >= T) continue;
      if constexpr (SPLIT) {
        *reinterpret_cast<float4*>(slab + ((size_t)blockIdx.z * T + m) * (NT * 16) + nt * 16 + g * 4) =
            make_float4(acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]);
      } else {
        epilogue<EPI>(e, m, nt * 16 + g * 4, acc[i][mt]);
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int KS, int T, int N, EpiArgs e) {
  const int q = blockIdx.x * 256 + threadIdx.x;      // one (m, 4 consecutive n) per thread
  const int nq = N >> 2;
  if (q >= T * nq) return;
  const int m = q / nq, n0 = (q - m * nq) * 4;
  f32x4_t v = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < KS; ++z) {
    const float4 p = *reinterpret_cast<const float4*>(slab + ((size_t)z * T + m) * N + n0);
    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
  }
  epilogue<EPI>(e, m, n0, v);
}

// sum a deferred K-split (SlabSum) the plain way: splitk_reduce + residual epilogue
int launch_splitk_flush(const SlabSum& sl, const float* resid_in, float* out, int ld_out, hipStream_t s) {
  EpiArgs e{};
  e.scale = sl.scale; e.bias = sl.bias; e.row_scale = sl.row_scale;
  e.out_f32 = out; e.resid_in = resid_in; e.ld_out = ld_out;
  hipLaunchKernelGGL((splitk_reduce_kernel<EPI_RESID>), dim3(ceil_div(sl.T * (sl.N / 4), 256)), dim3(256), 0, s, sl.slab, sl.KS,
                     sl.T, sl.N, e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

bool gemm_a8_supported(const LinearW& w) {
  return w.wd == MI_W_F8E4M3 && w.K % kA8BK == 0 && w.N % 16 == 0;
}

int launch_gemm_a8(const LinearW& w, int T, const uint8_t* x8, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                   float* splitk_ws, size_t splitk_ws_bytes, SlabSum* defer) {
  if (defer) *defer = SlabSum();
  MI_CHECK(T >= 1, "gemm_a8: T must be >= 1");
  MI_CHECK(gemm_a8_supported(w), "gemm_a8: needs fp8 weights, K % 128 == 0, N % 16 == 0");
  MI_CHECK(ldx >= T && e.row_scale != nullptr, "gemm_a8: x8 is the K-step-major image of >= T rows, with its row scales");
  const int NT = w.N / 16, KT = w.K / 64, nks = w.K / kA8BK;
  {   // The wide tile (MI355X_A8_WIDE: 0 never, 1 always, unset: by plan) where its one-work-group-per-CU grid fills one or two
      // rounds of the chip.  Measured on the Llama-8B shapes against the kernel below (us): 256 tokens gate|up 43.6 vs 45.9,
      // down 24.0 vs 25.6, QKV 17.3 vs 19.4; 512: 76.9 vs 86.0, 34.8 vs 39.6, 23.5 vs 28.0; 1024: down 58.4 vs 68.2, QKV 48.8 vs
      // 51.9 but gate|up (3.5 rounds) 145.0 vs 135.9; 2048: down 121.0 vs 136.1, O 46.2 vs 49.3, gate|up (7 rounds) 260.4 vs
      // 254.2 -- on long grids the three resident work-groups of the 128 x 128 kernel hide more latency than the ring does.
    static const int mode = [] { const char* v = getenv("MI355X_A8_WIDE"); return v ? atoi(v) : -1; }();
    int cus = 256;
    if (device_num_cu(&cus) != MI_OK || cus < 1) cus = 256;
    int best_ks = 0;
    double best_cost = 0.;
    for (int ks = 1; ks <= 8; ks *= 2) {
      if (nks % ks != 0 || nks / ks < 4) continue;
      if (ks > 1 && (!splitk_ws || (size_t)ks * T * w.N * sizeof(float) > splitk_ws_bytes)) continue;
      const int wgs = ceil_div(T, 128) * ceil_div(w.N, 256) * ks, rounds = ceil_div(wgs, cus);
      const bool ok = mode == 1 || (rounds <= 2 && (wgs < cus ? wgs * 10 >= cus * 7 : wgs * 5 >= rounds * cus * 4));
      if (!ok) continue;
      double cost = (double)rounds * (nks / ks) * 1.0;
      if (ks > 1) cost += 4.0 + (double)(ks + 1) * T * w.N * 4.0 / 4.0e6;
      if (best_ks == 0 || cost < best_cost) { best_ks = ks; best_cost = cost; }
    }
    if (mode != 0 && best_ks && T > 128) {
      const int KS = best_ks;
      dim3 grid(8 * ceil_div(T, 128) * ceil_div(ceil_div(w.N, 256), 8), 1, KS);
      const uint4* W = reinterpret_cast<const uint4*>(w.w);
      constexpr int lds = kWideStages * kA8wStage;
#define MI_A8W_L(EPI_, SPLIT_, WS_) \
  do { \
    MI_TRY_(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_a8_wide_kernel<EPI_, SPLIT_>), lds)); \
    hipLaunchKernelGGL((gemm_a8_wide_kernel<EPI_, SPLIT_>), grid, dim3(512), lds, s, W, NT, KT, T, w.K, x8, ldx, e, WS_); \
  } while (0)
#define MI_A8W(EPI_) \
  do { \
    if (KS == 1) { \
      MI_A8W_L(EPI_, false, nullptr); \
    } else { \
      MI_A8W_L(EPI_, true, splitk_ws); \
      if (defer && EPI_ == EPI_RESID) { *defer = SlabSum{splitk_ws, KS, T, w.N, e.scale, e.bias, e.row_scale}; } \
      else hipLaunchKernelGGL((splitk_reduce_kernel<EPI_>), dim3(ceil_div(T * (w.N / 4), 256)), dim3(256), 0, s, splitk_ws, KS, T, w.N, e); \
    } \
  } while (0)
      if (epi == EPI_QKV) MI_A8W(EPI_QKV);
      else if (epi == EPI_SWIGLU) MI_A8W(EPI_SWIGLU);
      else if (epi == EPI_RESID) MI_A8W(EPI_RESID);
      else MI_A8W(EPI_F32);
#undef MI_A8W
#undef MI_A8W_L
      MI_HIP(hipGetLastError());
      return MI_OK;
    }
  }
  // 128 x 128 tiles: 64 accumulator registers per lane leave room for three work-groups per CU,
  // which hide each other's load latency (the 256 x 128 tile ran one work-group per CU at 24 % of
  // the matrix-core rate: 29.7 ms for the 2048 bucket against 22.5 ms; forcing a fourth wave per
  // SIMD spills inside the K loop: 122 ms)
  constexpr int ntw = 2;
  const int bn = 64 * ntw;
  const int mtiles = ceil_div(T, kA8BM), ntiles = ceil_div(w.N, bn);
  // K-split only when the tile grid cannot occupy the chip and the slabs stay small
  // (768 work-groups are resident at once.  Measured on the Llama-8B shapes: below 256 tiles aim
  // for 512 work-groups, below 384 for 768; splitting larger grids costs more in slab traffic
  // than it wins -- 6.2 / 8.8 / 13.5 ms for the 256 / 512 / 1024 buckets.)
  const int KS = gemm_pick_splitk(mtiles * ntiles, nks, T, w.N, splitk_ws ? splitk_ws_bytes : 0);
  const uint4* W = reinterpret_cast<const uint4*>(w.w);
  dim3 grid(8 * mtiles * ceil_div(ntiles, 8), 1, KS);   // see the kernel: XCD-aware tile order
#define MI_A8(EPI_) \
  do { \
    if (KS == 1) { \
      hipLaunchKernelGGL((gemm_a8_kernel<EPI_, false, ntw>), grid, dim3(256), 0, s, W, NT, KT, T, w.K, x8, ldx, e, nullptr); \
    } else { \
      hipLaunchKernelGGL((gemm_a8_kernel<EPI_, true, ntw>), grid, dim3(256), 0, s, W, NT, KT, T, w.K, x8, ldx, e, splitk_ws); \
      if (defer && EPI_ == EPI_RESID) { *defer = SlabSum{splitk_ws, KS, T, w.N, e.scale, e.bias, e.row_scale}; } \
      else hipLaunchKernelGGL((splitk_reduce_kernel<EPI_>), dim3(ceil_div(T * (w.N / 4), 256)), dim3(256), 0, s, splitk_ws, KS, T, w.N, e); \
    } \
  } while (0)
  if (epi == EPI_QKV) MI_A8(EPI_QKV);
  else if (epi == EPI_SWIGLU) MI_A8(EPI_SWIGLU);
  else if (epi == EPI_RESID) MI_A8(EPI_RESID);
  else MI_A8(EPI_F32);
#undef MI_A8
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// =====================================================================================
// Load-time quantize + tile
// =====================================================================================
__global__ void rowmax_kernel(const float* __restrict__ w, int ld, float* __restrict__ rowmax) {
  const float* r = w + (size_t)blockIdx.x * ld;
  float m = 0.f;
  for (int i = threadIdx.x; i < ld; i += blockDim.x) m = fmaxf(m, fabsf(r[i]));
  __shared__ float sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) rowmax[blockIdx.x] = sm[0];
}
__global__ void allmax_kernel(const float* __restrict__ rowmax, int n, float* __restrict__ out) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, rowmax[i]);
  __shared__ float sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// bf16 activations -> e4m3 + per-token scale: s = amax / 448 (all-zero row: 1), q = rne(x / s).
// One work-group per row.
// The row stays in registers between the amax pass and the quantize pass (up to kRqMax 16-byte
// chunks per thread, K <= 16384); longer rows are re-read.
constexpr int kRqMax = 8;
__global__ __launch_bounds__(256) void rowquant_fp8_kernel(const uint16_t* __restrict__ x, int K, int ldx,
                                                           uint8_t* __restrict__ x8, float* __restrict__ row_scale) {
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const uint16_t* row = x + (size_t)t * ldx;
  const int nchunk = K / 8;
  u32x4_t v[kRqMax];
  float amax = 0.f;
#pragma unroll
  for (int q = 0; q < kRqMax; ++q) {
    const int c8 = tid + q * 256;
    v[q] = c8 < nchunk ? *reinterpret_cast<const u32x4_t*>(row + c8 * 8) : u32x4_t{0, 0, 0, 0};
  }
#pragma unroll
  for (int q = 0; q < kRqMax; ++q)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w = v[q][k];
      amax = fmaxf(amax, fmaxf(fabsf(bf16lo_to_f32(w)), fabsf(bf16hi_to_f32(w))));
    }
  for (int c8 = tid + kRqMax * 256; c8 < nchunk; c8 += 256) {
    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(row + c8 * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w = r[k];
      amax = fmaxf(amax, fmaxf(fabsf(bf16lo_to_f32(w)), fabsf(bf16hi_to_f32(w))));
    }
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  amax = fmaxf(amax, __shfl_xor(amax, 4));
  amax = fmaxf(amax, __shfl_xor(amax, 8));
  amax = fmaxf(amax, __shfl_xor(amax, 16));
  amax = fmaxf(amax, __shfl_xor(amax, 32));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = amax > 0.f ? amax / 448.f : 1.f;
  if (tid == 0) row_scale[t] = scale;
  auto emit = [&](const u32x4_t r, int c8) {
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w = r[k];
      const float a = fminf(fmaxf(bf16lo_to_f32(w) / scale, -448.f), 448.f);
      const float b = fminf(fmaxf(bf16hi_to_f32(w) / scale, -448.f), 448.f);
      o[k >> 1] |= (f32_to_e4m3fn(a) | (f32_to_e4m3fn(b) << 8)) << (16 * (k & 1));
    }
    // K-step-major image: [K / 128][T][128 B] (see gemm_a8_kernel: load_x)
    const int k0 = c8 * 8;
    *reinterpret_cast<uint2*>(x8 + ((size_t)(k0 >> 7) * gridDim.x + t) * 128 + (k0 & 127)) = make_uint2(o[0], o[1]);
  };
#pragma unroll
  for (int q = 0; q < kRqMax; ++q) {
    const int c8 = tid + q * 256;
    if (c8 < nchunk) emit(v[q], c8);
  }
  for (int c8 = tid + kRqMax * 256; c8 < nchunk; c8 += 256) emit(*reinterpret_cast<const u32x4_t*>(row + c8 * 8), c8);
}

int launch_rowquant_fp8(const uint16_t* x, int T, int K, int ldx, uint8_t* x8, float* row_scale, hipStream_t s) {
  MI_CHECK(K % 8 == 0 && ldx % 8 == 0, "rowquant: K and the row stride must be multiples of 8");
  hipLaunchKernelGGL(rowquant_fp8_kernel, dim3(T), dim3(256), 0, s, x, K, ldx, x8, row_scale);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

__device__ __forceinline__ int dst_to_src_row(const QuantJob& j, int r) {
  const int rel = r - j.dst_row0;
  if (rel < 0) return -1;
  int i;
  if (j.rowmap == ROWMAP_PLAIN) {
    i = rel;
  } else if (j.rowmap == ROWMAP_ROPE_PAIRS) {
    const int head = rel / j.hd, jj = rel % j.hd;
    i = head * j.hd + ((jj & 1) ? (jj >> 1) + (j.hd >> 1) : (jj >> 1));
  } else {
    if (rel & 1) return -1;
    i = rel >> 1;
  }
  return i < j.n_rows ? i : -1;
}

// one wave per destination tile
template <int WD>
__global__ __launch_bounds__(256) void quant_tile_kernel(QuantJob j, int nt_lo, int nt_cnt, int KT) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= nt_cnt * KT) return;
  const int nt = nt_lo + wave / KT, kt = wave % KT;
  const int r = nt * 16 + (lane & 15), gq = lane >> 4;
  const int i = dst_to_src_row(j, r);
  if (i < 0) return;
  constexpr int EPL = (WD == MI_W_BF16) ? 8 : 16;  // elements per lane
  const int col0 = kt * (EPL * 4) + gq * EPL;      // first of the lane's columns, relative to src_col0
  const bool pad_row = i >= j.n_rows - j.pad_rows;
  const int k_src = j.K - j.pad_cols;
  if (pad_row || col0 >= k_src) {                  // EPL divides every head size: a lane is all padding or none
    if (pad_row && kt == 0 && gq == 0) j.dst_scale[r] = 1.f;
    if (!pad_row && kt == 0 && gq == 0) {
      float sc = 1.f;
      if constexpr (WD != MI_W_BF16) {
        const float qm = (WD == MI_W_INT8) ? 127.f : 448.f;
        const float am = (j.quant_type == MI_Q_PER_TENSOR_SYMMETRIC) ? j.tmp_rowmax[j.src_rows_total] : j.tmp_rowmax[j.src_row0 + i];
        sc = am > 0.f ? am / qm : 1.f;
      }
      j.dst_scale[r] = sc;
    }
    reinterpret_cast<uint4*>(j.dst)[((size_t)nt * KT + kt) * 64 + lane] = make_uint4(0, 0, 0, 0);
    return;
  }
  const int srow = j.src_row0 + i;
  const float* src = j.src + (size_t)srow * j.ld + j.src_col0 + col0;
  float scale = 1.f;
  if constexpr (WD != MI_W_BF16) {
    const float qmax = (WD == MI_W_INT8) ? 127.f : 448.f;
    const float amax = (j.quant_type == MI_Q_PER_TENSOR_SYMMETRIC) ? j.tmp_rowmax[j.src_rows_total] : j.tmp_rowmax[srow];
    scale = amax > 0.f ? amax / qmax : 1.f;
  }
  if (kt == 0 && gq == 0) j.dst_scale[r] = scale;
  uint4 out;
  if constexpr (WD == MI_W_BF16) {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(src[2 * k], src[2 * k + 1]);
    out = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    const float qmax = (WD == MI_W_INT8) ? 127.f : 448.f;
    uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float t = src[k] / scale;
      t = fminf(fmaxf(t, -qmax), qmax);
      uint32_t b;
      if constexpr (WD == MI_W_INT8) b = (uint32_t)(int)rintf(t) & 0xffu;
      else b = f32_to_e4m3fn(t);
      o[k >> 2] |= b << (8 * (k & 3));
    }
    out = make_uint4(o[0], o[1], o[2], o[3]);
  }
  reinterpret_cast<uint4*>(j.dst)[((size_t)nt * KT + kt) * 64 + lane] = out;
}

int run_quant_job(const QuantJob& j, hipStream_t s) {
  MI_CHECK(j.K % 64 == 0, "quantize: K % 64 == 0 required");
  if (j.wd != MI_W_BF16) {
    hipLaunchKernelGGL(rowmax_kernel, dim3(j.src_rows_total), dim3(256), 0, s, j.src, j.ld, j.tmp_rowmax);
    hipLaunchKernelGGL(allmax_kernel, dim3(1), dim3(256), 0, s, j.tmp_rowmax, j.src_rows_total, j.tmp_rowmax + j.src_rows_total);
  }
  // destination rows touched: [dst_row0, dst_row0 + span)
  const int span = (j.rowmap == ROWMAP_EVERY_OTHER) ? 2 * j.n_rows : j.n_rows;
  const int nt_lo = j.dst_row0 / 16, nt_hi = ceil_div(j.dst_row0 + span, 16);
  const int KT = j.K / tile_k(j.wd);
  const int waves = (nt_hi - nt_lo) * KT;
  const dim3 grid(ceil_div(waves, 4)), block(256);
  if (j.wd == MI_W_BF16) hipLaunchKernelGGL((quant_tile_kernel<MI_W_BF16>), grid, block, 0, s, j, nt_lo, nt_hi - nt_lo, KT);
  else if (j.wd == MI_W_F8E4M3) hipLaunchKernelGGL((quant_tile_kernel<MI_W_F8E4M3>), grid, block, 0, s, j, nt_lo, nt_hi - nt_lo, KT);
  else hipLaunchKernelGGL((quant_tile_kernel<MI_W_INT8>), grid, block, 0, s, j, nt_lo, nt_hi - nt_lo, KT);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// tiled -> row-major (inspection / tests)
__global__ void untile_kernel(const uint4* __restrict__ tiled, int NT, int KT, int K, int eb, unsigned char* __restrict__ out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= NT * KT) return;
  const int nt = wave / KT, kt = wave % KT;
  const uint4 v = tiled[(size_t)wave * 64 + lane];
  const int r = nt * 16 + (lane & 15);
  // 16 bytes = 16 elements (1-byte types) or 8 elements (bf16)
  const size_t byte_col = (size_t)kt * 64 + (lane >> 4) * 16;
  *reinterpret_cast<uint4*>(out + (size_t)r * K * eb + byte_col) = v;
}

int launch_untile(const void* tiled, int N, int K, int wd, void* out, hipStream_t s) {
  MI_CHECK(N % 16 == 0 && K % 64 == 0, "untile: N % 16 == 0 and K % 64 == 0 required");
  const int NT = N / 16, KT = K / tile_k(wd);
  hipLaunchKernelGGL(untile_kernel, dim3(ceil_div(NT * KT, 4)), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(tiled), NT, KT, K, elem_bytes(wd), reinterpret_cast<unsigned char*>(out));
  MI_HIP(hipGetLastError());
  return MI_OK;
}

__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = bf16_to_f32(in[i]);
}
int launch_bf16_to_f32(const uint16_t* in, float* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(2048), dim3(256), 0, s, in, out, n);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
