// Paged-KV attention kernels for gfx950.  See attn_kernels.h for the contract.
//
// Token generation (attn_decode_kernel): HBM-bound.  One work-group per
// (context split, kv head, sequence); its 4 waves walk 32-token tiles of the sequence's
// blocks.  A tile row (one token, hd bf16) is read by hd/8 adjacent lanes as one 16-byte
// load each, so a wave-instruction moves 1 KiB of contiguous pool memory; 2 * 32*hd*2 bytes
// (K and V tile) are in flight per wave.  q.k uses v_dot2c_f32_bf16 and a DPP sum over the
// row's lanes; every lane group keeps its own online-softmax state, merged once at the end
// (wave shuffles, then LDS across the 4 waves).  Splits are merged by attn_combine_kernel in
// a fixed order (deterministic).
//
// Context encoding (attn_prefill_kernel): MFMA flash attention over the same pool.  Scores
// are computed transposed (S^T = K.Q^T) so the accumulator registers ARE the B operand of
// the P.V product (no LDS round trip for P); V is staged transposed in LDS.

#include "attn_kernels.h"

namespace mi {

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// exp2(x - m) with the -inf conventions of online softmax (m finite or -inf)
__device__ __forceinline__ float sexp2(float x, float m) {
  return (x == -INFINITY) ? 0.f : fexp2(x - m);
}

// =====================================================================================
// token generation
// =====================================================================================
template <int HD, int GP>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kpool, const uint16_t* __restrict__ vpool,
    int bs, const int32_t* __restrict__ block_table, int MB, const int32_t* __restrict__ ctx_lens,
    int nh, int nkv, int G, int NS, float* __restrict__ o_part, float* __restrict__ ml_part, float scale_log2e) {
  constexpr int LPR = HD / 8;    // lanes per token row
  constexpr int TPI = 64 / LPR;  // tokens per wave-instruction
  constexpr int NI = 32 / TPI;   // loads per 32-token tile
  __shared__ float sm_o[4][GP][HD];
  __shared__ float sm_ml[4][GP][2];

  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tg = lane / LPR, dl = lane % LPR;
  const int ctx = ctx_lens[b];
  const int nt = ceil_div(ctx, 32), tps = ceil_div(nt, NS);
  const int t_beg = split * tps, t_end = min(nt, t_beg + tps);

  uint4 qp[GP];
#pragma unroll
  for (int h = 0; h < GP; ++h)
    qp[h] = (h < G) ? *reinterpret_cast<const uint4*>(q + ((size_t)b * nh + kvh * G + h) * HD + dl * 8)
                    : make_uint4(0, 0, 0, 0);
  float m[GP], l[GP], o[GP][8];
#pragma unroll
  for (int h = 0; h < GP; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[h][e] = 0.f;
  }

  for (int tt = t_beg + wave; tt < t_end; tt += 4) {
    const int tok0 = tt * 32;
    const int blk = block_table[(size_t)b * MB + tok0 / bs];
    const size_t base = (((size_t)blk * nkv + kvh) * bs + (tok0 % bs)) * HD + dl * 8;
    uint4 kr[NI], vr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) kr[i] = *reinterpret_cast<const uint4*>(kpool + base + (size_t)(i * TPI + tg) * HD);
#pragma unroll
    for (int i = 0; i < NI; ++i) vr[i] = *reinterpret_cast<const uint4*>(vpool + base + (size_t)(i * TPI + tg) * HD);

    float p[NI][GP];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool valid = tok0 + i * TPI + tg < ctx;
#pragma unroll
      for (int h = 0; h < GP; ++h) {
        float s = 0.f;
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kr[i].x), __builtin_bit_cast(bf16x2_t, qp[h].x), s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kr[i].y), __builtin_bit_cast(bf16x2_t, qp[h].y), s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kr[i].z), __builtin_bit_cast(bf16x2_t, qp[h].z), s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kr[i].w), __builtin_bit_cast(bf16x2_t, qp[h].w), s, false);
        s = group_sum<LPR>(s);
        p[i][h] = valid ? s * scale_log2e : -INFINITY;
      }
    }
#pragma unroll
    for (int h = 0; h < GP; ++h) {
      float mx = p[0][h];
#pragma unroll
      for (int i = 1; i < NI; ++i) mx = fmaxf(mx, p[i][h]);
      const float mn = fmaxf(m[h], mx);
      const float alpha = sexp2(m[h], mn);  // mn == -inf only if m[h] == -inf -> 0, state is all-zero anyway
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        p[i][h] = sexp2(p[i][h], mn);
        ps += p[i][h];
      }
      l[h] = l[h] * alpha + ps;
      m[h] = mn;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[h][e] *= alpha;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float vf[8];
      vf[0] = bf16lo_to_f32(vr[i].x); vf[1] = bf16hi_to_f32(vr[i].x);
      vf[2] = bf16lo_to_f32(vr[i].y); vf[3] = bf16hi_to_f32(vr[i].y);
      vf[4] = bf16lo_to_f32(vr[i].z); vf[5] = bf16hi_to_f32(vr[i].z);
      vf[6] = bf16lo_to_f32(vr[i].w); vf[7] = bf16hi_to_f32(vr[i].w);
#pragma unroll
      for (int h = 0; h < GP; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[h][e] += p[i][h] * vf[e];
    }
  }

  // merge the TPI lane groups of this wave
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int h = 0; h < GP; ++h) {
      const float mo = __shfl_xor(m[h], off), lo = __shfl_xor(l[h], off);
      const float mn = fmaxf(m[h], mo);
      const float a = sexp2(m[h], mn), bb = sexp2(mo, mn);
      l[h] = l[h] * a + lo * bb;
      m[h] = mn;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[h][e] = o[h][e] * a + __shfl_xor(o[h][e], off) * bb;
    }
  }
  if (tg == 0) {
#pragma unroll
    for (int h = 0; h < GP; ++h) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sm_o[wave][h][dl * 8 + e] = o[h][e];
      if (dl == 0) {
        sm_ml[wave][h][0] = m[h];
        sm_ml[wave][h][1] = l[h];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < G * HD; idx += 256) {
    const int h = idx / HD, d = idx % HD;
    float M = sm_ml[0][h][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, sm_ml[w][h][0]);
    float acc = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = sexp2(sm_ml[w][h][0], M);
      acc += f * sm_o[w][h][d];
      L += f * sm_ml[w][h][1];
    }
    const size_t row = ((size_t)b * nh + kvh * G + h) * NS + split;
    o_part[row * HD + d] = acc;
    if (d == 0) {
      ml_part[row * 2] = M;
      ml_part[row * 2 + 1] = L;
    }
  }
}

// Merge the context splits of one (sequence, head) in split order.  All partials are
// fetched before the first use (kAttnMaxSplits is a compile-time bound), so the kernel is
// one L2 round trip deep.
template <int HD>
__global__ __launch_bounds__(HD) void attn_combine_kernel(const float* __restrict__ o_part, const float* __restrict__ ml_part,
                                                          int NS, int nh, uint16_t* __restrict__ out) {
  const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const size_t row0 = ((size_t)b * nh + head) * NS;
  float mv[kAttnMaxSplits], lv[kAttnMaxSplits], ov[kAttnMaxSplits];
#pragma unroll
  for (int s = 0; s < kAttnMaxSplits; ++s) {
    const bool ok = s < NS;
    const float2 ml = ok ? *reinterpret_cast<const float2*>(ml_part + (row0 + s) * 2) : make_float2(-INFINITY, 0.f);
    mv[s] = ml.x;
    lv[s] = ml.y;
    ov[s] = ok ? o_part[(row0 + s) * HD + d] : 0.f;
  }
  float M = -INFINITY;
#pragma unroll
  for (int s = 0; s < kAttnMaxSplits; ++s) M = fmaxf(M, mv[s]);
  float acc = 0.f, L = 0.f;
#pragma unroll
  for (int s = 0; s < kAttnMaxSplits; ++s) {
    const float f = sexp2(mv[s], M);
    acc += f * ov[s];
    L += f * lv[s];
  }
  out[((size_t)b * nh + head) * HD + d] = f32_to_bf16(acc / L);
}

int attn_decode_splits(int B, int nkv) {
  int ns = 256 / (B * nkv);  // ~one work-group per CU; >= 4 tiles (one per wave) each at 1k context
  return ns < 1 ? 1 : (ns > kAttnMaxSplits ? kAttnMaxSplits : ns);
}
size_t attn_scratch_bytes(int B, int nh, int hd) {
  return (size_t)B * nh * kAttnMaxSplits * (hd + 2) * sizeof(float);
}

template <int HD, int GP>
static int launch_decode_t(const uint16_t* q, const uint16_t* kpool, const uint16_t* vpool, int bs,
                           const int32_t* bt, int MB, const int32_t* ctx, int B, int nh, int nkv,
                           uint16_t* out, void* scratch, hipStream_t s) {
  const int G = nh / nkv, NS = attn_decode_splits(B, nkv);
  float* o_part = reinterpret_cast<float*>(scratch);
  float* ml_part = o_part + (size_t)B * nh * kAttnMaxSplits * HD;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
  hipLaunchKernelGGL((attn_decode_kernel<HD, GP>), dim3(NS, nkv, B), dim3(256), 0, s, q, kpool, vpool, bs, bt,
                     MB, ctx, nh, nkv, G, NS, o_part, ml_part, scale_log2e);
  hipLaunchKernelGGL((attn_combine_kernel<HD>), dim3(nh, B), dim3(HD), 0, s, o_part, ml_part, NS, nh, out);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_attn_decode(const uint16_t* q, const uint16_t* kpool, const uint16_t* vpool, int block_size,
                       const int32_t* block_table, int MB, const int32_t* ctx_lens, int B, int nh,
                       int nkv, int hd, uint16_t* out, void* scratch, hipStream_t s) {
  MI_CHECK(hd == 64 || hd == 128, "attention: head_dim must be 64 or 128");
  MI_CHECK(nh % nkv == 0 && nh / nkv <= 8, "attention: q heads per kv head must be 1..8");
  MI_CHECK(block_size % 32 == 0, "attention: block_size must be a multiple of 32");
  const int G = nh / nkv;
#define MI_DEC(HD_, GP_) return launch_decode_t<HD_, GP_>(q, kpool, vpool, block_size, block_table, MB, ctx_lens, B, nh, nkv, out, scratch, s)
  if (hd == 128) {
    if (G <= 1) MI_DEC(128, 1);
    if (G <= 2) MI_DEC(128, 2);
    if (G <= 4) MI_DEC(128, 4);
    MI_DEC(128, 8);
  } else {
    if (G <= 1) MI_DEC(64, 1);
    if (G <= 2) MI_DEC(64, 2);
    if (G <= 4) MI_DEC(64, 4);
    MI_DEC(64, 8);
  }
#undef MI_DEC
}

// =====================================================================================
// context encoding
// =====================================================================================
// Work-group = 4 waves = 64 queries of one head; wave w owns queries 16w..16w+15.
// Per 32-key tile:  S^T_u[key][q] = K_u . Q^T   (u = 0,1: two 16-key MFMA tiles)
//                   O^T[d][q]    += V^T[d][key] . P^T[key][q]
// The MFMA k-slot (g, j) of the P.V product is bound to key (j < 4 ? 4g + j : 16 + 4g + j - 4),
// which is where S^T_0 / S^T_1 already hold that key's score for lane (g, q) — so the
// exponentiated accumulators are the B operand as they stand, and V^T is staged with its
// keys in that order (perm below).
template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_kernel(
    const uint16_t* __restrict__ q, int T, int q_pos0, const uint16_t* __restrict__ kpool,
    const uint16_t* __restrict__ vpool, int bs, const int32_t* __restrict__ block_table, int nh, int nkv,
    uint16_t* __restrict__ out, float scale_log2e) {
  constexpr int KP = HD + 8;   // K row pitch (elements): +16 B
  constexpr int VP = 32 + 8;   // V^T row pitch
  constexpr int CPR = HD / 8;  // 16-byte chunks per row
  constexpr int DN = HD / 16;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[32 * KP];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[HD * VP];

  const int head = blockIdx.y, kvh = head / (nh / nkv);
  const int q0 = blockIdx.x * 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int qi = q0 + wave * 16 + c;            // this lane's query (column of S^T / O^T)
  const bool qvalid = qi < T;
  const int qpos = q_pos0 + (qvalid ? qi : 0);

  // Q^T fragments: lane (g, c) holds Q[q = c][32 ks + 8 g .. +8]
  uint4 qf[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks)
    qf[ks] = qvalid ? *reinterpret_cast<const uint4*>(q + ((size_t)qi * nh + head) * HD + ks * 32 + g * 8)
                    : make_uint4(0, 0, 0, 0);

  f32x4_t acc[DN];
#pragma unroll
  for (int dn = 0; dn < DN; ++dn) acc[dn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrow = -INFINITY, lrow = 0.f;

  const int last_q = min(q0 + 63, T - 1);
  const int ntiles = (q_pos0 + last_q) / 32 + 1;
  const int kv_len = q_pos0 + T;

  for (int tile = 0; tile < ntiles; ++tile) {
    __syncthreads();  // previous tile's LDS reads are done
    for (int idx = tid; idx < 32 * CPR; idx += 256) {
      const int tok = idx / CPR, ch = idx % CPR;
      const int ta = tile * 32 + tok;
      uint4 kv4 = make_uint4(0, 0, 0, 0), vv4 = kv4;
      if (ta < kv_len) {
        const int blk = block_table[ta / bs];
        const size_t src = (((size_t)blk * nkv + kvh) * bs + (ta % bs)) * HD + ch * 8;
        kv4 = *reinterpret_cast<const uint4*>(kpool + src);
        vv4 = *reinterpret_cast<const uint4*>(vpool + src);
      }
      *reinterpret_cast<uint4*>(&Ks[tok * KP + ch * 8]) = kv4;
      const int r = tok & 15;
      const int pcol = 8 * (r >> 2) + 4 * (tok >> 4) + (r & 3);  // key -> P.V k-slot
      const uint32_t vw[4] = {vv4.x, vv4.y, vv4.z, vv4.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        Vt[(ch * 8 + e) * VP + pcol] = (uint16_t)(vw[e >> 1] >> (16 * (e & 1)));
    }
    __syncthreads();

    // S^T_u = K_u . Q^T
    f32x4_t st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&Ks[(16 * u + c) * KP + ks * 32 + g * 8]);
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, qf[ks]), st[u], 0, 0, 0);
      }
    }
    // causal mask + online softmax for query c (its keys are spread over the 4 lanes g)
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = tile * 32 + 16 * u + 4 * g + i;
        const float sv = (key <= qpos) ? st[u][i] * scale_log2e : -INFINITY;
        st[u][i] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(mrow, mx);
    const float alpha = sexp2(mrow, mn);
    mrow = mn;
    float ps = 0.f;
    float pv[8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pe = sexp2(st[u][i], mn);
        pv[4 * u + i] = pe;
        ps += pe;
      }
    lrow = lrow * alpha + ps;
    union { uint32_t w[4]; bf16x8_t v; } pb;
#pragma unroll
    for (int k = 0; k < 4; ++k) pb.w[k] = pack_bf16x2(pv[2 * k], pv[2 * k + 1]);
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
      acc[dn][0] *= alpha; acc[dn][1] *= alpha; acc[dn][2] *= alpha; acc[dn][3] *= alpha;
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&Vt[(16 * dn + c) * VP + 8 * g]);
      acc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb.v, acc[dn], 0, 0, 0);
    }
  }
  lrow += __shfl_xor(lrow, 16);
  lrow += __shfl_xor(lrow, 32);
  if (qvalid) {
    const float inv = 1.f / lrow;
    uint16_t* orow = out + ((size_t)qi * nh + head) * HD;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
      const uint2 pk = make_uint2(pack_bf16x2(acc[dn][0] * inv, acc[dn][1] * inv),
                                  pack_bf16x2(acc[dn][2] * inv, acc[dn][3] * inv));
      *reinterpret_cast<uint2*>(orow + 16 * dn + 4 * g) = pk;
    }
  }
}

int launch_attn_prefill(const uint16_t* q, int T, int q_pos0, const uint16_t* kpool, const uint16_t* vpool,
                        int block_size, const int32_t* block_table, int nh, int nkv, int hd, uint16_t* out,
                        hipStream_t s) {
  MI_CHECK(hd == 64 || hd == 128, "attention: head_dim must be 64 or 128");
  MI_CHECK(T >= 1 && q_pos0 >= 0, "attention: bad T / q_pos0");
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
  const dim3 grid(ceil_div(T, 64), nh);
  if (hd == 128)
    hipLaunchKernelGGL((attn_prefill_kernel<128>), grid, dim3(256), 0, s, q, T, q_pos0, kpool, vpool, block_size, block_table, nh, nkv, out, scale_log2e);
  else
    hipLaunchKernelGGL((attn_prefill_kernel<64>), grid, dim3(256), 0, s, q, T, q_pos0, kpool, vpool, block_size, block_table, nh, nkv, out, scale_log2e);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// =====================================================================================
// standalone KV write (the model path writes K/V from the QKV epilogue instead)
// =====================================================================================
__global__ void kv_write_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                const int64_t* __restrict__ slots, int nkv, int hd, uint16_t* __restrict__ kpool,
                                uint16_t* __restrict__ vpool, int bs) {
  const int t = blockIdx.x;
  const long slot = slots[t];
  if (slot < 0) return;
  const int blk = (int)(slot / bs), off = (int)(slot % bs);
  const int chunks = nkv * hd / 8;
  for (int i = threadIdx.x; i < chunks; i += blockDim.x) {
    const int head = (i * 8) / hd, d = (i * 8) % hd;
    const size_t dst = (((size_t)blk * nkv + head) * bs + off) * hd + d;
    *reinterpret_cast<uint4*>(kpool + dst) = *reinterpret_cast<const uint4*>(k + (size_t)t * nkv * hd + i * 8);
    *reinterpret_cast<uint4*>(vpool + dst) = *reinterpret_cast<const uint4*>(v + (size_t)t * nkv * hd + i * 8);
  }
}

int launch_kv_write(const uint16_t* k, const uint16_t* v, const int64_t* slots, int T, int nkv, int hd,
                    uint16_t* kpool, uint16_t* vpool, int block_size, hipStream_t s) {
  MI_CHECK(hd % 8 == 0, "kv_write: head_dim % 8 == 0 required");
  if (T == 0) return MI_OK;
  hipLaunchKernelGGL(kv_write_kernel, dim3(T), dim3(128), 0, s, k, v, slots, nkv, hd, kpool, vpool, block_size);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // namespace mi
