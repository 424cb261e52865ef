// Paged-KV attention kernels for gfx950.  See attn_kernels.h for the contract.
//
// Token generation (attn_decode_kernel): HBM-bound.  One work-group per
// (context split, kv head, sequence); its waves walk 32-token tiles of the sequence's blocks,
// two tiles in flight per wave.  The query heads of the kv group are the columns of MFMA tiles
// (S^T = K.Q^T from K fragments loaded straight from the pool; O^T += V^T.P^T with V parked in a
// wave-private LDS tile and read transposed), so the arithmetic per byte is a few hundred
// cycles per 16 KiB tile -- the VALU dot-product form this replaced cost ~3000.  With fewer
// (sequence, kv head) pairs than CUs the context is split over work-groups and the splits are
// merged by attn_combine_kernel in a fixed order (deterministic); otherwise one work-group
// covers the whole sequence.
//
// Context encoding (attn_prefill_kernel): MFMA flash attention over the same pool.  Scores
// are computed transposed (S^T = K.Q^T) so the accumulator registers ARE the B operand of
// the P.V product (no LDS round trip for P); V is staged transposed in LDS.

#include "attn_kernels.h"

namespace mi {

#define MI_TRY_A(expr)             \
  do {                             \
    int _rc = (expr);              \
    if (_rc != MI_OK) return _rc;  \
  } while (0)

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses into 1 KiB of LDS starting at `l` (wave-uniform)
__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// exp2(x - m) with the -inf conventions of online softmax (m finite or -inf)
__device__ __forceinline__ float sexp2(float x, float m) {
  return (x == -INFINITY) ? 0.f : fexp2(x - m);
}

// =====================================================================================
// token generation
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// NW waves per work-group.  FINAL: the work-group covers the whole context of its (sequence, kv
// head) and writes the bf16 output itself (grid (1, nkv, B)); otherwise it covers context split
// blockIdx.x of NS and leaves (o, m, l) partials for attn_combine_kernel.
//
// A wave walks 32-token tiles.  The G <= 16 query heads of the kv group are the 16 columns of the
// MFMA tiles (the idle columns cost nothing: the kernel is bound by bytes, not by the matrix core):
//     S^T[token][head]  = K_tile . Q^T             2 x HD/32 MFMA 16x16x32 (two 16-token halves)
//     O^T[d][head]     += V_tile^T . P^T           HD/16 MFMA 16x16x32, k = the tile's 32 tokens
// K goes from the pool straight into A fragments (lane (g, c): token c, 16 B of the row).  V is
// read as whole rows, parked row-major in a wave-private LDS tile and picked up transposed by
// ds_read_b64_tr_b16.  The P.V k-slot (g, j) is bound to token (j < 4 ? 4g + j : 16 + 4g + j - 4),
// which is where the two S^T accumulators already hold that token's score, so the exponentiated
// scores are the B operand as they stand.
// Every wave keeps TWO tiles in flight (the next tile is requested before the current one is
// consumed); a tile past the wave's range is requested as one harmless line of the null block, so
// the loop has no predicated loads.  Rows of the last tile beyond the context are masked by score;
// the pool must not hold NaN/Inf there (the library zero-fills it and only ever writes finite values).
//
// MERGE (with !FINAL): the context splits of one (sequence, kv head) are merged by whichever of
// their NS work-groups finishes LAST, inside this launch -- no combine launch.  No work-group ever
// waits for another (nothing can hang): each publishes its partial with write-through (sc1) stores,
// every storing wave drains them, one lane takes a ticket from an agent-scope counter, and the
// work-group whose ticket is NS - 1 reads all NS partials back with sc1 loads (past its L1; no
// other reader of these lines exists in the launch, so no L2 holds a stale copy) and writes the
// bf16 output.  This is the hand-off form MI355X_MICROARCH.md lists as valid without fences
// ("the workgroup whose add came last, told by the value its add returned"; one work-group per
// CU: the launch pads its LDS request to keep it so).  The merge itself runs in split order, so the
// result does not depend on which work-group performs it.  The counter is left at 0 for the next launch.
template <int HD, int NW, bool FINAL, bool MERGE = false>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kpool, const uint16_t* __restrict__ vpool,
    int bs, const int32_t* __restrict__ block_table, int MB, const int32_t* __restrict__ ctx_lens,
    int nh, int nkv, int G, int NS, float* __restrict__ o_part, float* __restrict__ ml_part,
    uint16_t* __restrict__ out, float scale_log2e, unsigned int* __restrict__ tickets = nullptr, int R = 1,
    int num_blocks = 0) {
  // R > 1 (the target's pass of a speculation step): batch rows b R .. b R + R - 1 are the SAME sequence
  // at consecutive positions (one block-table row, context lengths growing by one).  Their query heads
  // sit side by side in the 16 MFMA columns (column = row-in-sequence * G + head, G R <= 16) and each
  // column masks by its own row's context length: the K/V of the sequence is read once, not R times.
  constexpr int KP = HD + 8;        // LDS row pitch (elements)
  constexpr int CPR = HD / 8;       // 16-byte chunks per row
  constexpr int KS = HD / 32, DN = HD / 16;
  constexpr int NV = 32 * CPR / 64; // 16-byte V loads per lane per tile
  constexpr int VT = 32 * KP;       // elements of one wave's V tile
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  uint16_t* Vs = reinterpret_cast<uint16_t*>(dsm);                       // [NW][32][KP] bf16
  float* sm_ml = reinterpret_cast<float*>(dsm + (size_t)NW * VT * 2);    // [NW][16][2]

  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z * R;   // b: first batch row of the sequence
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int Gc = G * R;                         // live MFMA columns
  const int cr = c / G, ch = c - cr * G;        // column c = (row of the sequence, head of the kv group)
  int ctx = 0;                                  // the longest context of the sequence's rows bounds the tiles walked
  for (int r = 0; r < R; ++r) ctx = max(ctx, ctx_lens[b + r]);
  const int ctx_c = c < Gc ? ctx_lens[b + cr] : 0;   // what this lane's column may attend to
  // Tiles are dealt round-robin: wave w of split s takes tiles s NW + w, + NS NW, ...  Which tile a wave STARTS
  // with therefore does not depend on the context length, so (num_blocks > 0) its first K/V request goes out
  // beside the load of the context length instead of behind it (one L2 round trip less on the way to the first
  // MFMA); what it fetched is masked by score, or never consumed when the tile lies beyond the context.
  const int nt = ceil_div(ctx, 32), stride = NS * NW;
  const int t_first = split * NW + wave, t_end = nt;
  uint16_t* Vw = Vs + wave * VT;

  u32x4_t kA[2 * KS], vA[NV], kB[2 * KS], vB[NV];
  // A 32-token tile is two 16-token halves; with block_size 16 (vLLM's default) they sit in two
  // different blocks, so each half looks its own block up (the same block for block_size % 32 == 0).
  auto issue = [&](u32x4_t (&kr)[2 * KS], u32x4_t (&vr)[NV], int tt) {
    const bool live = tt < t_end;
    const int tok0 = live ? tt * 32 : 0;
    size_t rowh[2];   // pool row of the first token of each half
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // the second half of the last tile may lie beyond the context (and beyond the table): masked by score, read from the null block
      const bool in = live && (tok0 + 16 * u) < ctx;
      const int blk = in ? block_table[(size_t)b * MB + (tok0 + 16 * u) / bs] : 0;
      rowh[u] = in ? ((size_t)blk * nkv + kvh) * bs + ((tok0 + 16 * u) % bs) : 0;
    }
    // K fragment (u, ks): token 16u + c, dims 32 ks + 8 g ..
    const size_t kk = live ? 32 : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t kbase = live ? (rowh[u] + c) * HD + g * 8 : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kr[u * KS + ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kpool + kbase + ks * kk));
    }
    // V rows: lane-linear 16-byte chunks (whole rows per instruction); load i covers rows
    // [i * 64 / CPR, (i + 1) * 64 / CPR) of the tile, i.e. half (i * 64 / CPR) / 16
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (lane + 64 * i) / CPR, ch = (lane + 64 * i) % CPR;   // row of the tile, 16-byte chunk of the row
      const size_t src = live ? (rowh[row >> 4] + (row & 15)) * HD + ch * 8 : 0;
      vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vpool + src));
    }
  };
  if (num_blocks > 0) {
    // speculative form of issue(): no use of the context length; table index and block id clamped into range
    const int tok0 = t_first * 32;
    size_t rowh[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bi = min((tok0 + 16 * u) / bs, MB - 1);
      const int blk = min(max(block_table[(size_t)b * MB + bi], 0), num_blocks - 1);
      rowh[u] = ((size_t)blk * nkv + kvh) * bs + ((tok0 + 16 * u) % bs);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t kbase = (rowh[u] + c) * HD + g * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kA[u * KS + ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kpool + kbase + ks * 32));
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (lane + 64 * i) / CPR, chn = (lane + 64 * i) % CPR;
      vA[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vpool + (rowh[row >> 4] + (row & 15)) * HD + chn * 8));
    }
  } else {
    issue(kA, vA, t_first);
  }

  u32x4_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = (c < Gc) ? *reinterpret_cast<const u32x4_t*>(q + ((size_t)(b + cr) * nh + kvh * G + ch) * HD + ks * 32 + g * 8)
                      : u32x4_t{0, 0, 0, 0};
  f32x4_t acc[DN];
#pragma unroll
  for (int dn = 0; dn < DN; ++dn) acc[dn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrow = -INFINITY, lrow = 0.f;   // head c: running max (same on its 4 lanes), this lane's share of the sum
  const int tr_row = (lane & 15) >> 2, tr_col = (lane & 3) * 4;

  auto update = [&](const u32x4_t (&kr)[2 * KS], const u32x4_t (&vr)[NV], int tt) {
    // park V (the previous tile's transposed reads are complete: their values fed MFMAs already)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = lane + 64 * i;
      *reinterpret_cast<u32x4_t*>(&Vw[(idx / CPR) * KP + (idx % CPR) * 8]) = vr[i];
    }
    asm volatile("" ::: "memory");   // LDS ops of one wave execute in order; keep the compiler from reordering them
    f32x4_t st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4_t kw = kr[u * KS + ks], qw = qf[ks];
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kw), __builtin_bit_cast(bf16x8_t, qw), st[u], 0, 0, 0);
      }
    }
    s16x4_t vlo[DN], vhi[DN];
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
      const uint16_t* r0 = &Vw[(4 * g + tr_row) * KP + 16 * dn + tr_col];
      vlo[dn] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)r0);
      vhi[dn] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(r0 + 16 * KP));
    }
    const int tok0 = tt * 32;
    float pv[8];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sv = st[u][i] * scale_log2e;
        if (tok0 + 16 * u + 4 * g + i >= ctx_c) sv = -INFINITY;
        pv[4 * u + i] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));       // -inf where the column has no live token in this tile (sexp2 conventions)
    const float mn = fmaxf(mrow, mx);
    const float alpha = sexp2(mrow, mn);
    float ps = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pv[k] = sexp2(pv[k], mn);
      ps += pv[k];
    }
    const bool moved = mn != mrow;
    mrow = mn;
    lrow = lrow * alpha + ps;
    union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk.w[k] = pack_bf16x2(pv[2 * k], pv[2 * k + 1]);
    if (__any(moved)) {
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        acc[dn][0] *= alpha; acc[dn][1] *= alpha; acc[dn][2] *= alpha; acc[dn][3] *= alpha;
      }
    }
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
      const s16x4_t lo = vlo[dn], hi = vhi[dn];
      const bf16x8_t a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pk.v, acc[dn], 0, 0, 0);
    }
  };

  for (int tt = t_first; tt < t_end; tt += 2 * stride) {
    issue(kB, vB, tt + stride);
    update(kA, vA, tt);
    issue(kA, vA, tt + 2 * stride);
    if (tt + stride < t_end) update(kB, vB, tt + stride);
  }

  // ---- merge the waves: O^T / m / l of wave w go to its own (now idle) LDS tile --------------
  lrow += __shfl_xor(lrow, 16);
  lrow += __shfl_xor(lrow, 32);
  asm volatile("" ::: "memory");
  float* so = reinterpret_cast<float*>(Vw);            // [16 heads][HD] f32 = 8 KiB at HD 128 <= the tile
#pragma unroll
  for (int dn = 0; dn < DN; ++dn) *reinterpret_cast<f32x4_t*>(&so[c * HD + 16 * dn + 4 * g]) = acc[dn];
  if (g == 0) {
    sm_ml[(wave * 16 + c) * 2] = mrow;
    sm_ml[(wave * 16 + c) * 2 + 1] = lrow;
  }
  __syncthreads();
  for (int idx = tid; idx < Gc * HD; idx += NW * 64) {
    const int h = idx / HD, d = idx % HD;       // h: MFMA column
    const int hr = h / G, hh = h - hr * G;      // -> (row of the sequence, head)
    float M = sm_ml[h * 2];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, sm_ml[(w * 16 + h) * 2]);
    float o = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {  // wave order: deterministic
      const float f = sexp2(sm_ml[(w * 16 + h) * 2], M);
      o += f * reinterpret_cast<const float*>(Vs + w * VT)[h * HD + d];
      L += f * sm_ml[(w * 16 + h) * 2 + 1];
    }
    if constexpr (FINAL) {
      out[((size_t)(b + hr) * nh + kvh * G + hh) * HD + d] = f32_to_bf16(L > 0.f ? o / L : 0.f);   // empty context -> zeros
    } else if constexpr (MERGE) {
      const size_t row = ((size_t)(b + hr) * nh + kvh * G + hh) * kAttnMaxSplits + split;   // fixed stride: a head's m/l block is one 128-byte line
      __hip_atomic_store(&o_part[row * HD + d], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store ... sc1
      if (d == 0) {
        __hip_atomic_store(&ml_part[row * 2], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ml_part[row * 2 + 1], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const size_t row = ((size_t)(b + hr) * nh + kvh * G + hh) * kAttnMaxSplits + split;   // fixed stride: a head's m/l block is one 128-byte line
      o_part[row * HD + d] = o;
      if (d == 0) {
        ml_part[row * 2] = M;
        ml_part[row * 2 + 1] = L;
      }
    }
  }
  if constexpr (MERGE && !FINAL) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
    __syncthreads();
    int* s_last = reinterpret_cast<int*>(sm_ml);        // the m/l staging is dead behind the barrier
    if (tid == 0) {
      unsigned int* tk = tickets + (size_t)b * nkv + kvh;
      const unsigned int t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = t == (unsigned int)(NS - 1);
      if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *s_last = last ? 1 : 0;
    }
    __syncthreads();
    if (*s_last) {
      for (int idx = tid; idx < G * HD; idx += NW * 64) {
        const int h = idx / HD, d = idx % HD;
        const size_t row0 = ((size_t)b * nh + kvh * G + h) * kAttnMaxSplits;
        float mv[kAttnMaxSplits], lv[kAttnMaxSplits], ov[kAttnMaxSplits];
#pragma unroll
        for (int sp = 0; sp < kAttnMaxSplits; ++sp) {
          const bool ok = sp < NS;
          mv[sp] = ok ? __hip_atomic_load(&ml_part[(row0 + sp) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
          lv[sp] = ok ? __hip_atomic_load(&ml_part[(row0 + sp) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
          ov[sp] = ok ? __hip_atomic_load(&o_part[(row0 + sp) * HD + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
        float M = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < kAttnMaxSplits; ++sp) M = fmaxf(M, mv[sp]);
        float acc_o = 0.f, L = 0.f;
#pragma unroll
        for (int sp = 0; sp < kAttnMaxSplits; ++sp) {   // split order: the same sums as attn_combine_kernel
          const float f = sexp2(mv[sp], M);
          acc_o += f * ov[sp];
          L += f * lv[sp];
        }
        out[((size_t)b * nh + kvh * G + h) * HD + d] = f32_to_bf16(L > 0.f ? acc_o / L : 0.f);
      }
    }
  }
}
template <int HD, int NW>
constexpr size_t attn_decode_lds() { return (size_t)NW * 32 * (HD + 8) * 2 + (size_t)NW * 16 * 2 * sizeof(float); }

// Merge the context splits of one (sequence, head) in split order.  All partials are
// fetched before the first use (kAttnMaxSplits is a compile-time bound), so the kernel is
// one L2 round trip deep.
template <int HD>
__global__ __launch_bounds__(HD) void attn_combine_kernel(const float* __restrict__ o_part, const float* __restrict__ ml_part,
                                                          int NS, int nh, uint16_t* __restrict__ out) {
  const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const size_t row0 = ((size_t)b * nh + head) * kAttnMaxSplits;
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
  out[((size_t)b * nh + head) * HD + d] = f32_to_bf16(L > 0.f ? acc / L : 0.f);   // empty context -> zeros
}

int attn_decode_splits(int B, int nkv) {
  int ns = 256 / (B * nkv);  // ~one work-group per CU; >= 4 tiles (one per wave) each at 1k context
  return ns < 1 ? 1 : (ns > kAttnMaxSplits ? kAttnMaxSplits : ns);
}
// [tickets: kAttnTickets counters, one per (sequence, kv head), zero between launches]
// [m/l partials: B x nh rows of one 128-byte line][o partials]  -- the ticket block and the m/l
// rows sit at offsets that do not depend on the batch size of the call
constexpr int kAttnTickets = 1024;
size_t attn_scratch_bytes(int B, int nh, int hd) {
  return kAttnTickets * sizeof(unsigned int) + (size_t)B * nh * kAttnMaxSplits * (hd + 2) * sizeof(float);
}
// Off by default: measured on Llama-3.1-8B decode (B 4, ctx 1024) the in-launch merge costs the
// last work-group ~5.4 us (write-through drain + returning atomic + sc1 read-back are three
// dependent trips to memory) against 4.8 us for the separate combine launch: 2.157 vs 2.138 ms per
// step.  MI355X_ATTN_MERGE=1 selects it (same results: tests/test_model_gpu.py runs both).
static bool attn_merge_enabled() {
  static const bool on = [] { const char* v = getenv("MI355X_ATTN_MERGE"); return v && v[0] == '1'; }();
  return on;
}

template <int HD>
static int launch_decode_t(const uint16_t* q, const uint16_t* kpool, const uint16_t* vpool, int bs,
                           const int32_t* bt, int MB, const int32_t* ctx, int B, int nh, int nkv,
                           uint16_t* out, void* scratch, hipStream_t s, bool tickets_zeroed, int R, int num_blocks) {
  const int G = nh / nkv, Bs = B / R, NS = attn_decode_splits(Bs, nkv);   // Bs sequences of R batch rows each
  // in-launch merge: needs every split's work-group resident on a CU of its own (grid <= CUs) and
  // the ticket counters at zero (the model keeps them so; the per-op entry cannot know)
  int num_cu = 0;
  if (device_num_cu(&num_cu) != MI_OK) return MI_EHIP;
  const bool merge = attn_merge_enabled() && R == 1 && NS > 1 && tickets_zeroed && NS * nkv * B <= num_cu && nkv * B <= kAttnTickets;
  unsigned int* tickets = reinterpret_cast<unsigned int*>(scratch);
  float* ml_part = reinterpret_cast<float*>(tickets + kAttnTickets);
  float* o_part = ml_part + (size_t)B * nh * kAttnMaxSplits * 2;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
  constexpr size_t lds8 = attn_decode_lds<HD, 8>(), lds4 = attn_decode_lds<HD, 4>();
  constexpr size_t kMergeLds = 96 * 1024;   // more than half a CU's LDS: one work-group per CU (see the kernel comment)
  {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(attn_decode_kernel<HD, 8, true>), (int)lds8);
    if (rc == MI_OK) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(attn_decode_kernel<HD, 4, false>), (int)lds4);
    if (rc == MI_OK) rc = ensure_dynamic_lds(reinterpret_cast<const void*>(attn_decode_kernel<HD, 4, false, true>), (int)kMergeLds);
    if (rc != MI_OK) return rc;
  }
  if (NS == 1) {
    // Enough (sequence, kv head) pairs to fill the chip: one launch, no partials.  (Measured: a
    // single CU sustains only ~32 GB/s from HBM however many loads its waves keep in flight, so
    // a short grid must split the context over CUs -- the one-launch form at B x nkv = 32 took
    // 16 us per 1k tokens of context against 8 + 5 us for split + combine.)
    hipLaunchKernelGGL((attn_decode_kernel<HD, 8, true>), dim3(1, nkv, Bs), dim3(512), lds8, s, q,
                       kpool, vpool, bs, bt, MB, ctx, nh, nkv, G, 1, nullptr, nullptr, out, scale_log2e, nullptr, R, num_blocks);
  } else if (merge) {
    hipLaunchKernelGGL((attn_decode_kernel<HD, 4, false, true>), dim3(NS, nkv, B), dim3(256), kMergeLds, s, q,
                       kpool, vpool, bs, bt, MB, ctx, nh, nkv, G, NS, o_part, ml_part, out, scale_log2e, tickets);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<HD, 4, false>), dim3(NS, nkv, Bs), dim3(256), lds4, s, q,
                       kpool, vpool, bs, bt, MB, ctx, nh, nkv, G, NS, o_part, ml_part, nullptr, scale_log2e, nullptr, R, num_blocks);
    hipLaunchKernelGGL((attn_combine_kernel<HD>), dim3(nh, B), dim3(HD), 0, s, o_part, ml_part, NS, nh, out);
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int launch_attn_decode(const uint16_t* q, const uint16_t* kpool, const uint16_t* vpool, int block_size,
                       const int32_t* block_table, int MB, const int32_t* ctx_lens, int B, int nh,
                       int nkv, int hd, uint16_t* out, void* scratch, hipStream_t s, bool tickets_zeroed, int rows_per_seq,
                       int num_blocks) {
  MI_CHECK(hd == 64 || hd == 128, "attention: head_dim must be 64 or 128");
  MI_CHECK(rows_per_seq >= 1 && B % rows_per_seq == 0 && (nh / nkv) * rows_per_seq <= 16,
           "attention: rows_per_seq must divide the batch and q heads per kv head x rows_per_seq must be <= 16");
  MI_CHECK(nh % nkv == 0 && nh / nkv <= 16, "attention: q heads per kv head must be 1..16");
  MI_CHECK(block_size % 16 == 0, "attention: block_size must be a multiple of 16");
  if (hd == 128) return launch_decode_t<128>(q, kpool, vpool, block_size, block_table, MB, ctx_lens, B, nh, nkv, out, scratch, s, tickets_zeroed, rows_per_seq, num_blocks);
  return launch_decode_t<64>(q, kpool, vpool, block_size, block_table, MB, ctx_lens, B, nh, nkv, out, scratch, s, tickets_zeroed, rows_per_seq, num_blocks);
}

// =====================================================================================
// context encoding
// =====================================================================================
// One work-group per (32*QB queries, kv head): wave w owns q head (w % Gp) of the kv group and
// query block (w / Gp), i.e. 32 queries = two 16-query MFMA tiles that share every K / V
// fragment; the G heads of the group share the K/V tiles staged in LDS (one HBM read per group).
// Per 64-key tile:  S^T_u[key][q] = K_u . Q^T        (u = 0..3: four 16-key MFMA tiles)
//                   O^T[d][q]    += V^T[d][key] . P^T[key][q]   (two k-steps of 32 keys)
// The MFMA k-slot (g, j) of P.V k-step kk is bound to key 32 kk + (j < 4 ? 4g + j : 16 + 4g + j - 4),
// which is where S^T_{2kk} / S^T_{2kk+1} already hold that key's score for lane (g, q): the
// exponentiated accumulators are the B operand as they stand (no LDS round trip for P).  V stays
// row-major in LDS; its transposed A fragments come from ds_read_b64_tr_b16 (4 keys x 16 dims
// per 16-lane group).  Next tile's K/V are prefetched into registers under the MFMAs.

template <int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_prefill_kernel(
    const uint16_t* __restrict__ q, int T, int q_pos0, const uint16_t* __restrict__ kpool,
    const uint16_t* __restrict__ vpool, int bs, const int32_t* __restrict__ block_table, int nh, int nkv, int G,
    int Gp, int paired, uint16_t* __restrict__ out, float scale_log2e) {
  // row pitch (elements): head_dim + 16, i.e. 288 B = 72 banks (head_dim 128) or 160 B = 40 banks (64).  Either way the 16 rows
  // of a ds_read_b128 lane group start on 16 different 4-bank windows (its lanes of the next 16-byte column take the windows in
  // between) and the 8 rows of a ds_read_b64_tr_b16 group own 8 banks each: no conflicts.  (At head_dim + 8 one pair of lanes
  // per group met on a bank: 37 % of the LDS cycles, SQ_LDS_BANK_CONFLICT.)
  constexpr int KP = HD + 16;
  constexpr int CPR = HD / 8;   // 16-byte chunks per row
  constexpr int DN = HD / 16, KS = HD / 32;
  constexpr int TK = 64;        // keys per tile
  constexpr int LD = (TK * CPR) / (WAVES * 64);  // 16-byte loads per thread per K (or V) tile
  __shared__ __attribute__((aligned(16))) uint16_t Ks[TK * KP];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[TK * KP];
  __shared__ int32_t bt_lds[kPrefillMaxBlocks];   // this sequence's block table: one L2 trip, not one per tile

  constexpr int nthr = WAVES * 64, waves = WAVES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int kvh = blockIdx.y, QB = waves / Gp;
  const int hl = wave % Gp, qb = wave / Gp;
  const bool head_ok = hl < G;
  const int head = kvh * G + min(hl, G - 1);
  const int kv_len = q_pos0 + T;
  for (int i = threadIdx.x; i < ceil_div(kv_len, bs); i += nthr) bt_lds[i] = block_table[i];

  // Causal work grows linearly with the query block index.  paired (long prompts): a work-group
  // takes block x and then its complement (nblk - 1 - x), every work-group the same number of key
  // tiles, one work-group per CU.  Otherwise one block per work-group, heaviest first: two
  // work-groups share a CU (a wave's softmax arithmetic runs under the other's MFMAs) and the
  // dispatcher hands out the light blocks as the heavy ones finish.  (Measured on Llama-8B shapes:
  // unpaired wins 1-3 % of the whole prefill up to the 1024 bucket, pairing wins 9 % at 2048.)
  const int nblk = ceil_div(T, 32 * QB);
  for (int pass = 0; pass < (paired ? 2 : 1); ++pass) {
  const int xb = !paired ? nblk - 1 - (int)blockIdx.x : (pass == 0 ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x);
  if (pass == 1 && xb <= (int)blockIdx.x) break;   // odd count: the middle block is done once
  const int wg_q0 = xb * 32 * QB;
  const int q0 = wg_q0 + 32 * qb;            // this wave's 32 queries

  u32x4_t qf[2][KS];
  int qpos[2];
  bool qok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qi = q0 + 16 * t + c;
    qok[t] = head_ok && qi < T;
    qpos[t] = q_pos0 + min(qi, T - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[t][ks] = qok[t] ? *reinterpret_cast<const u32x4_t*>(q + ((size_t)qi * nh + head) * HD + ks * 32 + g * 8)
                         : u32x4_t{0, 0, 0, 0};
  }
  f32x4_t acc[2][DN];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) acc[t][dn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};

  const int wg_last_pos = q_pos0 + min(wg_q0 + 32 * QB - 1, T - 1);
  const int ntiles = wg_last_pos / TK + 1;
  const int w_first_pos = q_pos0 + min(q0, T - 1), w_last_pos = q_pos0 + min(q0 + 31, T - 1);

  __syncthreads();   // block table staged (pass 0) / previous pass done with the K/V tiles
  // register prefetch of one K/V tile (each thread: LD chunks of K and of V)
  // Two register sets, two tiles ahead: with one wave per SIMD a tile's arithmetic (~1 us) is
  // shorter than an HBM round trip, so the tile after next is requested before this one is used.
  // (ext vectors: hipcc keeps arrays of them in VGPRs; HIP's uint4 struct goes to scratch)
  u32x4_t kregA[LD], vregA[LD], kregB[LD], vregB[LD];
  auto prefetch = [&](u32x4_t (&kreg)[LD], u32x4_t (&vreg)[LD], int tile) {
    tile = min(tile, ntiles - 1);
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int idx = tid + i * nthr;
      const int key = idx / CPR, ch = idx % CPR;
      const int ta = min(tile * TK + key, kv_len - 1);   // clamp: rows past the context are masked anyway
      const int blk = bt_lds[ta / bs];
      const size_t src = (((size_t)blk * nkv + kvh) * bs + (ta % bs)) * HD + ch * 8;
      kreg[i] = *reinterpret_cast<const u32x4_t*>(kpool + src);
      vreg[i] = *reinterpret_cast<const u32x4_t*>(vpool + src);
    }
  };
  prefetch(kregA, vregA, 0);
  prefetch(kregB, vregB, 1);

  auto do_tile = [&](u32x4_t (&kreg)[LD], u32x4_t (&vreg)[LD], int tile) {
    __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int idx = tid + i * nthr;
      const int key = idx / CPR, ch = idx % CPR;
      *reinterpret_cast<u32x4_t*>(&Ks[key * KP + ch * 8]) = kreg[i];
      *reinterpret_cast<u32x4_t*>(&Vs[key * KP + ch * 8]) = vreg[i];
    }
    __syncthreads();
    prefetch(kreg, vreg, tile + 2);
    if (tile * TK > w_last_pos) return;   // wave-uniform: every key of this tile is in this wave's future

    // ---- S^T = K . Q^T ------------------------------------------------------------------
    // One wave per SIMD: latency is hidden by issue order, not by other waves.  All 16 K
    // fragments are requested first, then the 32 MFMAs run back to back.
    bf16x8_t kf[4][KS];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kf[u][ks] = *reinterpret_cast<const bf16x8_t*>(&Ks[(16 * u + c) * KP + ks * 32 + g * 8]);
    f32x4_t st[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      st[u][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      st[u][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][ks], __builtin_bit_cast(bf16x8_t, qf[0][ks]), st[u][0], 0, 0, 0);
        st[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u][ks], __builtin_bit_cast(bf16x8_t, qf[1][ks]), st[u][1], 0, 0, 0);
      }
    }
    // V^T fragments for the P.V product: requested now, so their LDS latency runs under the
    // softmax arithmetic.  tr read: lane 4q' + p of a 16-lane group addresses row q' (a key),
    // columns 4p..4p+3 (dims); lane i receives column i (dim 16 dn + i) of the 4 rows = 4
    // consecutive k-slots of the A fragment.
    const int tr_row = (lane & 15) >> 2, tr_col = (lane & 3) * 4;
    s16x4_t vlo[2][DN], vhi[2][DN];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const uint16_t* r0 = &Vs[(32 * kk + 4 * g + tr_row) * KP + 16 * dn + tr_col];
        vlo[kk][dn] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)r0);
        vhi[kk][dn] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(r0 + 16 * KP));
      }
    // ---- causal mask + online softmax (query c of q-tile t; its keys sit on the 4 lanes g) --
    // Only the tiles that straddle the diagonal need the mask; every other tile takes the lean
    // path: max on the raw scores, then one fma + one exp2 per score.  The O^T rescale is skipped
    // (exactly: alpha == 1) when no lane's running maximum moved.
    const bool need_mask = tile * TK + TK - 1 > w_first_pos;
    bf16x8_t pb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pv[16];
      float mn, alpha, ps = 0.f;
      if (need_mask) {
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sv = st[u][t][i] * scale_log2e;
            if (tile * TK + 16 * u + 4 * g + i > qpos[t]) sv = -INFINITY;
            st[u][t][i] = sv;
            mx = fmaxf(mx, sv);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        mn = fmaxf(mrow[t], mx);
        alpha = sexp2(mrow[t], mn);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            pv[4 * u + i] = sexp2(st[u][t][i], mn);
            ps += pv[4 * u + i];
          }
      } else {
        float mx = st[0][t][0];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) mx = fmaxf(mx, st[u][t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        mn = fmaxf(mrow[t], mx * scale_log2e);      // finite: every key of the tile is visible
        alpha = fexp2(mrow[t] - mn);                // exp2(-inf) = 0 on the first tile
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            pv[4 * u + i] = fexp2(fmaf(st[u][t][i], scale_log2e, -mn));
            ps += pv[4 * u + i];
          }
      }
      const bool moved = mn != mrow[t];
      mrow[t] = mn;
      lrow[t] = lrow[t] * alpha + ps;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk.w[k] = pack_bf16x2(pv[8 * kk + 2 * k], pv[8 * kk + 2 * k + 1]);
        pb[t][kk] = pk.v;
      }
      if (__any(moved)) {
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) {
          acc[t][dn][0] *= alpha; acc[t][dn][1] *= alpha; acc[t][dn][2] *= alpha; acc[t][dn][3] *= alpha;
        }
      }
    }
    // ---- O^T += V^T . P^T ----------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const s16x4_t lo = vlo[kk][dn], hi = vhi[kk][dn];
        const bf16x8_t a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[0][dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[0][kk], acc[0][dn], 0, 0, 0);
        acc[1][dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[1][kk], acc[1][dn], 0, 0, 0);
      }
    }
  };
  for (int tile = 0; tile < ntiles; tile += 2) {
    do_tile(kregA, vregA, tile);
    if (tile + 1 < ntiles) do_tile(kregB, vregB, tile + 1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float lsum = lrow[t];
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (qok[t]) {
      const float inv = 1.f / lsum;
      uint16_t* orow = out + ((size_t)(q0 + 16 * t + c) * nh + head) * HD;
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const uint2 pk = make_uint2(pack_bf16x2(acc[t][dn][0] * inv, acc[t][dn][1] * inv),
                                    pack_bf16x2(acc[t][dn][2] * inv, acc[t][dn][3] * inv));
        *reinterpret_cast<uint2*>(orow + 16 * dn + 4 * g) = pk;
      }
    }
  }
  }  // pass
}


// =====================================================================================
// context encoding, second generation: 32x32x16 MFMAs, K/V tiles by LDS-DMA, two waves per SIMD
// =====================================================================================
// What the counters of the kernel above asked for (17 % MFMA-busy at one wave per
// SIMD, 408 registers, 144 accumulator moves and a ds_write staging pass per 64-key tile):
//   * a wave owns ONE 32-query block of one q head: S^T[key][q] = K . Q^T on v_mfma_f32_32x32x16_bf16
//     puts a query on a lane (its 32 scores of a 64-key tile in 32 registers of lanes l and l + 32), so
//     the row maximum / sum is 31 VALU ops and ONE v_permlane32_swap, and the exponentiated
//     accumulators, packed pairwise to bf16, ARE the B operand of O^T[d][q] += V^T . P^T (k-slot j of
//     lane half h <-> key 16 s + 8 (j >> 2) + 4 h + (j & 3): the transposed V reads are bound to that order);
//   * K and V tiles (64 keys) arrive by LDS-DMA (global_load_lds, 1 KiB per wave-instruction) into a
//     ring of three 32 KiB stages: no staging registers, no ds_write, two tiles in flight while one is
//     multiplied, retired by a counted vmcnt in front of ONE raw barrier per tile.  The paged pool is
//     addressed per piece (4 keys x 256 B at head_dim 128): the block id is a scalar load, the per-lane
//     part of the source address is constant for the whole kernel;
//   * the LDS image is rows of head_dim bf16 whose 16-byte chunks are XOR-swizzled by the key (applied
//     to the per-lane SOURCE address of the DMA and to the reads): conflict-free for the ds_read_b128
//     row reads of K (16 lanes = 16 keys = 16 different slots) and for the ds_read_b64_tr_b16
//     transposed reads of V (a half-wave's 4 keys land on 4 different 64-byte windows);
//   * ~170 registers: two waves per SIMD.  A work-group is NW waves = QB query blocks x Gp q heads of
//     one kv group, all sharing the K/V ring.  Causal work grows with the block index, so the QB blocks
//     of a work-group are complementary pairs (j, nblk - 1 - j): every SIMD of the chip gets the same
//     number of key tiles, and the light block's waves keep issuing DMAs for the heavy one's tail.
// Softmax: the running maximum used in the exponent is only moved when the tile's maximum exceeds it
// by more than kDeferLog2 (deferred rescale): p <= 2^kDeferLog2 instead of <= 1, O and l carry the
// same factor, the quotient O / l is unchanged up to rounding; the 64-register rescale of O^T becomes rare.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr float kDeferLog2 = 6.f;

// {v of this lane, v of lane ^ 32} in some order (max / sum do not care): one v_permlane32_swap.  The elements of the
// builtin's result are copied out before the bit cast: hipcc 7.2 reads element 0 for __builtin_bit_cast(float, r[1])
// (hipcc notes) -- which made a row's sum 2 x (its lower half's sum).
__device__ __forceinline__ f32x2_t swap32_pair(float v) {
  const unsigned a = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return f32x2_t{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}

template <int HD>
__device__ __forceinline__ int kv_swz(int key) {   // XOR key of the 16-byte chunk index of a K/V row in LDS
  if constexpr (HD == 128) return ((key & 3) << 2) | ((key >> 2) & 3);          // 256-byte rows, 16 chunks
  else return (((key >> 1) & 1) << 2) | ((key >> 2) & 3);                        // 128-byte rows,  8 chunks
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_prefill2_kernel(
    const uint16_t* __restrict__ q, int T, int q_pos0, const uint16_t* __restrict__ kpool,
    const uint16_t* __restrict__ vpool, int bs, int bs_shift, const int32_t* __restrict__ block_table, int nh, int nkv,
    int G, int Gp, int nblk, uint16_t* __restrict__ out, float scale_log2e) {
  constexpr int TK = 64;                       // keys per tile
  constexpr int ROWB = HD * 2;                 // bytes per K/V row
  constexpr int CPR = HD / 8;                  // 16-byte chunks per row
  constexpr int KPP = 1024 / ROWB;             // keys per 1 KiB DMA piece (4 / 8)
  constexpr int NP = TK / KPP;                 // pieces per K (or V) tile (16 / 8)
  constexpr int KBYTES = TK * ROWB;            // 16 / 8 KiB
  constexpr int STAGE = 2 * KBYTES;
  constexpr int NSTAGE = 3;
  constexpr int PPW = NP / NW > 0 ? NP / NW : 1;   // K pieces per wave and tile (V the same)
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  constexpr int CNT = 2 * PPW;                 // DMA instructions per wave and tile
  constexpr int KSTEPS = HD / 16, DB = HD / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kvh = blockIdx.y;
  const int QB = NW / Gp;
  const int hl = wave % Gp, slot = wave / Gp;
  const int head = kvh * G + min(hl, G - 1);
  const bool head_ok = hl < G;
  const int kv_len = q_pos0 + T;
  const int last_piece_key = ((kv_len - 1) / KPP) * KPP;
  const int qq = lane & 31, h = lane >> 5;
  const int npair = (nblk + 1) >> 1;

  // ---- per-lane constants of the LDS image -------------------------------------------------------
  // DMA: lane i of a piece fills slot (i % CPR) of key (i / CPR) of the piece: it fetches the chunk the swizzle puts there
  const int kip = lane / CPR, dslot = lane % CPR;
  int dma_sw;
  if constexpr (HD == 128) dma_sw = (kip << 2) | (wave & 3);                 // kv_swz(4 piece + kip), piece = wave (mod 4)
  else dma_sw = (((kip >> 1) & 1) << 2) | ((2 * (wave & 1) + (kip >> 2)) & 3);   // kv_swz(8 piece + kip), piece = wave (mod 2)
  const int dma_lane_off = kip * HD + ((dslot ^ dma_sw) * 8);                // elements, relative to the piece's first row
  // K row reads (A operand of S^T): key qq of a 32-key block, chunk 2 ks + h
  int kaddr[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) kaddr[ks] = qq * ROWB + (((2 * ks + h) ^ kv_swz<HD>(qq)) * 16);
  // V transposed reads (A operand of O^T): group gi = lane >> 4 covers dims 16 (gi & 1) .. + 15 of a 32-dim block for lane half
  // h = gi >> 1; lane 4 qr + pc of the group addresses key row (4 h + qr [+ 8]) at dims 4 pc .. 4 pc + 3
  const int gi = lane >> 4, qr = (lane & 15) >> 2, pc = lane & 3;
  int vaddr[DB][2];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = 4 * h + qr + 8 * u;                                     // (+ 32 b + 16 s: no effect on the swizzle key)
      const int chunk = 4 * db + 2 * (gi & 1) + (pc >> 1);
      vaddr[db][u] = key * ROWB + ((chunk ^ kv_swz<HD>(key)) * 16) + 8 * (pc & 1);
    }

  const int passes = QB == 1 ? 2 : 1;
  for (int pass = 0; pass < passes; ++pass) {
    // ---- which query block this wave owns --------------------------------------------------------
    int blk_q, pair;
    bool blk_ok;
    if (QB == 1) {                       // one block at a time: the heavy one of pair blockIdx.x, then the light one
      pair = blockIdx.x;
      blk_q = pass == 0 ? nblk - 1 - pair : pair;
      blk_ok = pair < npair && !(pass == 1 && pair == nblk - 1 - pair);
    } else {
      pair = (int)blockIdx.x * (QB >> 1) + (slot >> 1);
      blk_q = (slot & 1) ? nblk - 1 - pair : pair;
      blk_ok = pair < npair && !((slot & 1) && pair == nblk - 1 - pair);
    }
    const int q0 = blk_q * 32;
    const bool active = blk_ok && head_ok;
    // tiles: this wave's, and the work-group's (the heaviest block any of its waves owns)
    const int p_first = q_pos0 + min(q0, T - 1), p_last = q_pos0 + min(q0 + 31, T - 1);
    const int nt_wave = active ? p_last / TK + 1 : 0;
    int wg_pair0 = QB == 1 ? (int)blockIdx.x : (int)blockIdx.x * (QB >> 1);
    int wg_blk = QB == 1 ? (pass == 0 ? nblk - 1 - wg_pair0 : wg_pair0) : nblk - 1 - wg_pair0;   // heaviest block of the work-group
    wg_blk = max(min(wg_blk, nblk - 1), 0);
    const int nt = (q_pos0 + min(wg_blk * 32 + 31, T - 1)) / TK + 1;

    // ---- Q fragments (B operand of S^T): Q[q0 + qq][16 ks + 8 h ..] ---------------------------------
    const int qi = min(q0 + qq, T - 1);
    const int qpos = q_pos0 + qi;
    u32x4_t qf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      qf[ks] = active ? *reinterpret_cast<const u32x4_t*>(q + ((size_t)qi * nh + head) * HD + ks * 16 + h * 8) : u32x4_t{0, 0, 0, 0};

    f32x16_t o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_used = -INFINITY, lsum = 0.f;

    // Block ids are fetched (scalar loads) one tile ahead of the DMAs that use them: the round trip to the scalar cache / L2
    // runs under a tile's arithmetic instead of in front of the global_load_lds it feeds.
    int pf_blk[PPW], pf_off[PPW];
    auto fetch_ids = [&](int t) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int k0 = min(t * TK + (wave + NW * i) * KPP, last_piece_key);   // rows past the context: the last piece again (masked by position)
        int bi;
        if (bs_shift >= 0) { bi = k0 >> bs_shift; pf_off[i] = k0 & (bs - 1); }
        else { bi = k0 / bs; pf_off[i] = k0 - bi * bs; }
        pf_blk[i] = block_table[bi];
      }
    };
    auto issue = [&](int st) {                  // the tile whose ids fetch_ids loaded last
      unsigned char* base = smem + st * STAGE;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const size_t row = ((size_t)pf_blk[i] * nkv + kvh) * bs + pf_off[i];
        glds16(kpool + row * HD + dma_lane_off, base + piece * 1024);
        glds16(vpool + row * HD + dma_lane_off, base + KBYTES + piece * 1024);
      }
    };

    int st_issue = 0, st_read = 0;
    fetch_ids(0);
    issue(0);
    if (nt > 1) { fetch_ids(1); issue(1); }
    fetch_ids(2);
    st_issue = 2;
    // hipcc waits for a plain load at its first use -- inside the tile loop, as vmcnt(0) on every iteration, which would drain
    // the two tiles in flight each time.  Taking the Q fragments through an empty asm here puts that wait in front of the loop.
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(qf[ks]));
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // every wave's pieces of tile t are in; everyone is done reading tile t - 1
      if (t + 2 < nt) {
        issue(st_issue);
        st_issue = st_issue == NSTAGE - 1 ? 0 : st_issue + 1;
        fetch_ids(t + 3);
      }
      const unsigned char* Kst = smem + st_read * STAGE;
      const unsigned char* Vst = Kst + KBYTES;
      st_read = st_read == NSTAGE - 1 ? 0 : st_read + 1;
      if (t >= nt_wave) continue;                // wave-uniform: the tile lies in this block's future (or the slot is idle)

      // ---- S^T = K . Q^T : two 32-key blocks.  The K fragments of a block are requested one block ahead of the MFMAs
      // that take them (8 reads in flight; hipcc left to itself issues read -> wait -> MFMA one at a time) ----
      f32x16_t sc[2];
      {
        bf16x8_t kf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(Kst + kaddr[ks]);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[b][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            sc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], __builtin_bit_cast(bf16x8_t, qf[ks]), sc[b], 0, 0, 0);
            if (b == 0) kf[ks] = *reinterpret_cast<const bf16x8_t*>(Kst + 32 * ROWB + kaddr[ks]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, KSTEPS, 0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      }
      // ---- online softmax: this lane holds 32 scores of query qq (keys (r & 3) + 8 (r >> 2) + 4 h of each block) ----
      const bool need_mask = t * TK + TK - 1 > p_first;
      if (need_mask) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * TK + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h > qpos) sc[b][r] = -INFINITY;
      }
      float mx = sc[0][0];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[b][r]);
      {
        const f32x2_t sw = swap32_pair(mx);
        mx = fmaxf(sw[0], sw[1]);
      }
      const float ms = mx * scale_log2e;         // key 0 is visible to every query: finite from the first tile on
      const bool moved = ms > m_used + kDeferLog2;
      if (__any(moved)) {
        const float mn = moved ? ms : m_used;
        const float alpha = moved ? fexp2(m_used - mn) : 1.f;   // exp2(-inf) = 0 on the first tile
        m_used = mn;
        lsum *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
      // ---- p = exp2(s - m), packed to bf16 = the B operands of O^T += V^T . P^T.  The transposed V fragments of the first
      // kVAhead MFMAs are requested in front of the exponentials (no dependence), the rest kVAhead MFMAs ahead of their use ----
      // The transposed reads are inline asm: behind the builtin hipcc assumes the read may alias the LDS-DMAs in flight and
      // drains them (s_waitcnt vmcnt(0) per tile).  Their completion is counted by hand: LDS reads return in order, and between
      // the first V read of a tile and its last MFMA the wave issues no other LDS or scalar-memory instruction.
      constexpr int NPV = 4 * DB;               // MFMAs: (b, s2) outer, d-block inner
      constexpr int kVAhead = 4;
      s16x4_t vlo[NPV], vhi[NPV];
      unsigned vbase[DB][2];
      const unsigned vst_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Vst;
#pragma unroll
      for (int db = 0; db < DB; ++db) { vbase[db][0] = vst_lds + vaddr[db][0]; vbase[db][1] = vst_lds + vaddr[db][1]; }
#define MI_VREAD(i_)                                                                                                         \
      do {                                                                                                                   \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[i_]) : "v"(vbase[(i_) % DB][0]), "n"(16 * ((i_) / DB) * ROWB)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[i_]) : "v"(vbase[(i_) % DB][1]), "n"(16 * ((i_) / DB) * ROWB)); \
      } while (0)
      MI_VREAD(0); MI_VREAD(1); MI_VREAD(2); MI_VREAD(3);
      static_assert(kVAhead == 4, "the prefetch above is written out for 4");
      bf16x8_t pb[4];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = fexp2(fmaf(sc[b][r], scale_log2e, -m_used));   // masked: exp2(-inf) = 0
          lsum += pv[r];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
          for (int k = 0; k < 4; ++k) pk.w[k] = pack_bf16x2(pv[8 * s2 + 2 * k], pv[8 * s2 + 2 * k + 1]);
          pb[2 * b + s2] = pk.v;
        }
      }
      // ---- O^T += V^T . P^T --------------------------------------------------------------------------
      // before MFMA i: reads issued = 2 min(i + 1 + kVAhead - 1, NPV) ... all but the ones behind fragment i may be outstanding
#define MI_PV(i_)                                                                                                            \
      do {                                                                                                                   \
        constexpr int issued = 2 * ((i_) + kVAhead < NPV ? (i_) + kVAhead : NPV);                                            \
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vlo[i_]), "+v"(vhi[i_]) : "n"(issued - 2 * ((i_) + 1)));               \
        const s16x4_t lo = vlo[i_], hi = vhi[i_];                                                                            \
        const bf16x8_t a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                         \
        o[(i_) % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[(i_) / DB], o[(i_) % DB], 0, 0, 0);                     \
        if constexpr ((i_) + kVAhead < NPV) MI_VREAD((i_) + kVAhead);                                                        \
      } while (0)
      MI_PV(0); MI_PV(1); MI_PV(2); MI_PV(3); MI_PV(4); MI_PV(5); MI_PV(6); MI_PV(7);
      if constexpr (NPV == 16) { MI_PV(8); MI_PV(9); MI_PV(10); MI_PV(11); MI_PV(12); MI_PV(13); MI_PV(14); MI_PV(15); }
#undef MI_PV
#undef MI_VREAD
    }
    // ---- epilogue: O^T[d][q] / l -> out[q][head][d]; lanes l and l + 32 hold neighbouring 4-dim groups of one query:
    // one v_permlane32_swap per pair of groups makes 16-byte stores of them
    {
      const f32x2_t sw = swap32_pair(lsum);
      lsum = sw[0] + sw[1];
    }
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    const bool st_ok = active && q0 + qq < T;
    uint16_t* orow = out + ((size_t)(q0 + qq) * nh + head) * HD;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int k2 = 0; k2 < 4; k2 += 2) {      // dim groups k2, k2 + 1: dims 32 db + 8 k + 4 h + (0..3)
        uint32_t a0 = pack_bf16x2(o[db][4 * k2 + 0] * inv, o[db][4 * k2 + 1] * inv);
        uint32_t a1 = pack_bf16x2(o[db][4 * k2 + 2] * inv, o[db][4 * k2 + 3] * inv);
        uint32_t b0 = pack_bf16x2(o[db][4 * k2 + 4] * inv, o[db][4 * k2 + 5] * inv);
        uint32_t b1 = pack_bf16x2(o[db][4 * k2 + 6] * inv, o[db][4 * k2 + 7] * inv);
        const auto x = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        // lower half: [own group k2 | upper's group k2] = dims 8 k2 .. 8 k2 + 7; upper half: [lower's k2 + 1 | own k2 + 1]
        const u32x4_t v = {x[0], y[0], x[1], y[1]};
        if (st_ok) *reinterpret_cast<u32x4_t*>(orow + 32 * db + 8 * (k2 + h)) = v;
      }
    __syncthreads();   // the ring is reused by the next pass
  }
}

static bool attn_prefill_v1() {   // MI355X_ATTN_PREFILL_V1=1: the first-generation kernel (A/B)
  static const bool on = [] { const char* v = getenv("MI355X_ATTN_PREFILL_V1"); return v && v[0] == '1'; }();
  return on;
}

int launch_attn_prefill(const uint16_t* q, int T, int q_pos0, const uint16_t* kpool, const uint16_t* vpool,
                        int block_size, const int32_t* block_table, int nh, int nkv, int hd, uint16_t* out,
                        hipStream_t s) {
  MI_CHECK(hd == 64 || hd == 128, "attention: head_dim must be 64 or 128");
  MI_CHECK(T >= 1 && q_pos0 >= 0, "attention: bad T / q_pos0");
  MI_CHECK(nh % nkv == 0 && nh / nkv <= 8, "attention: q heads per kv head must be 1..8");
  MI_CHECK(ceil_div(q_pos0 + T, block_size) <= kPrefillMaxBlocks, "attention: context spans more than 4096 blocks");
  const int G = nh / nkv;
  const int Gp = G <= 1 ? 1 : (G <= 2 ? 2 : (G <= 4 ? 4 : 8));
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
  if (!attn_prefill_v1()) {
    // second generation: 8 waves = (8 / Gp) query blocks x Gp heads; blocks in complementary pairs
    const int nblk = ceil_div(T, 32), npair = (nblk + 1) / 2, qb = 8 / Gp;
    int bs_shift = -1;
    for (int sh = 4; sh < 20; ++sh) if ((1 << sh) == block_size) bs_shift = sh;
    const dim3 grid2(ceil_div(npair, qb >= 2 ? qb / 2 : 1), nkv);
    if (hd == 128) {
      constexpr int lds = 3 * 2 * 64 * 256;
      MI_TRY_A(ensure_dynamic_lds(reinterpret_cast<const void*>(attn_prefill2_kernel<128, 8>), lds));
      hipLaunchKernelGGL((attn_prefill2_kernel<128, 8>), grid2, dim3(512), lds, s, q, T, q_pos0, kpool, vpool, block_size, bs_shift,
                         block_table, nh, nkv, G, Gp, nblk, out, scale_log2e);
    } else {
      constexpr int lds = 3 * 2 * 64 * 128;
      MI_TRY_A(ensure_dynamic_lds(reinterpret_cast<const void*>(attn_prefill2_kernel<64, 8>), lds));
      hipLaunchKernelGGL((attn_prefill2_kernel<64, 8>), grid2, dim3(512), lds, s, q, T, q_pos0, kpool, vpool, block_size, bs_shift,
                         block_table, nh, nkv, G, Gp, nblk, out, scale_log2e);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  const int waves = Gp < 4 ? 4 : Gp, QB = waves / Gp;
  const int nqb = ceil_div(T, 32 * QB);
  const int paired = nqb * nkv > 384 ? 1 : 0;   // more blocks than 1.5 per CU: pair complementary ones
  const dim3 grid(paired ? ceil_div(nqb, 2) : nqb, nkv);
#define MI_PF(HD_, W_) hipLaunchKernelGGL((attn_prefill_kernel<HD_, W_>), grid, dim3(W_ * 64), 0, s, q, T, q_pos0, kpool, vpool, block_size, block_table, nh, nkv, G, Gp, paired, out, scale_log2e)
  if (hd == 128) { if (waves == 4) MI_PF(128, 4); else MI_PF(128, 8); }
  else { if (waves == 4) MI_PF(64, 4); else MI_PF(64, 8); }
#undef MI_PF
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
