// Launch-side declarations of the weight-quantized linear layer (reference K3 of
// the NxDI quantized QKV/O/MLP/lm_head linears reached through
// /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:339-348).
#pragma once
#include "mi_common.h"

namespace mi {

// What happens to the fp32 accumulator tile D[n][m] once the contraction is done.
enum { EPI_F32 = 0, EPI_QKV = 1, EPI_SWIGLU = 2, EPI_RESID = 3 };
// Where the bf16 activations come from.
enum { PRO_BF16 = 0, PRO_NORM = 1, PRO_NORM_PARTIAL = 2 };  // callers pass PRO_BF16 / PRO_NORM; PARTIAL is picked when ProArgs::partial is set

struct EpiArgs {
  const float* scale;  // [N] per-output-row dequant scale
  const float* row_scale;  // [M] per-token activation scale (FP8-activation GEMM) or nullptr
  const float* bias;   // [N] or nullptr
  // EPI_F32: out[m * ld_out + n] = y;  EPI_RESID: out = resid_in[m * ld_out + n] + y
  float* out_f32;
  int ld_out;
  const float* resid_in;
  // EPI_QKV: rows [0,q_dim) -> RoPE -> q_out; [q_dim, q_dim+kv_dim) -> RoPE -> K pool;
  // the rest -> V pool.  Rows of every q/k head are stored pair-interleaved
  // (2i <- d=i, 2i+1 <- d=i+hd/2) so a lane owns both halves of a rotation.
  uint16_t* q_out;  // [M, q_dim] bf16
  int q_dim, kv_dim, hd, nkv;
  const int32_t* pos;    // [M]
  const int32_t* slots;  // [M], -1 = no write
  const float* rope_cos; // [max_pos, hd/2]
  const float* rope_sin;
  uint16_t* kpool;       // this layer's [num_blocks][nkv][block_size][hd]
  uint16_t* vpool;
  int block_size;
  // EPI_SWIGLU: rows interleaved (gate_i, up_i); act[m * ld_act + i] = silu(gate)*up
  uint16_t* act_out;
  int ld_act;
};

struct ProArgs {
  // PRO_BF16
  const uint16_t* x;  // [M, K] bf16
  int ldx;
  // PRO_NORM: h = resid_in (+ partial); block 0 stores h to resid_out; x = rmsnorm(h) * gain
  const float* resid_in;
  const float* partial;  // nullptr = none
  float* resid_out;      // nullptr = do not store
  const float* gain;
  float eps;
};

struct LinearW {
  const void* w;  // tiled, see mi_common.h
  int N, K, wd;
};

// M <= 16 token-generation path (HBM-bound weight streaming).
int launch_gemv(const LinearW& w, int M, int pro, const ProArgs& p, int epi, const EpiArgs& e,
                hipStream_t s);
// M > 16 context-encoding path (MFMA-bound).  x is always bf16 [T, K].
// defer != null: a K-split residual projection (EPI_RESID) leaves its slabs unsummed and describes them in *defer
// (slab == null: nothing deferred); the caller hands them to the next row norm or to launch_splitk_flush
int launch_gemm(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e,
                hipStream_t s, float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0, SlabSum* defer = nullptr);
int launch_splitk_flush(const SlabSum& sl, const float* resid_in, float* out, int ld_out, hipStream_t s);
// the wide-N LDS-DMA variant of launch_gemm (1-byte weights); launch_gemm picks it by size
int launch_gemm_wide(const LinearW& w, int T, const uint16_t* x, int ldx, int epi, const EpiArgs& e, hipStream_t s,
                     float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0, SlabSum* defer = nullptr, int force_bm = 0,
                     int force_ks = 0);   // force_*: tests (token block 128 / 256, K slices)
// M > 16 with FP8 activations: x8 = K-step-major e4m3 image [K / 128][ldx >= T rows][128 B] as
// launch_rowquant_fp8 writes it (per-token scale in EpiArgs::row_scale), FP8 weights,
// MX-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) at twice the bf16 rate.
int launch_gemm_a8(const LinearW& w, int T, const uint8_t* x8, int ldx, int epi, const EpiArgs& e,
                   hipStream_t s, float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0, SlabSum* defer = nullptr);
bool gemm_a8_supported(const LinearW& w);
// bf16 rows -> K-step-major e4m3 image [K / 128][T][128 B] + per-row scale (amax / 448; all-zero row -> 1)
int launch_rowquant_fp8(const uint16_t* x, int T, int K, int ldx, uint8_t* x8, float* row_scale, hipStream_t s);
size_t gemv_lds_bytes(int M, int K);
bool gemv_fits(int M, int K);

// ---- load-time quantize + tile ------------------------------------------------------
enum { ROWMAP_PLAIN = 0, ROWMAP_ROPE_PAIRS = 1, ROWMAP_EVERY_OTHER = 2 };
struct QuantJob {
  const float* src;    // row-major [src_rows_total, ld] fp32 (the FULL HF tensor)
  int ld;              // source row stride (elements)
  int src_row0;        // first source row of this shard
  int src_col0;        // first source column of this shard
  int n_rows;          // rows taken from the source
  int K;               // columns taken from the source (= destination K)
  int rowmap;          // how source row i lands in the destination
  int hd;              // ROWMAP_ROPE_PAIRS: head size
  int dst_row0;        // first destination row (ROWMAP_EVERY_OTHER: row of element 0)
  int wd, quant_type;
  void* dst;           // tiled destination matrix [dstN, K]
  float* dst_scale;    // [dstN]
  float* tmp_rowmax;   // [src_rows_total + 1] scratch
  int src_rows_total;  // rows of the full tensor (per-tensor scale spans all of them)
  // zero padding (q heads a rank holds only as padding, model.hip): the LAST pad_rows of the n_rows taken and
  // the LAST pad_cols of the K columns taken are not read from the source: zeros (padding rows: scale 1)
  int pad_rows, pad_cols;
};
int run_quant_job(const QuantJob& j, hipStream_t s);
int launch_untile(const void* tiled, int N, int K, int wd, void* out, hipStream_t s);
int launch_bf16_to_f32(const uint16_t* in, float* out, size_t n, hipStream_t s);

}  // namespace mi
