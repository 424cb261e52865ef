// libmi355x_vllm: context, weight loading and the model call behind include/mi355x_vllm.h.
//
// mi_forward is the MI355X counterpart of the NxDI model __call__ that the reference reaches
// at /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:339-348 and of the
// logits[:, -1, :] slice at :363.  One host thread, one HIP stream, library-owned device
// memory (weights, paged KV pool, activations), caller-owned host arrays.
//
// Per decoder layer (token generation, 6 launches, captured in a hipGraph per batch size):
//   gemv[norm -> QKV -> bias -> RoPE -> q buffer + KV-pool scatter]
//   paged attention (split) ; combine
//   gemv[O proj -> fp32 partial]                     (+ RCCL all-reduce when tp_degree > 1)
//   gemv[residual add + norm -> gate|up -> SwiGLU]
//   gemv[down proj -> fp32 partial]                  (+ RCCL all-reduce)
// The residual stream is fp32 and ping-pongs between two buffers: the norm prologue of the
// NEXT kernel folds in the previous partial, so a row-parallel output needs no extra pass.

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <array>
#include <string>
#include <vector>

#include "attn_kernels.h"
#include "linear_kernels.h"
#include "misc_kernels.h"

namespace mi {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace mi

using namespace mi;

#define MI_TRY(expr)             \
  do {                           \
    int _rc = (expr);            \
    if (_rc != MI_OK) return _rc; \
  } while (0)

#include "model_internal.h"
#include "tp_group.h"

namespace {

// ---- memory helpers ---------------------------------------------------------------------
template <typename T>
int dmalloc(T** p, size_t n, size_t* tally = nullptr) {
  MI_HIP(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  if (tally) *tally += n * sizeof(T);
  return MI_OK;
}

int alloc_linear(mi_ctx* c, Linear& L, int N, int K, int wd, bool bias) {
  L.N = N; L.K = K; L.wd = wd;
  MI_CHECK(N % 16 == 0 && K % 64 == 0, "linear dims: N % 16 == 0 and K % 64 == 0 required (per TP shard)");
  MI_HIP(hipMalloc(&L.w, L.bytes()));
  c->weight_bytes += L.bytes();
  MI_TRY(dmalloc(&L.scale, (size_t)N, &c->weight_bytes));
  if (bias) {
    MI_TRY(dmalloc(&L.bias, (size_t)N, &c->weight_bytes));
    MI_HIP(hipMemsetAsync(L.bias, 0, (size_t)N * 4, c->stream));
  }
  return MI_OK;
}

// ---- profiling wrapper ---------------------------------------------------------------------
struct Scope {
  mi_ctx* c;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  Scope(mi_ctx* c_, int cls_) : c(c_), cls(cls_) {
    if (c->prof.on) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, c->stream);
    }
  }
  ~Scope() {
    if (c->prof.on) {
      hipEventRecord(b, c->stream);
      c->prof.ev.push_back(a);
      c->prof.ev.push_back(b);
      c->prof.cls.push_back(cls);
    }
  }
};

int prof_collect(mi_ctx* c) {
  if (!c->prof.on || c->prof.ev.empty()) return MI_OK;
  MI_HIP(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->prof.cls.size(); ++i) {
    float ms = 0.f;
    MI_HIP(hipEventElapsedTime(&ms, c->prof.ev[2 * i], c->prof.ev[2 * i + 1]));
    c->prof.ms[c->prof.cls[i]] += ms;
    c->prof.launches[c->prof.cls[i]] += 1;
    hipEventDestroy(c->prof.ev[2 * i]);
    hipEventDestroy(c->prof.ev[2 * i + 1]);
  }
  c->prof.ev.clear();
  c->prof.cls.clear();
  return MI_OK;
}

// ---- linear dispatch --------------------------------------------------------------------
// rows x K activations -> epilogue.  Token-generation sized inputs stream the weights
// (GEMV); larger ones go through rmsnorm rows + MFMA GEMM.
// a deferred K-split sum (SlabSum) that no row norm took over: sum it the plain way
int flush_pending(mi_ctx* c) {
  if (!c->pend.slab) return MI_OK;
  Scope sc(c, MI_K_GEMM);
  const SlabSum sl = c->pend;
  c->pend = SlabSum();
  return launch_splitk_flush(sl, c->pend_in, c->pend_out, sl.N, c->stream);
}

int run_linear(mi_ctx* c, const Linear& L, int rows, int pro, ProArgs p, int epi, EpiArgs e) {
  e.scale = L.scale;
  e.bias = L.bias;
  if (gemv_fits(rows, L.K)) {
    if (!c->runs(MI_K_GEMV)) return MI_OK;
    Scope sc(c, MI_K_GEMV);
    if (c->prof.on) c->prof.gemv_bytes += (double)L.bytes();
    return launch_gemv(L.view(), rows, pro, p, epi, e, c->stream);
  }
  const uint16_t* x = p.x;
  int ldx = p.ldx;
  const bool a8 = c->cfg.prefill_fp8_activations && gemm_a8_supported(L.view());
  if (pro == PRO_NORM) {
    Scope sc(c, MI_K_OTHER);
    // A K-split residual projection in front of this norm left its slabs unsummed: the norm sums them on the way
    // (sum, scale, bias, + residual: the arithmetic of splitk_reduce + epilogue) and writes the residual stream.
    const SlabSum* slabs = nullptr;
    const float* rin = p.resid_in;
    float* rout = p.resid_out;
    if (c->pend.slab) {
      if (p.resid_in == c->pend_out && !p.partial && !p.resid_out && rows == c->pend.T && L.K == c->pend.N) {
        slabs = &c->pend;
        rin = c->pend_in;
        rout = c->pend_out;
      } else {
        MI_TRY(flush_pending(c));
      }
    }
    if (a8) {   // norm and per-token quantization in one pass over the row
      MI_TRY(launch_norm_rows_fp8(rin, p.partial, rout, p.gain, rows, L.K, p.eps, c->x8, c->x8_scale, c->stream, slabs));
    } else {
      MI_TRY(launch_norm_rows(rin, p.partial, rout, p.gain, rows, L.K, p.eps, c->xn, c->stream, slabs));
      x = c->xn;
      ldx = L.K;
    }
    c->pend = SlabSum();
  }
  // a residual projection whose K-split sum the next norm can take over (one work-group per row there)
  static const bool fuse = [] { const char* v = getenv("MI355X_FUSE_SPLITK_NORM"); return !v || v[0] != '0'; }();
  // (from 384 rows on -- the 512 bucket holds 495: the norm runs one work-group per row -- with few rows the K-split sum is the wider kernel on its
  //  own; measured at 32 rows: 14.2 us fused against 5.0 + 4.6 us)
  static const int fuse_min_rows = [] { const char* v = getenv("MI355X_FUSE_SPLITK_MIN_ROWS"); return v ? atoi(v) : 384; }();
  SlabSum* defer = (fuse && epi == EPI_RESID && rows >= fuse_min_rows && !c->pend.slab) ? &c->pend : nullptr;
  auto note_deferred = [&]() {
    if (defer && c->pend.slab) { c->pend_in = e.resid_in; c->pend_out = e.out_f32; }
  };
  if (a8) {
    if (pro != PRO_NORM) {
      Scope sc(c, MI_K_OTHER);
      MI_TRY(launch_rowquant_fp8(x, rows, L.K, ldx, c->x8, c->x8_scale, c->stream));
    }
    e.row_scale = c->x8_scale;
    Scope sc(c, MI_K_GEMM);
    MI_TRY(launch_gemm_a8(L.view(), rows, c->x8, rows, epi, e, c->stream, c->splitk_ws, c->splitk_ws_bytes, defer));
    note_deferred();
    return MI_OK;
  }
  Scope sc(c, MI_K_GEMM);
  MI_TRY(launch_gemm(L.view(), rows, x, ldx, epi, e, c->stream, c->splitk_ws, c->splitk_ws_bytes, defer));
  note_deferred();
  return MI_OK;
}

// Host-side stage timing of the token-generation call (MI355X_HOST_TIMING=1: printed at destroy).
struct HostTiming {
  bool on = getenv("MI355X_HOST_TIMING") != nullptr;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  long calls = 0;
  std::chrono::steady_clock::time_point t0;
  void start() { if (on) t0 = std::chrono::steady_clock::now(); }
  void lap(int i) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    acc[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t0 = t1;
  }
  void report() const {
    if (!on || !calls) return;
    fprintf(stderr, "[mi355x host timing] %ld decode calls, us/call: stage+validate %.1f | H2D enqueue %.1f | launch %.1f | "
                    "sample/fetch enqueue %.1f | wait %.1f | copy-out %.1f\n", calls, acc[0] / calls, acc[1] / calls,
            acc[2] / calls, acc[3] / calls, acc[4] / calls, acc[5] / calls);
  }
};
static HostTiming g_ht;

int all_reduce_partial(mi_ctx* c, int rows) {
  if (!c->collective()) return MI_OK;
  Scope sc(c, MI_K_COMM);
  if (c->grp) return group_all_reduce(c, c->partial, (size_t)rows * c->H);
  if (c->xport_allreduce) {
    if (c->xport_allreduce(c->xport_user, c->partial, (size_t)rows * c->H, c->stream) != 0) {
      set_error("all-reduce transport callback failed");
      return MI_ECOMM;
    }
    return MI_OK;
  }
  ncclResult_t r = ncclAllReduce(c->partial, c->partial, (size_t)rows * c->H, ncclFloat, ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) {
    set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return MI_ECOMM;
  }
  return MI_OK;
}

// The decoder stack over `rows` token rows whose ids/positions/slots are already in
// d_ids/d_pos/d_slots.  decode: rows = B sequences (block tables d_bt [B, MB], lengths d_ctx).
// prefill: rows = new tokens of ONE sequence at positions q_pos0.. (block table d_bt[0, :]).
// segs != null: a RAGGED context-encoding batch (chunked prefill, reference runner.py:938-1051):
// the rows are the concatenated chunks of nseg requests; request i owns rows [row0, row0 + n) at
// positions pos0.. and block-table row i.  The projections see one [rows, K] matrix (the weights
// cross the chip once for all requests); attention runs per request; the logits are those of
// each request's last row (gathered through d_ctx, which holds the last-row indices).
struct Seg { int row0, n, pos0; };
static bool attn_early_kv() {   // MI355X_ATTN_EARLY_KV=0: the first K/V request of the decode attention waits for the context length
  static const bool on = [] { const char* v = getenv("MI355X_ATTN_EARLY_KV"); return !v || v[0] != '0'; }();
  return on;
}

int run_layers(mi_ctx* c, int rows, bool decode, int B, int MB, int q_pos0, int logits_rows,
               int logits_row0, const Seg* segs = nullptr, int nseg = 0) {
  const mi_model_config& k = c->cfg;
  hipStream_t s = c->stream;
  c->pend = SlabSum();   // (a call that failed half way may have left one behind)
  if (c->runs(MI_K_OTHER)) {
    Scope sc(c, MI_K_OTHER);
    MI_TRY(launch_embed(c->d_ids, c->embed, rows, c->H, c->resid[0], s));
  }
  int cur = 0;
  bool have_partial = false;
  for (int l = 0; l < k.num_layers; ++l) {
    const LayerW& W = c->layers[l];
    uint16_t* kpool = c->kv_pool + (size_t)l * 2 * c->kv_half;
    uint16_t* vpool = kpool + c->kv_half;
    {  // norm -> QKV -> RoPE -> q / KV pool
      ProArgs p{};
      p.resid_in = c->resid[cur];
      p.partial = have_partial ? c->partial : nullptr;
      p.resid_out = have_partial ? c->resid[cur ^ 1] : nullptr;
      p.gain = W.g_in;
      p.eps = k.rms_norm_eps;
      EpiArgs e{};
      e.q_out = c->qbuf; e.q_dim = c->q_dim; e.kv_dim = c->kv_dim; e.hd = c->hd; e.nkv = c->nkv_l;
      e.pos = c->d_pos; e.slots = c->d_slots; e.rope_cos = c->rope_cos; e.rope_sin = c->rope_sin;
      e.kpool = kpool; e.vpool = vpool; e.block_size = k.block_size;
      MI_TRY(run_linear(c, W.qkv, rows, PRO_NORM, p, EPI_QKV, e));
      if (have_partial) cur ^= 1;
    }
    if (decode && !c->runs(MI_K_ATTN_DECODE)) {
      // (class-masked replay: see mi_replay_decode_classes)
    } else if (decode) {
      Scope sc(c, MI_K_ATTN_DECODE);
      MI_TRY(launch_attn_decode(c->qbuf, kpool, vpool, k.block_size, c->d_bt, MB, c->d_ctx, B, c->nh_l,
                                c->nkv_l, c->hd, c->attn_out, c->attn_scratch, s, /*tickets_zeroed=*/true,
                                c->attn_rows_per_seq, attn_early_kv() ? k.num_blocks : 0));
    } else if (segs) {
      Scope sc(c, MI_K_ATTN_PREFILL);
      static const bool batch_decodes = [] { const char* v = getenv("MI355X_RAGGED_DECODE_ATTN"); return !v || v[0] != '0'; }();
      for (int i = 0; i < nseg;) {
        int j = i;
        while (batch_decodes && j < nseg && segs[j].n == 1) ++j;
        if (j > i) {
          // a run of requests that generate (one token each, consecutive rows and table rows): ONE split-context
          // decode-attention launch for all of them instead of a one-query context encoding per request
          MI_TRY(launch_attn_decode(c->qbuf + (size_t)segs[i].row0 * c->q_dim, kpool, vpool, k.block_size,
                                    c->d_bt + (size_t)i * MB, MB, c->d_seg_ctx + i, j - i, c->nh_l, c->nkv_l, c->hd,
                                    c->attn_out + (size_t)segs[i].row0 * c->q_dim, c->attn_scratch, s, false, 1,
                                    attn_early_kv() ? k.num_blocks : 0));
          i = j;
          continue;
        }
        MI_TRY(launch_attn_prefill(c->qbuf + (size_t)segs[i].row0 * c->q_dim, segs[i].n, segs[i].pos0, kpool, vpool,
                                   k.block_size, c->d_bt + (size_t)i * MB, c->nh_l, c->nkv_l, c->hd,
                                   c->attn_out + (size_t)segs[i].row0 * c->q_dim, s));
        ++i;
      }
    } else {
      Scope sc(c, MI_K_ATTN_PREFILL);
      MI_TRY(launch_attn_prefill(c->qbuf, rows, q_pos0, kpool, vpool, k.block_size, c->d_bt, c->nh_l,
                                 c->nkv_l, c->hd, c->attn_out, s));
    }
    const bool tp = c->collective();  // a communicator (even of one rank) selects the collective path
    {  // O projection.  TP = 1: straight into the residual stream (resid' = resid + y);
       // TP > 1: fp32 partial -> RCCL all-reduce -> folded in by the next norm prologue.
      ProArgs p{};
      p.x = c->attn_out; p.ldx = c->q_dim;
      EpiArgs e{};
      e.ld_out = c->H;
      if (tp) {
        e.out_f32 = c->partial;
        MI_TRY(run_linear(c, W.o, rows, PRO_BF16, p, EPI_F32, e));
        MI_TRY(all_reduce_partial(c, rows));
      } else {
        e.out_f32 = c->resid[cur ^ 1]; e.resid_in = c->resid[cur];
        MI_TRY(run_linear(c, W.o, rows, PRO_BF16, p, EPI_RESID, e));
        cur ^= 1;
      }
    }
    {  // (residual +) norm -> gate|up -> SwiGLU
      ProArgs p{};
      p.resid_in = c->resid[cur];
      p.partial = tp ? c->partial : nullptr;
      p.resid_out = tp ? c->resid[cur ^ 1] : nullptr;
      p.gain = W.g_post; p.eps = k.rms_norm_eps;
      EpiArgs e{};
      e.act_out = c->act; e.ld_act = c->I_l;
      MI_TRY(run_linear(c, W.gu, rows, PRO_NORM, p, EPI_SWIGLU, e));
      if (tp) cur ^= 1;
    }
    {  // down projection
      ProArgs p{};
      p.x = c->act; p.ldx = c->I_l;
      EpiArgs e{};
      e.ld_out = c->H;
      if (tp) {
        e.out_f32 = c->partial;
        MI_TRY(run_linear(c, W.down, rows, PRO_BF16, p, EPI_F32, e));
        MI_TRY(all_reduce_partial(c, rows));
        have_partial = true;
      } else {
        e.out_f32 = c->resid[cur ^ 1]; e.resid_in = c->resid[cur];
        MI_TRY(run_linear(c, W.down, rows, PRO_BF16, p, EPI_RESID, e));
        cur ^= 1;
      }
    }
  }
  MI_TRY(flush_pending(c));   // the last down projection's K-split sum: the final norm reads other rows (or a gather of rows)
  {  // final norm + lm_head on the rows that are sampled (loader.py:363: logits[:, -1, :])
    ProArgs p{};
    p.resid_in = c->resid[cur] + (size_t)logits_row0 * c->H;
    p.partial = have_partial ? c->partial + (size_t)logits_row0 * c->H : nullptr;
    if (segs) {   // the last row of every request, made contiguous in the K-split workspace
      MI_CHECK((size_t)2 * nseg * c->H * 4 <= c->splitk_ws_bytes, "ragged batch: gather workspace too small");
      Scope sc(c, MI_K_OTHER);
      float* g0 = c->splitk_ws;
      float* g1 = g0 + (size_t)nseg * c->H;
      MI_TRY(launch_gather_rows(c->resid[cur], c->d_ctx, nseg, c->H, g0, s));
      if (have_partial) MI_TRY(launch_gather_rows(c->partial, c->d_ctx, nseg, c->H, g1, s));
      p.resid_in = g0;
      p.partial = have_partial ? g1 : nullptr;
    }
    p.resid_out = nullptr;
    p.gain = c->g_final; p.eps = k.rms_norm_eps;
    EpiArgs e{};
    e.out_f32 = c->logits; e.ld_out = c->V_l;
    MI_TRY(run_linear(c, c->lm_head, logits_rows, PRO_NORM, p, EPI_F32, e));
  }
  if (c->grp) {
    // in-process group: every shard hands its vocabulary slice to the host (or to rank 0's sampler) itself
  } else if (c->xport_allgather) {
    Scope sc(c, MI_K_COMM);
    if (c->xport_allgather(c->xport_user, c->logits, c->logits_all, (size_t)k.max_num_seqs * c->V_l, s) != 0) {
      set_error("all-gather transport callback failed");
      return MI_ECOMM;
    }
  } else if (c->comm) {
    Scope sc(c, MI_K_COMM);
    ncclResult_t r = ncclAllGather(c->logits, c->logits_all, (size_t)k.max_num_seqs * c->V_l, ncclFloat, c->comm, s);
    if (r != ncclSuccess) {
      set_error(std::string("ncclAllGather: ") + ncclGetErrorString(r));
      return MI_ECOMM;
    }
  }
  return MI_OK;
}

int fetch_logits(mi_ctx* c, int nrows, float* out) {
  const mi_model_config& k = c->cfg;
  const int V = k.vocab_size;
  if (!c->collective()) {
    MI_HIP(hipMemcpyAsync(c->h_logits, c->logits, (size_t)nrows * V * 4, hipMemcpyDeviceToHost, c->stream));
    g_ht.lap(3);
    MI_HIP(hipStreamSynchronize(c->stream));
    g_ht.lap(4);
    if (out != c->h_logits) memcpy(out, c->h_logits, (size_t)nrows * V * 4);   // mi_logits_buffer callers read in place
    g_ht.lap(5);
    return MI_OK;
  }
  if (c->grp) {   // this shard's vocabulary slice, straight into its columns of the caller's rows
    MI_HIP(hipMemcpyAsync(c->h_logits, c->logits, (size_t)nrows * c->V_l * 4, hipMemcpyDeviceToHost, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    MI_TRY(group_check_errors(c));
    for (int b = 0; b < nrows; ++b)
      memcpy(out + (size_t)b * V + (size_t)k.tp_rank * c->V_l, c->h_logits + (size_t)b * c->V_l, (size_t)c->V_l * 4);
    return MI_OK;
  }
  const size_t per_rank = (size_t)k.max_num_seqs * c->V_l;
  MI_HIP(hipMemcpyAsync(c->h_logits, c->logits_all, per_rank * k.tp_degree * 4, hipMemcpyDeviceToHost, c->stream));
  MI_HIP(hipStreamSynchronize(c->stream));
  for (int r = 0; r < k.tp_degree; ++r)
    for (int b = 0; b < nrows; ++b)
      memcpy(out + (size_t)b * V + (size_t)r * c->V_l, c->h_logits + r * per_rank + (size_t)b * c->V_l, (size_t)c->V_l * 4);
  return MI_OK;
}

// ---- weight loading ------------------------------------------------------------------
bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int ensure_stage(mi_ctx* c, size_t elems, int rows) {
  if (elems > c->stage_elems) {
    if (c->stage_raw) hipFree(c->stage_raw);
    if (c->stage_f32) hipFree(c->stage_f32);
    MI_HIP(hipMalloc(&c->stage_raw, elems * 2));
    MI_HIP(hipMalloc(reinterpret_cast<void**>(&c->stage_f32), elems * 4));
    c->stage_elems = elems;
  }
  (void)rows;
  return MI_OK;
}

// src: fp32 [rows_total, ld] on the device (the FULL HF tensor)
int place_matrix(mi_ctx* c, Linear& L, const float* src, int rows_total, int ld, int src_row0, int n_rows,
                 int src_col0, int rowmap, int dst_row0, int pad_rows = 0, int pad_cols = 0) {
  QuantJob j{};
  j.pad_rows = pad_rows; j.pad_cols = pad_cols;
  j.src = src; j.ld = ld; j.src_row0 = src_row0; j.src_col0 = src_col0; j.n_rows = n_rows; j.K = L.K;
  j.rowmap = rowmap; j.hd = c->hd; j.dst_row0 = dst_row0; j.wd = L.wd; j.quant_type = c->cfg.quant_type;
  j.dst = L.w; j.dst_scale = L.scale; j.tmp_rowmax = c->rowmax; j.src_rows_total = rows_total;
  return run_quant_job(j, c->stream);
}

int place_vector(mi_ctx* c, float* dst, const float* host_f32, int n) {
  MI_HIP(hipMemcpyAsync(dst, host_f32, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  MI_HIP(hipStreamSynchronize(c->stream));
  return MI_OK;
}

// the tensor is on the device as fp32 [rows, cols] (cols = 1 for vectors is handled by the caller)
int route_matrix(mi_ctx* c, const std::string& name, const float* dsrc, int rows, int cols) {
  const mi_model_config& k = c->cfg;
  const int T = k.tp_degree, r = k.tp_rank;
  if (name == "model.embed_tokens.weight") {
    MI_CHECK(rows == k.vocab_size && cols == c->H, "embed_tokens shape");
    MI_TRY(launch_f32_to_bf16(dsrc, c->embed, (size_t)rows * cols, c->stream));
    if (k.tie_word_embeddings) {
      MI_TRY(place_matrix(c, c->lm_head, dsrc, rows, cols, r * c->V_l, c->V_l, 0, ROWMAP_PLAIN, 0));
      c->have_lm_head = true;
    }
    return MI_OK;
  }
  if (name == "lm_head.weight") {
    MI_CHECK(rows == k.vocab_size && cols == c->H, "lm_head shape");
    MI_TRY(place_matrix(c, c->lm_head, dsrc, rows, cols, r * c->V_l, c->V_l, 0, ROWMAP_PLAIN, 0));
    c->have_lm_head = true;
    return MI_OK;
  }
  int l = -1;
  if (sscanf(name.c_str(), "model.layers.%d.", &l) != 1 || l < 0 || l >= k.num_layers) return MI_OK;  // not ours
  LayerW& W = c->layers[l];
  const int I = k.intermediate_size;
  if (ends_with(name, "self_attn.q_proj.weight")) {
    MI_CHECK(rows == k.num_heads * c->hd && cols == c->H, "q_proj shape");
    return place_matrix(c, W.qkv, dsrc, rows, cols, c->qh0 * c->hd, c->q_dim, 0, ROWMAP_ROPE_PAIRS, 0, (c->nh_l - c->nh_real) * c->hd, 0);
  }
  if (ends_with(name, "self_attn.k_proj.weight")) {
    MI_CHECK(rows == k.num_kv_heads * c->hd && cols == c->H, "k_proj shape");
    return place_matrix(c, W.qkv, dsrc, rows, cols, c->kvh0 * c->hd, c->kv_dim, 0, ROWMAP_ROPE_PAIRS, c->q_dim);
  }
  if (ends_with(name, "self_attn.v_proj.weight")) {
    MI_CHECK(rows == k.num_kv_heads * c->hd && cols == c->H, "v_proj shape");
    return place_matrix(c, W.qkv, dsrc, rows, cols, c->kvh0 * c->hd, c->kv_dim, 0, ROWMAP_PLAIN, c->q_dim + c->kv_dim);
  }
  if (ends_with(name, "self_attn.o_proj.weight")) {
    MI_CHECK(rows == c->H && cols == k.num_heads * c->hd, "o_proj shape");
    return place_matrix(c, W.o, dsrc, rows, cols, 0, c->H, c->qh0 * c->hd, ROWMAP_PLAIN, 0, 0, (c->nh_l - c->nh_real) * c->hd);
  }
  if (ends_with(name, "mlp.gate_proj.weight")) {
    MI_CHECK(rows == I && cols == c->H, "gate_proj shape");
    return place_matrix(c, W.gu, dsrc, rows, cols, r * c->I_l, c->I_l, 0, ROWMAP_EVERY_OTHER, 0);
  }
  if (ends_with(name, "mlp.up_proj.weight")) {
    MI_CHECK(rows == I && cols == c->H, "up_proj shape");
    return place_matrix(c, W.gu, dsrc, rows, cols, r * c->I_l, c->I_l, 0, ROWMAP_EVERY_OTHER, 1);
  }
  if (ends_with(name, "mlp.down_proj.weight")) {
    MI_CHECK(rows == c->H && cols == I, "down_proj shape");
    return place_matrix(c, W.down, dsrc, rows, cols, 0, c->H, r * c->I_l, ROWMAP_PLAIN, 0);
  }
  (void)T;
  return MI_OK;
}

int route_vector(mi_ctx* c, const std::string& name, const std::vector<float>& v) {
  const mi_model_config& k = c->cfg;
  const int n = (int)v.size();
  if (name == "model.norm.weight") {
    MI_CHECK(n == c->H, "model.norm shape");
    return place_vector(c, c->g_final, v.data(), n);
  }
  int l = -1;
  if (sscanf(name.c_str(), "model.layers.%d.", &l) != 1 || l < 0 || l >= k.num_layers) return MI_OK;
  LayerW& W = c->layers[l];
  if (ends_with(name, "input_layernorm.weight")) { MI_CHECK(n == c->H, "input_layernorm shape"); return place_vector(c, W.g_in, v.data(), n); }
  if (ends_with(name, "post_attention_layernorm.weight")) { MI_CHECK(n == c->H, "post_attention_layernorm shape"); return place_vector(c, W.g_post, v.data(), n); }
  const bool qb = ends_with(name, "self_attn.q_proj.bias"), kb = ends_with(name, "self_attn.k_proj.bias"),
             vb = ends_with(name, "self_attn.v_proj.bias");
  if (qb || kb || vb) {
    MI_CHECK(W.qkv.bias != nullptr, "bias tensor given but the config has qkv_bias = 0");
    const int hd = c->hd;
    const int src0 = qb ? c->qh0 * hd : c->kvh0 * hd;
    const int cnt = qb ? c->nh_real * hd : c->kv_dim;   // padding q heads keep the zero bias of alloc_linear
    const int dst0 = qb ? 0 : (kb ? c->q_dim : c->q_dim + c->kv_dim);
    MI_CHECK(n == (qb ? k.num_heads : k.num_kv_heads) * hd, "qkv bias shape");
    std::vector<float> out(cnt);
    if (cnt == 0) return MI_OK;
    for (int i = 0; i < cnt; ++i) {
      int j = i;  // destination index of source element i
      if (!vb) {
        const int head = i / hd, d = i % hd;
        j = head * hd + (d < hd / 2 ? 2 * d : 2 * (d - hd / 2) + 1);
      }
      out[j] = v[src0 + i];
    }
    return place_vector(c, W.qkv.bias + dst0, out.data(), cnt);
  }
  return MI_OK;
}

float bf16_bits_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int capture_or_launch_decode(mi_ctx* c, int B, int MB) {
  const int key = (B * 65536 + MB) ^ (int)((c->class_mask & 0xff) << 24) ^ ((c->attn_rows_per_seq - 1) << 12);
  // a caller-supplied transport runs on the host thread: never captured
  if (!c->cfg.use_graphs || c->prof.on || c->xport_allreduce) return run_layers(c, B, true, B, MB, 0, B, 0);
  auto it = c->graphs.find(key);
  if (it == c->graphs.end()) {
    // one eager pass first: sets every kernel's attributes outside of capture
    MI_TRY(run_layers(c, B, true, B, MB, 0, B, 0));
    MI_HIP(hipStreamSynchronize(c->stream));
    hipGraph_t g = nullptr;
    MI_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = run_layers(c, B, true, B, MB, 0, B, 0);
    hipError_t ce = hipStreamEndCapture(c->stream, &g);
    if (rc != MI_OK) return rc;
    MI_HIP(ce);
    hipGraphExec_t ge = nullptr;
    MI_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    c->graphs[key] = ge;
    return MI_OK;  // the eager pass already produced this step's logits
  }
  MI_HIP(hipGraphLaunch(it->second, c->stream));
  return MI_OK;
}

}  // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char* mi_last_error(void) { return mi::g_err.c_str(); }
int mi_version(void) { return 4; }   // 3: + in-process tensor parallelism (tp_rank = MI_TP_ALL_RANKS, tp_device_ids, tp_transport)

int mi_tp_plan(const mi_model_config* cfg, int32_t rank, mi_tp_plan_t* out) {
  MI_CHECK(cfg && out, "null argument");
  const mi_model_config& k = *cfg;
  const int T = k.tp_degree;
  MI_CHECK(T >= 1 && rank >= 0 && rank < T, "mi_tp_plan: rank out of range");
  MI_CHECK(k.num_heads > 0 && k.num_kv_heads > 0 && k.num_heads % k.num_kv_heads == 0, "num_heads must be a multiple of num_kv_heads");
  MI_CHECK(k.num_kv_heads >= T ? k.num_kv_heads % T == 0 : T % k.num_kv_heads == 0,
           "num_kv_heads must be a multiple or a divisor of tp_degree");
  MI_CHECK(k.intermediate_size % T == 0 && k.vocab_size % T == 0, "intermediate_size and vocab_size must divide by tp_degree");
  const int G = k.num_heads / k.num_kv_heads;
  mi_tp_plan_t p{};
  if (k.num_kv_heads >= T) {       // whole kv groups per rank
    p.kv_heads_local = k.num_kv_heads / T;
    p.q_heads_local = p.q_heads_real = p.kv_heads_local * G;
    p.kv_head0 = rank * p.kv_heads_local;
    p.q_head0 = p.kv_head0 * G;
  } else {                         // every kv head on R ranks, its G q heads dealt ceil(G / R) at a time, the tail zero-weight padding
    const int R = T / k.num_kv_heads, Gl = (G + R - 1) / R, sub = rank % R;
    p.kv_heads_local = 1;
    p.q_heads_local = Gl;
    p.kv_head0 = rank / R;
    p.q_head0 = std::min(p.kv_head0 * G + sub * Gl, (p.kv_head0 + 1) * G);
    p.q_heads_real = std::max(0, std::min(Gl, G - sub * Gl));
  }
  p.inter_local = k.intermediate_size / T;
  p.inter0 = rank * p.inter_local;
  p.vocab_local = k.vocab_size / T;
  p.vocab0 = rank * p.vocab_local;
  *out = p;
  return MI_OK;
}

int mi_ctx_create(const mi_model_config* cfg, mi_ctx** out) {
  MI_CHECK(cfg && out, "null argument");
  const mi_model_config& k = *cfg;
  MI_CHECK(k.num_layers > 0 && k.hidden_size > 0 && k.num_heads > 0 && k.num_kv_heads > 0, "bad geometry");
  MI_CHECK(k.head_dim == 64 || k.head_dim == 128, "head_dim must be 64 or 128");
  if (k.tp_rank == MI_TP_ALL_RANKS && k.tp_degree > 1) {   // every rank shard inside this process (tp_group.h)
    mi_ctx* f = new mi_ctx();
    f->cfg = k;
    int rc = group_create(k, f);
    if (rc != MI_OK) {
      const std::string keep = mi::g_err;
      group_destroy(f->owned_group);
      delete f;
      set_error(keep);
      return rc;
    }
    *out = f;
    return MI_OK;
  }
  MI_CHECK(k.tp_degree >= 1 && (k.tp_rank >= 0 || k.tp_degree == 1) && k.tp_rank < k.tp_degree, "bad tp_degree / tp_rank");
  MI_CHECK(k.num_heads % k.num_kv_heads == 0, "num_heads must be a multiple of num_kv_heads");
  MI_CHECK(k.intermediate_size % k.tp_degree == 0 && k.vocab_size % k.tp_degree == 0, "intermediate/vocab must divide by tp_degree");
  MI_CHECK(k.block_size > 0 && k.block_size % 16 == 0, "block_size must be a positive multiple of 16");
  MI_CHECK(k.num_blocks >= 2, "num_blocks must include the null block and at least one real block");
  // token-generation batches of up to 16 rows stream the weights (GEMV); larger ones take the
  // context-encoding GEMMs (vLLM's default max_num_seqs for this platform is 32, platform.py)
  MI_CHECK(k.max_num_seqs >= 1 && k.max_num_seqs <= 256, "max_num_seqs must be 1..256");
  MI_CHECK(k.max_model_len >= 1, "max_model_len");
  MI_CHECK(ceil_div(k.max_model_len, k.block_size) <= kPrefillMaxBlocks,
           "max_model_len spans more than 4096 blocks per sequence: raise block_size (context encoding stages a sequence's block table in LDS)");
  MI_CHECK(k.weight_dtype >= MI_W_BF16 && k.weight_dtype <= MI_W_INT8, "weight_dtype");
  MI_CHECK(k.quant_type == MI_Q_PER_TENSOR_SYMMETRIC || k.quant_type == MI_Q_PER_CHANNEL_SYMMETRIC, "quant_type");
  int ndev = 0;
  MI_HIP(hipGetDeviceCount(&ndev));
  MI_CHECK(ndev > 0, "no HIP device: this library has no CPU fallback");
  MI_CHECK(k.device_id >= 0 && k.device_id < ndev, "device_id out of range");
  MI_HIP(hipSetDevice(k.device_id));

  mi_ctx* c = new mi_ctx();
  c->cfg = k;
  if (c->cfg.tp_rank < 0) c->cfg.tp_rank = 0;   // MI_TP_ALL_RANKS of one rank
  MI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->H = k.hidden_size;
  c->hd = k.head_dim;
  // Head sharding.  kv heads >= ranks: whole kv groups per rank.  Fewer kv heads than ranks: every kv head is
  // replicated on R = T / nkv ranks and its G q heads are dealt to them ceil(G / R) at a time; where R does not
  // divide G the last rank(s) of a group hold ZERO-WEIGHT padding heads (q rows, q bias and o_proj columns zero:
  // they add nothing to the row-parallel sum) -- Qwen2.5-7B at TP 8: 28 q / 4 kv heads -> 4 + 3(+1 pad) per kv
  // head.  The reference runs such models because it skips vLLM's divisibility check (platform.py:58-64) and
  // its model library pads heads the same way.
  mi_tp_plan_t plan;
  MI_TRY(mi_tp_plan(&c->cfg, c->cfg.tp_rank, &plan));   // (host arithmetic only; exported so that the plan can be tested without a GPU)
  c->nkv_l = plan.kv_heads_local;
  c->nh_l = plan.q_heads_local;
  c->nh_real = plan.q_heads_real;
  c->kvh0 = plan.kv_head0;
  c->qh0 = plan.q_head0;
  MI_CHECK(c->nh_l % c->nkv_l == 0 && c->nh_l / c->nkv_l <= 8, "q heads per kv head (per rank) must be 1..8");
  c->q_dim = c->nh_l * c->hd;
  c->kv_dim = c->nkv_l * c->hd;
  c->I_l = plan.inter_local;
  c->V_l = plan.vocab_local;
  int maxb = k.max_model_len;
  for (int i = 0; i < k.num_ctx_buckets && i < 8; ++i) maxb = std::max(maxb, k.ctx_buckets[i]);
  c->max_rows = std::max(maxb, k.max_num_seqs);

  const int wd = k.weight_dtype;
  c->layers.resize(k.num_layers);
  for (auto& W : c->layers) {
    MI_TRY(alloc_linear(c, W.qkv, c->q_dim + 2 * c->kv_dim, c->H, wd, k.qkv_bias != 0));
    MI_TRY(alloc_linear(c, W.o, c->H, c->q_dim, wd, false));
    MI_TRY(alloc_linear(c, W.gu, 2 * c->I_l, c->H, wd, false));
    MI_TRY(alloc_linear(c, W.down, c->H, c->I_l, wd, false));
    MI_TRY(dmalloc(&W.g_in, (size_t)c->H, &c->weight_bytes));
    MI_TRY(dmalloc(&W.g_post, (size_t)c->H, &c->weight_bytes));
  }
  MI_TRY(alloc_linear(c, c->lm_head, c->V_l, c->H, k.quantize_lm_head ? wd : MI_W_BF16, false));
  MI_TRY(dmalloc(&c->g_final, (size_t)c->H, &c->weight_bytes));
  MI_TRY(dmalloc(&c->embed, (size_t)k.vocab_size * c->H, &c->weight_bytes));
  MI_TRY(dmalloc(&c->rowmax, (size_t)std::max(std::max(k.vocab_size, k.hidden_size), std::max(k.intermediate_size, k.num_heads * k.head_dim)) + 16));
  *out = c;
  return MI_OK;
}

int mi_ctx_destroy(mi_ctx* c) {
  if (!c) return MI_OK;
  if (c->owned_group) {
    group_destroy(c->owned_group);
    delete c;
    return MI_OK;
  }
  g_ht.report();
  hipSetDevice(c->cfg.device_id);
  // a borrowed stream (a draft model runs on its target's shard-0 stream, mi_forward_spec) may be gone already when the
  // target was closed first: wait for the device instead of naming it
  if (c->stream_owned) hipStreamSynchronize(c->stream);
  else hipDeviceSynchronize();
  for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->grp && c->grp->lockstep && c->cfg.tp_rank != 0) hipStreamSynchronize(c->stream);
  auto fl = [](Linear& L) { hipFree(L.w); hipFree(L.scale); hipFree(L.bias); };
  for (auto& W : c->layers) { fl(W.qkv); fl(W.o); fl(W.gu); fl(W.down); hipFree(W.g_in); hipFree(W.g_post); }
  fl(c->lm_head);
  void* ptrs[] = {c->g_final, c->embed, c->rowmax, c->kv_pool, c->rope_cos, c->rope_sin, c->stage_raw, c->stage_f32,
                  c->resid[0], c->resid[1], c->partial, c->xn, c->qbuf, c->attn_out, c->act, c->logits, c->logits_all,
                  c->attn_scratch, c->d_inputs, c->d_dec, c->d_dec_bt, c->d_sparams, c->d_tokens, c->x8, c->x8_scale, c->splitk_ws, c->d_spec, c->d_sample_scratch};
  for (void* p : ptrs) hipFree(p);
  void* hptrs[] = {c->h_inputs, c->h_dec, c->h_dec_bt, c->h_sparams, c->h_tokens, c->h_logits, c->h_spec};
  for (void* p : hptrs) if (p) hipHostFree(p);
  if (c->stream_owned) hipStreamDestroy(c->stream);
  delete c;
  return MI_OK;
}

int mi_load_weight(mi_ctx* c, const char* name_c, const void* host, int32_t dtype, const int64_t* shape, int32_t ndim) {
  MI_CHECK(c && name_c && host && shape, "null argument");
  if (c->owned_group)
    return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_load_weight(rc, name_c, host, dtype, shape, ndim); });
  MI_CHECK(!c->finalized, "mi_load_weight after mi_finalize");
  MI_CHECK(dtype == MI_F32 || dtype == MI_BF16, "weights must be fp32 or bf16 on the host");
  MI_CHECK(ndim == 1 || ndim == 2, "weights must be 1-D or 2-D");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  const std::string name(name_c);
  if (ndim == 1) {
    std::vector<float> v((size_t)shape[0]);
    if (dtype == MI_F32) memcpy(v.data(), host, v.size() * 4);
    else for (size_t i = 0; i < v.size(); ++i) v[i] = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(host)[i]);
    return route_vector(c, name, v);
  }
  const size_t elems = (size_t)shape[0] * (size_t)shape[1];
  MI_TRY(ensure_stage(c, elems, (int)shape[0]));
  if (dtype == MI_F32) {
    MI_HIP(hipMemcpyAsync(c->stage_f32, host, elems * 4, hipMemcpyHostToDevice, c->stream));
  } else {
    MI_HIP(hipMemcpyAsync(c->stage_raw, host, elems * 2, hipMemcpyHostToDevice, c->stream));
    MI_TRY(launch_bf16_to_f32(reinterpret_cast<const uint16_t*>(c->stage_raw), c->stage_f32, elems, c->stream));
  }
  MI_TRY(route_matrix(c, name, c->stage_f32, (int)shape[0], (int)shape[1]));
  MI_HIP(hipStreamSynchronize(c->stream));  // the host buffer / staging may be reused by the caller
  return MI_OK;
}

int mi_init_synthetic_weights(mi_ctx* c, uint64_t seed, float std) {
  MI_CHECK(c, "null argument");
  if (c->owned_group)
    return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_init_synthetic_weights(rc, seed, std); });
  MI_CHECK(!c->finalized, "bad state");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  const mi_model_config& k = c->cfg;
  uint64_t tid = 1;
  auto gen = [&](const std::string& name, int rows, int cols, float sd) -> int {
    const size_t elems = (size_t)rows * cols;
    MI_TRY(ensure_stage(c, elems, rows));
    MI_TRY(launch_randn(c->stage_f32, elems, seed, tid++, sd, c->stream));
    return route_matrix(c, name, c->stage_f32, rows, cols);
  };
  MI_TRY(gen("model.embed_tokens.weight", k.vocab_size, c->H, 1.0f));
  char buf[128];
  for (int l = 0; l < k.num_layers; ++l) {
    auto nm = [&](const char* suf) { snprintf(buf, sizeof buf, "model.layers.%d.%s", l, suf); return std::string(buf); };
    MI_TRY(gen(nm("self_attn.q_proj.weight"), k.num_heads * c->hd, c->H, std));
    MI_TRY(gen(nm("self_attn.k_proj.weight"), k.num_kv_heads * c->hd, c->H, std));
    MI_TRY(gen(nm("self_attn.v_proj.weight"), k.num_kv_heads * c->hd, c->H, std));
    MI_TRY(gen(nm("self_attn.o_proj.weight"), c->H, k.num_heads * c->hd, std));
    MI_TRY(gen(nm("mlp.gate_proj.weight"), k.intermediate_size, c->H, std));
    MI_TRY(gen(nm("mlp.up_proj.weight"), k.intermediate_size, c->H, std));
    MI_TRY(gen(nm("mlp.down_proj.weight"), c->H, k.intermediate_size, std));
    if (k.qkv_bias) {   // Qwen2: FULL bias vectors in logical coordinates, each rank routes its slice
      const char* bn[3] = {"self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias"};
      for (int i = 0; i < 3; ++i) {
        const int n = (i == 0 ? k.num_heads : k.num_kv_heads) * c->hd;
        MI_TRY(ensure_stage(c, (size_t)n, n));
        MI_TRY(launch_randn(c->stage_f32, (size_t)n, seed, tid++, 0.25f, c->stream));
        std::vector<float> hb((size_t)n);
        MI_HIP(hipMemcpyAsync(hb.data(), c->stage_f32, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        MI_HIP(hipStreamSynchronize(c->stream));
        MI_TRY(route_vector(c, nm(bn[i]), hb));
      }
    }
    MI_TRY(launch_fill_f32(c->layers[l].g_in, c->H, 1.0f, c->stream));
    MI_TRY(launch_fill_f32(c->layers[l].g_post, c->H, 1.0f, c->stream));
  }
  MI_TRY(launch_fill_f32(c->g_final, c->H, 1.0f, c->stream));
  if (!k.tie_word_embeddings) MI_TRY(gen("lm_head.weight", k.vocab_size, c->H, std));
  MI_HIP(hipStreamSynchronize(c->stream));
  return MI_OK;
}

// ---- weight artifacts: the device images (quantized, tiled, sharded) of one rank, on disk --------
// The counterpart of the reference's compiled-artifact directory (loader.py:160-226): a restart
// skips reading the bf16 checkpoint, quantizing and re-tiling, and streams the ready-made images
// straight into HBM.  One file per rank: header (format + everything the images depend on), then
// every tensor in a fixed order as [uint64 nbytes][bytes].
namespace {
struct ArtifactHeader {
  char magic[8];            // "MI355XW\0"
  int32_t version;          // of this file layout
  int32_t tile_format;      // of the weight tiling (mi_common.h): bump when the kernels' layout changes
  int32_t geo[9];           // layers, hidden, heads, kv heads, head dim, intermediate, vocab, qkv_bias, tie
  int32_t weight_dtype, quant_type, quantize_lm_head, tp_degree, tp_rank;
};
constexpr int32_t kArtifactVersion = 1, kTileFormat = 1;

void make_header(const mi_ctx* c, ArtifactHeader& h) {
  const mi_model_config& k = c->cfg;
  h = ArtifactHeader{};
  memcpy(h.magic, "MI355XW", 8);
  h.version = kArtifactVersion;
  h.tile_format = kTileFormat;
  const int32_t g[9] = {k.num_layers, k.hidden_size, k.num_heads, k.num_kv_heads, k.head_dim, k.intermediate_size,
                        k.vocab_size, k.qkv_bias, k.tie_word_embeddings};
  memcpy(h.geo, g, sizeof g);
  h.weight_dtype = k.weight_dtype; h.quant_type = k.quant_type; h.quantize_lm_head = k.quantize_lm_head;
  h.tp_degree = k.tp_degree; h.tp_rank = k.tp_rank;
}

// every device tensor of a rank, in file order
void artifact_tensors(mi_ctx* c, std::vector<std::pair<void*, size_t>>& out) {
  auto lin = [&](Linear& L) {
    out.push_back({L.w, L.bytes()});
    out.push_back({L.scale, (size_t)L.N * 4});
    if (L.bias) out.push_back({L.bias, (size_t)L.N * 4});
  };
  for (auto& W : c->layers) {
    lin(W.qkv); lin(W.o); lin(W.gu); lin(W.down);
    out.push_back({W.g_in, (size_t)c->H * 4});
    out.push_back({W.g_post, (size_t)c->H * 4});
  }
  lin(c->lm_head);
  out.push_back({c->g_final, (size_t)c->H * 4});
  out.push_back({c->embed, (size_t)c->cfg.vocab_size * c->H * 2});
}

void artifact_file(const mi_ctx* c, const char* dir, std::string& out) {
  out = std::string(dir) + "/rank" + std::to_string(c->cfg.tp_rank) + "_of" + std::to_string(c->cfg.tp_degree) + ".miw";
}
constexpr size_t kArtifactChunk = (size_t)64 << 20;
}  // namespace

int mi_save_weights(mi_ctx* c, const char* dir) {
  MI_CHECK(c && dir, "null argument");
  if (c->owned_group) return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_save_weights(rc, dir); });
  MI_CHECK(c->have_lm_head, "mi_save_weights: the weights are not loaded yet");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  std::string path;
  artifact_file(c, dir, path);
  const std::string tmp = path + ".part";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { set_error("mi_save_weights: cannot create " + tmp); return MI_EINVAL; }
  void* stage = nullptr;
  if (hipHostMalloc(&stage, kArtifactChunk, hipHostMallocDefault) != hipSuccess) { fclose(f); set_error("mi_save_weights: pinned staging"); return MI_ENOMEM; }
  ArtifactHeader h;
  make_header(c, h);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  std::vector<std::pair<void*, size_t>> ts;
  artifact_tensors(c, ts);
  MI_HIP(hipStreamSynchronize(c->stream));
  for (auto& t : ts) {
    const uint64_t n = t.second;
    ok = ok && fwrite(&n, 8, 1, f) == 1;
    for (size_t off = 0; ok && off < t.second; off += kArtifactChunk) {
      const size_t m = std::min(kArtifactChunk, t.second - off);
      if (hipMemcpy(stage, (char*)t.first + off, m, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
      else ok = fwrite(stage, 1, m, f) == m;
    }
  }
  hipHostFree(stage);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); set_error("mi_save_weights: write to " + tmp + " failed"); return MI_EINVAL; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { set_error("mi_save_weights: cannot rename to " + path); return MI_EINVAL; }
  return MI_OK;
}

// MI_EINVAL (-> ValueError in the Python mirror, like a config mismatch in the reference) when the
// file is missing or was made for another model / quantization / sharding / tile format.
int mi_load_weights_file(mi_ctx* c, const char* dir) {
  MI_CHECK(c && dir, "null argument");
  if (c->owned_group) return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_load_weights_file(rc, dir); });
  MI_CHECK(!c->finalized, "mi_load_weights_file after mi_finalize");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  std::string path;
  artifact_file(c, dir, path);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { set_error("weight artifacts not found: " + path); return MI_EINVAL; }
  ArtifactHeader h{}, want;
  make_header(c, want);
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(&h, &want, sizeof h) != 0) {
    fclose(f);
    set_error("weight artifacts at " + path + " were made for another model / quantization / sharding / format");
    return MI_EINVAL;
  }
  void* stage = nullptr;
  if (hipHostMalloc(&stage, kArtifactChunk, hipHostMallocDefault) != hipSuccess) { fclose(f); set_error("pinned staging"); return MI_ENOMEM; }
  std::vector<std::pair<void*, size_t>> ts;
  artifact_tensors(c, ts);
  bool ok = true;
  for (auto& t : ts) {
    uint64_t n = 0;
    ok = ok && fread(&n, 8, 1, f) == 1 && n == t.second;
    for (size_t off = 0; ok && off < t.second; off += kArtifactChunk) {
      const size_t m = std::min(kArtifactChunk, t.second - off);
      ok = fread(stage, 1, m, f) == m && hipMemcpy((char*)t.first + off, stage, m, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) break;
  }
  hipHostFree(stage);
  fclose(f);
  if (!ok) { set_error("weight artifacts at " + path + " are truncated or do not match this context's tensors"); return MI_EINVAL; }
  c->have_lm_head = true;
  return MI_OK;
}

int mi_set_num_blocks(mi_ctx* c, int32_t num_blocks) {
  MI_CHECK(c, "null argument");
  if (c->owned_group) {
    c->cfg.num_blocks = num_blocks;
    return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_set_num_blocks(rc, num_blocks); });
  }
  MI_CHECK(!c->finalized, "mi_set_num_blocks after mi_finalize");
  MI_CHECK(num_blocks >= 2, "num_blocks must include the null block and at least one real block");
  c->cfg.num_blocks = num_blocks;
  return MI_OK;
}

// bytes mi_finalize allocates besides the KV pool (activations, exchange buffers, scratch): what
// has to be kept out of the memory offered to vLLM for KV blocks
static size_t workspace_estimate(const mi_ctx* c) {
  const mi_model_config& k = c->cfg;
  const size_t R = (size_t)c->max_rows, H = (size_t)c->H;
  size_t b = 3 * R * H * 4 + R * std::max<size_t>(H, c->I_l) * 2 + 2 * R * c->q_dim * 2 + R * c->I_l * 2;
  if (k.prefill_fp8_activations) b += R * std::max<size_t>(std::max<size_t>(H, c->I_l), c->q_dim) + R * 4;
  b += (size_t)64 << 20;                                                   // split-K workspace
  b += (size_t)k.max_num_seqs * c->V_l * 4 * (1 + (k.tp_degree > 1 ? k.tp_degree : 0));
  b += attn_scratch_bytes(k.max_num_seqs, c->nh_l, c->hd);
  b += (size_t)k.max_model_len * c->hd * 4;                                 // rotary tables
  if (k.tp_degree > 1) b += 8 * R * H + 4 * (R * H / k.tp_degree + 1024);   // exchange slots (tp_group.hip)
  return b + ((size_t)8 << 20);                                             // inputs, small buffers
}

int64_t mi_kv_bytes_per_block(mi_ctx* c) {
  if (!c) return 0;
  if (c->owned_group) c = c->owned_group->ranks[0];
  return (int64_t)c->cfg.num_layers * 2 * c->nkv_l * c->cfg.block_size * c->hd * 2;
}

int mi_finalize(mi_ctx* c) {
  MI_CHECK(c, "null argument");
  if (c->owned_group) {
    MI_CHECK(!c->finalized, "bad state");
    MI_TRY(group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_finalize(rc); }));
    MI_TRY(group_selftest(c->owned_group));
    c->finalized = true;
    return MI_OK;
  }
  MI_CHECK(!c->finalized, "bad state");
  MI_CHECK(c->have_lm_head, "lm_head.weight (or tied embed_tokens) was never loaded");
  MI_CHECK(c->cfg.tp_degree == 1 || c->collective(), "tp_degree > 1: call mi_tp_init before mi_finalize");
  const mi_model_config& k = c->cfg;
  MI_HIP(hipSetDevice(k.device_id));
  hipStream_t s = c->stream;
  if (c->stage_raw) { hipFree(c->stage_raw); c->stage_raw = nullptr; }
  if (c->stage_f32) { hipFree(c->stage_f32); c->stage_f32 = nullptr; }
  c->stage_elems = 0;
  // KV pool, zero-filled: stale or never-written rows must hold finite values
  c->kv_half = (size_t)k.num_blocks * c->nkv_l * k.block_size * c->hd;
  const size_t kv_elems = c->kv_half * 2 * k.num_layers;
  {
    size_t fr = 0, tot = 0;
    MI_HIP(hipMemGetInfo(&fr, &tot));
    const size_t want = kv_elems * 2 + workspace_estimate(c);
    if (want > fr) {
      set_error("KV pool of " + std::to_string(k.num_blocks) + " blocks (" + std::to_string(kv_elems * 2 >> 20) + " MiB over " +
                std::to_string(k.num_layers) + " layers) + " + std::to_string(workspace_estimate(c) >> 20) +
                " MiB of workspace do not fit the " + std::to_string(fr >> 20) + " MiB free on the device: lower "
                "num_gpu_blocks_override / gpu_memory_utilization");
      return MI_ENOMEM;
    }
  }
  MI_TRY(dmalloc(&c->kv_pool, kv_elems, &c->kv_bytes));
  MI_HIP(hipMemsetAsync(c->kv_pool, 0, kv_elems * 2, s));
  // RoPE tables (fp32 angles as HF computes them; llama3 rescale of the inverse frequencies)
  {
    const int half = c->hd / 2, P = k.max_model_len;
    std::vector<float> inv(half), cs((size_t)P * half), sn((size_t)P * half);
    for (int i = 0; i < half; ++i) {
      float f = 1.0f / powf(k.rope_theta, (float)(2 * i) / (float)c->hd);
      if (k.rope_type == MI_ROPE_LLAMA3) {
        const float old = (float)k.rope_original_max_position;
        const float wavelen = 2.0f * (float)M_PI / f;
        const float lo_w = old / k.rope_low_freq_factor, hi_w = old / k.rope_high_freq_factor;
        const float scaled = wavelen > lo_w ? f / k.rope_factor : f;
        const float smooth = (old / wavelen - k.rope_low_freq_factor) / (k.rope_high_freq_factor - k.rope_low_freq_factor);
        const float mid = (1.0f - smooth) * scaled / k.rope_factor + smooth * scaled;
        const bool is_mid = !(wavelen < hi_w) && !(wavelen > lo_w);
        f = is_mid ? mid : scaled;
      }
      inv[i] = f;
    }
    for (int p = 0; p < P; ++p)
      for (int i = 0; i < half; ++i) {
        const float ang = (float)p * inv[i];
        cs[(size_t)p * half + i] = cosf(ang);
        sn[(size_t)p * half + i] = sinf(ang);
      }
    MI_TRY(dmalloc(&c->rope_cos, cs.size(), &c->workspace_bytes));
    MI_TRY(dmalloc(&c->rope_sin, sn.size(), &c->workspace_bytes));
    MI_HIP(hipMemcpyAsync(c->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(c->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  const size_t R = (size_t)c->max_rows;
  size_t* ws = &c->workspace_bytes;
  MI_TRY(dmalloc(&c->resid[0], R * c->H, ws));
  MI_TRY(dmalloc(&c->resid[1], R * c->H, ws));
  MI_TRY(dmalloc(&c->partial, R * c->H, ws));
  MI_TRY(dmalloc(&c->xn, R * std::max(c->H, c->I_l), ws));
  MI_TRY(dmalloc(&c->qbuf, R * c->q_dim, ws));
  MI_TRY(dmalloc(&c->attn_out, R * c->q_dim, ws));
  MI_TRY(dmalloc(&c->act, R * c->I_l, ws));
  if (k.prefill_fp8_activations) {
    MI_TRY(dmalloc(&c->x8, R * std::max(std::max(c->H, c->I_l), c->q_dim), ws));
    MI_TRY(dmalloc(&c->x8_scale, R, ws));
  }
  c->splitk_ws_bytes = (size_t)64 << 20;   // fp32 K-slice slabs of the short-prompt GEMMs
  MI_TRY(dmalloc(&c->splitk_ws, c->splitk_ws_bytes / 4, ws));
  MI_TRY(dmalloc(&c->logits, (size_t)k.max_num_seqs * c->V_l, ws));
  if (c->collective()) MI_TRY(dmalloc(&c->logits_all, (size_t)k.tp_degree * k.max_num_seqs * c->V_l, ws));
  MI_HIP(hipMalloc(&c->attn_scratch, attn_scratch_bytes(k.max_num_seqs, c->nh_l, c->hd)));
  MI_HIP(hipMemsetAsync(c->attn_scratch, 0, attn_scratch_bytes(k.max_num_seqs, c->nh_l, c->hd), s));   // incl. the merge tickets
  *ws += attn_scratch_bytes(k.max_num_seqs, c->nh_l, c->hd);
  c->MB_cap = ceil_div(k.max_model_len, k.block_size) + 1;
  const size_t nbt = (size_t)k.max_num_seqs * c->MB_cap;
  // inputs: [block tables][context lengths][ids][positions][slots], one block on each side
  c->inputs_elems = nbt + (size_t)k.max_num_seqs + 3 * R + (size_t)k.max_num_seqs;   // [tables][per-request ints][ids][pos][slots][per-request context lengths]
  MI_TRY(dmalloc(&c->d_inputs, c->inputs_elems, ws));
  MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_inputs), c->inputs_elems * 4, hipHostMallocDefault));
  memset(c->h_inputs, 0, c->inputs_elems * 4);
  auto carve = [&](int32_t* base) {
    int32_t* p = base;
    int32_t* bt = p; p += nbt;
    int32_t* ctx = p; p += k.max_num_seqs;
    int32_t* ids = p; p += R;
    int32_t* pos = p; p += R;
    int32_t* slots = p;
    return std::array<int32_t*, 5>{bt, ctx, ids, pos, slots};
  };
  {
    auto d = carve(c->d_inputs), h = carve(c->h_inputs);
    c->d_bt = d[0]; c->d_ctx = d[1]; c->d_ids = d[2]; c->d_pos = d[3]; c->d_slots = d[4];
    c->h_bt = h[0]; c->h_ctx = h[1]; c->h_ids = h[2]; c->h_pos = h[3]; c->h_slots = h[4];
  }
  {   // token-generation inputs: small per-step block + device-resident block tables
    const size_t ms = (size_t)k.max_num_seqs;
    MI_TRY(dmalloc(&c->d_dec, 4 * ms, ws));
    MI_TRY(dmalloc(&c->d_dec_bt, nbt, ws));
    MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_dec), 4 * ms * 4, hipHostMallocDefault));
    MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_dec_bt), nbt * 4, hipHostMallocDefault));
    memset(c->h_dec, 0, 4 * ms * 4);
    memset(c->h_dec_bt, 0, nbt * 4);
    MI_HIP(hipMemsetAsync(c->d_dec_bt, 0, nbt * 4, s));
    c->bt_checked.assign(ms, 0);
  }
  MI_TRY(dmalloc(&c->d_sparams, (size_t)k.max_num_seqs * 3, ws));
  MI_TRY(dmalloc(&c->d_tokens, (size_t)k.max_num_seqs, ws));
  {
    unsigned char* p = nullptr;
    MI_TRY(dmalloc(&p, sample_scratch_bytes(k.max_num_seqs), ws));
    c->d_sample_scratch = p;
  }
  MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_sparams), (size_t)k.max_num_seqs * 3 * 4, hipHostMallocDefault));
  MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_tokens), (size_t)k.max_num_seqs * 4, hipHostMallocDefault));
  MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_logits), (size_t)k.max_num_seqs * k.vocab_size * 4, hipHostMallocDefault));
  MI_HIP(hipStreamSynchronize(s));
  if (c->grp) MI_TRY(group_alloc_exchange(c));
  c->finalized = true;
  return MI_OK;
}

// current views of the step inputs (run_layers and a captured graph read these pointers)
static void set_input_views(mi_ctx* c, bool decode_step) {
  const mi_model_config& k = c->cfg;
  const size_t ms = (size_t)k.max_num_seqs, R = (size_t)c->max_rows, nbt = ms * c->MB_cap;
  if (decode_step) {
    c->d_ctx = c->d_dec; c->d_ids = c->d_dec + ms; c->d_pos = c->d_dec + 2 * ms; c->d_slots = c->d_dec + 3 * ms;
    c->h_ctx = c->h_dec; c->h_ids = c->h_dec + ms; c->h_pos = c->h_dec + 2 * ms; c->h_slots = c->h_dec + 3 * ms;
    c->d_bt = c->d_dec_bt; c->h_bt = c->h_dec_bt;
  } else {
    c->d_bt = c->d_inputs; c->d_ctx = c->d_inputs + nbt; c->d_ids = c->d_ctx + ms; c->d_pos = c->d_ids + R; c->d_slots = c->d_pos + R;
    c->h_bt = c->h_inputs; c->h_ctx = c->h_inputs + nbt; c->h_ids = c->h_ctx + ms; c->h_pos = c->h_ids + R; c->h_slots = c->h_pos + R;
    c->d_seg_ctx = c->d_slots + R; c->h_seg_ctx = c->h_slots + R;
  }
}

// ---- staging of a token-generation batch: 4 ints per row + the block-table rows that changed ----
// (block tables are resident on the device; a row is re-sent only when it differs from the host
//  shadow of what the device holds, and only entries that became live are range-checked)
struct DecodeStage { int row_lo = 0, row_hi = -1; };
// A staging pass that fails half way (a bad row behind good ones) has already recorded the good rows in the host
// shadow without sending them: forget the shadow, the next call re-sends every row.
struct ShadowGuard {
  mi_ctx* c;
  bool pushed = false;
  explicit ShadowGuard(mi_ctx* c_) : c(c_) {}
  ~ShadowGuard() { if (!pushed) c->bt_shadow_MB = -1; }
};
static void decode_stage_begin(mi_ctx* c, int MB, int B, DecodeStage& st) {
  const mi_model_config& k = c->cfg;
  if (c->bt_shadow_MB != MB) {   // another table width: nothing on the device can be reused
    c->bt_shadow.assign((size_t)k.max_num_seqs * MB, INT64_MIN);
    c->bt_shadow_MB = MB;
    std::fill(c->bt_checked.begin(), c->bt_checked.end(), 0);
  }
  st.row_lo = B;   // rows whose device copy is stale: [row_lo, row_hi]
  st.row_hi = -1;
}
static int decode_stage_row(mi_ctx* c, int b, int MB, int64_t id, int64_t pos, int64_t slot, int full, const int64_t* row,
                            DecodeStage& st) {
  const mi_model_config& k = c->cfg;
  const int bs = k.block_size;
  MI_CHECK(full >= 1 && full <= k.max_model_len, "full_context_lens out of range");
  const int need = ceil_div(full, bs);
  MI_CHECK(need <= MB, "block_table narrower than the context");
  int64_t* shadow = c->bt_shadow.data() + (size_t)b * MB;
  if (memcmp(row, shadow, (size_t)MB * 8) != 0) {   // new request in this row / block appended: re-send the row
    memcpy(shadow, row, (size_t)MB * 8);
    for (int j = 0; j < MB; ++j) c->h_bt[(size_t)b * MB + j] = (int32_t)row[j];
    c->bt_checked[b] = 0;
    st.row_lo = std::min(st.row_lo, b);
    st.row_hi = std::max(st.row_hi, b);
    c->bt_rows_sent += 1;
  } else {
    c->bt_rows_kept += 1;
  }
  for (int j = c->bt_checked[b]; j < need; ++j)   // only the entries that became live
    MI_CHECK(row[j] >= 0 && row[j] < k.num_blocks, "block_table entry out of range inside the live context");
  c->bt_checked[b] = std::max(c->bt_checked[b], need);
  MI_CHECK(pos >= 0 && pos < k.max_model_len, "position out of range");
  MI_CHECK(id >= 0 && id < k.vocab_size, "token id out of range");
  MI_CHECK(slot >= -1 && slot < (int64_t)k.num_blocks * bs, "slot out of range");
  c->h_ids[b] = (int32_t)id;
  c->h_pos[b] = (int32_t)pos;
  c->h_slots[b] = (int32_t)slot;
  c->h_ctx[b] = full;
  return MI_OK;
}
static int decode_stage_push(mi_ctx* c, int MB, const DecodeStage& st) {   // 4 x max_num_seqs ints, + the table rows that changed
  hipStream_t s = c->stream;
  MI_HIP(hipMemcpyAsync(c->d_dec, c->h_dec, (size_t)c->cfg.max_num_seqs * 16, hipMemcpyHostToDevice, s));
  if (st.row_hi >= st.row_lo)
    MI_HIP(hipMemcpyAsync(c->d_dec_bt + (size_t)st.row_lo * MB, c->h_dec_bt + (size_t)st.row_lo * MB,
                          (size_t)(st.row_hi - st.row_lo + 1) * MB * 4, hipMemcpyHostToDevice, s));
  return MI_OK;
}

// logits_out != null: the reference's CPU-sampling contract (fp32 logits of the last token of every
// row to the host).  tokens_out != null: on-device sampling -- the ids are sampled from the device
// logits and only B integers cross PCIe.
static int sample_on_device(mi_ctx* c, int nrows, int row0, const float* sampling_params, uint64_t seed,
                            int64_t* tokens_out) {
  const mi_model_config& k = c->cfg;
  hipStream_t s = c->stream;
  if (c->grp) {   // vocabulary slices -> rank 0's gather buffer (peer copy), rank 0 samples
    mi_ctx* c0 = c->grp->ranks[0];
    const size_t per_rank = (size_t)k.max_num_seqs * c->V_l;
    float* dst = c0->logits_all + (size_t)k.tp_rank * per_rank;
    if (c->grp->same_device) MI_HIP(hipMemcpyAsync(dst, c->logits, per_rank * 4, hipMemcpyDeviceToDevice, s));
    else MI_HIP(hipMemcpyPeerAsync(dst, c0->cfg.device_id, c->logits, k.device_id, per_rank * 4, s));
    MI_HIP(hipStreamSynchronize(s));
    MI_TRY(group_check_errors(c));
    MI_CHECK(c->grp->bar->wait(), "tensor-parallel group aborted");
    if (k.tp_rank != 0) return MI_OK;
  }
  const float* dparams = nullptr;
  bool all_greedy = true;   // (top_k = 1 rows draw nothing: the split argmax serves the whole batch)
  if (sampling_params)
    for (int i = 0; i < nrows; ++i) all_greedy = all_greedy && sampling_params[(size_t)(row0 + i) * 3] == 1.f;
  if (sampling_params && !all_greedy) {
    memcpy(c->h_sparams, sampling_params + (size_t)row0 * 3, (size_t)nrows * 3 * 4);
    MI_HIP(hipMemcpyAsync(c->d_sparams, c->h_sparams, (size_t)nrows * 3 * 4, hipMemcpyHostToDevice, s));
    dparams = c->d_sparams;
  }
  {
    Scope sc(c, MI_K_OTHER);
    if (c->collective()) MI_TRY(launch_sample_rows(c->logits_all, k.tp_degree, k.max_num_seqs, c->V_l, nrows, dparams, seed, row0, c->d_tokens, s, c->d_sample_scratch));
    else MI_TRY(launch_sample_rows(c->logits, 1, k.max_num_seqs, c->V_l, nrows, dparams, seed, row0, c->d_tokens, s, c->d_sample_scratch));
  }
  MI_HIP(hipMemcpyAsync(c->h_tokens, c->d_tokens, (size_t)nrows * 4, hipMemcpyDeviceToHost, s));
  g_ht.lap(3);
  MI_HIP(hipStreamSynchronize(s));
  g_ht.lap(4);
  for (int i = 0; i < nrows; ++i) tokens_out[row0 + i] = c->h_tokens[i];
  g_ht.lap(5);
  return MI_OK;
}

static int forward_impl(mi_ctx* c, int32_t B, int32_t S, const int64_t* input_ids, const int64_t* position_ids,
                        const int64_t* seq_ids, const int64_t* block_table, int32_t MB, const int64_t* slot_mapping,
                        int32_t SM, const int64_t* full_context_lens, const int64_t* computed_context_lens,
                        float* logits_out, const float* sampling_params, uint64_t seed, int64_t* tokens_out) {
  if (c && c->owned_group)   // the same call on every rank shard, each on its own thread; the shards fill disjoint columns
    return group_run(c->owned_group, [&](mi_ctx* rc, int) {
      return forward_impl(rc, B, S, input_ids, position_ids, seq_ids, block_table, MB, slot_mapping, SM, full_context_lens,
                          computed_context_lens, logits_out, sampling_params, seed, tokens_out);
    });
  MI_CHECK(c && c->finalized, "mi_forward before mi_finalize");
  MI_CHECK(input_ids && position_ids && block_table && slot_mapping && full_context_lens && computed_context_lens, "null argument");
  MI_CHECK((logits_out != nullptr) != (tokens_out != nullptr), "exactly one of logits_out / tokens_out");
  (void)seq_ids;  // block-layout KV: rows are addressed through block_table, not batch lines
  const mi_model_config& k = c->cfg;
  MI_CHECK(B >= 1 && B <= k.max_num_seqs, "batch size exceeds max_num_seqs");
  MI_CHECK(S >= 1 && MB >= 1 && MB <= c->MB_cap && SM >= 1, "bad S / MB / SM");
  MI_HIP(hipSetDevice(k.device_id));
  hipStream_t s = c->stream;
  const int V = k.vocab_size;
  const int bs = k.block_size;
  if (tokens_out && sampling_params)
    for (int b = 0; b < B; ++b) {
      const float tk = sampling_params[b * 3], tp = sampling_params[b * 3 + 1], tt = sampling_params[b * 3 + 2];
      MI_CHECK(tk >= 1.f && tp > 0.f && tp <= 1.f && tt > 0.f, "sampling_params rows must be (top_k >= 1, 0 < top_p <= 1, temperature > 0)");
    }

  auto check_row = [&](int b, int full, int n_new) -> int {
    MI_CHECK(full >= 1 && full <= k.max_model_len, "full_context_lens out of range");
    MI_CHECK(n_new >= 1, "nothing to compute: computed_context_lens >= full_context_lens");
    MI_CHECK(ceil_div(full, bs) <= MB, "block_table narrower than the context");
    for (int j = 0; j < ceil_div(full, bs); ++j) {
      const int64_t blk = block_table[(size_t)b * MB + j];
      MI_CHECK(blk >= 0 && blk < k.num_blocks, "block_table entry out of range inside the live context");
    }
    return MI_OK;
  };
  auto use_inputs = [&](bool decode_step) { set_input_views(c, decode_step); };

  if (S == 1) {  // ---- token generation -------------------------------------------------
    g_ht.start();
    use_inputs(true);
    DecodeStage st;
    ShadowGuard guard(c);
    decode_stage_begin(c, MB, B, st);
    for (int b = 0; b < B; ++b)
      MI_TRY(decode_stage_row(c, b, MB, input_ids[b], position_ids[b], slot_mapping[(size_t)b * SM], (int)full_context_lens[b],
                              block_table + (size_t)b * MB, st));
    auto push_inputs = [&]() -> int { return decode_stage_push(c, MB, st); };
    g_ht.lap(0);
    MI_TRY(push_inputs());
    guard.pushed = true;
    g_ht.lap(1);
    MI_TRY(capture_or_launch_decode(c, B, MB));
    g_ht.lap(2);
    c->last_B = B;
    c->last_MB = MB;
    if (tokens_out) MI_TRY(sample_on_device(c, B, 0, sampling_params, seed, tokens_out));
    else MI_TRY(fetch_logits(c, B, logits_out));
    g_ht.calls += 1;
    return prof_collect(c);
  }

  // ---- context encoding: one sequence at a time (the reference schedules ctx_batch_size 1,
  //      scheduler.py:116; loader.py:754-755) ------------------------------------------------
  use_inputs(false);
  auto push_inputs = [&]() -> int {   // the whole context-encoding input block in one copy
    MI_HIP(hipMemcpyAsync(c->d_inputs, c->h_inputs, c->inputs_elems * 4, hipMemcpyHostToDevice, s));
    return MI_OK;
  };
  for (int b = 0; b < B; ++b) {
    const int full = (int)full_context_lens[b], comp = (int)computed_context_lens[b];
    const int n_new = full - comp;
    MI_CHECK(comp >= 0 && full <= S, "context lengths inconsistent with the padded prompt");
    MI_TRY(check_row(b, full, n_new));
    MI_CHECK(n_new <= c->max_rows && n_new <= SM, "prompt longer than the largest bucket / slot_mapping");
    for (int t = 0; t < n_new; ++t) {
      const int64_t id = input_ids[(size_t)b * S + comp + t], pos = position_ids[(size_t)b * S + comp + t];
      const int64_t slot = slot_mapping[(size_t)b * SM + t];
      MI_CHECK(id >= 0 && id < V, "token id out of range");
      MI_CHECK(pos >= 0 && pos < k.max_model_len, "position out of range");
      MI_CHECK(slot >= -1 && slot < (int64_t)k.num_blocks * bs, "slot out of range");
      c->h_ids[t] = (int32_t)id;
      c->h_pos[t] = (int32_t)pos;
      c->h_slots[t] = (int32_t)slot;
    }
    for (int j = 0; j < MB; ++j) c->h_bt[j] = (int32_t)block_table[(size_t)b * MB + j];
    MI_TRY(push_inputs());
    MI_TRY(run_layers(c, n_new, false, 1, MB, comp, 1, n_new - 1));
    // (the copy engine reads the pinned block asynchronously: the next row's staging must not
    //  start before this row's copy has been consumed -- both fetch paths synchronize the stream)
    if (tokens_out) MI_TRY(sample_on_device(c, 1, b, sampling_params, seed, tokens_out));
    else MI_TRY(fetch_logits(c, 1, logits_out + (size_t)b * V));
  }
  return prof_collect(c);
}

int mi_forward(mi_ctx* c, int32_t B, int32_t S, const int64_t* input_ids, const int64_t* position_ids,
               const int64_t* seq_ids, const int64_t* block_table, int32_t MB, const int64_t* slot_mapping,
               int32_t SM, const int64_t* full_context_lens, const int64_t* computed_context_lens,
               float* logits_out) {
  MI_CHECK(logits_out, "null argument");
  return forward_impl(c, B, S, input_ids, position_ids, seq_ids, block_table, MB, slot_mapping, SM, full_context_lens,
                      computed_context_lens, logits_out, nullptr, 0, nullptr);
}

int mi_forward_tokens(mi_ctx* c, int32_t B, int32_t S, const int64_t* input_ids, const int64_t* position_ids,
                      const int64_t* seq_ids, const int64_t* block_table, int32_t MB, const int64_t* slot_mapping,
                      int32_t SM, const int64_t* full_context_lens, const int64_t* computed_context_lens,
                      const float* sampling_params, uint64_t seed, int64_t* tokens_out) {
  MI_CHECK(tokens_out, "null argument");
  return forward_impl(c, B, S, input_ids, position_ids, seq_ids, block_table, MB, slot_mapping, SM, full_context_lens,
                      computed_context_lens, nullptr, sampling_params, seed, tokens_out);
}

int mi_forward_chunked(mi_ctx* c, int32_t n_req, int32_t total, const int64_t* input_ids, const int64_t* position_ids,
                       const int64_t* slot_mapping, const int64_t* block_table, int32_t MB,
                       const int64_t* full_context_lens, const int64_t* computed_context_lens, float* logits_out,
                       const float* sampling_params, uint64_t seed, int64_t* tokens_out) {
  if (c && c->owned_group)
    return group_run(c->owned_group, [&](mi_ctx* rc, int) {
      return mi_forward_chunked(rc, n_req, total, input_ids, position_ids, slot_mapping, block_table, MB, full_context_lens,
                                computed_context_lens, logits_out, sampling_params, seed, tokens_out);
    });
  MI_CHECK(c && c->finalized, "mi_forward_chunked before mi_finalize");
  MI_CHECK(input_ids && position_ids && slot_mapping && block_table && full_context_lens && computed_context_lens, "null argument");
  MI_CHECK((logits_out != nullptr) != (tokens_out != nullptr), "exactly one of logits_out / tokens_out");
  const mi_model_config& k = c->cfg;
  MI_CHECK(n_req >= 1 && n_req <= k.max_num_seqs, "more requests than max_num_seqs");
  MI_CHECK(total >= n_req && total <= c->max_rows, "token batch larger than the largest context-encoding bucket");
  MI_CHECK(MB >= 1 && MB <= c->MB_cap, "bad block-table width");
  static const bool decode_shortcut = [] { const char* v = getenv("MI355X_CHUNKED_DECODE_SHORTCUT"); return !v || v[0] != '0'; }();
  if (decode_shortcut && total == n_req) {
    // every request contributes exactly one token: this IS a token-generation step (vLLM's native scheduler
    // sends those through the same ragged record once all prompts are encoded) -- the weight-streaming
    // kernels, the split-context attention and the step's hipGraph instead of a 1-row context encoding per request
    bool one_each = true;
    for (int i = 0; i < n_req; ++i) one_each = one_each && full_context_lens[i] - computed_context_lens[i] == 1;
    if (one_each)
      return forward_impl(c, n_req, 1, input_ids, position_ids, nullptr, block_table, MB, slot_mapping, 1, full_context_lens,
                          computed_context_lens, logits_out, sampling_params, seed, tokens_out);
  }
  MI_HIP(hipSetDevice(k.device_id));
  hipStream_t s = c->stream;
  const int bs = k.block_size, V = k.vocab_size;
  set_input_views(c, false);
  std::vector<Seg> segs((size_t)n_req);
  int row = 0;
  for (int i = 0; i < n_req; ++i) {
    const int full = (int)full_context_lens[i], comp = (int)computed_context_lens[i];
    const int n = full - comp;
    MI_CHECK(comp >= 0 && n >= 1 && full <= k.max_model_len, "chunk bounds: need 0 <= computed < full <= max_model_len");
    MI_CHECK(row + n <= total, "chunk lengths exceed the token batch");
    MI_CHECK(ceil_div(full, bs) <= MB, "block_table narrower than the context");
    for (int j = 0; j < MB; ++j) {
      const int64_t blk = block_table[(size_t)i * MB + j];
      if (j < ceil_div(full, bs)) MI_CHECK(blk >= 0 && blk < k.num_blocks, "block_table entry out of range inside the live context");
      c->h_bt[(size_t)i * MB + j] = (int32_t)blk;
    }
    for (int t = 0; t < n; ++t) {
      const int64_t id = input_ids[row + t], pos = position_ids[row + t], slot = slot_mapping[row + t];
      MI_CHECK(id >= 0 && id < V, "token id out of range");
      MI_CHECK(pos == comp + t, "positions of a chunk must continue its computed context");
      MI_CHECK(slot >= -1 && slot < (int64_t)k.num_blocks * bs, "slot out of range");
      c->h_ids[row + t] = (int32_t)id;
      c->h_pos[row + t] = (int32_t)pos;
      c->h_slots[row + t] = (int32_t)slot;
    }
    segs[i] = Seg{row, n, comp};
    c->h_ctx[i] = row + n - 1;   // the request's last row: where its logits come from
    c->h_seg_ctx[i] = full;      // keys the request attends to (the decode-attention launch of 1-token requests)
    row += n;
  }
  MI_CHECK(row == total, "chunk lengths do not add up to the token batch");
  MI_HIP(hipMemcpyAsync(c->d_inputs, c->h_inputs, c->inputs_elems * 4, hipMemcpyHostToDevice, s));
  MI_TRY(run_layers(c, total, false, 1, MB, 0, n_req, 0, segs.data(), n_req));
  if (tokens_out) MI_TRY(sample_on_device(c, n_req, 0, sampling_params, seed, tokens_out));
  else MI_TRY(fetch_logits(c, n_req, logits_out));
  return prof_collect(c);
}

// Fused speculation: k chained token-generation steps of the DRAFT context (its greedy token feeds its
// next step on the device), ONE token-generation pass of the TARGET over the B * k candidate rows
// (row b * k + i = candidate i of sequence b at position pos_b + i: the M <= 16 weight-streaming
// kernels read every weight once for all of them), greedy acceptance on the device.  Reference: NxDI's
// fused speculation behind /root/reference/vllm_neuron/worker/neuronx_distributed_model_loader.py:349-355
// (output contract: accepted tokens 0-padded to the speculation length + next position ids, re-masked
// by _remask_fused_spec_output :308-333) and the runner's slots for the speculated positions
// (neuronx_distributed_model_runner.py:825-830; here taken from the block table, so a window may
// cross a block boundary).  The draft runs k - 1 steps (candidates 1 .. k - 1); when a step accepts
// every candidate, the last one (position pos - 1 of the next call) has not been through the draft:
// the caller names it in draft_catchup_ids and the draft's FIRST step of the next call takes it as an
// extra row (same launch: its K/V is written before the attention of the sequence's own row reads
// it).  A missing catch-up never changes the text, only what the draft proposes.  K/V written for
// rejected candidates is overwritten by the next call.
int mi_forward_spec(mi_ctx* t, mi_ctx* d, int32_t B, int32_t k, const int64_t* input_ids, const int64_t* position_ids,
                    const int64_t* block_table, int32_t MB, const int64_t* draft_catchup_ids, int64_t* accepted_out,
                    int64_t* next_pos_out) {
  MI_CHECK(t && d && t != d && t->finalized && d->finalized, "mi_forward_spec: two finalized contexts (target, draft)");
  MI_CHECK(!d->owned_group && !d->grp && !d->collective() && !t->grp && (t->owned_group || !t->collective()),
           "mi_forward_spec: the draft is a single-GPU context; the target a single-GPU context or an in-process tensor-parallel group");
  MI_CHECK(input_ids && position_ids && block_table && accepted_out && next_pos_out, "null argument");
  // the target's rank shards (one: the context itself).  Everything the step keeps on the device besides the
  // shards' own inputs lives with shard 0, on whose GPU (and stream) the draft runs.
  mi_group* g = t->owned_group;
  std::vector<mi_ctx*> ranks = g ? g->ranks : std::vector<mi_ctx*>{t};
  mi_ctx* t0 = ranks[0];
  const mi_model_config& kt = t0->cfg;
  const mi_model_config& kd = d->cfg;
  MI_CHECK(kt.device_id == kd.device_id && kt.block_size == kd.block_size && kt.num_blocks == kd.num_blocks &&
           kt.vocab_size == kd.vocab_size && kt.max_model_len == kd.max_model_len,
           "mi_forward_spec: target (rank 0) and draft must share device, block_size, num_blocks, vocab_size and max_model_len");
  MI_CHECK(B >= 1 && k >= 1 && B * k <= kt.max_num_seqs && B <= kd.max_num_seqs,
           "mi_forward_spec: B * k rows exceed the target's max_num_seqs (or B the draft's)");
  int n_catch = 0;
  if (draft_catchup_ids)
    for (int b = 0; b < B; ++b) n_catch += draft_catchup_ids[b] >= 0 && position_ids[b] >= 1;
  MI_CHECK(B + n_catch <= kd.max_num_seqs, "mi_forward_spec: catch-up rows exceed the draft's max_num_seqs (create it with 2 x B rows)");
  MI_CHECK(MB >= 1 && MB <= t0->MB_cap && MB <= d->MB_cap, "bad block-table width");
  for (int b = 0; b < B; ++b) {
    const int64_t pos = position_ids[b];
    MI_CHECK(pos >= 0 && pos < kt.max_model_len, "position out of range");
    MI_CHECK(ceil_div((int)pos + (int)std::max<int64_t>(1, std::min<int64_t>(k, kt.max_model_len - pos - 1)), kt.block_size) <= MB,
             "block_table narrower than the speculation window");
  }
  const size_t ms = (size_t)kt.max_num_seqs;
  const int bs = kt.block_size;
  // Candidate rows that fit the model length.  A step that feeds position pos holds pos + 1 tokens and may
  // yield one token per candidate row, and vLLM caps a sequence at max_model_len tokens (its runner's
  // token table is that wide): at most max_model_len - pos - 1 rows, and never fewer than the plain step's one.
  auto lim_of = [&](int b) { return (int)std::max<int64_t>(1, std::min<int64_t>(k, kt.max_model_len - position_ids[b] - 1)); };
  auto slot_of = [&](int b, int64_t p) { return block_table[(size_t)b * MB + p / bs] * bs + p % bs; };
  static const bool shared_cols = [] { const char* v = getenv("MI355X_SPEC_SHARED_ATTN"); return !v || v[0] != '0'; }();
  // the k rows of a sequence share the attention's MFMA columns where they fit (q heads per kv head x k <= 16)
  const int rows_per_seq = (shared_cols && (t0->nh_l / t0->nkv_l) * k <= 16) ? k : 1;

  // every shard: its copy of the B * k candidate rows (ids of candidates 1.. arrive on the device)
  auto stage_target = [&](mi_ctx* rc) -> int {
    set_input_views(rc, true);
    DecodeStage st;
    ShadowGuard guard(rc);
    decode_stage_begin(rc, MB, B * k, st);
    for (int b = 0; b < B; ++b) {
      const int64_t pos = position_ids[b];
      const int lim = lim_of(b);
      for (int i = 0; i < k; ++i) {
        const int64_t p = pos + std::min(i, lim - 1);
        MI_TRY(decode_stage_row(rc, b * k + i, MB, i == 0 ? input_ids[b] : 0, p, i < lim ? slot_of(b, p) : -1, (int)p + 1,
                                block_table + (size_t)b * MB, st));
      }
    }
    MI_TRY(decode_stage_push(rc, MB, st));
    guard.pushed = true;
    return MI_OK;
  };
  // shard 0: the draft's k - 1 chained steps; candidate i + 1 lands in shard 0's input ids and in `cand`
  auto draft_chain = [&]() -> int {
    MI_HIP(hipSetDevice(kt.device_id));
    if (d->stream != t0->stream) {   // one stream for the draft and shard 0: the chain is ordered by enqueue order alone
      if (d->stream_owned) {
        MI_HIP(hipStreamSynchronize(d->stream));
        hipStreamDestroy(d->stream);
      } else {
        MI_HIP(hipDeviceSynchronize());   // the stream of an earlier target, possibly closed since
      }
      d->stream = t0->stream;
      d->stream_owned = false;
    }
    hipStream_t s = t0->stream;
    if (!t0->d_spec) {   // [cand][limit][pos0][out][next_pos], max_num_seqs ints each
      MI_HIP(hipMalloc(&t0->d_spec, 5 * ms * 4));
      MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&t0->h_spec), 5 * ms * 4, hipHostMallocDefault));
      memset(t0->h_spec, 0, 5 * ms * 4);
    }
    int32_t *d_cand = t0->d_spec, *d_limit = t0->d_spec + ms;
    int32_t *h_limit = t0->h_spec + ms, *h_pos0 = t0->h_spec + 2 * ms;
    set_input_views(d, true);
    DecodeStage st_d;
    ShadowGuard guard_d(d);
    decode_stage_begin(d, MB, B + n_catch, st_d);
    int catch_row = B;
    for (int b = 0; b < B; ++b) {
      const int64_t pos = position_ids[b];
      const int64_t* row = block_table + (size_t)b * MB;
      MI_TRY(decode_stage_row(d, b, MB, input_ids[b], pos, slot_of(b, pos), (int)pos + 1, row, st_d));
      if (draft_catchup_ids && draft_catchup_ids[b] >= 0 && pos >= 1)   // the token in front of it, not yet in the draft's K/V
        MI_TRY(decode_stage_row(d, catch_row++, MB, draft_catchup_ids[b], pos - 1, slot_of(b, pos - 1), (int)pos, row, st_d));
      h_limit[b] = lim_of(b);
      h_pos0[b] = (int32_t)pos;
    }
    MI_TRY(decode_stage_push(d, MB, st_d));
    guard_d.pushed = true;
    MI_HIP(hipMemcpyAsync(d_limit, h_limit, 2 * ms * 4, hipMemcpyHostToDevice, s));   // limit + pos0 (adjacent)
    for (int step = 0; step + 1 < k; ++step) {
      MI_TRY(capture_or_launch_decode(d, step == 0 ? B + n_catch : B, MB));   // the catch-up rows ride on the first step only
      MI_TRY(launch_sample_rows(d->logits, 1, kd.max_num_seqs, d->V_l, B, nullptr, 0, 0, d->d_tokens, s, d->d_sample_scratch));
      MI_TRY(launch_spec_advance(B, k, step, d->d_tokens, d->d_dec, kd.max_num_seqs, d->d_dec_bt, MB, bs, d_limit, t0->d_ids,
                                 d_cand, s));
    }
    d->last_B = B;
    d->last_MB = MB;
    return MI_OK;
  };
  auto target_pass = [&](mi_ctx* rc) -> int {
    rc->attn_rows_per_seq = rows_per_seq;
    const int rc_pass = capture_or_launch_decode(rc, B * k, MB);
    rc->attn_rows_per_seq = 1;
    rc->last_B = B * k;
    rc->last_MB = MB;
    return rc_pass;
  };
  auto accept = [&]() -> int {   // shard 0: its d_tokens hold the target's greedy choice of every row
    hipStream_t s = t0->stream;
    int32_t *d_cand = t0->d_spec, *d_limit = t0->d_spec + ms, *d_pos0 = t0->d_spec + 2 * ms, *d_out = t0->d_spec + 3 * ms;
    int32_t* h_out = t0->h_spec + 3 * ms;
    MI_TRY(launch_spec_accept(B, k, t0->d_tokens, d_cand, d_limit, d_pos0, d_out, d_out + ms, s));
    MI_HIP(hipMemcpyAsync(h_out, d_out, 2 * ms * 4, hipMemcpyDeviceToHost, s));   // out + next_pos (adjacent)
    MI_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < B * k; ++i) accepted_out[i] = h_out[i];
    for (int b = 0; b < B; ++b) next_pos_out[b] = h_out[ms + b];
    return MI_OK;
  };

  if (!g) {
    MI_HIP(hipSetDevice(kt.device_id));
    MI_TRY(stage_target(t0));
    MI_TRY(draft_chain());
    MI_TRY(target_pass(t0));
    MI_TRY(launch_sample_rows(t0->logits, 1, kt.max_num_seqs, t0->V_l, B * k, nullptr, 0, 0, t0->d_tokens, t0->stream,
                              t0->d_sample_scratch));
    return accept();
  }
  // ---- tensor-parallel target: the same phases, every shard on its own thread ----
  hipEvent_t ids_ready = nullptr;
  if (!g->lockstep) {
    MI_HIP(hipSetDevice(kt.device_id));   // the event is recorded on shard 0's stream: it belongs to shard 0's device
    MI_HIP(hipEventCreateWithFlags(&ids_ready, hipEventDisableTiming));
  }
  int rc = group_run(g, [&](mi_ctx* rcx, int r) -> int {
    MI_TRY(stage_target(rcx));
    if (r != 0) return MI_OK;
    MI_TRY(draft_chain());
    if (ids_ready) MI_HIP(hipEventRecord(ids_ready, t0->stream));
    return MI_OK;
  });
  if (rc == MI_OK)
    rc = group_run(g, [&](mi_ctx* rcx, int r) -> int {
      if (r != 0) {   // the candidates' ids: from shard 0's inputs into this shard's (one stream and GPU in loopback mode)
        if (ids_ready) {
          MI_HIP(hipStreamWaitEvent(rcx->stream, ids_ready, 0));
          MI_HIP(hipMemcpyPeerAsync(rcx->d_ids, rcx->cfg.device_id, t0->d_ids, kt.device_id, (size_t)B * k * 4, rcx->stream));
        } else {
          MI_HIP(hipMemcpyAsync(rcx->d_ids, t0->d_ids, (size_t)B * k * 4, hipMemcpyDeviceToDevice, rcx->stream));
        }
      }
      return target_pass(rcx);
    });
  std::vector<int64_t> greedy((size_t)B * k);
  if (rc == MI_OK)   // vocabulary-parallel logits -> shard 0's sampler (the on-device sampling path of the group)
    rc = group_run(g, [&](mi_ctx* rcx, int) -> int { return sample_on_device(rcx, B * k, 0, nullptr, 0, greedy.data()); });
  if (ids_ready) hipEventDestroy(ids_ready);
  MI_TRY(rc);
  MI_HIP(hipSetDevice(kt.device_id));
  return accept();
}

int mi_replay_decode(mi_ctx* c, int32_t steps, float* elapsed_ms) {
  MI_CHECK(c && c->finalized && elapsed_ms, "bad argument");
  if (c->owned_group) {   // every shard replays its step; the slowest rank is the step
    std::vector<float> ms(c->owned_group->T, 0.f);
    MI_TRY(group_run(c->owned_group, [&](mi_ctx* rc, int r) { return mi_replay_decode(rc, steps, &ms[r]); }));
    *elapsed_ms = 0.f;
    for (float v : ms) *elapsed_ms = std::max(*elapsed_ms, v);
    return MI_OK;
  }
  MI_CHECK(c->last_B > 0, "mi_replay_decode needs a preceding token-generation mi_forward");
  MI_CHECK(steps >= 1, "steps must be >= 1");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  set_input_views(c, true);
  hipEvent_t a, b;
  MI_HIP(hipEventCreate(&a));
  MI_HIP(hipEventCreate(&b));
  MI_HIP(hipEventRecord(a, c->stream));
  for (int i = 0; i < steps; ++i) MI_TRY(capture_or_launch_decode(c, c->last_B, c->last_MB));
  MI_HIP(hipEventRecord(b, c->stream));
  MI_HIP(hipEventSynchronize(b));
  MI_HIP(hipEventElapsedTime(elapsed_ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  return prof_collect(c);
}

int mi_replay_decode_classes(mi_ctx* c, int32_t steps, uint32_t class_mask, float* elapsed_ms) {
  MI_CHECK(c && c->finalized && elapsed_ms && class_mask != 0, "bad argument");
  MI_CHECK(!c->owned_group && !c->collective(), "mi_replay_decode_classes: single-GPU contexts only");
  c->class_mask = class_mask;
  int rc = mi_replay_decode(c, 2, elapsed_ms);          // capture (eager pass + graph) and warm
  if (rc == MI_OK) rc = mi_replay_decode(c, steps, elapsed_ms);
  c->class_mask = 0xffffffffu;
  return rc;
}

int mi_kv_stats(mi_ctx* c, mi_kv_stats_t* o) {
  MI_CHECK(c && o, "null argument");
  if (c->owned_group)   // per-GPU figures of rank 0 (every shard holds the same amounts)
    return group_run(c->owned_group, [&](mi_ctx* rc, int r) { return r == 0 ? mi_kv_stats(rc, o) : MI_OK; });
  MI_HIP(hipSetDevice(c->cfg.device_id));
  size_t fr = 0, tot = 0;
  MI_HIP(hipMemGetInfo(&fr, &tot));
  o->kv_bytes = (int64_t)c->kv_bytes; o->weight_bytes = (int64_t)c->weight_bytes;
  // before mi_finalize: what it WILL allocate besides the KV pool (the worker subtracts it from the
  // free memory it reports to vLLM); afterwards: what it did allocate
  o->workspace_bytes = c->finalized ? (int64_t)c->workspace_bytes : (int64_t)workspace_estimate(c);
  o->device_free_bytes = (int64_t)fr; o->device_total_bytes = (int64_t)tot;
  o->num_blocks = c->cfg.num_blocks; o->block_size = c->cfg.block_size;
  o->num_kv_heads_local = c->nkv_l; o->head_dim = c->hd; o->num_layers = c->cfg.num_layers;
  o->block_table_rows_sent = c->bt_rows_sent; o->block_table_rows_kept = c->bt_rows_kept;
  return MI_OK;
}

float* mi_logits_buffer(mi_ctx* c) {
  if (!c || c->owned_group || c->collective() || !c->finalized) return nullptr;
  return c->h_logits;
}

void* mi_stream(mi_ctx* c) {
  if (c && c->owned_group) return (void*)c->owned_group->ranks[0]->stream;
  return c ? (void*)c->stream : nullptr;
}

int mi_profile_enable(mi_ctx* c, int32_t on) {
  MI_CHECK(c, "null argument");
  if (c->owned_group) return group_run(c->owned_group, [&](mi_ctx* rc, int) { return mi_profile_enable(rc, on); });
  MI_TRY(prof_collect(c));
  c->prof.on = on != 0;
  if (on) {
    memset(c->prof.launches, 0, sizeof c->prof.launches);
    memset(c->prof.ms, 0, sizeof c->prof.ms);
    c->prof.gemv_bytes = 0;
  }
  return MI_OK;
}
int mi_profile_read(mi_ctx* c, int32_t* launches, float* ms, double* gemv_weight_bytes) {
  MI_CHECK(c && launches && ms, "null argument");
  if (c->owned_group)   // rank 0's timeline (the shards run the same launches)
    return group_run(c->owned_group, [&](mi_ctx* rc, int r) {
      int32_t l2[MI_K_NUM];
      float m2[MI_K_NUM];
      return r == 0 ? mi_profile_read(rc, launches, ms, gemv_weight_bytes) : mi_profile_read(rc, l2, m2, nullptr);
    });
  MI_TRY(prof_collect(c));
  for (int i = 0; i < MI_K_NUM; ++i) { launches[i] = c->prof.launches[i]; ms[i] = c->prof.ms[i]; }
  if (gemv_weight_bytes) *gemv_weight_bytes = c->prof.gemv_bytes;
  return MI_OK;
}

int mi_tp_info(mi_ctx* c, mi_tp_info_t* out) {
  MI_CHECK(c && out, "null argument");
  MI_CHECK(c->owned_group != nullptr, "mi_tp_info: not an in-process tensor-parallel context (tp_rank = MI_TP_ALL_RANKS, tp_degree > 1)");
  return group_info(c->owned_group, out);
}

int mi_tp_unique_id(void* out128) {
  MI_CHECK(out128, "null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) { set_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); return MI_ECOMM; }
  memcpy(out128, &id, 128);
  return MI_OK;
}
int mi_tp_init(mi_ctx* c, const void* id128) {
  MI_CHECK(c && id128, "null argument");
  MI_CHECK(!c->owned_group && !c->grp, "mi_tp_init on an in-process tensor-parallel context");
  MI_CHECK(!c->comm, "mi_tp_init called twice");
  MI_HIP(hipSetDevice(c->cfg.device_id));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&c->comm, c->cfg.tp_degree, id, c->cfg.tp_rank);
  if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); return MI_ECOMM; }
  return MI_OK;
}

int mi_tp_init_transport(mi_ctx* c, mi_allreduce_fn all_reduce, mi_allgather_fn all_gather, void* user) {
  MI_CHECK(c && all_reduce && all_gather, "null argument");
  MI_CHECK(!c->owned_group, "mi_tp_init_transport on an in-process tensor-parallel context");
  MI_CHECK(!c->collective(), "tensor-parallel transport already set");
  MI_CHECK(!c->finalized, "mi_tp_init_transport must precede mi_finalize");
  c->xport_allreduce = all_reduce;
  c->xport_allgather = all_gather;
  c->xport_user = user;
  return MI_OK;
}

#ifdef MI_TRACE
// dev builds: copy an internal activation buffer to the host (0 resid[0], 1 resid[1], 2 q, 3 attn_out, 4 act)
int mi_debug_copy(mi_ctx* c, int which, void* host, size_t bytes) {
  const void* src = which == 0 ? (const void*)c->resid[0] : which == 1 ? (const void*)c->resid[1] : which == 2 ? (const void*)c->qbuf
                    : which == 3 ? (const void*)c->attn_out : (const void*)c->act;
  MI_HIP(hipStreamSynchronize(c->stream));
  MI_HIP(hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost));
  return MI_OK;
}
#endif

// ---- per-kernel entry points ------------------------------------------------------------
int mi_op_tp_all_reduce(mi_ctx* c, float* const* bufs, size_t count) {
  MI_CHECK(c && c->owned_group && c->finalized && bufs, "mi_op_tp_all_reduce needs a finalized in-process tensor-parallel context");
  return group_run(c->owned_group, [&](mi_ctx* rc, int r) -> int {
    MI_TRY(group_all_reduce(rc, bufs[r], count));
    MI_HIP(hipStreamSynchronize(rc->stream));
    return group_check_errors(rc);
  });
}

int mi_op_sample(const float* logits, int32_t B, int32_t V, const float* sampling_params, uint64_t seed,
                 int32_t* tokens_out, void* stream) {
  MI_CHECK(logits && tokens_out && B >= 1 && V >= 1, "bad argument");
  return launch_sample_rows(logits, 1, B, V, B, sampling_params, seed, 0, tokens_out, (hipStream_t)stream);
}
size_t mi_op_sample_scratch_bytes(int32_t B) { return B >= 1 ? sample_scratch_bytes(B) : 0; }
int mi_op_sample_ws(const float* logits, int32_t B, int32_t V, const float* sampling_params, uint64_t seed,
                    int32_t* tokens_out, void* scratch, size_t scratch_bytes, void* stream) {
  MI_CHECK(logits && tokens_out && B >= 1 && V >= 1, "bad argument");
  MI_CHECK(scratch && scratch_bytes >= sample_scratch_bytes(B), "mi_op_sample_ws: scratch smaller than mi_op_sample_scratch_bytes(B)");
  return launch_sample_rows(logits, 1, B, V, B, sampling_params, seed, 0, tokens_out, (hipStream_t)stream, scratch);
}

int mi_op_quantize_weight(const float* w, int32_t N, int32_t K, int32_t wd, int32_t qt, void* tiled_out,
                          float* scale_out, void* stream) {
  MI_CHECK(w && tiled_out && scale_out, "null argument");
  MI_CHECK(N % 16 == 0 && K % 64 == 0, "N % 16 == 0 and K % 64 == 0 required");
  float* rowmax = nullptr;
  MI_HIP(hipMalloc(reinterpret_cast<void**>(&rowmax), ((size_t)N + 16) * 4));
  QuantJob j{};
  j.src = w; j.ld = K; j.src_row0 = 0; j.src_col0 = 0; j.n_rows = N; j.K = K; j.rowmap = ROWMAP_PLAIN; j.hd = 0;
  j.dst_row0 = 0; j.wd = wd; j.quant_type = qt; j.dst = tiled_out; j.dst_scale = scale_out; j.tmp_rowmax = rowmax;
  j.src_rows_total = N;
  int rc = run_quant_job(j, (hipStream_t)stream);
  hipStreamSynchronize((hipStream_t)stream);
  hipFree(rowmax);
  return rc;
}
int mi_op_untile_weight(const void* tiled, int32_t N, int32_t K, int32_t wd, void* q_out, void* stream) {
  MI_CHECK(tiled && q_out, "null argument");
  return launch_untile(tiled, N, K, wd, q_out, (hipStream_t)stream);
}
int mi_op_qlinear(const void* x, int32_t M, const void* w_tiled, const float* scale, const float* bias, int32_t N,
                  int32_t K, int32_t wd, float* y, int32_t force_path, void* stream) {
  MI_CHECK(x && w_tiled && scale && y, "null argument");
  LinearW W{w_tiled, N, K, wd};
  EpiArgs e{};
  e.scale = scale; e.bias = bias; e.out_f32 = y; e.ld_out = N;
  ProArgs p{};
  p.x = reinterpret_cast<const uint16_t*>(x); p.ldx = K;
  const bool gemv = force_path == 1 || (force_path == 0 && gemv_fits(M, K));
  if (gemv) return launch_gemv(W, M, PRO_BF16, p, EPI_F32, e, (hipStream_t)stream);
  if (force_path == 3) return launch_gemm_wide(W, M, p.x, K, EPI_F32, e, (hipStream_t)stream);
  if (force_path >= 4 && force_path <= 7) {   // the wide GEMM with its K dimension split: 4 / 5 = 2 / 4 slices at 128-token blocks, 6 / 7 at 256
    const int ks = (force_path & 1) ? 4 : 2, bm = force_path >= 6 ? 256 : 128;
    float* slab = nullptr;
    const size_t bytes = (size_t)ks * M * N * 4;
    MI_HIP(hipMalloc(reinterpret_cast<void**>(&slab), bytes));
    int rc = launch_gemm_wide(W, M, p.x, K, EPI_F32, e, (hipStream_t)stream, slab, bytes, nullptr, bm, ks);
    hipStreamSynchronize((hipStream_t)stream);
    hipFree(slab);
    return rc;
  }
  return launch_gemm(W, M, p.x, K, EPI_F32, e, (hipStream_t)stream);
}
int mi_op_qlinear_a8(const void* x, int32_t M, const void* w_tiled, const float* scale, const float* bias, int32_t N,
                     int32_t K, float* y, void* stream) {
  MI_CHECK(x && w_tiled && scale && y, "null argument");
  LinearW W{w_tiled, N, K, MI_W_F8E4M3};
  MI_CHECK(gemm_a8_supported(W), "qlinear_a8: needs K % 128 == 0 and N % 16 == 0");
  uint8_t* x8 = nullptr;
  float* xs = nullptr;
  MI_HIP(hipMalloc(reinterpret_cast<void**>(&x8), (size_t)M * K));
  MI_HIP(hipMalloc(reinterpret_cast<void**>(&xs), (size_t)M * 4));
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_rowquant_fp8(reinterpret_cast<const uint16_t*>(x), M, K, K, x8, xs, s);
  if (rc == MI_OK) {
    EpiArgs e{};
    e.scale = scale; e.bias = bias; e.out_f32 = y; e.ld_out = N; e.row_scale = xs;
    rc = launch_gemm_a8(W, M, x8, M, EPI_F32, e, s);
  }
  hipStreamSynchronize(s);
  hipFree(x8);
  hipFree(xs);
  return rc;
}
int mi_op_rmsnorm(const float* x, const float* g, int32_t T, int32_t H, float eps, void* y, void* stream) {
  MI_CHECK(x && g && y, "null argument");
  return launch_norm_rows(x, nullptr, nullptr, g, T, H, eps, reinterpret_cast<uint16_t*>(y), (hipStream_t)stream);
}
int mi_op_kv_write(const void* k, const void* v, const int64_t* slots, int32_t T, int32_t nkv, int32_t hd, void* pool,
                   int32_t num_blocks, int32_t block_size, void* stream) {
  MI_CHECK(k && v && slots && pool, "null argument");
  uint16_t* kp = reinterpret_cast<uint16_t*>(pool);
  uint16_t* vp = kp + (size_t)num_blocks * nkv * block_size * hd;
  return launch_kv_write(reinterpret_cast<const uint16_t*>(k), reinterpret_cast<const uint16_t*>(v), slots, T, nkv, hd,
                         kp, vp, block_size, (hipStream_t)stream);
}
int64_t mi_op_attn_scratch_bytes(int32_t B, int32_t nh, int32_t hd) { return (int64_t)attn_scratch_bytes(B, nh, hd); }
int mi_op_paged_attn_decode(const void* q, const void* pool, int32_t num_blocks, int32_t block_size,
                            const int32_t* block_table, int32_t MB, const int32_t* ctx_lens, int32_t B, int32_t nh,
                            int32_t nkv, int32_t hd, void* out, void* scratch, void* stream) {
  MI_CHECK(q && pool && block_table && ctx_lens && out && scratch, "null argument");
  const uint16_t* kp = reinterpret_cast<const uint16_t*>(pool);
  const uint16_t* vp = kp + (size_t)num_blocks * nkv * block_size * hd;
  return launch_attn_decode(reinterpret_cast<const uint16_t*>(q), kp, vp, block_size, block_table, MB, ctx_lens, B, nh,
                            nkv, hd, reinterpret_cast<uint16_t*>(out), scratch, (hipStream_t)stream);
}
int mi_op_paged_attn_prefill(const void* q, int32_t T, int32_t q_pos0, const void* pool, int32_t num_blocks,
                             int32_t block_size, const int32_t* block_table, int32_t MB, int32_t nh, int32_t nkv,
                             int32_t hd, void* out, void* stream) {
  MI_CHECK(q && pool && block_table && out, "null argument");
  MI_CHECK(ceil_div(q_pos0 + T, block_size) <= MB, "block_table narrower than the context");
  const uint16_t* kp = reinterpret_cast<const uint16_t*>(pool);
  const uint16_t* vp = kp + (size_t)num_blocks * nkv * block_size * hd;
  return launch_attn_prefill(reinterpret_cast<const uint16_t*>(q), T, q_pos0, kp, vp, block_size, block_table, nh, nkv,
                             hd, reinterpret_cast<uint16_t*>(out), (hipStream_t)stream);
}

}  // extern "C"
