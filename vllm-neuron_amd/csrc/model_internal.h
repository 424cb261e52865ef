// Internal state of a libmi355x_vllm context, shared by model.hip (the model call) and
// tp_group.hip (tensor parallelism inside one process).
#pragma once
#include <rccl/rccl.h>

#include <map>
#include <string>
#include <vector>

#include "linear_kernels.h"

namespace mi {

struct Linear {
  void* w = nullptr;
  float* scale = nullptr;
  float* bias = nullptr;
  int N = 0, K = 0, wd = MI_W_BF16;
  LinearW view() const { return LinearW{w, N, K, wd}; }
  size_t bytes() const { return (size_t)N * K * elem_bytes(wd); }
};

struct LayerW {
  Linear qkv, o, gu, down;
  float* g_in = nullptr;
  float* g_post = nullptr;
};

struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int> cls;
  int launches[MI_K_NUM] = {0};
  float ms[MI_K_NUM] = {0};
  double gemv_bytes = 0;
};

}  // namespace mi

struct mi_group;

using mi::LayerW;
using mi::Linear;
using mi::Prof;

struct mi_ctx {
  mi_model_config cfg;
  hipStream_t stream = nullptr;
  bool finalized = false;
  // per-rank geometry
  int H = 0, hd = 0, nh_l = 0, nh_real = 0, qh0 = 0, nkv_l = 0, kvh0 = 0, q_dim = 0, kv_dim = 0, I_l = 0, V_l = 0, max_rows = 0;
  std::vector<LayerW> layers;
  uint16_t* embed = nullptr;
  Linear lm_head;
  float* g_final = nullptr;
  bool have_lm_head = false;
  // KV pool: [L][2][NB][nkv_l][bs][hd] bf16
  uint16_t* kv_pool = nullptr;
  size_t kv_half = 0;  // elements of one K (or V) pool of one layer
  float *rope_cos = nullptr, *rope_sin = nullptr;
  // weight-load staging
  void* stage_raw = nullptr;
  float* stage_f32 = nullptr;
  float* rowmax = nullptr;
  size_t stage_elems = 0;
  // activations
  float* resid[2] = {nullptr, nullptr};
  float* partial = nullptr;
  uint16_t *xn = nullptr, *qbuf = nullptr, *attn_out = nullptr, *act = nullptr;
  uint8_t* x8 = nullptr;        // FP8-activation GEMM input [rows, K]
  float* x8_scale = nullptr;    // its per-token scales [rows]
  float* splitk_ws = nullptr;   // fp32 K-split slabs of short-prompt GEMMs
  mi::SlabSum pend;               // a residual projection's K-split sum left for the next row norm (run_linear)
  const float* pend_in = nullptr; // ... the residual it adds to
  float* pend_out = nullptr;      // ... and where residual + projection goes
  size_t splitk_ws_bytes = 0;
  float* logits = nullptr;      // [max_num_seqs, V_l]
  float* logits_all = nullptr;  // [tp, max_num_seqs, V_l] (tp > 1)
  void* attn_scratch = nullptr;
  // step inputs
  int32_t *d_ids = nullptr, *d_pos = nullptr, *d_slots = nullptr, *d_bt = nullptr, *d_ctx = nullptr;
  int32_t *d_seg_ctx = nullptr, *h_seg_ctx = nullptr;   // ragged records: context length of every request
  int32_t *h_ids = nullptr, *h_pos = nullptr, *h_slots = nullptr, *h_bt = nullptr, *h_ctx = nullptr;
  // d_* / h_* above are the CURRENT views: context encoding uses slices of one block (one H2D per
  // call: [block table row][ids][positions][slots]); token generation uses a second, small block
  // [context lengths][ids][positions][slots] (one H2D of 4 x max_num_seqs ints per step) and block
  // tables that STAY on the device: a row is re-sent only when the caller's row differs from the
  // host shadow of what the device holds (a new request in the row, a block appended), and only
  // the entries that became live since the last call are validated (the reference
  // rebuilds and re-validates every table every step, runner.py:798-832, 887-917).
  int32_t *d_inputs = nullptr, *h_inputs = nullptr;      // context-encoding block
  size_t inputs_elems = 0;
  int32_t *d_dec = nullptr, *h_dec = nullptr;            // token-generation block: [ctx][ids][pos][slots], max_num_seqs each
  int32_t *d_dec_bt = nullptr, *h_dec_bt = nullptr;      // resident block tables [max_num_seqs][MB of the last call]
  std::vector<int64_t> bt_shadow;                        // the caller's rows the device tables were made from
  std::vector<int> bt_checked;                           // per row: leading entries validated so far
  int bt_shadow_MB = 0;
  long bt_rows_sent = 0, bt_rows_kept = 0;               // statistics (mi_kv_stats)
  float* h_logits = nullptr;
  // on-device sampling: (top_k, top_p, temperature) rows and the sampled ids
  float *d_sparams = nullptr, *h_sparams = nullptr;
  int32_t *d_tokens = nullptr, *h_tokens = nullptr;
  int attn_rows_per_seq = 1;                        // token generation: consecutive batch rows that are ONE sequence (speculation)
  void* d_sample_scratch = nullptr;                 // partial (value, index) pairs of the split argmax
  int32_t *d_spec = nullptr, *h_spec = nullptr;   // fused speculation (mi_forward_spec): candidates, limits, outputs
  int MB_cap = 0;
  size_t weight_bytes = 0, workspace_bytes = 0, kv_bytes = 0;
  std::map<int, hipGraphExec_t> graphs;  // token-generation graph per (B * 65536 + MB)
  int last_B = 0, last_MB = 0;           // shape of the last token-generation call (mi_replay_decode)
  Prof prof;
  // kernel classes the token-generation step launches (mi_replay_decode_classes measures a step with
  // some of them left out; the results of such a step are meaningless, its timing is the point)
  uint32_t class_mask = 0xffffffffu;
  bool runs(int cls) const { return (class_mask >> cls) & 1u; }
  ncclComm_t comm = nullptr;
  // caller-supplied collectives in place of RCCL (mi_tp_init_transport)
  mi_allreduce_fn xport_allreduce = nullptr;
  mi_allgather_fn xport_allgather = nullptr;
  void* xport_user = nullptr;
  // tensor parallelism inside this process (tp_group.h): a rank shard points at its group; the
  // context the caller holds (tp_rank = MI_TP_ALL_RANKS) owns the group and nothing else
  mi_group* grp = nullptr;
  mi_group* owned_group = nullptr;
  bool stream_owned = true;
  bool collective() const { return comm != nullptr || xport_allreduce != nullptr || grp != nullptr; }
};

