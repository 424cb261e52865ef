/* mi355x_vllm.h — C ABI of libmi355x_vllm.so
 *
 * The inner drop-in boundary of the MI355X vLLM plugin: everything the reference
 * (vllm-project/vllm-neuron @ 2025-11-21) obtains from the third-party NxDI model
 * object is obtained from this library instead.  Plain pointers and sizes only, no
 * torch / C++ types.  All calls arrive on ONE host thread (vLLM's "uni" executor,
 * reference platform.py:120-121,166-167); a context is not thread-safe.
 *
 * Every function returns 0 on success and a negative MI_E* code on failure;
 * mi_last_error() then describes the failure (the Python mirror raises it as
 * RuntimeError / ValueError exactly where the reference raises).
 *
 * Reference interface each entry point replaces (paths relative to /root/reference):
 *   mi_ctx_create      NxDI model-class ctor + neuron_config
 *                      vllm_neuron/worker/neuronx_distributed_model_loader.py:216-218,236
 *                      (fields: _get_default_neuron_config :725-793, overrides :870-900)
 *   mi_load_weight     checkpoint load + save_quantized_state_dict      loader.py:234-241
 *   mi_init_synthetic_weights   (bench only: no checkpoints exist offline)
 *   mi_finalize        model.compile() + model.load()                    loader.py:240-241
 *   mi_forward         NxDI model __call__ + logits[:, -1, :]            loader.py:339-363
 *   mi_kv_stats        torch.classes.neuron.Runtime().get_vnc_memory_stats()
 *                      vllm_neuron/worker/neuron_worker.py:51-63
 *   mi_tp_*            NxDI tp_degree collectives                        loader.py:752-753
 *   mi_op_*            (no reference counterpart: per-kernel entry points for the
 *                      parity tests and micro-benchmarks; device pointers)
 */
#ifndef MI355X_VLLM_H
#define MI355X_VLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument / unsupported shape            */
#define MI_EHIP (-2)     /* HIP runtime error (message has the string) */
#define MI_ESTATE (-3)   /* call order violated (e.g. forward before finalize) */
#define MI_ENOMEM (-4)
#define MI_ECOMM (-5)    /* RCCL error */

/* element types for host/device tensors crossing the ABI */
enum { MI_F32 = 0, MI_BF16 = 1, MI_F8E4M3 = 2, MI_I8 = 3, MI_I64 = 4, MI_I32 = 5 };
/* weight storage dtypes ("quantization_dtype", loader.py:895-896) */
enum { MI_W_BF16 = 0, MI_W_F8E4M3 = 1, MI_W_INT8 = 2 };
/* "quantization_type" (loader.py:892-894) */
enum { MI_Q_PER_TENSOR_SYMMETRIC = 0, MI_Q_PER_CHANNEL_SYMMETRIC = 1 };
enum { MI_ROPE_DEFAULT = 0, MI_ROPE_LLAMA3 = 1 };

typedef struct mi_ctx mi_ctx;

typedef struct mi_model_config {
  /* decoder geometry (HF config; loader.py:612-631 derives nkv / head_dim the same way) */
  int32_t num_layers, hidden_size, num_heads, num_kv_heads, head_dim;
  int32_t intermediate_size, vocab_size;
  float rms_norm_eps;
  float rope_theta;
  int32_t rope_type; /* MI_ROPE_* */
  float rope_factor, rope_low_freq_factor, rope_high_freq_factor;
  int32_t rope_original_max_position;
  int32_t qkv_bias;            /* Qwen2 */
  int32_t tie_word_embeddings; /* lm_head shares embed_tokens */
  /* KV pool ("pa_num_blocks" incl. the null block 0, "pa_block_size": loader.py:741-745,775-778) */
  int32_t num_blocks, block_size;
  int32_t max_num_seqs;  /* "batch_size"  loader.py:738,756 */
  int32_t max_model_len; /* "seq_len"     loader.py:760-761 */
  /* "context_encoding_buckets" (README.md:80); 0 entries = one bucket of max_model_len */
  int32_t num_ctx_buckets;
  int32_t ctx_buckets[8];
  /* quantization keys (loader.py:886-898) */
  int32_t weight_dtype; /* MI_W_* ; MI_W_BF16 = not quantized */
  int32_t quant_type;   /* MI_Q_* */
  int32_t quantize_lm_head; /* 0 = "lm_head" listed in modules_to_not_convert */
  /* tensor parallelism ("tp_degree" loader.py:752-753): one process per GPU */
  int32_t tp_degree, tp_rank;
  int32_t device_id;
  int32_t use_graphs; /* capture token-generation steps into hipGraphs */
  /* Context-encoding GEMMs (rows > 16) on FP8 weights take per-token dynamically quantized FP8
   * activations and run on the MX-scaled MFMA at twice the bf16 rate; token generation (HBM-bound)
   * keeps bf16 activations.  0 = weight-only everywhere. */
  int32_t prefill_fp8_activations;
  /* Tensor parallelism inside ONE process, the reference's process model (a single worker drives
   * every core: platform.py:166-167, neuron_worker.py:106-121): tp_rank = MI_TP_ALL_RANKS makes the
   * context a group of tp_degree rank shards, rank r on GPU tp_device_ids[r] with a host thread of
   * its own; every entry point below fans out and rank 0's result is returned.  All ids equal =
   * every shard on one GPU (the single-GPU tests).  tp_transport picks the exchange between the
   * shards: the library's own all-reduce over peer-mapped memory (xGMI), or RCCL. */
  int32_t tp_device_ids[16];
  int32_t tp_transport; /* MI_TP_TRANSPORT_* */
} mi_model_config;
#define MI_TP_ALL_RANKS (-1)
enum { MI_TP_TRANSPORT_P2P = 0, MI_TP_TRANSPORT_RCCL = 1 };

const char* mi_last_error(void);
int mi_version(void);   /* 4 = this header (+ mi_tp_plan, mi_tp_info); 3 lacked them; 2 lacked tp_device_ids / tp_transport / MI_TP_ALL_RANKS; 1 lacked mi_forward_tokens, mi_op_sample, mi_tp_init_transport */

int mi_ctx_create(const mi_model_config* cfg, mi_ctx** out);
int mi_ctx_destroy(mi_ctx* ctx);

/* One HF-named tensor ("model.layers.3.self_attn.q_proj.weight", "lm_head.weight", ...),
 * row-major on the host, dtype MI_F32 or MI_BF16, FULL (unsharded) shape.  The library
 * slices its TP shard, quantizes (weight_dtype / quant_type) and re-tiles for the kernels. */
int mi_load_weight(mi_ctx* ctx, const char* name, const void* host, int32_t dtype,
                   const int64_t* shape, int32_t ndim);
/* Bench/smoke only: N(0, std) matrices from a counter-based RNG in LOGICAL coordinates
 * (identical values whatever the sharding); norm gains 1. */
int mi_init_synthetic_weights(mi_ctx* ctx, uint64_t seed, float std);
/* Weight artifacts -- the counterpart of the reference's compiled-artifact directory
 * (loader.py:160-226, keyed by a config hash there; the Python mirror keys the directory the same
 * way).  mi_save_weights writes this context's device images (quantized, tiled, sharded: what
 * mi_load_weight made) to <dir>/rank<r>_of<T>.miw; mi_load_weights_file streams such a file back
 * into HBM in place of every mi_load_weight call, and returns MI_EINVAL when the file is missing
 * or was made for another model, quantization, sharding or tile format (the caller then loads the
 * checkpoint the long way and saves again).  Also the home of quantized_checkpoints_path
 * (loader.py:888-891). */
int mi_save_weights(mi_ctx* ctx, const char* dir);
int mi_load_weights_file(mi_ctx* ctx, const char* dir);

/* Re-size the KV pool before mi_finalize: vLLM decides the block count after the weights
 * are resident (worker.determine_available_memory -> KVCacheConfig.num_blocks; the reference's
 * NxDI takes it at compile time as pa_num_blocks, loader.py:775-776). */
int mi_set_num_blocks(mi_ctx* ctx, int32_t num_blocks);
int mi_finalize(mi_ctx* ctx);

/* One model call.  Host arrays, caller-owned, int64 like the reference's CPU tensors:
 *   input_ids, position_ids [B, S]; seq_ids [B]; block_table [B, MB];
 *   slot_mapping [B, SM]; full_context_lens, computed_context_lens [B];
 *   logits_out [B, vocab_size] fp32 (last-token logits, loader.py:363).
 * S == 1 -> token generation;  S > 1 -> context encoding of tokens
 * computed..full-1 of each row (the FULL prompt is passed, runner.py:721-726,754).
 * Slot -1 = no write; block-table entries past full_context_lens are never read. */
int mi_forward(mi_ctx* ctx, int32_t B, int32_t S, const int64_t* input_ids,
               const int64_t* position_ids, const int64_t* seq_ids, const int64_t* block_table,
               int32_t MB, const int64_t* slot_mapping, int32_t SM,
               const int64_t* full_context_lens, const int64_t* computed_context_lens,
               float* logits_out);

/* The same call with on-device sampling (the reference's on_device_sampling_config path: the model
 * returns sampled ids instead of logits, loader.py:352-356, 367-375).  sampling_params: [B, 3] fp32
 * rows (top_k, top_p, temperature) as the reference packs them (runner.py:1106-1140; greedy
 * requests arrive as top_k = 1), or NULL for all-greedy.  top_k is capped at 256.  top_k == 1 is
 * argmax with the lowest index on ties (== torch.argmax on the logits mi_forward would return);
 * otherwise the id is drawn from the temperature-scaled softmax over the top_k logits cut to the
 * top_p nucleus, with u = splitmix64(seed, row) -- a pure function of (logits, params, seed, row),
 * restated in oracle/sampling.py.  tokens_out [B] int64. */
int mi_forward_tokens(mi_ctx* ctx, int32_t B, int32_t S, const int64_t* input_ids,
                      const int64_t* position_ids, const int64_t* seq_ids, const int64_t* block_table,
                      int32_t MB, const int64_t* slot_mapping, int32_t SM,
                      const int64_t* full_context_lens, const int64_t* computed_context_lens,
                      const float* sampling_params, uint64_t seed, int64_t* tokens_out);

/* Chunked prefill (the reference's path under vLLM's native scheduler: runner.py:938-1051,
 * loader.py:357-361): ONE ragged batch of `total` tokens = the concatenated chunks of n_req
 * requests -- request i contributes tokens computed_context_lens[i] .. full_context_lens[i] - 1
 * (a prompt chunk, or the single next token of a request that is generating).  input_ids,
 * position_ids, slot_mapping: [total] in request order; block_table [n_req, MB].  Every token's
 * K/V is written through slot_mapping, each request attends to its own blocks, the projections run
 * once over all rows.  logits_out [n_req, vocab_size] = logits of each request's LAST scheduled
 * token (the caller ignores rows whose prompt is not complete: prefill_completion_state);
 * or tokens_out [n_req] with sampling_params as in mi_forward_tokens.  total may not exceed the
 * largest context-encoding bucket (pass max_num_batched_tokens as a bucket). */
int mi_forward_chunked(mi_ctx* ctx, int32_t n_req, int32_t total, const int64_t* input_ids,
                       const int64_t* position_ids, const int64_t* slot_mapping, const int64_t* block_table,
                       int32_t MB, const int64_t* full_context_lens, const int64_t* computed_context_lens,
                       float* logits_out, const float* sampling_params, uint64_t seed, int64_t* tokens_out);

/* Fused speculation (the reference reaches it through NxDI's fused draft + target graph:
 * loader.py:349-355; output contract re-masked by _remask_fused_spec_output, loader.py:308-333; slots
 * of the speculated positions, runner.py:825-830).  One call = `k - 1` greedy token-generation steps of
 * the DRAFT context chained on the device, ONE token-generation pass of the TARGET context over the
 * B * k candidate rows, greedy acceptance.  input_ids / position_ids [B]: the last accepted token
 * of every sequence and its position; block_table [B, MB] must back positions up to
 * position + k - 1 (clipped at max_model_len) in BOTH contexts (same block ids, separate pools).
 * accepted_out [B, k]: the 1..k tokens generated this step, 0-padded (NxDI's
 * accepted_tokens_with_padding); next_pos_out [B] = position + number of tokens generated.  The
 * tokens are exactly those the target alone would generate greedily, one per step.
 * draft_catchup_ids [B] or NULL: the draft runs k - 1 steps, so after a step that generated k tokens
 * the token at position - 1 has not been through the draft; pass it here (-1 = nothing to catch up)
 * and the draft's first step takes it as an extra row.  Leaving it out never changes the output,
 * only the quality of the draft's proposals.  Needs
 * target max_num_seqs >= B * k, draft max_num_seqs >= B + catch-up rows (2 B), and both contexts created with the same
 * device (the target's rank 0 GPU when the target is an in-process tensor-parallel group; the draft is never
 * sharded), block_size, num_blocks, vocab_size and max_model_len.  From the first call on the
 * draft context runs on the target's stream: destroy the draft before the target. */
int mi_forward_spec(mi_ctx* target, mi_ctx* draft, int32_t B, int32_t k, const int64_t* input_ids,
                    const int64_t* position_ids, const int64_t* block_table, int32_t MB,
                    const int64_t* draft_catchup_ids, int64_t* accepted_out, int64_t* next_pos_out);

/* Replay the LAST token-generation call `steps` times with its inputs left resident in HBM
 * (no host round trip in between) and return the elapsed time measured with HIP events on the
 * context's stream.  For benchmarking the hot path itself: mi_forward adds one small H2D copy
 * and a [B, V] fp32 D2H copy per step. */
int mi_replay_decode(mi_ctx* ctx, int32_t steps, float* elapsed_ms);

/* The same replay with only the kernel classes of `class_mask` (bit MI_K_*) launched: e.g.
 * 1 << MI_K_GEMV replays the weight-streaming launches of a step back to back, in a graph of their
 * own -- elapsed / (steps x launches) is the average launch duration of that kernel, boundary
 * included, which is how rocprofv3's kernel trace counts it.  Timing only: the step's results are
 * not meaningful.  Single-GPU contexts. */
int mi_replay_decode_classes(mi_ctx* ctx, int32_t steps, uint32_t class_mask, float* elapsed_ms);

typedef struct mi_kv_stats_t {
  int64_t kv_bytes, weight_bytes;
  int64_t workspace_bytes;   /* before mi_finalize: what it will allocate besides the KV pool; after: what it did */
  int64_t device_free_bytes, device_total_bytes;
  int32_t num_blocks, block_size, num_kv_heads_local, head_dim, num_layers;
  /* token-generation block tables are device-resident: rows re-sent because the caller's row
   * changed (new request, block appended) vs rows found unchanged since the previous step */
  int64_t block_table_rows_sent, block_table_rows_kept;
} mi_kv_stats_t;
int mi_kv_stats(mi_ctx* ctx, mi_kv_stats_t* out);
/* Bytes ONE block costs on a GPU across all layers (K and V, this rank's kv heads): vLLM sizes the
 * cache from a single-layer spec (runner.get_kv_cache_spec), the pool holds num_layers of them. */
int64_t mi_kv_bytes_per_block(mi_ctx* ctx);

/* The pinned host buffer [max_num_seqs][vocab_size] fp32 the logits land in.  Passing it as
 * mi_forward's logits_out skips the final host copy: the caller reads the logits in place, valid
 * until the next call on this context.  NULL for tensor-parallel contexts (their shards fill the
 * caller's rows directly) and before mi_finalize. */
float* mi_logits_buffer(mi_ctx* ctx);

/* hipStream_t the context launches on (for event timing by the caller). */
void* mi_stream(mi_ctx* ctx);

/* Per-kernel-class HIP-event timing of the following mi_forward calls (eager launches,
 * no graph).  mi_profile_read returns, per class, launches and summed milliseconds. */
enum { MI_K_GEMV = 0, MI_K_GEMM = 1, MI_K_ATTN_DECODE = 2, MI_K_ATTN_PREFILL = 3, MI_K_OTHER = 4,
       MI_K_COMM = 5, MI_K_NUM = 6 };
int mi_profile_enable(mi_ctx* ctx, int32_t on);
int mi_profile_read(mi_ctx* ctx, int32_t* launches /*[MI_K_NUM]*/, float* ms /*[MI_K_NUM]*/,
                    double* gemv_weight_bytes);

/* Tensor parallel bring-up: rank 0 calls mi_tp_unique_id, the host layer broadcasts the
 * 128 bytes, every rank calls mi_tp_init (ncclCommInitRank on the context's device). */
int mi_tp_unique_id(void* out128);
int mi_tp_init(mi_ctx* ctx, const void* id128);

/* The same, with the two collectives supplied by the caller instead of RCCL (a custom xGMI
 * transport; the single-GPU loopback that tests/test_tp_loopback_gpu.py uses to run every
 * rank's shard on one device).  all_reduce: in-place fp32 sum of buf[count] over the ranks;
 * all_gather: recv[rank * count .. ] = rank's send[count].  Both are called from mi_forward on
 * the calling thread, must be ordered after the work already queued on `stream` and must leave
 * their result visible to work queued on it afterwards; return 0 on success.  Decode steps are
 * launched eagerly (no graph capture) on a context with collectives. */
typedef int (*mi_allreduce_fn)(void* user, void* buf, size_t count, void* stream);
typedef int (*mi_allgather_fn)(void* user, const void* send, void* recv, size_t count, void* stream);
int mi_tp_init_transport(mi_ctx* ctx, mi_allreduce_fn all_reduce, mi_allgather_fn all_gather, void* user);

/* The sharding plan of rank `rank` of cfg->tp_degree: which slices of the UNSHARDED checkpoint tensors the rank holds
 * (tp_degree = tensor_parallel_size, loader.py:752-753; head counts that do not divide are allowed, the reference skips
 * vLLM's divisibility check at platform.py:58-64).  Host only: no device, no context -- the same function mi_ctx_create
 * runs.  q heads [q_head0, q_head0 + q_heads_real) are real; the rank computes q_heads_local >= q_heads_real heads, the
 * surplus being zero-weight padding heads (q rows, q bias and o_proj columns zero).  kv heads [kv_head0, + kv_heads_local);
 * gate / up rows and down_proj columns [inter0, + inter_local); lm_head rows (logit columns) [vocab0, + vocab_local). */
typedef struct mi_tp_plan {
  int32_t q_head0, q_heads_real, q_heads_local;
  int32_t kv_head0, kv_heads_local;
  int32_t inter0, inter_local;
  int32_t vocab0, vocab_local;
} mi_tp_plan_t;
int mi_tp_plan(const mi_model_config* cfg, int32_t rank, mi_tp_plan_t* out);

/* What an in-process tensor-parallel context (tp_rank = MI_TP_ALL_RANKS, finalized) actually runs on -- so that a benchmark
 * record can tell "N ranks over peer memory, in hipGraphs" from "fell back to RCCL, eager". */
typedef struct mi_tp_info {
  int32_t tp_degree;
  int32_t transport;       /* MI_TP_TRANSPORT_* in use after the self-test */
  int32_t selftest;        /* 1: passed on the transport in use; -1: the peer-memory exchange failed it (transport is RCCL now); 0: not run */
  int32_t graphs;          /* 1: token-generation steps, exchange kernels included, replay from a hipGraph per shard */
  int32_t mode;            /* 0: one GPU per rank; 1: every shard on one GPU in lockstep (one stream, host barriers between the
                              exchange kernels); 2: every shard on one GPU, one stream per shard, the kernels wait on device flags */
  int32_t timeout_ms;      /* bound of a flag wait; on expiry the output is poisoned with NaN and MI_ECOMM is reported */
  int32_t device_ids[16];
  int32_t peer_access[16]; /* bit p of word r: rank r's device maps rank p's memory */
} mi_tp_info_t;
int mi_tp_info(mi_ctx* ctx, mi_tp_info_t* out);

/* ---- per-kernel entry points (device pointers; stream may be NULL) -------------------- */

/* The exchange step of an in-process tensor-parallel context (tp_rank = MI_TP_ALL_RANKS, finalized) on
 * caller-supplied data: bufs[r] = rank r's fp32 [count] on rank r's GPU, count % 8 == 0 and at most
 * max rows x hidden_size.  In place: every rank ends with the fp32 sum over ranks, taken in rank
 * order (bit-identical on every rank); messages of 512 KiB (as bf16) and more -- context encoding --
 * travel as bf16, go reduce-scatter + all-gather and round the sum to bf16 once more.  Returns
 * after the exchange has completed on every rank. */
int mi_op_tp_all_reduce(mi_ctx* ctx, float* const* bufs, size_t count);

/* The sampler of mi_forward_tokens on caller-supplied logits: logits [B, V] fp32, sampling_params
 * [B, 3] fp32 (top_k, top_p, temperature) or NULL = greedy, tokens_out [B] int32 -- all device. */
int mi_op_sample(const float* logits, int32_t B, int32_t V, const float* sampling_params, uint64_t seed,
                 int32_t* tokens_out, void* stream);
/* The same with the engine's scratch buffer (mi_op_sample_scratch_bytes(B) bytes of device memory): the form
 * mi_forward_tokens runs -- all-greedy batches split the argmax over the chip, top-k / top-p rows have the
 * vocabulary pre-selected by 32 work-groups per row.  Same ids as mi_op_sample. */
size_t mi_op_sample_scratch_bytes(int32_t B);
int mi_op_sample_ws(const float* logits, int32_t B, int32_t V, const float* sampling_params, uint64_t seed,
                    int32_t* tokens_out, void* scratch, size_t scratch_bytes, void* stream);


/* Quantize + re-tile a row-major fp32 [N, K] device matrix.  scale_out [N] fp32.
 * tiled_out: N*K bytes (fp8/int8) or 2*N*K (bf16), kernel-native 16-row tiles. */
int mi_op_quantize_weight(const float* w, int32_t N, int32_t K, int32_t weight_dtype,
                          int32_t quant_type, void* tiled_out, float* scale_out, void* stream);
/* Inverse of the tiling for inspection: q_out row-major [N, K] (1 or 2 bytes/elem). */
int mi_op_untile_weight(const void* tiled, int32_t N, int32_t K, int32_t weight_dtype,
                        void* q_out, void* stream);
/* y[M, N] fp32 = (x[M, K] bf16 . Wq^T) * scale (+ bias).  M <= 16: weight-streaming GEMV;
 * else MFMA GEMM.  force_path: 0 auto, 1 GEMV, 2 GEMM (picked by size), 3 the wide-N LDS-DMA GEMM (1-byte weights),
 * 4 / 5 that GEMM with K split over 2 / 4 slices at 128-token blocks, 6 / 7 at 256-token blocks (K / 64 a multiple of the slices). */
int mi_op_qlinear(const void* x_bf16, int32_t M, const void* w_tiled, const float* scale,
                  const float* bias, int32_t N, int32_t K, int32_t weight_dtype, float* y,
                  int32_t force_path, void* stream);
/* FP8-activation form (M > 16, fp8 weights, K % 128 == 0): x is quantized per token to e4m3
 * (scale amax/448) and multiplied on v_mfma_scale_f32_16x16x128_f8f6f4. */
int mi_op_qlinear_a8(const void* x_bf16, int32_t M, const void* w_tiled, const float* scale,
                     const float* bias, int32_t N, int32_t K, float* y, void* stream);
/* y[T, H] bf16 = rmsnorm(x[T, H] fp32) * g */
int mi_op_rmsnorm(const float* x, const float* g, int32_t T, int32_t H, float eps, void* y_bf16,
                  void* stream);
/* Paged KV write.  k, v [T, nkv, hd] bf16; slots [T] (int64; -1 skips).
 * pool layout (library-native): [2][num_blocks][nkv][block_size][hd] bf16. */
int mi_op_kv_write(const void* k, const void* v, const int64_t* slots, int32_t T, int32_t nkv,
                   int32_t hd, void* pool, int32_t num_blocks, int32_t block_size, void* stream);
/* Token-generation attention.  q [B, nh, hd] bf16; block_table [B, MB] int32;
 * ctx_lens [B] int32; out [B, nh*hd] bf16.  scratch: >= mi_op_attn_scratch_bytes. */
int64_t mi_op_attn_scratch_bytes(int32_t B, int32_t nh, int32_t hd);
int mi_op_paged_attn_decode(const void* q, const void* pool, int32_t num_blocks,
                            int32_t block_size, const int32_t* block_table, int32_t MB,
                            const int32_t* ctx_lens, int32_t B, int32_t nh, int32_t nkv,
                            int32_t hd, void* out, void* scratch, void* stream);
/* Context-encoding attention for ONE sequence.  q [T, nh, hd] bf16 are the new tokens at
 * absolute positions q_pos0 .. q_pos0+T-1; keys 0 .. q_pos0+T-1 are read from the pool
 * through block_table [MB] int32 (the new tokens' K/V must already be written). */
int mi_op_paged_attn_prefill(const void* q, int32_t T, int32_t q_pos0, const void* pool,
                             int32_t num_blocks, int32_t block_size, const int32_t* block_table,
                             int32_t MB, int32_t nh, int32_t nkv, int32_t hd, void* out,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_VLLM_H */
