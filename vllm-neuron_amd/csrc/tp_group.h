// Tensor parallelism inside ONE process: a context created with tp_rank = MI_TP_ALL_RANKS owns
// tp_degree rank shards (ordinary contexts, rank r on device tp_device_ids[r]), one host thread
// each, and fans every entry point of include/mi355x_vllm.h out to them.  This is the process
// model of the reference: vLLM's "uni" executor starts ONE worker and the model object drives
// every core (/root/reference/vllm_neuron/platform.py:166-167, worker/neuron_worker.py:106-121,
// tp_degree = tensor_parallel_size at worker/neuronx_distributed_model_loader.py:752-753).
//
// The exchange step of a row-parallel projection is a hand-written all-reduce over
// peer-mapped device memory (xGMI between GPUs; plain device memory when every shard sits on one
// GPU, which is how the single-GPU tests run the same kernels):
//   publish   fp32 partial -> exchange buffer of this rank; system-scope release; one flag
//             per (peer, work-group) written into the PEER's flag block
//   reduce    wait for the peers' flags of this work-group's range (bounded spin), read the
//             range from every rank's exchange buffer in rank order, sum in fp32 -> partial
// Token-generation messages (M x H, tens of KiB) travel as fp32: the exchange is latency-bound and
// the result is the one-GPU sum up to its order.  Messages of kArTwoShotBytes and more (context
// encoding) travel as bf16 and go reduce-scatter + all-gather (each rank reduces 1/T of the message
// and publishes the reduced slice, again bf16), which moves 2 (T-1)/T of the message per rank
// instead of (T-1) times it, at half the bytes.  Epochs are device-resident counters, so a captured hipGraph replays
// the exchange unchanged.  The sum runs in rank order on every rank: ranks agree bit for bit.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mi_common.h"

struct mi_ctx;

namespace mi {

constexpr int kArMaxRanks = 16;
constexpr int kArMaxBlocks = 64;                    // work-groups of an exchange kernel
constexpr size_t kArTwoShotBytes = 512 * 1024;      // bf16 message size from which reduce-scatter + all-gather pays

// what every rank's kernels need to know about every rank
struct ArPeers {
  uint16_t* xbuf[kArMaxRanks];   // [2 slots][cap] fp32-sized: the rank's published partial (fp32, or bf16 in the front half of the slot)
  uint16_t* ybuf[kArMaxRanks];   // [2 slots][cap / T + pad] bf16: the slice the rank reduced (two-shot)
  uint32_t* flag1[kArMaxRanks];  // [T peers][kArMaxBlocks]: "peer p, work-group w: partial of epoch e published"
  uint32_t* flag2[kArMaxRanks];  // the same for the reduced slices
};

// abortable host barrier for the rank threads
class HostBarrier {
 public:
  explicit HostBarrier(int n) : n_(n) {}
  bool wait();   // false when the barrier was aborted
  void abort();
  void reset();

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  unsigned gen_ = 0;
  bool aborted_ = false;
};

}  // namespace mi

struct mi_group {
  int T = 0;
  bool same_device = false;              // every shard on ONE device (single-GPU loopback)
  bool lockstep = false;                 // ... with a shared stream and host barriers between the exchange kernels (the default loopback form)
  bool host_barrier = false;             // concurrent loopback with a host barrier between publish and reduce (eager launches)
  int selftest = 0;                      // 1 passed, -1 peer-memory exchange failed (RCCL took over), -2 both failed, 0 not run
  int device_ids[16] = {0};
  int peer_access[16] = {0};             // bit p of word r: rank r's device maps rank p's memory
  std::vector<mi_ctx*> ranks;
  hipStream_t shared_stream = nullptr;   // lockstep only
  mi::HostBarrier* bar = nullptr;
  // exchange state
  mi::ArPeers peers{};
  size_t cap = 0;                        // elements per exchange slot
  std::vector<uint32_t*> counters;       // per rank, device: [0..3] epochs of the four kernels, [4..7] their tickets, [8] error word
  int use_rccl = 0;                      // in-process RCCL (ncclCommInitAll) instead of the hand-written exchange
  // worker threads
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<int(mi_ctx*, int)> job;
  unsigned job_gen = 0;
  int pending = 0;
  bool stop = false;
  std::vector<int> rc;
  std::vector<std::string> err;
};

namespace mi {

// run fn(rank context, rank) on every rank thread; returns the first failing status (message in mi_last_error)
int group_run(mi_group* g, const std::function<int(mi_ctx*, int)>& fn);
int group_create(const mi_model_config& cfg, mi_ctx* facade);
void group_destroy(mi_group* g);
// allocate the exchange buffers (after the shards know their activation sizes); called on every rank thread
int group_alloc_exchange(mi_ctx* c);
// sum `count` fp32 values of `buf` over the ranks, in place, on c->stream (called from rank r's thread)
int group_all_reduce(mi_ctx* c, float* buf, size_t count);
// one exchange on known data after the buffers exist; falls back to RCCL when the peer-memory path fails it
int group_selftest(mi_group* g);
// device-side error word of rank c (nonzero: an exchange kernel gave up waiting)
int group_check_errors(mi_ctx* c);
int group_info(mi_group* g, mi_tp_info_t* out);

}  // namespace mi
