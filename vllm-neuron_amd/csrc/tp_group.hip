// Tensor parallelism inside one process: rank threads, the hand-written all-reduce over
// peer-mapped memory, and the fan-out of the C ABI.  See tp_group.h.

#include "tp_group.h"

#include <cstdio>
#include <cstring>

#include "model_internal.h"

extern "C" const char* mi_last_error(void);

#define MI_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != MI_OK) return _rc; \
  } while (0)

namespace mi {

// =====================================================================================
// host barrier
// =====================================================================================
bool HostBarrier::wait() {
  std::unique_lock<std::mutex> lk(mu_);
  if (aborted_) return false;
  const unsigned gen = gen_;
  if (++count_ == n_) {
    count_ = 0;
    ++gen_;
    cv_.notify_all();
    return true;
  }
  cv_.wait(lk, [&] { return gen_ != gen || aborted_; });
  return !aborted_;
}
void HostBarrier::abort() {
  std::lock_guard<std::mutex> lk(mu_);
  aborted_ = true;
  cv_.notify_all();
}
void HostBarrier::reset() {
  std::lock_guard<std::mutex> lk(mu_);
  aborted_ = false;
  count_ = 0;
}

// =====================================================================================
// exchange kernels
// =====================================================================================
constexpr int kArThreads = 256;
// A flag wait is bounded in TIME (s_memrealtime: the 100 MHz constant clock), not in iterations: a peer's rank thread
// parked by the host scheduler for tens of milliseconds, or its first graph instantiation, must not trip it, a dead peer must.
// On expiry the kernel sets the error word AND poisons what it would have written (NaN): a partial sum never passes as data;
// the host reports MI_ECOMM after the step (group_check_errors).  MI355X_TP_TIMEOUT_MS, default 5000.
constexpr unsigned long long kArTicksPerMs = 100000ull;

enum { AR_EPOCH_PUB = 0, AR_EPOCH_RS = 1, AR_TICKET_PUB = 4, AR_TICKET_RS = 5, AR_ERROR = 8, AR_TOUCH = 12, AR_WORDS = 16 };

struct ArGeom {
  int T, r, G;          // ranks, this rank, work-groups
  int pubG;             // work-groups of the publish kernel of this exchange
  unsigned chunks;      // 16-byte (8 x bf16) chunks of the message
  unsigned cpw;         // chunks per work-group (one-shot) / per work-group inside a slice (two-shot)
  unsigned cps;         // chunks per rank slice (two-shot)
  size_t cap;           // elements per exchange slot
  size_t ycap;          // elements per reduced-slice slot
  unsigned long long timeout;   // flag-wait bound in 100 MHz ticks
};

__device__ __forceinline__ bool flag_reached(const uint32_t* p, uint32_t e) {
  const uint32_t v = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
  return (int32_t)(v - e) >= 0;
}
// one flag, bounded in time; false = gave up (error word set)
__device__ __forceinline__ bool wait_flag(const uint32_t* p, uint32_t e, unsigned long long timeout, uint32_t* err) {
  if (flag_reached(p, e)) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(8);
    if (flag_reached(p, e)) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}
// wait until work-groups 0 .. nG-1 of every peer have signalled epoch e in this rank's flag block; false (for the whole
// work-group) when any wait gave up
__device__ __forceinline__ bool wait_flags(const uint32_t* flags, int T, int self, int nG, uint32_t e, unsigned long long timeout,
                                           uint32_t* err) {
  __shared__ int s_gave_up;
  if (threadIdx.x == 0) s_gave_up = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < T * nG; i += blockDim.x) {
    const int p = i / nG, w = i - p * nG;
    if (p == self) continue;   // a rank does not signal itself
    if (!wait_flag(flags + p * kArMaxBlocks + w, e, timeout, err)) s_gave_up = 1;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope, every wave: drop stale lines of the peers' buffers
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return s_gave_up == 0;
}
__device__ __forceinline__ void store8_nan(float* o) {
  const float n = __builtin_nanf("");
  *reinterpret_cast<float4*>(o) = make_float4(n, n, n, n);
  *reinterpret_cast<float4*>(o + 4) = make_float4(n, n, n, n);
}
// the last work-group of a kernel advances its epoch counter (every work-group read it at entry)
__device__ __forceinline__ void advance_epoch(uint32_t* ctr, int epoch_idx, int ticket_idx, uint32_t e, int G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ctr + ticket_idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (uint32_t)(G - 1)) {
      __hip_atomic_store(ctr + ticket_idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr + epoch_idx, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__device__ __forceinline__ uint32_t read_epoch(const uint32_t* ctr, int idx) {
  return __hip_atomic_load(ctr + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u32x4_t pack8(const float4 a, const float4 b) {
  return u32x4_t{pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w)};
}
__device__ __forceinline__ void add8(float (&acc)[8], const u32x4_t v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[2 * k] += bf16lo_to_f32(v[k]);
    acc[2 * k + 1] += bf16hi_to_f32(v[k]);
  }
}
__device__ __forceinline__ void store8(float* o, const float (&a)[8]) {
  *reinterpret_cast<float4*>(o) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(a[4], a[5], a[6], a[7]);
}

// Wire format of an exchange slot: F32 = the fp32 partial as it is (token generation: the message is
// latency-bound, the sum equals the one-GPU sum up to its order); otherwise bf16 (context encoding:
// 2 (T-1)/T x 32 MiB per rank and exchange at N = 2048, H = 8192 -- half the bytes on the links).
// `base` is the slot in units of the wire element.
template <bool F32>
__device__ __forceinline__ void wire_store8(void* base, size_t ch, const float4 a, const float4 b) {
  if constexpr (F32) {
    float* p = reinterpret_cast<float*>(base) + ch * 8;
    *reinterpret_cast<float4*>(p) = a;
    *reinterpret_cast<float4*>(p + 4) = b;
  } else {
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(base) + ch * 8) = pack8(a, b);
  }
}
template <bool F32>
__device__ __forceinline__ void wire_add8(float (&acc)[8], const void* base, size_t ch) {
  if constexpr (F32) {
    const float* p = reinterpret_cast<const float*>(base) + ch * 8;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
  } else {
    add8(acc, *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(base) + ch * 8));
  }
}
template <bool F32>
__device__ __forceinline__ const void* wire_slot(const uint16_t* buf, uint32_t e, size_t cap) {
  return reinterpret_cast<const unsigned char*>(buf) + (size_t)(e & 1) * cap * 4;   // slots are fp32-sized whatever travels
}

// fp32 partial -> this rank's exchange slot; per work-group: release, then one flag per peer
template <bool F32>
__global__ __launch_bounds__(kArThreads) void ar_publish_kernel(const float* __restrict__ partial, ArPeers P, ArGeom ge,
                                                                uint32_t* __restrict__ ctr) {
  const int w = blockIdx.x;
  const uint32_t e = read_epoch(ctr, AR_EPOCH_PUB) + 1;
  void* slot = const_cast<void*>(wire_slot<F32>(P.xbuf[ge.r], e, ge.cap));
  const unsigned c0 = (unsigned)w * ge.cpw, c1 = min(c0 + ge.cpw, ge.chunks);
  for (unsigned ch = c0 + threadIdx.x; ch < c1; ch += kArThreads) {
    const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)ch * 8);
    const float4 b = *reinterpret_cast<const float4*>(partial + (size_t)ch * 8 + 4);
    wire_store8<F32>(slot, ch, a, b);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope, every storing thread
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < ge.T && (int)threadIdx.x != ge.r)
    __hip_atomic_store(P.flag1[threadIdx.x] + ge.r * kArMaxBlocks + w, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  advance_epoch(ctr, AR_EPOCH_PUB, AR_TICKET_PUB, e, ge.G);
}

// one-shot: sum the range of this work-group over all ranks, in rank order
template <bool F32>
__global__ __launch_bounds__(kArThreads) void ar_reduce_kernel(float* __restrict__ out, ArPeers P, ArGeom ge,
                                                               uint32_t* __restrict__ ctr) {
  const int w = blockIdx.x;
  const uint32_t e = read_epoch(ctr, AR_EPOCH_PUB);   // the publish kernel in front of this one has advanced it
  // the peers' flags for work-group w: word p * kArMaxBlocks + w of this rank's block
  __shared__ int s_gave_up;
  if (threadIdx.x == 0) s_gave_up = 0;
  __syncthreads();
  {
    const int t = threadIdx.x;
    if (t < ge.T && t != ge.r && !wait_flag(P.flag1[ge.r] + t * kArMaxBlocks + w, e, ge.timeout, ctr + AR_ERROR)) s_gave_up = 1;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope, every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const bool ok = s_gave_up == 0;
  const unsigned c0 = (unsigned)w * ge.cpw, c1 = min(c0 + ge.cpw, ge.chunks);
  for (unsigned ch = c0 + threadIdx.x; ch < c1; ch += kArThreads) {
    if (!ok) { store8_nan(out + (size_t)ch * 8); continue; }   // a peer never published: no partial sum leaves this kernel
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < ge.T; ++p) wire_add8<F32>(acc, wire_slot<F32>(P.xbuf[p], e, ge.cap), ch);
    store8(out + (size_t)ch * 8, acc);
  }
}

// two-shot, step 1: this rank reduces ITS slice [r * cps, (r + 1) * cps) and publishes the bf16 result
__global__ __launch_bounds__(kArThreads) void ar_reduce_scatter_kernel(float* __restrict__ out, ArPeers P, ArGeom ge,
                                                                       uint32_t* __restrict__ ctr) {
  const int w = blockIdx.x;
  const uint32_t e = read_epoch(ctr, AR_EPOCH_PUB);
  const uint32_t e2 = read_epoch(ctr, AR_EPOCH_RS) + 1;
  // any publishing work-group of a peer may have written part of this range: wait for all of them
  const bool ok = wait_flags(P.flag1[ge.r], ge.T, ge.r, ge.pubG, e, ge.timeout, ctr + AR_ERROR);
  uint16_t* yslot = P.ybuf[ge.r] + (size_t)(e2 & 1) * ge.ycap;
  const unsigned s0 = min((unsigned)ge.r * ge.cps, ge.chunks), s1 = min(s0 + ge.cps, ge.chunks);
  const unsigned c0 = min(s0 + (unsigned)w * ge.cpw, s1), c1 = min(c0 + ge.cpw, s1);
  const float poison = ok ? 0.f : __builtin_nanf("");   // gave up: the slice this rank publishes (and keeps) is NaN, the protocol goes on
  for (unsigned ch = c0 + threadIdx.x; ch < c1; ch += kArThreads) {
    float acc[8] = {poison, poison, poison, poison, poison, poison, poison, poison};
    for (int p = 0; p < ge.T; ++p) wire_add8<false>(acc, wire_slot<false>(P.xbuf[p], e, ge.cap), ch);
    const u32x4_t pk = pack8(make_float4(acc[0], acc[1], acc[2], acc[3]), make_float4(acc[4], acc[5], acc[6], acc[7]));
    *reinterpret_cast<u32x4_t*>(yslot + (size_t)(ch - s0) * 8) = pk;
    float rounded[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // the owner keeps what the others will read
    add8(rounded, pk);
    store8(out + (size_t)ch * 8, rounded);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < ge.T && (int)threadIdx.x != ge.r)
    __hip_atomic_store(P.flag2[threadIdx.x] + ge.r * kArMaxBlocks + w, e2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  advance_epoch(ctr, AR_EPOCH_RS, AR_TICKET_RS, e2, ge.G);
}

// two-shot, step 2: fetch every other rank's reduced slice
__global__ __launch_bounds__(kArThreads) void ar_gather_kernel(float* __restrict__ out, ArPeers P, ArGeom ge,
                                                               uint32_t* __restrict__ ctr) {
  const int w = blockIdx.x;
  const uint32_t e2 = read_epoch(ctr, AR_EPOCH_RS);
  const bool ok = wait_flags(P.flag2[ge.r], ge.T, ge.r, ge.G, e2, ge.timeout, ctr + AR_ERROR);
  for (int p = 0; p < ge.T; ++p) {
    if (p == ge.r) continue;
    const uint16_t* yslot = P.ybuf[p] + (size_t)(e2 & 1) * ge.ycap;
    const unsigned s0 = min((unsigned)p * ge.cps, ge.chunks), s1 = min(s0 + ge.cps, ge.chunks);
    const unsigned c0 = min(s0 + (unsigned)w * ge.cpw, s1), c1 = min(c0 + ge.cpw, s1);
    for (unsigned ch = c0 + threadIdx.x; ch < c1; ch += kArThreads) {
      if (!ok) { store8_nan(out + (size_t)ch * 8); continue; }
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      add8(v, *reinterpret_cast<const u32x4_t*>(yslot + (size_t)(ch - s0) * 8));
      store8(out + (size_t)ch * 8, v);
    }
  }
}

// One word of every rank's slots and flag blocks read from THIS rank (and thereby mapped / first-touched from it) before
// the first exchange: see group_alloc_exchange.
__global__ void ar_touch_kernel(ArPeers P, int T, size_t cap, size_t ycap, uint32_t* __restrict__ ctr) {
  const int p = threadIdx.x;
  if (p >= T) return;
  uint32_t acc = 0;
  acc += __hip_atomic_load(reinterpret_cast<const uint32_t*>(P.xbuf[p]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  acc += __hip_atomic_load(reinterpret_cast<const uint32_t*>(P.xbuf[p]) + cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // second slot (fp32-sized)
  acc += __hip_atomic_load(reinterpret_cast<const uint32_t*>(P.ybuf[p]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  acc += __hip_atomic_load(reinterpret_cast<const uint32_t*>(P.ybuf[p] + ycap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  acc += __hip_atomic_load(P.flag1[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  acc += __hip_atomic_load(P.flag2[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  atomicAdd(ctr + AR_TOUCH, acc);     // everything was zero-filled by its owner: the word stays 0
}

static unsigned long long ar_timeout_ticks(int* ms_out = nullptr) {
  static const int ms = [] { const char* v = getenv("MI355X_TP_TIMEOUT_MS"); const int m = v ? atoi(v) : 5000; return m > 0 ? m : 5000; }();
  if (ms_out) *ms_out = ms;
  return (unsigned long long)ms * kArTicksPerMs;
}

// =====================================================================================
// group plumbing
// =====================================================================================
static void rank_thread(mi_group* g, int r, int device) {
  hipSetDevice(device);
  unsigned seen = 0;
  for (;;) {
    std::function<int(mi_ctx*, int)> fn;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_job.wait(lk, [&] { return g->stop || g->job_gen != seen; });
      if (g->stop) return;
      seen = g->job_gen;
      fn = g->job;
    }
    int rc = MI_EHIP;
    std::string msg;
    try {
      rc = fn(g->ranks[r], r);
      if (rc != MI_OK) msg = mi_last_error();
    } catch (const std::exception& ex) {
      msg = ex.what();
    }
    if (rc != MI_OK && g->bar) g->bar->abort();   // peers waiting at an exchange barrier must not wait for ever
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->rc[r] = rc;
      g->err[r] = msg;
      if (--g->pending == 0) g->cv_done.notify_all();
    }
  }
}

int group_run(mi_group* g, const std::function<int(mi_ctx*, int)>& fn) {
  {
    std::unique_lock<std::mutex> lk(g->mu);
    g->job = fn;
    g->pending = g->T;
    ++g->job_gen;
    g->cv_job.notify_all();
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
  }
  int first = MI_OK;
  std::string msg;
  for (int r = 0; r < g->T; ++r)
    if (g->rc[r] != MI_OK && first == MI_OK) {
      first = g->rc[r];
      msg = "tp rank " + std::to_string(r) + ": " + g->err[r];
    }
  if (first != MI_OK) {
    set_error(msg);
    if (g->bar) g->bar->reset();
  }
  return first;
}

int group_create(const mi_model_config& cfg, mi_ctx* facade) {
  const int T = cfg.tp_degree;
  MI_CHECK(T >= 2 && T <= kArMaxRanks, "in-process tensor parallelism: tp_degree must be 2..16");
  int ndev = 0;
  MI_HIP(hipGetDeviceCount(&ndev));
  bool same = true, distinct = true;
  for (int r = 0; r < T; ++r) {
    MI_CHECK(cfg.tp_device_ids[r] >= 0 && cfg.tp_device_ids[r] < ndev, "tp_device_ids entry out of range");
    if (cfg.tp_device_ids[r] != cfg.tp_device_ids[0]) same = false;
    for (int q = 0; q < r; ++q)
      if (cfg.tp_device_ids[q] == cfg.tp_device_ids[r]) distinct = false;
  }
  MI_CHECK(same || distinct, "tp_device_ids must be all distinct (one GPU per rank) or all equal (single-GPU loopback)");
  mi_group* g = new mi_group();
  g->T = T;
  g->same_device = same;
  // Single-GPU loopback, two forms.  Lockstep (default): one stream for all shards, host barriers between the exchange
  // kernels -- a reducer never waits on the device.  Concurrent (MI355X_TP_LOOPBACK_CONCURRENT=1): one stream per shard, no
  // host barrier, hipGraphs as configured -- publish / reduce kernels of different shards run side by side and wait on the
  // device flags exactly as they will across GPUs (the waits are bounded, so a scheduling problem is a failed call, not a hang).
  const bool concurrent = same && getenv("MI355X_TP_LOOPBACK_CONCURRENT") && getenv("MI355X_TP_LOOPBACK_CONCURRENT")[0] == '1';
  g->lockstep = same && !concurrent;
  // Measured on one MI355X (tools/tp_concurrent_probe.py): two shard streams run side by side, and the exchange passes its
  // self-test and the model tests from hipGraphs; with four or eight, a reduce kernel that waits on a flag ends up IN FRONT of
  // the publish kernel it waits for in one of the device's hardware queues (whatever GPU_MAX_HW_QUEUES says) and the wait runs
  // into its time bound -- a property of several dependent streams on ONE device, not of the protocol: with every publish
  // enqueued before any reduce (MI355X_TP_CONCURRENT_HOSTBAR=1: a host barrier between the two launches, hence eager launches)
  // four and eight concurrent shard streams pass.  On a node every rank has a device, and queues, of its own.
  g->host_barrier = concurrent && getenv("MI355X_TP_CONCURRENT_HOSTBAR") != nullptr;
  g->use_rccl = (!same && cfg.tp_transport == MI_TP_TRANSPORT_RCCL) ? 1 : 0;
  for (int r = 0; r < T; ++r) { g->device_ids[r] = cfg.tp_device_ids[r]; g->peer_access[r] = same ? (1 << T) - 1 : (1 << r); }
  g->bar = new HostBarrier(T);
  g->ranks.assign(T, nullptr);
  g->rc.assign(T, MI_OK);
  g->err.assign(T, "");
  g->counters.assign(T, nullptr);
  facade->owned_group = g;
  if (!same) {   // peer mappings, both directions
    for (int r = 0; r < T; ++r) {
      MI_HIP(hipSetDevice(cfg.tp_device_ids[r]));
      for (int p = 0; p < T; ++p) {
        if (p == r) continue;
        int can = 0;
        MI_HIP(hipDeviceCanAccessPeer(&can, cfg.tp_device_ids[r], cfg.tp_device_ids[p]));
        if (!can && !g->use_rccl) {   // no peer mapping between this pair: the hand-written exchange cannot run
          fprintf(stderr, "[mi355x] GPUs %d and %d have no peer access: tensor-parallel exchange over RCCL instead\n",
                  cfg.tp_device_ids[r], cfg.tp_device_ids[p]);
          g->use_rccl = 1;
        }
        if (can) {
          hipError_t pe = hipDeviceEnablePeerAccess(cfg.tp_device_ids[p], 0);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) MI_HIP(pe);
          (void)hipGetLastError();
          g->peer_access[r] |= 1 << p;
        }
      }
    }
  }
  for (int r = 0; r < T; ++r) g->threads.emplace_back(rank_thread, g, r, cfg.tp_device_ids[r]);
  // every shard is an ordinary context of rank r, created on its own thread / device
  std::vector<mi_ctx*> made(T, nullptr);
  int rc = group_run(g, [&](mi_ctx*, int r) -> int {
    mi_model_config k = cfg;
    k.tp_rank = r;
    k.device_id = cfg.tp_device_ids[r];
    if (g->lockstep || g->host_barrier) k.use_graphs = 0;   // host barriers sit between the exchange kernels
    if (g->use_rccl) k.use_graphs = 0;   // ncclAllReduce under stream capture on T threads at once: never run anywhere, so not relied on
    mi_ctx* c = nullptr;
    int rc2 = mi_ctx_create(&k, &c);
    if (rc2 != MI_OK) return rc2;
    c->grp = g;
    made[r] = c;
    return MI_OK;
  });
  for (int r = 0; r < T; ++r) g->ranks[r] = made[r];
  if (rc != MI_OK) return rc;
  if (g->lockstep) {   // one stream for every shard: the order of the exchange kernels is the enqueue order
    g->shared_stream = g->ranks[0]->stream;
    for (int r = 1; r < T; ++r) {
      hipStreamDestroy(g->ranks[r]->stream);
      g->ranks[r]->stream = g->shared_stream;
      g->ranks[r]->stream_owned = false;
    }
  }
  if (g->use_rccl) {
    std::vector<ncclComm_t> comms(T);
    std::vector<int> devs(cfg.tp_device_ids, cfg.tp_device_ids + T);
    ncclResult_t nr = ncclCommInitAll(comms.data(), T, devs.data());
    if (nr != ncclSuccess) {
      set_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
      return MI_ECOMM;
    }
    for (int r = 0; r < T; ++r) g->ranks[r]->comm = comms[r];
  }
  return MI_OK;
}

void group_destroy(mi_group* g) {
  if (!g) return;
  if (!g->threads.empty()) {
    // every shard drains its stream first; the lockstep form shares rank 0's stream, so rank 0 (its owner) goes last
    group_run(g, [&](mi_ctx* c, int) -> int {
      if (c) hipStreamSynchronize(c->stream);
      return MI_OK;
    });
    for (int phase = 0; phase < 2; ++phase)
      group_run(g, [&](mi_ctx* c, int r) -> int {
        if (c && (phase == 0) == (r != 0)) {
          if (g->counters[r]) hipFree(g->counters[r]);
          void* p[] = {g->peers.xbuf[r], g->peers.ybuf[r], g->peers.flag1[r], g->peers.flag2[r]};
          for (void* q : p) if (q) hipFree(q);
          mi_ctx_destroy(c);
        }
        return MI_OK;
      });
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->stop = true;
      g->cv_job.notify_all();
    }
    for (auto& t : g->threads) t.join();
  }
  delete g->bar;
  delete g;
}

// Flag blocks are UNCACHED device memory: a peer's polling loads and the publisher's flag stores
// must meet in this GPU's memory.  The message slots are ordinary device memory, kept coherent by
// the release (write back L2) in front of the flag store and the acquire (drop non-local lines)
// behind the flag wait.  Uncached slots were tried and dropped: on one GPU the FIRST exchange after
// the allocation read zeros for 8 KiB..64 KiB of the peers' slots in about one process in six
// (profiles/README.md, "exchange self-test"); ordinary memory never did in the same loop.
static int alloc_exchange_mem(void** p, size_t bytes, bool uncached) {
  if (uncached && !getenv("MI355X_TP_FLAGS_CACHED")) {
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached) == hipSuccess) return MI_OK;
    (void)hipGetLastError();
  }
  MI_HIP(hipMalloc(p, bytes));
  return MI_OK;
}

int group_alloc_exchange(mi_ctx* c) {
  mi_group* g = c->grp;
  const int r = c->cfg.tp_rank, T = g->T;
  if (!g->use_rccl) {
    const size_t cap = (size_t)c->max_rows * c->H;
    const size_t ycap = ((cap / 8 + T - 1) / T + kArMaxBlocks) * 8;
    void *xb = nullptr, *yb = nullptr, *f1 = nullptr, *f2 = nullptr, *ct = nullptr;
    const bool slots_uncached = getenv("MI355X_TP_SLOTS_UNCACHED") != nullptr;   // (the round-2 layout, for the one-off check of the clear above)
    if (alloc_exchange_mem(&xb, 2 * cap * 4, slots_uncached) != MI_OK) return MI_EHIP;   // two slots, fp32-sized (token generation sends fp32)
    if (alloc_exchange_mem(&yb, 2 * ycap * 2, slots_uncached) != MI_OK) return MI_EHIP;
    const size_t fbytes = (size_t)T * kArMaxBlocks * 4;
    if (alloc_exchange_mem(&f1, fbytes, true) != MI_OK) return MI_EHIP;
    if (alloc_exchange_mem(&f2, fbytes, true) != MI_OK) return MI_EHIP;
    MI_HIP(hipMalloc(&ct, AR_WORDS * 4));
    // EVERY exchange allocation is written by its owner before its address is published: round 2 lost the peers'
    // contribution in the FIRST exchange after allocation (4 KiB-aligned ranges, data in place when the host looked later)
    // with never-written slots in uncached memory, and never with the flag blocks, which differed only in being cleared here.
    // What fits that evidence is the allocator's own (lazy, page-granular) initialisation of a fresh allocation landing
    // after the first peer writes; a completed owner-side clear closes that window whatever the memory type.
    MI_HIP(hipMemsetAsync(xb, 0, 2 * cap * 4, c->stream));
    MI_HIP(hipMemsetAsync(yb, 0, 2 * ycap * 2, c->stream));
    MI_HIP(hipMemsetAsync(f1, 0, fbytes, c->stream));
    MI_HIP(hipMemsetAsync(f2, 0, fbytes, c->stream));
    MI_HIP(hipMemsetAsync(ct, 0, AR_WORDS * 4, c->stream));
    MI_HIP(hipStreamSynchronize(c->stream));
    g->peers.xbuf[r] = (uint16_t*)xb;
    g->peers.ybuf[r] = (uint16_t*)yb;
    g->peers.flag1[r] = (uint32_t*)f1;
    g->peers.flag2[r] = (uint32_t*)f2;
    g->counters[r] = (uint32_t*)ct;
    g->cap = cap;
    c->workspace_bytes += 8 * cap + 4 * ycap + 2 * fbytes;
  }
  // every rank's pointers are in the table before anyone launches an exchange
  MI_CHECK(g->bar->wait(), "tensor-parallel group aborted");
  if (!g->use_rccl) {
    // ... and every rank has read one word of every peer's slots and flags (mapped and touched from here) before the first exchange
    const size_t ycap = ((g->cap / 8 + T - 1) / T + kArMaxBlocks) * 8;
    hipLaunchKernelGGL(ar_touch_kernel, dim3(1), dim3(64), 0, c->stream, g->peers, T, g->cap, ycap, g->counters[r]);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(c->stream));
    MI_CHECK(g->bar->wait(), "tensor-parallel group aborted");
  }
  return MI_OK;
}

int group_all_reduce(mi_ctx* c, float* buf, size_t count) {
  mi_group* g = c->grp;
  const int T = g->T, r = c->cfg.tp_rank;
  hipStream_t s = c->stream;
  if (g->use_rccl) {
    ncclResult_t nr = ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, c->comm, s);
    if (nr != ncclSuccess) {
      set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(nr));
      return MI_ECOMM;
    }
    return MI_OK;
  }
  MI_CHECK(count % 8 == 0 && count <= g->cap, "all-reduce: message not a multiple of 8 elements or larger than the exchange slot");
  ArGeom ge{};
  ge.T = T; ge.r = r; ge.chunks = (unsigned)(count / 8); ge.cap = g->cap;
  ge.ycap = ((g->cap / 8 + T - 1) / T + kArMaxBlocks) * 8;
  ge.timeout = ar_timeout_ticks();
  const bool two_shot = count * 2 >= kArTwoShotBytes && T > 2;
  uint32_t* ctr = g->counters[r];
  // publish: one chunk per thread where the message allows it
  ge.G = (int)std::min<unsigned>(kArMaxBlocks, std::max<unsigned>(1, ceil_div((int)ge.chunks, kArThreads)));
  ge.cpw = (ge.chunks + ge.G - 1) / ge.G;
  ge.pubG = ge.G;
  // test hook (tests/test_tp_group_gpu.py: a peer that never publishes): rank MI355X_TP_TEST_MUTE_RANK skips its publish
  const char* mute = getenv("MI355X_TP_TEST_MUTE_RANK");
  if (mute && atoi(mute) == r) {
  } else if (two_shot) hipLaunchKernelGGL(ar_publish_kernel<false>, dim3(ge.G), dim3(kArThreads), 0, s, buf, g->peers, ge, ctr);
  else hipLaunchKernelGGL(ar_publish_kernel<true>, dim3(ge.G), dim3(kArThreads), 0, s, buf, g->peers, ge, ctr);
  const bool hostbar = g->host_barrier;   // concurrent loopback beyond two shards: every publish enqueued before any reduce (group_create)
  if (g->lockstep || hostbar) MI_CHECK(g->bar->wait(), "tensor-parallel group aborted");   // every rank's publish is in the stream
  if (!two_shot) {
    hipLaunchKernelGGL(ar_reduce_kernel<true>, dim3(ge.G), dim3(kArThreads), 0, s, buf, g->peers, ge, ctr);
  } else {
    ArGeom g2 = ge;
    g2.cps = (ge.chunks + T - 1) / T;
    g2.G = (int)std::min<unsigned>(kArMaxBlocks, std::max<unsigned>(1, ceil_div((int)g2.cps, kArThreads)));
    g2.cpw = (g2.cps + g2.G - 1) / g2.G;
    hipLaunchKernelGGL(ar_reduce_scatter_kernel, dim3(g2.G), dim3(kArThreads), 0, s, buf, g->peers, g2, ctr);
    if (g->lockstep || hostbar) MI_CHECK(g->bar->wait(), "tensor-parallel group aborted");
    hipLaunchKernelGGL(ar_gather_kernel, dim3(g2.G), dim3(kArThreads), 0, s, buf, g->peers, g2, ctr);
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// One exchange on known data, checked on the host: rank r contributes r + 1 (+ a per-element
// ramp), every rank must end with the same exact sums.  Run once after the exchange buffers
// exist; the caller falls back to RCCL when the peer-memory path does not deliver (a fabric /
// driver configuration this code has never met), instead of failing at the first model call.
static int selftest_rank(mi_ctx* c, size_t n, int it) {
  mi_group* g = c->grp;
  const int T = g->T, r = c->cfg.tp_rank;
  // pinned staging, copies on the shard's own stream.  The ramp moves with the iteration, so a slot
  // that still shows an earlier exchange (a stale cache line) does not add up to this one's sums.
  auto ramp = [it](size_t i) { return (float)((i + 7 * (size_t)it) % 31) * 0.25f; };
  float* h = nullptr;
  MI_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), n * 4, hipHostMallocDefault));
  for (size_t i = 0; i < n; ++i) h[i] = (float)(r + 1) + ramp(i);   // exact in bf16 too (T = 2, 4, 8, 16)
  int rc = MI_OK;
  if (hipMemcpyAsync(c->partial, h, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MI_EHIP;
  if (rc == MI_OK) rc = group_all_reduce(c, c->partial, n);
  if (rc == MI_OK && hipMemcpyAsync(h, c->partial, n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MI_EHIP;
  if (rc == MI_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MI_EHIP;
  if (rc == MI_OK) rc = group_check_errors(c);
  if (rc != MI_OK) {
    hipHostFree(h);
    if (rc == MI_EHIP) set_error("tensor-parallel exchange self-test: HIP copy failed");
    return rc;
  }
  for (size_t i = 0; i < n; ++i) {
    const float want = (float)(T * (T + 1) / 2) + (float)T * ramp(i);
    if (h[i] != want) {
      std::string dbg;
      if (!g->use_rccl) {
        uint32_t ep = 0;
        (void)hipMemcpy(&ep, g->counters[r] + AR_EPOCH_PUB, 4, hipMemcpyDeviceToHost);
        dbg = " epoch " + std::to_string(ep) + " peers:";
        for (int p = 0; p < T; ++p) {
          float v[2] = {-1.f, -1.f};
          (void)hipMemcpy(v, reinterpret_cast<const unsigned char*>(g->peers.xbuf[p]) + (size_t)(ep & 1) * g->cap * 4 + i * 4, 4, hipMemcpyDeviceToHost);
          (void)hipMemcpy(v + 1, reinterpret_cast<const unsigned char*>(g->peers.xbuf[p]) + (size_t)((ep + 1) & 1) * g->cap * 4 + i * 4, 4, hipMemcpyDeviceToHost);
          dbg += " " + std::to_string(v[0]) + "/" + std::to_string(v[1]);
        }
        size_t bad = 0, last = i;
        for (size_t j = i; j < n; ++j) if (h[j] != (float)(T * (T + 1) / 2) + (float)T * ramp(j)) { ++bad; last = j; }
        dbg += " bad " + std::to_string(bad) + " last " + std::to_string(last) + " cap " + std::to_string(g->cap);
      }
      set_error("tensor-parallel exchange self-test (" + std::to_string(n) + " elements): rank " + std::to_string(r) +
                " element " + std::to_string(i) + " is " + std::to_string(h[i]) + ", expected " + std::to_string(want) + dbg);
      hipHostFree(h);
      return MI_ECOMM;
    }
  }
  hipHostFree(h);
  return MI_OK;
}

int group_selftest(mi_group* g) {
  auto run = [&]() {
    // a token-generation sized message (one-shot) and, when the slots allow, a context-encoding sized one (two-shot)
    int rc = MI_OK;
    // each slot of the double buffers is used, then REUSED with other data (a reader that still holds the
    // line of two exchanges ago fails the second round)
    const int iters = getenv("MI355X_ST_ITERS") ? atoi(getenv("MI355X_ST_ITERS")) : 4;
    for (int it = 0; it < iters && rc == MI_OK; ++it) {
      rc = group_run(g, [&](mi_ctx* c, int) { return selftest_rank(c, std::min<size_t>(g->use_rccl ? (size_t)c->max_rows * c->H : g->cap, 4 * 4096), it); });
      if (rc != MI_OK) fprintf(stderr, "[mi355x] self-test failed at iteration %d\n", it);
    }
    for (int it = 0; it < iters && rc == MI_OK && !g->use_rccl && g->cap >= kArTwoShotBytes; ++it)
      rc = group_run(g, [&](mi_ctx* c, int) { return selftest_rank(c, std::min<size_t>(g->cap, kArTwoShotBytes) / 8 * 8, iters + it); });
    return rc;
  };
  int rc = run();
  g->selftest = rc == MI_OK ? 1 : -1;
  if (rc == MI_OK || g->use_rccl || g->same_device) return rc;
  // the peer-memory exchange did not deliver on this machine: RCCL instead
  fprintf(stderr, "[mi355x] tensor-parallel exchange over peer memory failed its self-test (%s); falling back to RCCL\n",
          mi_last_error());
  std::vector<ncclComm_t> comms(g->T);
  std::vector<int> devs;
  for (int r = 0; r < g->T; ++r) devs.push_back(g->ranks[r]->cfg.device_id);
  ncclResult_t nr = ncclCommInitAll(comms.data(), g->T, devs.data());
  if (nr != ncclSuccess) {
    set_error(std::string("peer-memory exchange failed its self-test and ncclCommInitAll failed too: ") + ncclGetErrorString(nr));
    return MI_ECOMM;
  }
  for (int r = 0; r < g->T; ++r) {
    g->ranks[r]->comm = comms[r];
    g->ranks[r]->cfg.use_graphs = 0;   // as for an RCCL group from the start (group_create)
  }
  g->use_rccl = 1;
  g->bar->reset();
  rc = run();
  if (rc != MI_OK) g->selftest = -2;   // neither transport delivered
  return rc;
}

int group_info(mi_group* g, mi_tp_info_t* o) {
  memset(o, 0, sizeof(*o));
  o->tp_degree = g->T;
  o->transport = g->use_rccl ? MI_TP_TRANSPORT_RCCL : MI_TP_TRANSPORT_P2P;
  o->selftest = g->selftest;
  o->graphs = g->ranks.empty() || !g->ranks[0] ? 0 : g->ranks[0]->cfg.use_graphs;
  o->mode = !g->same_device ? 0 : (g->lockstep ? 1 : 2);
  ar_timeout_ticks(&o->timeout_ms);
  for (int r = 0; r < g->T && r < 16; ++r) { o->device_ids[r] = g->device_ids[r]; o->peer_access[r] = g->peer_access[r]; }
  return MI_OK;
}

int group_check_errors(mi_ctx* c) {
  mi_group* g = c->grp;
  if (!g || g->use_rccl) return MI_OK;
  uint32_t w = 0;
  MI_HIP(hipMemcpy(&w, g->counters[c->cfg.tp_rank] + AR_ERROR, 4, hipMemcpyDeviceToHost));
  if (w != 0) {
    set_error("tensor-parallel exchange: a rank gave up waiting for a peer's flag (peer fault or P2P visibility problem)");
    return MI_ECOMM;
  }
  return MI_OK;
}

}  // namespace mi
