// rc_runtime.h -- what the units of the C-ABI (include/rnacode_hip.h) share: error reporting, buffers that only grow and the pool
// that keeps them between batches, the context, a batch, a stream of batches, and the functions that cross unit boundaries.
//   rc_context.cpp   contexts: streams, constant tables, host thread pool
//   rc_batch.cpp     a batch's life: prepare -> upload -> run -> wait
//   rc_schedule.cpp  which kernels a run launches, with what grids, LDS and scratch (the launch plans)
//   rc_stream.cpp    streams of sub-batches
//   rc_results.cpp   accessors, backtrack, code tables
//   rc_trees_api.cpp the tree estimator's entry points, stand-alone fits and the MT19937 accessor
// (Until round 5 all of this was one file, rc_api.cpp.)
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/rnacode_hip.h"
#include "rc_host.h"
#include "rc_launch.h"

using namespace rc;


inline thread_local std::string g_err;
inline int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(RC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define RC_TRY(expr) do { int r_ = (expr); if (r_ != RC_OK) return r_; } while (0)


// RC_TRACE=1: host-side time stamps of the pipeline stages on stderr
inline bool trace_on() { static const bool on = std::getenv("RC_TRACE") != nullptr; return on; }
// CLOCK_MONOTONIC in microseconds (what Python's time.monotonic() reads as well: a driver script can put its own stamps on the same axis)
inline double trace_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void trace(const char *what, const void *obj) {
  if (!trace_on()) return;
  std::fprintf(stderr, "[rc %14.1f us] %-14s %p\n", trace_now_us(), what, obj);
}
// a runtime call that may block (allocation, release, synchronisation): under RC_TRACE=1 every one that takes longer than 0.5 ms
// leaves a line with its duration and size -- how the step that took 300 times the median was found (tools/stall_hunt.py)
template <typename F> auto timed_call(const char *what, size_t bytes, F f) -> decltype(f()) {
  if (!trace_on()) return f();
  const double t0 = trace_now_us();
  auto r = f();
  const double dt = trace_now_us() - t0;
  if (dt > 500.0) std::fprintf(stderr, "[rc %14.1f us] slow call     %s: %.3f ms (%zu bytes)\n", trace_now_us(), what, dt * 1e-3, bytes);
  return r;
}

// hipEvent created on first use, destroyed with its owner
struct Event {
  hipEvent_t e = nullptr;
  ~Event() { if (e) (void)hipEventDestroy(e); }
  hipError_t record(hipStream_t st) {
    if (!e) { hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; }
    return hipEventRecord(e, st);
  }
  operator hipEvent_t() const { return e; }
};

// Buffers that only grow (batches of a stream are recycled: the steady state allocates nothing).  Growing never frees on the
// spot: hipFree and hipHostFree wait for the whole device, and a stream whose sub-batches grow (a small first one, then doubling)
// would stop at every submit until the kernels of the batches in flight had finished (10 ms per submit in the stand-alone driver).
// The old allocation is retired instead and released when nothing of this library is in flight.
struct Retired {
  std::mutex m;
  std::vector<std::pair<void *, bool>> list;   // pointer, pinned host memory?
  size_t bytes = 0;
  std::atomic<bool> ctxGone{false};            // set by rc_ctx_destroy: a batch that outlives its context must not touch it
  void add(void *p, bool host, size_t n) { std::lock_guard<std::mutex> lk(m); list.emplace_back(p, host); bytes += n; }
  void drain() {
    std::vector<std::pair<void *, bool>> take;
    { std::lock_guard<std::mutex> lk(m); take.swap(list); bytes = 0; }
    for (auto &e : take) { if (e.second) (void)timed_call("retired hipHostFree", 0, [&] { return hipHostFree(e.first); }); else (void)timed_call("retired hipFree", 0, [&] { return hipFree(e.first); }); }
  }
  bool empty() { std::lock_guard<std::mutex> lk(m); return list.empty(); }
  size_t held() { std::lock_guard<std::mutex> lk(m); return bytes; }
};
// Buffers that belong to no context yet retire here; a batch's buffers are pointed at their context's list when the batch is bound
// to it (rc_batch::adopt): one context's wait then never frees -- and device-synchronises for -- another context's leftovers.
inline Retired g_retired;

// An allocation that fails while outgrown buffers are still parked (a stream kept at depth 3 never has "nothing in flight") gets a
// second try after the device has drained and the parked buffers are gone.
template <typename Alloc> hipError_t alloc_with_retry(Retired *ret, Alloc alloc) {
  hipError_t r = alloc();
  if (r == hipSuccess || (ret->empty() && g_retired.empty())) return r;
  (void)hipGetLastError();
  (void)timed_call("alloc retry hipDeviceSynchronize", 0, [&] { return hipDeviceSynchronize(); });
  ret->drain();
  g_retired.drain();
  return alloc();
}

// Buffers of destroyed batches, kept by their context for the next batch: hipFree / hipHostFree synchronise the device and
// hipMalloc / hipHostMalloc take 50-300 us each, and a caller that makes one batch per alignment block (the reference's main loop through
// the shim, RNAcode.c:164-216) paid both fourteen times per block -- 0.6 ms of the 1.6 ms a one-block batch took, 1.1 of 6.3 ms for
// genomic-preprocessed.maf.  A buffer is handed out again for requests between a quarter of its size and its size.  Bounded
// (a sixteenth of the device memory, 1 GB of pinned memory, 256 buffers); what does not fit is freed as before.  The pool outlives
// its context while batches still point at it (a batch may be destroyed after its context).
// Every live pool of the process: an allocation that fails on a device gives back what the pools of the OTHER contexts on that device keep,
// too, before it gives up (several contexts on one GPU: rnacode_hip --gpus N --devices 0,0,..., the tests).
struct BufPool;
struct PoolRegistry {
  std::mutex m;
  std::vector<BufPool *> pools;
  void add(BufPool *p) { std::lock_guard<std::mutex> lk(m); pools.push_back(p); }
  void remove(BufPool *p) { std::lock_guard<std::mutex> lk(m); pools.erase(std::remove(pools.begin(), pools.end(), p), pools.end()); }
  inline void drain_device(int device);
};
inline PoolRegistry g_pools;

struct BufPool {
  struct Slot { void *p; size_t cap; };
  std::mutex m;
  std::vector<Slot> dev, host;
  size_t devBytes = 0, hostBytes = 0, devCap = static_cast<size_t>(1) << 30, hostCap = static_cast<size_t>(1) << 30;
  bool closed = false;   // the context is gone: nothing is kept any more
  int device = -1;
  BufPool() { g_pools.add(this); }
  void *take(bool pinned, size_t want, size_t *got) {
    std::lock_guard<std::mutex> lk(m);
    std::vector<Slot> &v = pinned ? host : dev;
    int best = -1;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].cap >= want && v[i].cap / 4 <= want && (best < 0 || v[i].cap < v[static_cast<size_t>(best)].cap)) best = static_cast<int>(i);
    if (best < 0) return nullptr;
    Slot s = v[static_cast<size_t>(best)];
    v.erase(v.begin() + best);
    (pinned ? hostBytes : devBytes) -= s.cap;
    *got = s.cap;
    return s.p;
  }
  bool give(bool pinned, void *p, size_t cap) {
    std::lock_guard<std::mutex> lk(m);
    std::vector<Slot> &v = pinned ? host : dev;
    size_t &bytes = pinned ? hostBytes : devBytes;
    if (closed || v.size() >= 256 || bytes + cap > (pinned ? hostCap : devCap)) return false;
    v.push_back(Slot{p, cap});
    bytes += cap;
    return true;
  }
  void drain(bool close) {
    std::vector<Slot> d, h;
    { std::lock_guard<std::mutex> lk(m); d.swap(dev); h.swap(host); devBytes = hostBytes = 0; closed = closed || close; }
    for (auto &s : d) (void)timed_call("pool hipFree", s.cap, [&] { return hipFree(s.p); });
    for (auto &s : h) (void)timed_call("pool hipHostFree", s.cap, [&] { return hipHostFree(s.p); });
  }
  ~BufPool() { g_pools.remove(this); drain(true); }
};
inline void PoolRegistry::drain_device(int device) {
  std::lock_guard<std::mutex> lk(m);   // (a pool cannot go away while it is being drained: its destructor takes this lock first)
  for (BufPool *p : pools) if (p->device == device) p->drain(false);
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  Retired *retired = &g_retired;
  BufPool *pool = nullptr;
  ~DevBuf() { release(); }
  void release() { if (p && !(pool && pool->give(false, p, cap))) (void)timed_call("hipFree", cap, [&] { return hipFree(p); }); p = nullptr; cap = 0; }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) retired->add(p, false, cap);
    p = nullptr; cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    if (pool && (p = pool->take(false, want, &cap))) return hipSuccess;
    hipError_t r = alloc_with_retry(retired, [&] { return timed_call("hipMalloc", want, [&] { return hipMalloc(&p, want); }); });
    if (r != hipSuccess && pool) { pool->drain(false); r = hipMalloc(&p, want); }
    if (r != hipSuccess && pool) { g_pools.drain_device(pool->device); r = hipMalloc(&p, want); }
    if (r == hipSuccess) cap = want; else p = nullptr;
    return r;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  Retired *retired = &g_retired;
  BufPool *pool = nullptr;
  ~PinBuf() { release(); }
  void release() { if (p && !(pool && pool->give(true, p, cap))) (void)timed_call("hipHostFree", cap, [&] { return hipHostFree(p); }); p = nullptr; cap = 0; }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) retired->add(p, true, cap);
    p = nullptr; cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 4096);
    if (pool && (p = pool->take(true, want, &cap))) return hipSuccess;
    hipError_t r = alloc_with_retry(retired, [&] { return timed_call("hipHostMalloc", want, [&] { return hipHostMalloc(&p, want, hipHostMallocDefault); }); });
    if (r != hipSuccess && pool) { pool->drain(false); r = hipHostMalloc(&p, want, hipHostMallocDefault); }
    if (r == hipSuccess) cap = want; else p = nullptr;
    return r;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

// CPUs this process may actually use: affinity mask and the cgroup CPU quota (a container sees all cores of the
// host in hardware_concurrency but is throttled to its quota)
inline int effective_cpus() {
  int n = static_cast<int>(std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, CPU_COUNT(&set));
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64]; long long period = 0;
    if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
      const long long quota = std::atoll(q);
      if (quota > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    }
    std::fclose(f);
  }
  return std::max(1, n);
}

// persistent worker threads for the block preparation: run(n, fn) calls fn(i) for every i in [0, n), the caller
// takes part, items are handed out one by one (blocks differ in size)
class Pool {
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  std::atomic<int> next_{0};
  int total_ = 0, busy_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= total_) break;
      (*job_)(i);
    }
  }
  void worker() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--busy_ == 0) done_.notify_all();
      }
    }
  }

 public:
  explicit Pool(int nthreads) { for (int t = 1; t < nthreads; t++) th_.emplace_back([this] { worker(); }); }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return static_cast<int>(th_.size()) + 1; }
  void run(int n, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    if (th_.empty() || n < 4) { for (int i = 0; i < n; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &fn; total_ = n; next_.store(0); busy_ = static_cast<int>(th_.size()); gen_++;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return busy_ == 0; });
    job_ = nullptr;
  }
};

inline size_t al256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }


struct rc_ctx {
  int device = 0;
  int numCU = 0;
  size_t totalMem = 0;             // device memory (bytes)
  size_t togetherWords = static_cast<size_t>(2) << 30;   // uint32 of staging scratch the side-by-side class launches of a batch may take (RC_TOGETHER_MAX_MB)
  DevBuf treeJobs, treeIn, treeWork, treeRes;   // rc_fit_trees_device
  PinBuf treeInPin, treeResPin;
  std::mutex treeMutex;
  int gridCapPerCU = 0;       // RC_GRID_CAP_PER_CU: at most this many k_null workgroups per CU (experiments on cache residency of the code scratch); 0 = what fits
  double splitFactor = 0.5;   // a k_null launch with fewer (block, sample group) items than this fraction of the chip's wavefront slots splits every item into its six strand x frame parts (RC_SPLIT_FACTOR)
  long long fatStreamMinItems = 16384;   // RC_FAT_STREAM_MIN_ITEMS: a stream's sub-batch of fat classes takes the two-row path from this many (block, 64-sample group) items on: four rounds of the chip's wavefront slots (1024 blocks at n = 1000)
  int tailSharing = 1;        // k_null: items taken near the end of the queues are shared part by part with workgroups that have run out of work (RC_TAIL_SHARING=0: off)
  int nativeWavesPerCU = 8;   // persistent wavefronts of k_native_dp per CU (RC_NATIVE_WAVES_PER_CU): it runs beside k_null and should not crowd it out
  size_t genericScratchWords = static_cast<size_t>(1) << 30;   // uint32 of staging scratch the wide-block kernels (k_generic_sim / k_generic_dp) may take per round of items: a twelfth of the device memory (24 GB of 288)
  static constexpr int kComputeStreams = 3;
  hipStream_t compute[kComputeStreams] = {nullptr, nullptr, nullptr};   // null sampling + fit; consecutive batches take them in turn, so the tail of one overlaps the head of the next
  static constexpr int computeStreams = 2;       // consecutive batches alternate between two (a third was measured: the chip is busy with the second batch by then, 1250-block shard streamed 7.85 ms either way)
  hipStream_t native = nullptr;    // table preparation and native-block kernels: small and latency-bound, they run beside the null sampling
  hipStream_t copy = nullptr;      // host -> device blob copies
  hipStream_t aux = nullptr;       // synchronous helpers (second-stage result copies, backtrack rows, stand-alone fits): never behind queued batches
  hipStream_t tree2 = nullptr;     // ... and the second stream its class launches alternate with
  Event treeFork, treeJoin;
  hipStream_t tree = nullptr;      // rc_fit_trees_device: 25-40 ms kernels of a driver's tree-fit thread, kept off the helper stream
  int prioLow = 0, prioHigh = 0;   // stream priorities (compute streams low, the small latency-bound work high)
  int classPrio = 0;               // priority of the class streams
  int expMode = 0;                 // the EVD fit's exp: the host C library's algorithm and variant, if it is one the library knows (rc_refexp.h)
  std::mutex streamMutex;          // ctx_stream
  std::vector<hipStream_t> classStreams;   // k_null launches of different row-count classes, when a batch has several: these beside the run's own stream
  int classStreamCount = 3;                // ... that many in all (RC_CLASS_STREAMS)
  unsigned seq = 0;
  CodeTables ct62{62}, ct90{90};
  PairTable pt62, pt90;
  DevBuf d_tables;                 // pair tables, genetic code, BLOSUM matrices, code infos (see table_ptrs)
  // MT19937 stream cache
  uint32_t *d_U = nullptr;
  size_t U_cap = 0;
  uint32_t U_seed = 0;
  int U_Spad = 0, U_D = 0;
  bool U_valid = false;
  Event U_ready;
  float tieThr = 0.0f;
  std::map<std::pair<int, size_t>, int> occ;   // (N-1, LDS bytes) -> resident workgroups per CU of k_null
  struct PlainPlan { int occ; size_t lds; int hiOcc; };
  std::map<std::tuple<int, int, int>, PlainPlan> plainPlans;   // (N-1, tree nodes, high-occupancy wanted) -> launch shape of a k_null that reads its codes from L2
  size_t ldsPerCU = 160 * 1024;
  int rowSplit = 1;             // RC_ROW_SPLIT=0: never split a strand x frame part's rows over workgroups (tiny batches; A/B)
  int highOccupancy = 1;        // RC_HIGH_OCCUPANCY: 1 k_null_occ (one more wavefront per SIMD) for batches of one row-count class, 0 never, 2 always (A/B)
  size_t ldsMaxBytes = 13312;   // per wavefront: >= 12 wavefronts per CU; measured crossover against reading the codes from L2 (cols 150: 52.0 vs 54.0 ms, cols 180: 41.5 vs 39.3 ms)
  size_t ldsMaxBytesOneRow = 10752;   // the one-row kernels: staging only where it leaves >= 15 workgroups per CU (6 rows x 120 / 135 / 150 columns: staged 19.6 / 25.8 / 34.0 ms, from L2 with the LDS sigma tables 20.2 / 24.9 / 30.2)
  bool stageManyRows = false;   // RC_LDS_MAX_BYTES set: classes of six or more other sequences may stage their codes as well (by default they never do: their look-ups
                                // are ds_bpermute_b32 then, and from L2 with the LDS sigma tables they are 0..30 % faster at every block length measured)
  int debugSkip = 0;
  DevBuf d_cellStats;              // RC_PROFILING builds with RC_CELL_STATS=1: k_null's cell counters, printed when the context goes
  int serialNative = 0;
  // blocks with at least this many rows take the generic kernels (RC_GENERIC_MIN_ROWS: tests run everything through them).  Where the crossover lies is
  // measured (tools/rows_sweep.py, profiles/r06/rows_sweep_*.txt): k_null<N-1> holds 3 (N-1) states in registers -- 2 wavefronts per SIMD up to 36 rows,
  // ONE from 37 on (0.11..0.18 of the issue roof at 40..64 rows), where k_generic_dp (states parked in memory, any N) runs at 0.19..0.24
  // ... and blocks of tiledMinRows rows and more (omega <= 0, short enough) k_tiled_dp (rc_null_tiled.h; RC_TILED_MIN_ROWS, 65 = never; RC_TILED_MAX_L,
  // RC_TILED_MAX_ROWS, RC_TILED_WIDE_MAX_L).  rc_device.h, block_class
  ClassRule rule = kDefaultClassRule;
  int stopRounds = 6;                    // at most this many rounds (RC_STOP_ROUNDS; <= kMaxRounds)
  long long stopTwoRoundsMinItems = 0;   // --stop-early runs two rounds of sample groups only for batches with at least this many (block, group) items
  int hostThreads = 0;
  std::unique_ptr<Pool> pool;
  std::atomic<int> inflight{0};   // batches between run_async and wait
  std::shared_ptr<Retired> retired = std::make_shared<Retired>();   // outgrown buffers of this context's batches (shared: a batch may outlive its context)
  std::shared_ptr<BufPool> bufPool = std::make_shared<BufPool>();   // buffers of destroyed batches, for the next ones
  size_t retiredCapBytes = 0;     // more than this parked while batches are in flight: the next wait drains (and stalls) once; 1/16 of the device memory
};

// the context's streams, created on first use (see rc_ctx_create); which: 0..2 compute, 3 native, 4 copy, 5 aux, 6 and 7 tree fits
// nullptr if the stream cannot be created: callers return RC_ERR_DEVICE (a null handle would be the legacy default stream, which
// does not order against the non-blocking ones)
inline hipStream_t ctx_stream(rc_ctx *c, int which) {
  hipStream_t *slot = which < rc_ctx::kComputeStreams ? &c->compute[which] : which == 3 ? &c->native : which == 4 ? &c->copy : which == 5 ? &c->aux : which == 6 ? &c->tree : &c->tree2;
  std::lock_guard<std::mutex> lk(c->streamMutex);
  if (!*slot) {
    (void)hipSetDevice(c->device);
    if (hipStreamCreateWithPriority(slot, hipStreamNonBlocking, which < rc_ctx::kComputeStreams ? c->prioLow : c->prioHigh) != hipSuccess) *slot = nullptr;
  }
  return *slot;
}
inline hipStream_t stream_native(rc_ctx *c) { return ctx_stream(c, 3); }
inline hipStream_t stream_copy(rc_ctx *c) { return ctx_stream(c, 4); }
inline hipStream_t stream_aux(rc_ctx *c) { return ctx_stream(c, 5); }
inline hipStream_t stream_tree(rc_ctx *c) { return ctx_stream(c, 6); }
inline hipStream_t stream_tree2(rc_ctx *c) { return ctx_stream(c, 7); }
#define RC_STREAM_TRY(var, expr)                                                               \
  hipStream_t var = (expr);                                                                    \
  if (!var) return fail(RC_ERR_DEVICE, "hipStreamCreateWithPriority failed")

// layout of rc_ctx::d_tables
struct TablePtrs { const uint8_t *pair; const int *pep; const int *blosum; const CodeInfo *info; };
inline TablePtrs table_ptrs(const rc_ctx *c, int blosum) {
  const uint8_t *base = c->d_tables.as<uint8_t>();
  const size_t per = 4096 + 400 * sizeof(int) + 64 * sizeof(CodeInfo);
  const uint8_t *m = base + 64 * sizeof(int) + (blosum == 90 ? per : 0);
  return TablePtrs{m, reinterpret_cast<const int *>(base), reinterpret_cast<const int *>(m + 4096),
                   reinterpret_cast<const CodeInfo *>(m + 4096 + 400 * sizeof(int))};
}

struct rc_stream;

struct rc_batch {
  // Declared FIRST, hence destroyed LAST: the DevBuf / PinBuf members below hand their memory to the pool (or park it in the list)
  // from their destructors, and a batch may be destroyed after its context -- these references are then the last ones.
  std::shared_ptr<Retired> retiredKeep;   // the context's list of outgrown buffers
  std::shared_ptr<BufPool> poolKeep;      // ... and its pool of buffers of destroyed batches
  rc_ctx *ctx = nullptr;
  int device = 0;
  rc_params par{};
  int n = 0;
  enum State { EMPTY, READY, RUNNING, DONE } state = EMPTY;
  std::vector<BlockMeta> meta;
  std::vector<DevBlock> db;                  // host copies of the block headers (offsets for the accessors)
  std::map<int, std::string> errs;           // blocks that were rejected: index -> message
  std::vector<int> okBlocks;                 // batch indices with status RC_OK
  std::map<int, std::vector<int>> classes;   // class (N-1, or kGenericClass for blocks of more than 64 rows) -> batch indices
  int classOff[kClassSlots] = {0};           // start of a class in the class-ordered lists
  int maxL = 0, maxDraws = 0, hssCap = 8;
  bool allExact = false;                     // gap parameters outside the fast kernels' range: EXACT instantiation for every block
  // blob
  PinBuf hblob;
  size_t hostCap = 0, hostUsed = 0, devCap = 0;
  size_t oDblocks = 0, oOk = 0, oClass = 0, oPrefix = 0;   // tail arrays inside the host part
  int nModels = 0;
  // device
  DevBuf dblob, dmaxima, dhss, dhssCount, dhssPacked, dhssOffsets, dfit, dflags, dexact, dcounters, dscratch, dnativeScratch, dnativeTile, dnativeAll, dsteal, dsbuf;
  size_t sbufNeed = 0;      // floats of row buffer the rounds of the current run need (launch_null_groups, sizing pass)
  float *maxPtr = nullptr;
  bool maximaExternal = false;
  PinBuf hstage, hpacked;
  Event evUp, evPrep0, evPrep, evStart, evMt0, evMt1, evN0, evN1, evS0, evS1, evMid, evFit, evDone;
  std::vector<std::unique_ptr<Event>> classDone;
  bool mtLaunched = false;
  size_t packedAhead = 0;   // HSS records copied to the host together with the counts
  // host results
  std::vector<float> maxima;
  bool maximaFetched = false;
  std::vector<FitOut> fit;
  std::vector<int> hssCount, hssOff;         // per (block, strand x frame) slot: records and where they start in hssRec
  std::vector<DevHss> hssRec;
  unsigned long long clamped = 0;
  float t[6] = {0, 0, 0, 0, 0, 0};
  int nl[6] = {0, 0, 0, 0, 0, 0};
  double prepHostMs = 0.0;
  std::string nullKernel;   // the k_null instantiation that did most of the last run's sampling, as rocprofv3 prints it
  double nullKernelCost = 0.0;
  rc_stream *owner = nullptr;
  // bind the batch to its context: from now on its buffers, when outgrown, are parked in that context's list, and come from and go
  // back to its pool
  void adopt(rc_ctx *c) {
    ctx = c; device = c->device; retiredKeep = c->retired; poolKeep = c->bufPool;
    Retired *r = retiredKeep.get();
    for (DevBuf *d : {&dblob, &dmaxima, &dhss, &dhssCount, &dhssPacked, &dhssOffsets, &dfit, &dflags, &dexact, &dcounters, &dscratch, &dnativeScratch,
                      &dnativeTile, &dnativeAll, &dsteal, &dsbuf}) { d->retired = r; d->pool = poolKeep.get(); }
    for (PinBuf *h : {&hblob, &hstage, &hpacked}) { h->retired = r; h->pool = poolKeep.get(); }
  }
};

struct rc_stream {
  rc_ctx *ctx = nullptr;
  rc_params par{};
  int depth = 2;
  std::deque<rc_batch *> inflight;
  std::vector<rc_batch *> spare;
};

// layout of rc_batch::dcounters (uint32 units)
constexpr size_t kCntExact = 0;                                  // [kClassSlots] int: blocks flagged by k_prep_lut, by class
constexpr int kMaxRounds = 6;                                    // --stop-early: rounds of sample groups (each doubles the samples simulated so far)
constexpr size_t kCntWork = kClassSlots + 1;                     // [kMaxRounds][3: fast / exact / split simulation][kClassSlots][8]
constexpr size_t kCntClamp = kCntWork + static_cast<size_t>(kMaxRounds) * 3 * kClassSlots * 8;  // one u64 (8-byte aligned: see static_assert)
constexpr size_t kCntTotal = kCntClamp + 2;
static_assert((kCntClamp % 2) == 0, "clamp counter alignment");

// one run's environment, handed from rc_batch.cpp to the launch plans of rc_schedule.cpp
struct RunEnv {
  rc_batch *b;
  rc_ctx *c;
  hipStream_t cs;     // this run's compute stream
  int n, sampleN, groups, Spad, slots;
  TablePtrs tp;
  const uint8_t *blob;
  const DevBlock *dblocks;
  const int *okList, *classList;
  bool streaming;     // a sub-batch of a stream: others are queued behind it
  bool nativeFirst;   // the native-block kernels run on this run's compute stream in front of k_null, with the chip to themselves
  int nativeMode;     // native_grid's mode
  bool fat;           // every class of the batch takes the two-row k_null (fat_class) and nothing runs beside it
};

// functions that cross unit boundaries
void want_hw_queues();
int batch_prepare(rc_batch *b, const rc_block *blocks, int32_t n_blocks, const rc_params *par);   // rc_batch.cpp
int batch_upload(rc_batch *b);
int batch_run_async(rc_batch *b, bool streaming = false);
int batch_wait(rc_batch *b);
void trace_device(rc_batch *b);
extern bool g_no_stream_cache;
size_t native_grid(const rc_ctx *c, size_t items, int smax, int mode = 0);                        // rc_schedule.cpp
bool fat_class(const rc_batch *b, const rc_ctx *c, int NK, int maxL);
int launch_native_block(const RunEnv &R);
size_t steal_slots(const rc_ctx *c);
size_t steal_words(const rc_ctx *c);
int launch_null_groups(const RunEnv &R, int gLo, int gHi, int phase, uint32_t extraSkip, size_t *sizeOnly = nullptr);
