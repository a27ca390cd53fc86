// rc_api.cpp -- the C-ABI of include/rnacode_hip.h: contexts, batches resident in HBM, launches, streams of batches.
//
// No CPU implementation of the hot path lives here: scoring, sampling, scan and fit are
// kernels in rc_kernels.hip; this file prepares block constants (rc_host.cpp, host threads), moves bytes,
// launches, and formats results.  Every compute entry point fails with RC_ERR_DEVICE when no
// HIP device is usable.
//
// A batch goes through: prepare (host threads write the block tables straight into one pinned blob) ->
// upload (ONE asynchronous copy + the kernels that derive the expected-score tables on the device) ->
// run (all launches asynchronous, results staged into pinned memory) -> wait.  rc_batch_create/rc_batch_run
// are the synchronous forms; rc_stream_* keeps several batches in flight so that the host prepares batch
// i+1 while the GPU scores batch i (RNAcode.c:115-221 is a stream of blocks).
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

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(RC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define RC_TRY(expr) do { int r_ = (expr); if (r_ != RC_OK) return r_; } while (0)

namespace {

// RC_TRACE=1: host-side time stamps of the pipeline stages on stderr
bool trace_on() { static const bool on = std::getenv("RC_TRACE") != nullptr; return on; }
// CLOCK_MONOTONIC in microseconds (what Python's time.monotonic() reads as well: a driver script can put its own stamps on the same axis)
double trace_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void trace(const char *what, const void *obj) {
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
Retired g_retired;

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
struct BufPool {
  struct Slot { void *p; size_t cap; };
  std::mutex m;
  std::vector<Slot> dev, host;
  size_t devBytes = 0, hostBytes = 0, devCap = static_cast<size_t>(1) << 30, hostCap = static_cast<size_t>(1) << 30;
  bool closed = false;   // the context is gone: nothing is kept any more
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
  ~BufPool() { drain(true); }
};

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
int effective_cpus() {
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

size_t al256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

}  // namespace

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
  int dualRows = 1;           // RC_DUAL_ROWS=0: never the two-row instantiations of k_null (A/B)
  int nativeScanLanes = 1;    // RC_NATIVE_SCAN_LANES=0: getHSS of the native block inside k_native_dp, one wavefront per matrix (A/B)
  int genericSplit = 1;       // RC_GENERIC_SPLIT=0: blocks of more than 64 rows simulated and scored by one kernel (A/B)
  int dualFromL2 = 1;         // RC_DUAL_FROM_L2: the two-row k_null reads its codes from L2 behind a suffix cache where staging them would leave <= 12 workgroups per CU (1, default), never (0), wherever the two-row kernel runs (2)
  long long fatStreamMinItems = 16384;   // RC_FAT_STREAM_MIN_ITEMS: a stream's sub-batch of fat classes takes the two-row path from this many (block, 64-sample group) items on: four rounds of the chip's wavefront slots (1024 blocks at n = 1000)
  int nativeFirst = -1;       // RC_NATIVE_FIRST: 1 always queue a batch's native-block kernels in front of its k_null, 0 never (and no two-row k_null for fat classes); -1: for batches that take the fat path (see fat_class)
  int copyResults = 0;        // RC_COPY_RESULTS=1: the small results go back with hipMemcpyAsync instead of k_results_out (A/B)
  int tailRelease = 1;        // how k_null publishes a late item's codes: 1 agent-scope release (the documented cross-CU hand-off), 0 workgroup scope (RC_TAIL_RELEASE=workgroup: what gfx950 needs in practice, 1-1.5 % faster on streams of small sub-batches, not guaranteed)
  int tailSharing = 1;        // k_null: items taken near the end of the queues are shared part by part with workgroups that have run out of work (RC_TAIL_SHARING=0: off)
  int nativeWavesPerCU = 8;   // persistent wavefronts of k_native_dp per CU (RC_NATIVE_WAVES_PER_CU): it runs beside k_null and should not crowd it out
  size_t genericScratchWords = static_cast<size_t>(1) << 30;   // uint32 of staging scratch one k_null_generic launch may take: a twelfth of the device memory (24 GB of 288)
  static constexpr int kComputeStreams = 3;
  hipStream_t compute[kComputeStreams] = {nullptr, nullptr, nullptr};   // null sampling + fit; consecutive batches take them in turn, so the tail of one overlaps the head of the next
  int computeStreams = 2;                        // RC_COMPUTE_STREAMS (1..3).  A third one was measured: the third batch of a stream no longer waits behind the first one's tail and fit, but the chip is busy with the second by then (1250-block shard streamed: 7.85 ms either way)
  hipStream_t native = nullptr;    // table preparation and native-block kernels: small and latency-bound, they run beside the null sampling
  hipStream_t copy = nullptr;      // host -> device blob copies
  hipStream_t aux = nullptr;       // synchronous helpers (second-stage result copies, backtrack rows, stand-alone fits): never behind queued batches
  hipStream_t tree = nullptr;      // rc_fit_trees_device: 25-40 ms kernels of a driver's tree-fit thread, kept off the helper stream
  int prioLow = 0, prioHigh = 0;   // stream priorities (compute streams low, the small latency-bound work high)
  int classPrio = 0;               // priority of the class streams
  std::mutex streamMutex;          // ctx_stream
  std::vector<hipStream_t> classStreams;   // k_null launches of different row-count classes, when a batch has several
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
  int suffixCache = 1;          // RC_SUFFIX_CACHE=0: the launches that read their codes from L2 keep none of them in LDS (A/B)
  int highOccupancy = 1;        // RC_HIGH_OCCUPANCY: 1 k_null_occ (one more wavefront per SIMD) for batches of one row-count class, 0 never, 2 always (A/B)
  size_t ldsMaxBytes = 13312;   // per wavefront: >= 12 wavefronts per CU; measured crossover against reading the codes from L2 (cols 150: 52.0 vs 54.0 ms, cols 180: 41.5 vs 39.3 ms)
  size_t ldsMaxBytesOneRow = 10752;   // the one-row kernels: staging only where it leaves >= 15 workgroups per CU (6 rows x 120 / 135 / 150 columns: staged 19.6 / 25.8 / 34.0 ms, from L2 with the LDS sigma tables 20.2 / 24.9 / 30.2)
  bool stageManyRows = false;   // RC_LDS_MAX_BYTES set: classes of six or more other sequences may stage their codes as well (by default they never do: their look-ups
                                // are ds_bpermute_b32 then, and from L2 with the LDS sigma tables they are 0..30 % faster at every block length measured)
  int debugSkip = 0;
  DevBuf d_cellStats;              // RC_PROFILING builds with RC_CELL_STATS=1: k_null's cell counters, printed when the context goes
  int serialNative = 0;
  int genericMinRows = kTemplRows + 1;   // blocks with at least this many rows take the generic kernels (RC_GENERIC_MIN_ROWS: tests run everything through them)
  int stopRounds = 6;                    // at most this many rounds (RC_STOP_ROUNDS; <= kMaxRounds)
  long long stopTwoRoundsMinItems = 0;   // --stop-early runs two rounds of sample groups only for batches with at least this many (block, group) items
  int hostThreads = 0;
  std::unique_ptr<Pool> pool;
  std::atomic<int> inflight{0};   // batches between run_async and wait
  std::shared_ptr<Retired> retired = std::make_shared<Retired>();   // outgrown buffers of this context's batches (shared: a batch may outlive its context)
  std::shared_ptr<BufPool> bufPool = std::make_shared<BufPool>();   // buffers of destroyed batches, for the next ones (RC_BUFFER_POOL=0: off)
  int poolEnabled = 1;
  size_t retiredCapBytes = 0;     // more than this parked while batches are in flight: the next wait drains (and stalls) once; 1/16 of the device memory
};

// the context's streams, created on first use (see rc_ctx_create); which: 0..2 compute, 3 native, 4 copy, 5 aux, 6 tree fits
// nullptr if the stream cannot be created: callers return RC_ERR_DEVICE (a null handle would be the legacy default stream, which
// does not order against the non-blocking ones)
static hipStream_t ctx_stream(rc_ctx *c, int which) {
  hipStream_t *slot = which < rc_ctx::kComputeStreams ? &c->compute[which] : which == 3 ? &c->native : which == 4 ? &c->copy : which == 5 ? &c->aux : &c->tree;
  std::lock_guard<std::mutex> lk(c->streamMutex);
  if (!*slot) {
    (void)hipSetDevice(c->device);
    if (hipStreamCreateWithPriority(slot, hipStreamNonBlocking, which < rc_ctx::kComputeStreams ? c->prioLow : c->prioHigh) != hipSuccess) *slot = nullptr;
  }
  return *slot;
}
static hipStream_t stream_native(rc_ctx *c) { return ctx_stream(c, 3); }
static hipStream_t stream_copy(rc_ctx *c) { return ctx_stream(c, 4); }
static hipStream_t stream_aux(rc_ctx *c) { return ctx_stream(c, 5); }
static hipStream_t stream_tree(rc_ctx *c) { return ctx_stream(c, 6); }
#define RC_STREAM_TRY(var, expr)                                                               \
  hipStream_t var = (expr);                                                                    \
  if (!var) return fail(RC_ERR_DEVICE, "hipStreamCreateWithPriority failed")

// layout of rc_ctx::d_tables
struct TablePtrs { const uint8_t *pair; const int *pep; const int *blosum; const CodeInfo *info; };
static TablePtrs table_ptrs(const rc_ctx *c, int blosum) {
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
  DevBuf dblob, dmaxima, dhss, dhssCount, dhssPacked, dhssOffsets, dfit, dflags, dexact, dcounters, dscratch, dnativeScratch, dnativeTile, dnativeAll, dsteal;
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
    ctx = c; device = c->device; retiredKeep = c->retired; poolKeep = c->poolEnabled ? c->bufPool : nullptr;
    Retired *r = retiredKeep.get();
    for (DevBuf *d : {&dblob, &dmaxima, &dhss, &dhssCount, &dhssPacked, &dhssOffsets, &dfit, &dflags, &dexact, &dcounters, &dscratch, &dnativeScratch,
                      &dnativeTile, &dnativeAll, &dsteal}) { d->retired = r; d->pool = poolKeep.get(); }
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
namespace {
constexpr size_t kCntExact = 0;                                  // [kClassSlots] int: blocks flagged by k_prep_lut, by class
constexpr int kMaxRounds = 6;                                    // --stop-early: rounds of sample groups (each doubles the samples simulated so far)
constexpr size_t kCntWork = kClassSlots + 1;                     // [kMaxRounds][2 fast/exact][kClassSlots][8]
constexpr size_t kCntClamp = kCntWork + static_cast<size_t>(kMaxRounds) * 2 * kClassSlots * 8;  // one u64 (8-byte aligned: see static_assert)
constexpr size_t kCntTotal = kCntClamp + 2;
static_assert((kCntClamp % 2) == 0, "clamp counter alignment");
}  // namespace

extern "C" {

const char *rc_last_error(void) { return g_err.c_str(); }

void rc_default_params(rc_params *p) {
  p->Delta = -10.0f; p->Omega = -4.0f; p->omega = -2.0f;
  p->stopPenalty_0 = -9999.0f; p->stopPenalty_k = -8.0f;
  p->blosum = 62; p->sampleN = 100; p->cutoff = 1.0f; p->stopEarly = 0; p->seed_base = 42;
}

// A batch with several row-count classes puts its class launches on separate streams, and a stream of sub-batches keeps two batches in
// flight: more kernels at once than the four hardware queues the HIP runtime gives a process by default, and kernels that share a queue run
// one after the other.  Ask for eight before the runtime starts (no effect, and no harm, if the host program has already initialised it;
// an explicit setting wins).  10 000 blocks of 3..12 rows as 1024-block sub-batches: 148 -> 116 ms.
static void want_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int rc_device_count(void) {
  want_hw_queues();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rc_host_cpus(void) { return effective_cpus(); }

void rc_ctx_destroy(rc_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->retired->ctxGone.store(true);
  c->retired->drain();
  g_retired.drain();
  c->bufPool->drain(true);
  c->pool.reset();
  if (kProfiling && c->d_cellStats.p) {
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    if (hipMemcpy(st, c->d_cellStats.p, sizeof st, hipMemcpyDeviceToHost) == hipSuccess && st[0])
      std::fprintf(stderr, "[rc cell stats] wavefront-cells %llu, with no positive lane %llu (%.4f %%), positive lane-cells %.2f %%\n", st[0], st[1],
                   100.0 * st[1] / st[0], 100.0 * st[2] / (64.0 * st[0])),
      std::fprintf(stderr, "[rc cell stats] before the row's first event %.1f %%, event codons %.1f %%, after its last event %.1f %%, between events %.1f %%\n",
                   100.0 * st[3] / st[0], 100.0 * st[4] / st[0], 100.0 * st[5] / st[0], 100.0 * (st[0] - st[3] - st[4] - st[5]) / st[0]);
  }
  if (c->d_U) (void)hipFree(c->d_U);
  for (hipStream_t st : c->compute) if (st) (void)hipStreamDestroy(st);
  if (c->native) (void)hipStreamDestroy(c->native);
  if (c->copy) (void)hipStreamDestroy(c->copy);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  if (c->tree) (void)hipStreamDestroy(c->tree);
  for (hipStream_t st : c->classStreams) (void)hipStreamDestroy(st);
  delete c;
}

int rc_ctx_create(int device, rc_ctx **out) {
  if (!out) return fail(RC_ERR_ARG, "null out pointer");
  *out = nullptr;
  want_hw_queues();
  int n = 0;
  trace("ctx: start", nullptr);
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(RC_ERR_DEVICE, "no HIP device available: the scoring path has no CPU fallback");
  trace("ctx: runtime up", nullptr);
  if (device < 0 || device >= n) return fail(RC_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<rc_ctx, void (*)(rc_ctx *)> guard(new rc_ctx(), rc_ctx_destroy);
  rc_ctx *c = guard.get();
  c->device = device;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->numCU = prop.multiProcessorCount;
  c->totalMem = prop.totalGlobalMem;
  c->retiredCapBytes = prop.totalGlobalMem / 16;
  c->bufPool->devCap = prop.totalGlobalMem / 16;
  if (const char *e = std::getenv("RC_BUFFER_POOL")) c->poolEnabled = std::atoi(e) != 0;
  for (DevBuf *d : {&c->treeJobs, &c->treeIn, &c->treeWork, &c->treeRes, &c->d_tables, &c->d_cellStats}) d->retired = c->retired.get();
  for (PinBuf *h : {&c->treeInPin, &c->treeResPin}) h->retired = c->retired.get();
  c->genericScratchWords = std::max<size_t>(static_cast<size_t>(1) << 30, prop.totalGlobalMem / 12 / sizeof(uint32_t));
  c->togetherWords = std::max<size_t>(static_cast<size_t>(1) << 30, prop.totalGlobalMem / 36 / sizeof(uint32_t));   // 8 GB of 288: allocating more costs a short run more than it gains (tools/cli_mixed_sweep.sh)
  if (const char *e = std::getenv("RC_TOGETHER_MAX_MB")) c->togetherWords = static_cast<size_t>(std::max(0ll, std::atoll(e))) * (1u << 20) / sizeof(uint32_t);
  if (const char *e = std::getenv("RC_GRID_CAP_PER_CU")) c->gridCapPerCU = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("RC_SPLIT_FACTOR")) c->splitFactor = std::max(0.0, std::atof(e));
  if (const char *e = std::getenv("RC_NATIVE_WAVES_PER_CU")) c->nativeWavesPerCU = std::max(1, std::min(32, std::atoi(e)));
  if (const char *e = std::getenv("RC_TAIL_SHARING")) c->tailSharing = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_TAIL_RELEASE")) c->tailRelease = std::strcmp(e, "workgroup") != 0;
  if (const char *e = std::getenv("RC_SUFFIX_CACHE")) c->suffixCache = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_HIGH_OCCUPANCY")) c->highOccupancy = std::max(0, std::min(2, std::atoi(e)));
  if (prop.maxSharedMemoryPerMultiProcessor > 0) c->ldsPerCU = prop.maxSharedMemoryPerMultiProcessor;
  if (const char *e = std::getenv("RC_COPY_RESULTS")) c->copyResults = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_NATIVE_FIRST")) c->nativeFirst = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_DUAL_ROWS")) c->dualRows = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_FAT_STREAM_MIN_ITEMS")) c->fatStreamMinItems = std::max(0ll, std::atoll(e));
  if (const char *e = std::getenv("RC_GENERIC_SCRATCH_MB")) c->genericScratchWords = std::max<size_t>(1, static_cast<size_t>(std::atoll(e))) * (1u << 20) / sizeof(uint32_t);
  // non-blocking streams: no implicit ordering against the legacy default stream that a host framework in
  // the same process (torch, RCCL) may be using; the native stream gets the higher priority so that the
  // runtime puts it on a hardware queue of its own (two plain streams can end up sharing one, which
  // serialises the native-block kernels behind k_null again -- seen under torch.distributed)
  int prioLow = 0, prioHigh = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
  if (const char *e = std::getenv("RC_COMPUTE_STREAMS")) c->computeStreams = std::max(1, std::min(rc_ctx::kComputeStreams, std::atoi(e)));
  // The streams are created when they are first used (ctx_stream): a HIP stream with a hardware queue of its own costs 10 ms to create,
  // six of them were 60 ms of every context -- a fifth of the stand-alone driver's run on a 10 000-block input -- and a run of fat batches
  // (rc_api.cpp, fat_class) never touches the native-block stream, a single batch never the second compute stream.
  c->prioLow = prioLow; c->prioHigh = prioHigh;
  c->classPrio = 0;
  if (const char *e = std::getenv("RC_STREAM_PRIO")) {   // experiment: "flat" every stream at the default priority, "classlow" the class streams with the compute streams
    if (!std::strcmp(e, "flat")) c->prioLow = c->prioHigh = c->classPrio = 0;
    else if (!std::strcmp(e, "classlow")) c->classPrio = prioLow;
    else if (!std::strcmp(e, "two")) { c->prioHigh = 0; c->classPrio = prioLow; }   // two levels: compute + class low, the rest default
  }
  trace("ctx: streams", nullptr);
  if (const char *e = std::getenv("RC_DUAL_FROM_L2")) c->dualFromL2 = std::atoi(e);
  if (const char *e = std::getenv("RC_GENERIC_SPLIT")) c->genericSplit = std::atoi(e);
  if (const char *e = std::getenv("RC_NATIVE_SCAN_LANES")) c->nativeScanLanes = std::atoi(e);
  c->serialNative = std::getenv("RC_SERIAL_NATIVE") ? 1 : 0;   // profiling: one kernel at a time, clean per-kernel intervals
  c->stopTwoRoundsMinItems = 4LL * c->numCU * 16;   // the chip filled four times over
  if (const char *e = std::getenv("RC_STOP_MIN_ITEMS")) c->stopTwoRoundsMinItems = std::atoll(e);
  if (const char *e = std::getenv("RC_GENERIC_MIN_ROWS")) c->genericMinRows = std::max(3, std::min(kTemplRows + 1, std::atoi(e)));
  if (const char *e = std::getenv("RC_STOP_ROUNDS")) c->stopRounds = std::max(2, std::min(kMaxRounds, std::atoi(e)));
  c->pt62.build(c->ct62);
  c->pt90.build(c->ct90);
  {   // constant tables: genetic code, then per matrix: pair table, BLOSUM, code infos
    const size_t per = 4096 + 400 * sizeof(int) + 64 * sizeof(CodeInfo);
    std::vector<uint8_t> h(64 * sizeof(int) + 2 * per);
    int *pep = reinterpret_cast<int *>(h.data());
    for (int i = 0; i < 64; i++) pep[i] = c->ct62.pep[i];
    for (int w = 0; w < 2; w++) {
      const PairTable &pt = w ? c->pt90 : c->pt62;
      const CodeTables &ct = w ? c->ct90 : c->ct62;
      uint8_t *m = h.data() + 64 * sizeof(int) + w * per;
      std::memcpy(m, pt.pair, 4096);
      int *bl = reinterpret_cast<int *>(m + 4096);
      for (int p = 0; p < 20; p++) for (int q = 0; q < 20; q++) bl[20 * p + q] = ct.blosum[p][q];
      std::memcpy(m + 4096 + 400 * sizeof(int), pt.info, 64 * sizeof(CodeInfo));
    }
    HIP_TRY(c->d_tables.ensure(h.size()));
    HIP_TRY(hipMemcpy(c->d_tables.p, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  trace("ctx: tables", nullptr);
  c->tieThr = float_threshold_lt(0.0001);
  if (const char *e = std::getenv("RC_LDS_MAX_BYTES")) { c->ldsMaxBytes = c->ldsMaxBytesOneRow = static_cast<size_t>(std::atol(e)); c->stageManyRows = true; }
  if (kProfiling) if (const char *e = std::getenv("RC_DEBUG_SKIP")) c->debugSkip = std::atoi(e);   // RC_PROFILING builds only: results are wrong
  if (kProfiling && std::getenv("RC_CELL_STATS")) {
    HIP_TRY(c->d_cellStats.ensure(6 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_cellStats.p, 0, 6 * sizeof(unsigned long long)));
  }
  c->hostThreads = std::min(effective_cpus(), 32);
  if (const char *e = std::getenv("RC_HOST_THREADS")) c->hostThreads = std::max(1, std::atoi(e));
  *out = guard.release();
  return RC_OK;
}

int rc_ctx_set_host_threads(rc_ctx *c, int32_t threads) {
  if (!c || threads < 1) return fail(RC_ERR_ARG, "bad argument");
  if (c->inflight.load() != 0) return fail(RC_ERR_ARG, "batches are in flight");
  c->hostThreads = threads;
  c->pool.reset();
  return RC_OK;
}

int rc_ctx_host_threads(const rc_ctx *c) { return c ? c->hostThreads : 0; }

void rc_batch_destroy(rc_batch *b) {
  if (!b) return;
  trace("destroy", b);
  (void)hipSetDevice(b->device);
  const bool ctxAlive = b->ctx && !(b->retiredKeep && b->retiredKeep->ctxGone.load());
  if (b->state == rc_batch::RUNNING) { if (b->evDone.e) (void)hipEventSynchronize(b->evDone); if (ctxAlive) b->ctx->inflight.fetch_sub(1); }
  else if (b->state == rc_batch::READY && b->evPrep.e) (void)hipEventSynchronize(b->evPrep);   // the upload reads the pinned blob
  delete b;
  trace("destroyed", b);
}

// ------------------------------------------------------------------------------------------ prepare + upload

static int batch_prepare(rc_batch *b, const rc_block *blocks, int32_t n_blocks, const rc_params *par) {
  rc_ctx *ctx = b->ctx;
  if (par->blosum != 62 && par->blosum != 90) return fail(RC_ERR_ARG, "blosum must be 62 or 90");
  if (par->sampleN < 1) return fail(RC_ERR_ARG, "sampleN must be >= 1");
  const auto t0 = std::chrono::steady_clock::now();
  b->par = *par; b->n = n_blocks;
  b->state = rc_batch::EMPTY;
  b->meta.assign(n_blocks, BlockMeta());
  b->db.resize(n_blocks);
  b->errs.clear(); b->okBlocks.clear(); b->classes.clear();
  b->maxL = 0; b->maxDraws = 0; b->hssCap = 8;
  b->allExact = !params_in_fast_range(*par);
  const PairTable &pt = (par->blosum == 90) ? ctx->pt90 : ctx->pt62;

  // capacity of the two blob parts from the shapes alone, then one pinned buffer
  size_t hostSum = 0, devSum = 0;
  for (int i = 0; i < n_blocks; i++) {
    size_t h = 0, d = 0;
    if (blocks[i].n_rows >= 3 && blocks[i].n_rows <= kMaxRows && blocks[i].n_cols >= 3 && blocks[i].n_cols <= RC_MAX_COLS)
      block_footprint(blocks[i].n_rows, blocks[i].n_cols, &h, &d);
    hostSum += h; devSum += d;
  }
  const size_t nn = static_cast<size_t>(std::max(n_blocks, 1));
  const size_t tail = al256(nn * sizeof(DevBlock)) + 2 * al256(nn * sizeof(int)) + al256(nn * sizeof(uint64_t)) + al256((nn + 1) * sizeof(int));
  b->hostCap = al256(hostSum) + tail + 256;
  b->devCap = al256(devSum) + 256;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(b->hblob.ensure(b->hostCap));
  BlobArena arena;
  arena.host = b->hblob.as<uint8_t>();
  arena.hostCap = b->hostCap;          // the device-only arena starts here in the device blob
  arena.devCap = b->devCap;

  // block-constant preparation, blocks are independent -> host threads
  {
    if (!ctx->pool || ctx->pool->size() != ctx->hostThreads) ctx->pool.reset(new Pool(ctx->hostThreads));
    std::mutex errMu;
    ctx->pool->run(n_blocks, [&](int i) {
      std::string err;
      const int st = prepare_block(blocks[i], *par, pt, arena, static_cast<uint32_t>(i), b->db[i], b->meta[i], err);
      if (st != RC_OK && st != RC_ERR_SKIP) {
        std::lock_guard<std::mutex> lk(errMu);
        b->errs[i] = err;
      }
    });
  }

  // lists and per-block result areas
  bool sameL = true;
  const size_t SpadBytes = (static_cast<size_t>(par->sampleN) + kWave - 1) / kWave * kWave * sizeof(uint32_t);
  for (int i = 0; i < n_blocks; i++) {
    BlockMeta &m = b->meta[i];
    if (m.status != RC_OK) continue;
    // The MT19937 streams of a batch are U[nodes x cols][samples padded to 64] (4 bytes each), one array for the batch's largest
    // block: a block whose array would not fit a quarter of the device memory (288 GB: nodes x cols x samples > 1.8e10, e.g.
    // 500 rows x 18 000 columns at n = 1000) gets a status of its own instead of failing the whole batch's allocation.
    if (static_cast<double>(b->db[i].nnodes) * m.cols * static_cast<double>(SpadBytes) > static_cast<double>(ctx->totalMem) / 4.0) {
      m.status = RC_ERR_UNSUPPORTED; b->db[i].status = RC_ERR_UNSUPPORTED;
      b->errs[i] = "rows x columns x samples too large for the device: the block's random-number streams (4 bytes x tree nodes x columns x samples) exceed a quarter of its memory; split the block (breakmaf) or lower --num-samples";
      continue;
    }
    b->okBlocks.push_back(i);
    auto &cls = b->classes[m.N < ctx->genericMinRows ? m.NK : kGenericClass];
    if (!cls.empty() && b->meta[cls.front()].L != m.L) sameL = false;
    cls.push_back(i);
    b->maxL = std::max(b->maxL, m.L);
    b->maxDraws = std::max(b->maxDraws, b->db[i].nnodes * m.cols);
    const size_t smax = static_cast<size_t>(m.L) / 3;
    b->hssCap = std::max(b->hssCap, static_cast<int>(smax) / 3 + 2);
  }
  // longest blocks first: k_null's workgroups pull items from a shared counter, so the order of the
  // list is the order of dispatch (longest-processing-time-first keeps the tail short)
  if (!sameL)
    for (auto &kv : b->classes) std::stable_sort(kv.second.begin(), kv.second.end(), [&](int x, int y) { return b->meta[x].L > b->meta[y].L; });

  // tail of the host part: headers and lists, so that ONE copy moves everything
  uint8_t *base = arena.host;
  size_t o = al256(arena.hostUsed.load());
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  b->oDblocks = take(nn * sizeof(DevBlock));
  b->oOk = take(nn * sizeof(int));
  b->oClass = take(nn * sizeof(int));
  b->oPrefix = take((nn + 1) * sizeof(int));
  if (o > b->hostCap) return fail(RC_ERR_ARG, "internal: blob tail does not fit");
  b->hostUsed = o;
  if (n_blocks > 0) std::memcpy(base + b->oDblocks, b->db.data(), sizeof(DevBlock) * n_blocks);
  if (!b->okBlocks.empty()) std::memcpy(base + b->oOk, b->okBlocks.data(), sizeof(int) * b->okBlocks.size());
  {
    int *cl = reinterpret_cast<int *>(base + b->oClass);
    int at = 0;
    std::fill(b->classOff, b->classOff + kClassSlots, 0);
    for (auto &kv : b->classes) {
      b->classOff[kv.first] = at;
      std::memcpy(cl + at, kv.second.data(), sizeof(int) * kv.second.size());
      at += static_cast<int>(kv.second.size());
    }
    int *pf = reinterpret_cast<int *>(base + b->oPrefix);
    int run = 0;
    for (size_t k = 0; k < b->okBlocks.size(); k++) { pf[k] = run; run += 2 * b->meta[b->okBlocks[k]].N; }
    pf[b->okBlocks.size()] = run;
    b->nModels = run;
  }
  b->prepHostMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return RC_OK;
}

// one asynchronous copy of the host part, then the kernels that derive the expected-score tables
static int batch_upload(rc_batch *b) {
  rc_ctx *c = b->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int n = b->n;
  const size_t nn = static_cast<size_t>(std::max(n, 1));
  HIP_TRY(b->dblob.ensure(b->hostCap + b->devCap));
  if (!b->maximaExternal) { HIP_TRY(b->dmaxima.ensure(nn * b->par.sampleN * sizeof(float))); b->maxPtr = b->dmaxima.as<float>(); }
  HIP_TRY(b->dhss.ensure(nn * 6 * b->hssCap * sizeof(DevHss)));
  HIP_TRY(b->dhssCount.ensure(nn * 6 * sizeof(int)));
  HIP_TRY(b->dhssPacked.ensure(nn * 6 * b->hssCap * sizeof(DevHss)));
  HIP_TRY(b->dhssOffsets.ensure((nn * 6 + 1) * sizeof(int)));
  HIP_TRY(b->dfit.ensure(nn * sizeof(FitOut)));
  HIP_TRY(b->dflags.ensure(nn * sizeof(uint32_t)));
  HIP_TRY(b->dexact.ensure(nn * sizeof(int)));
  HIP_TRY(b->dcounters.ensure(kCntTotal * sizeof(uint32_t)));
  // copy and table kernels share one stream of their own: batch i+1's tables are made while batch i is being scored
  hipStream_t ps = stream_copy(c);
  if (!ps) return fail(RC_ERR_DEVICE, "hipStreamCreate failed");
  HIP_TRY(hipMemcpyAsync(b->dblob.p, b->hblob.p, b->hostUsed, hipMemcpyHostToDevice, ps));
  HIP_TRY(b->evUp.record(ps));
  HIP_TRY(b->evPrep0.record(ps));
  HIP_TRY(hipMemsetAsync(b->dflags.p, 0, nn * sizeof(uint32_t), ps));
  HIP_TRY(hipMemsetAsync(b->dcounters.p, 0, kCntTotal * sizeof(uint32_t), ps));
  if (!b->okBlocks.empty()) {
    const TablePtrs tp = table_ptrs(c, b->par.blosum);
    PrepArgs pa{};
    uint8_t *blob = b->dblob.as<uint8_t>();
    pa.blob = blob;
    pa.dblocks = reinterpret_cast<const DevBlock *>(blob + b->oDblocks);
    pa.blocks = reinterpret_cast<const int *>(blob + b->oOk);
    pa.nBlocks = static_cast<int>(b->okBlocks.size());
    pa.modelPrefix = reinterpret_cast<const int *>(blob + b->oPrefix);
    pa.nModels = b->nModels;
    pa.pep = tp.pep; pa.blosum = tp.blosum; pa.codeInfo = tp.info;
    pa.flags = b->dflags.as<uint32_t>();
    pa.exactList = b->dexact.as<int>();
    pa.exactCount = b->dcounters.as<int>() + kCntExact;
    std::memcpy(pa.classOff, b->classOff, sizeof pa.classOff);
    pa.genericMinRows = c->genericMinRows;
    launch_prep(pa, ps);
  }
  HIP_TRY(b->evPrep.record(ps));
  HIP_TRY(hipGetLastError());
  b->state = rc_batch::READY;
  b->maximaFetched = false;
  return RC_OK;
}

int rc_batch_create(rc_ctx *ctx, const rc_block *blocks, int32_t n_blocks, const rc_params *par, rc_batch **out) {
  if (!ctx || (!blocks && n_blocks > 0) || !par || !out || n_blocks < 0) return fail(RC_ERR_ARG, "bad argument");
  *out = nullptr;
  std::unique_ptr<rc_batch, void (*)(rc_batch *)> guard(new rc_batch(), rc_batch_destroy);   // released on success only
  rc_batch *b = guard.get();
  b->adopt(ctx);
  trace("create", b);
  RC_TRY(batch_prepare(b, blocks, n_blocks, par));
  trace("prepared", b);
  RC_TRY(batch_upload(b));
  trace("uploaded", b);
  *out = guard.release();
  return RC_OK;
}

// ------------------------------------------------------------------------------------------ run

// When set, the MT19937 stream is regenerated on every run (bench: the stream belongs to a step).
static bool g_no_stream_cache = false;
void rc_set_stream_cache(int enabled) { g_no_stream_cache = !enabled; }

// U[draw][sample] for seeds seed .. seed + Spad - 1, at least D draws; generated on `st`, other streams order
// themselves behind U_ready
static int ensure_mt_stream(rc_ctx *c, rc_batch *b, hipStream_t st, uint32_t seed, int Spad, int D) {
  b->mtLaunched = false;
  if (c->U_valid && c->U_seed == seed && c->U_Spad == Spad && c->U_D >= D) {
    if (c->U_ready.e) HIP_TRY(hipStreamWaitEvent(st, c->U_ready, 0));
    return RC_OK;
  }
  if (c->inflight.load() > 0) HIP_TRY(hipDeviceSynchronize());   // other batches still read the old stream (rare: a batch with more draws)
  const size_t need = static_cast<size_t>(Spad) * D;
  if (need > c->U_cap) {
    if (c->d_U) (void)hipFree(c->d_U);
    c->d_U = nullptr; c->U_cap = 0;
    HIP_TRY(hipMalloc(&c->d_U, need * sizeof(uint32_t)));
    c->U_cap = need;
  }
  HIP_TRY(b->evMt0.record(st));
  launch_mt_stream(seed, Spad, D, c->d_U, st);
  HIP_TRY(b->evMt1.record(st));
  HIP_TRY(c->U_ready.record(st));
  b->mtLaunched = true;
  c->U_seed = seed; c->U_Spad = Spad; c->U_D = D; c->U_valid = true;
  return RC_OK;
}

namespace {
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
}  // namespace

// persistent workgroups of k_native_dp for `items` (block, strand x frame) items whose longest block has smax codons: each keeps 64 rows
// of S (64 * smax floats), all of them together at most 2 GiB (very long blocks run on fewer workgroups)
// mode 0: beside a k_null that leaves registers free on every SIMD (RC_NATIVE_WAVES_PER_CU, 8); 1: with the chip to itself, every wavefront slot
static size_t native_grid(const rc_ctx *c, size_t items, int smax, int mode = 0) {
  const size_t perGroup = static_cast<size_t>(kWave) * std::max(1, smax) * sizeof(float);
  const size_t byMemory = std::max<size_t>(1, (static_cast<size_t>(2) << 30) / perGroup);
  const size_t perCU = mode == 1 ? 32 : static_cast<size_t>(c->nativeWavesPerCU);
  return std::max<size_t>(1, std::min(std::min(items, perCU * c->numCU), byMemory));
}

// A k_null launch that walks two rows per pass (rc_null_kernel.h; N-1 <= kDualRowsMaxNK, codes staged in LDS) needs up to 127 VGPRs:
// four such wavefronts take a SIMD's whole register file, and any other kernel's wavefront on that SIMD displaces one of them (measured:
// the native-block kernels beside it, even one workgroup per CU, 42.5 -> 50.3 ms; an LDS request sized for 15 workgroups per CU did
// not help).  So the two-row instantiations are used where nothing has to run beside them: in batches made of such classes only
// ("fat": every class 3 <= N-1 <= 5 with LDS staging; N-1 = 2 needs 95 VGPRs, leaves a quarter of the registers free and is always
// two-row), whose native-block kernels are queued in front of k_null on the same stream with every wavefront slot to themselves
// (1.9 ms instead of 2.9 at C3) -- resident batches, and sub-batches of a stream that are long enough for the bubble this leaves at
// every sub-batch boundary to be paid back (RC_FAT_STREAM_MIN_ITEMS).  Everything else keeps the one-row instantiations (74 VGPRs)
// and the small kernels beside them.
static bool fat_class(const rc_batch *b, const rc_ctx *c, int NK, int maxL) {
  if (NK < 3 || NK > kDualRowsMaxNK || b->allExact) return false;
  const size_t lds = static_cast<size_t>(maxL / 3) * ((NK + 4) / 5) * kWave * sizeof(uint32_t);
  return lds <= c->ldsMaxBytes;   // (longer blocks: the two-row kernel from L2 gains 3-4 % on the launch and loses it again to the native-block kernels queued in front)
}

static int launch_native_block(const RunEnv &R) {
  rc_batch *b = R.b; rc_ctx *c = R.c;
  // ---- native block: sigma -> DP per N-1 class -> scan, on the native stream: these kernels are small and
  // latency-bound, the null sampling fills the chip beside them; the fit waits for both
  hipStream_t st = R.nativeFirst ? R.cs : stream_native(c);
  if (!st) return fail(RC_ERR_DEVICE, "hipStreamCreate failed");
  if (!R.nativeFirst) HIP_TRY(hipStreamWaitEvent(st, b->evStart, 0));
  HIP_TRY(b->evN0.record(st));
  NativeArgs na{};
  na.blob = b->dblob.as<uint8_t>(); na.dblocks = R.dblocks; na.blocks = R.okList; na.pair = R.tp.pair;
  na.hss = b->dhss.as<DevHss>(); na.hssCount = b->dhssCount.as<int>(); na.hssCap = b->hssCap; na.tieThr = c->tieThr;
  HIP_TRY(hipMemsetAsync(b->dhssCount.p, 0, static_cast<size_t>(R.n) * 6 * sizeof(int), st));
  launch_native_sigma(na, static_cast<int>(b->okBlocks.size()), st); b->nl[3]++;
  {   // one 64-row buffer per persistent workgroup, shared by the class launches (same stream)
    size_t need = 0;
    for (auto &kv : b->classes) {
      if (kv.first == kGenericClass) continue;
      int smaxClass = 1;
      for (int bi : kv.second) smaxClass = std::max(smaxClass, b->meta[bi].L / 3);
      const size_t grid = native_grid(c, kv.second.size() * 6, smaxClass, R.nativeMode);
      need = std::max(need, grid * kWave * smaxClass);
    }
    HIP_TRY(b->dnativeTile.ensure(std::max<size_t>(need, 4) * sizeof(float)));
  }
  for (auto &kv : b->classes) {
    NativeArgs nc = na;
    nc.blocks = R.classList + b->classOff[kv.first];
    if (kv.first == kGenericClass) {   // more than 64 rows: states in a scratch, at most 256 blocks per launch share it
      int maxNK = 0, smaxG = 1;
      for (int bi : kv.second) { maxNK = std::max(maxNK, b->meta[bi].NK); smaxG = std::max(smaxG, b->meta[bi].L / 3); }
      const size_t stride = static_cast<size_t>(3) * maxNK * kWave + static_cast<size_t>(kWave) * smaxG;   // states, then 64 rows of S
      const int chunk = 256;
      HIP_TRY(b->dnativeScratch.ensure(stride * 6 * std::min<size_t>(chunk, kv.second.size()) * sizeof(float)));
      for (size_t at = 0; at < kv.second.size(); at += chunk) {
        NativeArgs ng = nc;
        ng.blocks = nc.blocks + at;
        launch_native_dp_generic(ng, static_cast<int>(std::min<size_t>(chunk, kv.second.size() - at)), b->dnativeScratch.as<float>(), stride, st);
        b->nl[3]++;
      }
      continue;
    }
    // up to 64 rows: DP and getHSS fused, 64 rows of one matrix at a time through a per-workgroup buffer (persistent grid)
    int smaxClass = 1;
    for (int bi : kv.second) smaxClass = std::max(smaxClass, b->meta[bi].L / 3);
    nc.nItems = static_cast<int>(kv.second.size()) * 6;
    nc.tileStride = static_cast<size_t>(kWave) * smaxClass;
    const int grid = static_cast<int>(native_grid(c, static_cast<size_t>(nc.nItems), smaxClass, R.nativeMode));
    nc.tile = b->dnativeTile.as<float>();
    // every matrix of the class kept for a scan with one lane per matrix, where that fits in a sixteenth of the device memory
    // (headline: 10 000 blocks x 6 x 40 x 40 floats = 384 MB); otherwise DP and scan fused, 64 rows at a time
    const size_t allFloats = static_cast<size_t>(nc.nItems) * smaxClass * smaxClass;
    if (c->nativeScanLanes && allFloats * sizeof(float) <= c->totalMem / 16 && nc.nItems >= 6 * 64 && (kv.first <= 16 || kv.first > 32)) {   // (17..32: see k_native_dp)
      HIP_TRY(b->dnativeAll.ensure(std::max<size_t>(allFloats, 4) * sizeof(float)));
      nc.sAll = b->dnativeAll.as<float>(); nc.sAllSites = smaxClass;
    }
    if (!launch_native_dp(kv.first, nc, grid, st))
      return fail(RC_ERR_UNSUPPORTED, "no native DP kernel for this number of rows");
    b->nl[3]++;
    if (nc.sAll) { launch_native_scan(nc, st); b->nl[3]++; }
  }
  HIP_TRY(hipMemsetAsync(b->dhssOffsets.as<int>() + R.slots, 0, sizeof(int), st));
  launch_hss_pack(b->dhss.as<DevHss>(), b->dhssCount.as<int>(), b->hssCap, R.slots, b->dhssPacked.as<DevHss>(), b->dhssOffsets.as<int>(),
                  b->dhssOffsets.as<int>() + R.slots, st);
  HIP_TRY(b->evN1.record(st));
  return RC_OK;
}

// null sampling of the sample groups [gLo, gHi) for every class; phase 0 or 1 selects the work counters
// sizeOnly: no launch, *sizeOnly = uint32 of staging scratch this round needs (the rounds of a run share one allocation made up front)
// tail sharing of k_null (rc_null_kernel.h): per launch one claim word and a (block, group) pair per workgroup
static size_t steal_slots(const rc_ctx *c) { return static_cast<size_t>(c->numCU) * 32; }
static size_t steal_words(const rc_ctx *c) { return 16 + 4 * steal_slots(c); }   // header, claim words, (block, group) pairs, list of published slots

// Launch shape of a k_null that reads its code words from L2 (rc_null_kernel.h): which build -- k_null_occ, one more wavefront per SIMD,
// for batches of one row-count class: round 3 measured +1..7 % there and -5 % on a stream of ten classes, whose small kernels ran in the
// registers those wavefronts take --, how many workgroups per CU, and how much LDS each gets: what phase A needs at least, and with the
// suffix cache everything the occupancy leaves (160 KB / workgroups per CU), which phase B fills with the most re-read code words.
static rc_ctx::PlainPlan plain_plan(rc_ctx *c, int NK, int maxNodes, int wantHi) {   // wantHi: 1 the k_null_occ build, 2 the two-row kernel
  const auto key = std::make_tuple(NK, maxNodes, wantHi);
  auto it = c->plainPlans.find(key);
  if (it != c->plainPlans.end()) return it->second;
  rc_ctx::PlainPlan pp{0, 0, 0};
  for (int hi = wantHi; hi >= 0 && pp.occ <= 0; hi = hi == 2 ? 0 : hi - 1) {
    const size_t minLds = hi ? al256(static_cast<size_t>(maxNodes) * 64 + 64 * 64) : static_cast<size_t>(kPhaseALds);
    const int occ = null_occupancy_plain(NK, minLds, hi);
    if (occ <= 0) continue;
    size_t lds = minLds;
    if (c->suffixCache) {
      size_t budget = (c->ldsPerCU / static_cast<size_t>(occ)) & ~static_cast<size_t>(255);
      while (budget > minLds && null_occupancy_plain(NK, budget, hi) < occ) budget -= 256;
      lds = std::max(budget, minLds);
    }
    pp = rc_ctx::PlainPlan{occ, lds, hi};
  }
  c->plainPlans[key] = pp;
  return pp;
}

static int launch_null_groups(const RunEnv &R, int gLo, int gHi, int phase, uint32_t extraSkip, size_t *sizeOnly = nullptr) {
  rc_batch *b = R.b; rc_ctx *c = R.c;
  const int groupsHere = gHi - gLo;
  struct Plan { int NK, grid, gridExact, comboSplit; size_t stride, ldsBytes, need; double itemCost; int hiOcc, cacheSites; size_t plainLds;
                int simGrid, roundBlocks; size_t codesBytes, stateBytes; };   // (the last four: the wide-block class in two launches)
  std::vector<Plan> plans;
  size_t needSum = 0, needMax = 0;
  // Several row-count classes: first planned as launches that run side by side (no strand x frame split: together they fill the chip);
  // if their staging areas do not fit side by side, planned again as one launch after the other, each filling the chip by itself.
  bool together = b->classes.size() > 1 && !c->serialNative;
  // ... unless all of them together are too few to fill the chip: then every item is split into its six parts, as for a single small class
  const bool splitAll = static_cast<double>(b->okBlocks.size()) * groupsHere <= c->splitFactor * c->numCU * 16;
 plan_again:
  plans.clear(); needSum = 0; needMax = 0;
  for (auto &kv : b->classes) {
    Plan p{};
    p.NK = kv.first;
    int maxLc = 0, maxNKc = 0;
    for (int bi : kv.second) { maxLc = std::max(maxLc, b->meta[bi].L); maxNKc = std::max(maxNKc, b->meta[bi].NK); }
    p.itemCost = static_cast<double>(maxNKc) * maxLc * maxLc;   // (cell, sequence) steps of the class's longest item, up to a factor
    if (kv.first == kGenericClass) {   // k_null_generic: one launch for every block of more than 64 rows, whatever its N
      int maxN = 0, maxNodes = 0;
      for (int bi : kv.second) { maxN = std::max(maxN, b->meta[bi].N); maxNodes = std::max(maxNodes, b->db[bi].nnodes); }
      p.stride = (null_generic_scratch_bytes(maxN, maxLc, maxNodes) + 3) / 4;   // in uint32 like the others
      p.ldsBytes = null_generic_lds_bytes(maxN, maxNodes);
      const std::pair<int, size_t> key(kGenericClass, p.ldsBytes);
      if (!c->occ.count(key)) c->occ[key] = null_generic_occupancy(p.ldsBytes);
      const int occ = std::max(1, c->occ[key]);
      long long items = static_cast<long long>(kv.second.size()) * groupsHere;
      p.comboSplit = (together ? splitAll : static_cast<double>(items) <= c->splitFactor * c->numCU * occ) ? 1 : 0;
      if (p.comboSplit) items *= 6;
      p.grid = static_cast<int>(std::min<long long>(items, static_cast<long long>(c->numCU) * occ));
      // every wavefront keeps the sigma codes of its 64 samples (N x L x 2 bytes per sample): the scratch, not the registers, bounds the occupancy
      p.grid = static_cast<int>(std::max<size_t>(1, std::min<size_t>(p.grid, c->genericScratchWords / std::max<size_t>(p.stride, 1))));
      p.need = p.stride * p.grid;
      if (c->genericSplit && p.comboSplit && static_cast<size_t>(kv.second.size()) * groupsHere * null_generic_codes_bytes(maxN, maxLc, maxNodes) <= c->genericScratchWords * sizeof(uint32_t)) {
        // Few items (every item split into its six strand x frame parts to fill the chip): two launches -- k_generic_sim leaves the
        // codes of every (block, sample group) item in the scratch, k_generic_dp's six parts of an item read them there, instead of
        // each part simulating the item again (16 blocks of 100 x 300: 67.7 -> 53.1 ms).  Not for large batches: the simulation's
        // LDS (node states, codon windows) allows no more wavefronts per CU than the DP's registers do, so nothing is gained, and the
        // codes of all items at once (5 MB each) would have to go through the scratch in rounds (1024 blocks: 1.03 -> 1.33 s).
        p.codesBytes = null_generic_codes_bytes(maxN, maxLc, maxNodes);
        p.stateBytes = null_generic_state_bytes(maxN, maxLc, maxNodes);
        const int occD = std::max(1, generic_dp_occupancy(p.ldsBytes)), occS = std::max(1, generic_sim_occupancy(p.ldsBytes));
        long long itemsB = static_cast<long long>(kv.second.size()) * groupsHere;
        const size_t budget = c->genericScratchWords * sizeof(uint32_t);
        const size_t slots = std::max<size_t>(static_cast<size_t>(groupsHere), budget / p.codesBytes);
        p.roundBlocks = static_cast<int>(std::max<size_t>(1, std::min<size_t>(kv.second.size(), slots / static_cast<size_t>(groupsHere))));
        const long long roundItems = static_cast<long long>(p.roundBlocks) * groupsHere;
        (void)itemsB;   // (comboSplit stays as planned above: it is what brought us here)
        p.grid = static_cast<int>(std::min<long long>(roundItems * (p.comboSplit ? 6 : 1), static_cast<long long>(c->numCU) * occD));
        p.simGrid = static_cast<int>(std::min<long long>(roundItems, static_cast<long long>(c->numCU) * occS));
        p.need = (static_cast<size_t>(roundItems) * p.codesBytes + static_cast<size_t>(p.grid) * p.stateBytes + 3) / 4 + 64;
      }
      needSum += p.need; needMax = std::max(needMax, p.need);
      plans.push_back(p);
      continue;
    }
    const int NCW = (p.NK + 4) / 5;   // 5 six-bit sigma codes per word
    p.stride = static_cast<size_t>(2) * (maxLc + 1) * code_pos_words(p.NK);   // both strands, positions 0..L, [word][lane] with a narrow last word (rc_device.h)
    // sigma codes of one strand x frame are staged in LDS when that still leaves >= 12 wavefronts per
    // CU (RC_LDS_MAX_BYTES overrides the per-wavefront budget); otherwise they are read from the
    // per-workgroup scratch in global memory.
    p.ldsBytes = static_cast<size_t>(maxLc / 3) * NCW * kWave * sizeof(uint32_t);
    const bool exactOnly = b->allExact || p.NK >= kFastRows;   // wide blocks (N > 32) only have the EXACT instantiation
    // The two-row kernel with its codes staged in LDS runs 160 KB / staged bytes workgroups per CU, and its time goes almost with
    // the inverse of that number (6 / 8 / 10 / 12 / 16 per CU: 93.5 / 71.6 / 59.6 / 51.6 / 42.5 ms at the headline's shape).  Where
    // staging leaves twelve or fewer (blocks of more than ~135 columns), the same kernel reading its codes from L2 behind a suffix
    // cache runs sixteen and is faster: 6 rows x 150 columns 77.3 -> 71.7 ms; at 120 columns (fifteen staged) it is slower, 42.6 -> 47.0.
    bool dualL2 = false;
    if (c->dualFromL2 && c->dualRows && !exactOnly && p.NK >= 3 && p.NK <= kDualRowsMaxNK && R.fat && p.ldsBytes <= c->ldsMaxBytes)
      dualL2 = c->dualFromL2 == 2 || c->ldsPerCU / std::max<size_t>(p.ldsBytes, kPhaseALds) <= 12;
    const bool twoRows = c->dualRows && p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat);   // (the launch's own test, below)
    if (p.ldsBytes > (twoRows ? c->ldsMaxBytes : std::min(c->ldsMaxBytes, c->ldsMaxBytesOneRow)) || exactOnly || dualL2 || (p.NK > kDualRowsMaxNK && !c->stageManyRows)) p.ldsBytes = 0;
    else p.ldsBytes = std::max<size_t>(p.ldsBytes, kPhaseALds);   // phase A keeps the tree's threshold table and the pair table there
    const std::pair<int, size_t> key(p.NK, p.ldsBytes);
    if (!c->occ.count(key)) c->occ[key] = null_occupancy(p.NK, p.ldsBytes);
    if (c->occ[key] <= 0 && p.ldsBytes) { p.ldsBytes = 0; const std::pair<int, size_t> k2(p.NK, 0); if (!c->occ.count(k2)) c->occ[k2] = null_occupancy(p.NK, 0); }
    int occ = std::max(1, c->occ[std::pair<int, size_t>(p.NK, p.ldsBytes)]);
    if (!p.ldsBytes && !exactOnly) {   // codes from L2
      int maxNodes = 0;
      for (int bi : kv.second) maxNodes = std::max(maxNodes, b->db[bi].nnodes);
      const int wantHi = dualL2 ? 2 : (c->highOccupancy == 2 || (c->highOccupancy == 1 && b->classes.size() == 1)) ? 1 : 0;
      const rc_ctx::PlainPlan pp = plain_plan(c, p.NK, maxNodes, wantHi);
      if (pp.occ > 0) {
        occ = pp.occ; p.hiOcc = pp.hiOcc; p.plainLds = pp.lds;
        // (the first NK x 256 bytes of that LDS hold the sigma tables during the DP: the look-ups of these launches read them there)
        const size_t tables = static_cast<size_t>(p.NK) * kWave * sizeof(float);
        p.cacheSites = c->suffixCache && pp.lds > tables ? static_cast<int>((pp.lds - tables) / (static_cast<size_t>(code_pos_words(p.NK)) * sizeof(uint32_t))) : 0;
      }
    }
    if (c->gridCapPerCU > 0) occ = std::min(occ, c->gridCapPerCU);
    int items = static_cast<int>(kv.second.size()) * groupsHere;
    // small batches: split every item into its six strand x frame parts to fill the chip
    p.comboSplit = (together ? splitAll : static_cast<double>(items) <= c->splitFactor * c->numCU * occ) ? 1 : 0;
    if (p.comboSplit) items *= 6;
    p.grid = std::min(items, c->numCU * occ);
    // very long blocks: bound the sigma-code staging area (4 GiB of uint32) by running fewer workgroups
    p.grid = static_cast<int>(std::max<size_t>(1, std::min<size_t>(p.grid, (static_cast<size_t>(1) << 30) / std::max<size_t>(p.stride, 1))));
    // blocks flagged by k_prep_lut go through the EXACT instantiation in a second launch on the same stream; their
    // number is only known on the device, the launch is a few idle workgroups when there are none
    p.gridExact = exactOnly ? 0 : std::min(p.grid, c->numCU);
    p.need = p.stride * std::max(p.grid, p.gridExact);
    needSum += p.need; needMax = std::max(needMax, p.need);
    plans.push_back(p);
  }
  // Several row-count classes: their launches are independent (own work queues, own part of the staging
  // scratch), so they go on separate streams and share the chip -- a small class no longer waits for the tail
  // of the previous one.  One class, a scratch that would exceed 4 GiB, or RC_SERIAL_NATIVE: this run's stream only.
  // the classes with the longest items first (longest-processing-time-first across the launches too)
  std::stable_sort(plans.begin(), plans.end(), [](const Plan &x, const Plan &y) { return x.itemCost > y.itemCost; });
  if (together && needSum > c->togetherWords) { together = false; goto plan_again; }
  const bool concurrent = together;
  const size_t need = concurrent ? needSum : needMax;
  if (sizeOnly) { *sizeOnly = need; return RC_OK; }
  if (need * sizeof(uint32_t) > b->dscratch.cap) return fail(RC_ERR_ARG, "internal: staging scratch was not sized for this round");
  Event &fork = phase ? b->evMid : b->evS0;
  size_t scratchOff = 0;
  if (concurrent) while (b->classDone.size() < static_cast<size_t>(kMaxRounds) * plans.size()) b->classDone.emplace_back(new Event());
  for (size_t pi = 0; pi < plans.size(); pi++) {
    const Plan &p = plans[pi];
    const auto &members = b->classes[p.NK];
    hipStream_t st = R.cs;
    if (concurrent) {
      while (c->classStreams.size() < std::min<size_t>(plans.size(), 8)) {
        hipStream_t ns = nullptr;
        HIP_TRY(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, c->classPrio));
        c->classStreams.push_back(ns);
      }
      st = c->classStreams[pi % c->classStreams.size()];
      HIP_TRY(hipStreamWaitEvent(st, fork, 0));   // the memsets / the stop marks
    }
    NullArgs a{};
    a.blob = R.blob; a.dblocks = R.dblocks;
    a.classBlocks = R.classList + b->classOff[p.NK];
    a.nClassBlocks = static_cast<int>(members.size());
    a.flags = b->dflags.as<uint32_t>();
    a.gLo = gLo; a.gHi = gHi; a.sampleN = R.sampleN; a.Spad = R.Spad;
    a.U = c->d_U; a.pair = R.tp.pair; a.scratch = b->dscratch.as<uint32_t>() + scratchOff; a.scratchStride = p.stride;
    a.maxima = b->maxPtr; a.clampCount = reinterpret_cast<unsigned long long *>(b->dcounters.as<uint32_t>() + kCntClamp);
    a.tieThr = c->tieThr; a.debugSkip = c->debugSkip; a.comboSplit = p.comboSplit;
    a.cellStats = c->d_cellStats.as<unsigned long long>();
    unsigned int *work = b->dcounters.as<unsigned int>() + kCntWork + static_cast<size_t>(phase) * 2 * kClassSlots * 8;
    {
      const double cost = p.itemCost * static_cast<double>(members.size()) * groupsHere;
      if (cost > b->nullKernelCost) {
        b->nullKernelCost = cost;
        if (p.NK == kGenericClass) b->nullKernel = "rc::k_null_generic";
        else {
          const bool ex = b->allExact || p.NK >= kFastRows;
          const bool dual = !ex && c->dualRows && p.ldsBytes && p.ldsBytes <= 48 * 1024 && p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat);
          if (!ex && !p.ldsBytes && p.hiOcc == 2) b->nullKernel = "rc::k_null<" + std::to_string(p.NK) + ", false, false, true>";
          else if (!ex && !p.ldsBytes && p.hiOcc) b->nullKernel = "rc::k_null_occ<" + std::to_string(p.NK) + ">";
          else b->nullKernel = "rc::k_null<" + std::to_string(p.NK) + (!ex && p.ldsBytes ? ", true" : ", false") + (ex ? ", true" : ", false") + (dual ? ", true>" : ", false>");
        }
      }
    }
    if (p.NK == kGenericClass) {
      a.skipMask = extraSkip;
      a.scratchStride = p.stride * sizeof(uint32_t);   // bytes for this kernel
      a.workCounter = work + p.NK * 8;
      if (p.codesBytes) {
        uint8_t *base = reinterpret_cast<uint8_t *>(a.scratch);
        const size_t codesAllBytes = (static_cast<size_t>(p.roundBlocks) * groupsHere * p.codesBytes + 255) & ~static_cast<size_t>(255);
        a.codesAll = base; a.codesStride = p.codesBytes;
        a.scratchStride = p.stateBytes;
        unsigned int *simWork = work + kClassSlots * 8 + p.NK * 8;   // (the class's unused "exact" queue counters)
        const int *list = a.classBlocks;
        for (size_t at = 0; at < members.size(); at += static_cast<size_t>(p.roundBlocks)) {
          if (at) {   // the queues of the round before are spent
            HIP_TRY(hipMemsetAsync(a.workCounter, 0, 8 * sizeof(unsigned int), st));
            HIP_TRY(hipMemsetAsync(simWork, 0, 8 * sizeof(unsigned int), st));
          }
          NullArgs r = a;
          r.classBlocks = list + at;
          r.nClassBlocks = static_cast<int>(std::min<size_t>(static_cast<size_t>(p.roundBlocks), members.size() - at));
          NullArgs sim = r;
          sim.workCounter = simWork;
          launch_generic_sim(sim, p.simGrid, p.ldsBytes, st);
          launch_generic_dp(r, p.grid, p.ldsBytes, base + codesAllBytes, st);
          b->nl[2]++;
        }
      } else
      launch_null_generic(a, p.grid, p.ldsBytes, reinterpret_cast<uint8_t *>(a.scratch), st);
      b->nl[2]++;
    } else if (b->allExact || p.NK >= kFastRows) {
      a.skipMask = extraSkip;
      a.workCounter = work + kClassSlots * 8 + p.NK * 8;
      if (!launch_null(p.NK, a, p.grid, 0, true, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
      b->nl[2]++;
    } else {
      a.skipMask = kFlagExact | extraSkip;
      a.workCounter = work + p.NK * 8;
      a.dual = (c->dualRows && p.ldsBytes && p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat)) ? 1 : 0;
      a.hiOcc = p.hiOcc; a.plainLds = static_cast<int>(p.plainLds); a.cacheSites = p.cacheSites;
      // tail sharing: one slot per workgroup of this launch in the zeroed dsteal area (see batch_run_async)
      if (c->tailSharing && !p.comboSplit && static_cast<size_t>(p.grid) <= steal_slots(c) && b->dsteal.p) {
        const size_t idx = static_cast<size_t>(phase) * b->classes.size() + pi;
        if ((idx + 1) * steal_words(c) * sizeof(uint32_t) <= b->dsteal.cap) a.steal = b->dsteal.as<unsigned int>() + idx * steal_words(c);
        a.stealWait = R.streaming ? 0 : 1;
        a.stealRelease = c->tailRelease;
      }
      if (!launch_null(p.NK, a, p.grid, p.ldsBytes, false, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
      b->nl[2]++;
      NullArgs e = a;
      e.classBlocks = b->dexact.as<int>() + b->classOff[p.NK];
      e.nBlocksPtr = b->dcounters.as<int>() + kCntExact + p.NK;
      e.nClassBlocks = 0;
      e.skipMask = extraSkip;
      e.comboSplit = 0;
      e.steal = nullptr;
      e.dual = 0; e.hiOcc = 0; e.plainLds = 0; e.cacheSites = 0;
      e.workCounter = work + kClassSlots * 8 + p.NK * 8;
      if (!launch_null(p.NK, e, p.gridExact, 0, true, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
    }
    if (concurrent) {
      scratchOff += p.need;
      Event &done = *b->classDone[static_cast<size_t>(phase) * plans.size() + pi];
      HIP_TRY(done.record(st));
      HIP_TRY(hipStreamWaitEvent(R.cs, done, 0));
    }
  }
  return RC_OK;
}

static int batch_run_async(rc_batch *b, bool streaming = false) {
  rc_ctx *c = b->ctx;
  if (b->state != rc_batch::READY && b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch is not ready to run");
  HIP_TRY(hipSetDevice(c->device));
  std::fill(b->t, b->t + 6, 0.0f);
  std::fill(b->nl, b->nl + 6, 0);
  b->maximaFetched = false;
  b->nullKernel.clear(); b->nullKernelCost = 0.0;
  const int n = b->n, sampleN = b->par.sampleN;
  const int groups = (sampleN + kWave - 1) / kWave, Spad = groups * kWave;
  b->clamped = 0;
  hipStream_t cs = ctx_stream(c, static_cast<int>(c->seq++ % static_cast<unsigned>(c->computeStreams)));
  if (!cs) return fail(RC_ERR_DEVICE, "hipStreamCreate failed");
  const uint8_t *blob = b->dblob.as<uint8_t>();
  bool nativeFirst = c->serialNative != 0, allFat = !b->classes.empty(), wide = false;
  for (auto &kv : b->classes) {
    int maxLc = 0;
    for (int bi : kv.second) maxLc = std::max(maxLc, b->meta[bi].L);
    allFat = allFat && kv.first != kGenericClass && fat_class(b, c, kv.first, maxLc);
    wide = wide || kv.first == kGenericClass;
  }
  bool fat = allFat && c->nativeFirst != 0 && (!streaming || static_cast<long long>(b->okBlocks.size()) * groups >= c->fatStreamMinItems);
  if (c->nativeFirst == 1) nativeFirst = true;
  if (fat) nativeFirst = true;
  // k_null_generic's persistent grid takes every register of the chip as well (128 VGPRs, four wavefronts per SIMD): native-block
  // kernels queued beside it either wait for its tail or, worse, get in first and keep part of its workgroups out for their whole run
  if (wide && c->nativeFirst != 0) nativeFirst = true;
  const int nativeMode = c->serialNative ? 0 : nativeFirst ? 1 : 0;
  const RunEnv R{b, c, cs, n, sampleN, groups, Spad, n * 6, table_ptrs(c, b->par.blosum), blob,
                 reinterpret_cast<const DevBlock *>(blob + b->oDblocks), reinterpret_cast<const int *>(blob + b->oOk),
                 reinterpret_cast<const int *>(blob + b->oClass), streaming, nativeFirst, nativeMode, fat};
  HIP_TRY(hipStreamWaitEvent(cs, b->evPrep, 0));   // upload + expected-score tables
  HIP_TRY(b->evStart.record(cs));
  if (!b->okBlocks.empty()) {
    RC_TRY(ensure_mt_stream(c, b, cs, b->par.seed_base, Spad, b->maxDraws));
    // ---- null sampling per N-1 class
    uint32_t *cnt = b->dcounters.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(cnt + kCntWork, 0, (kCntTotal - kCntWork) * sizeof(uint32_t), cs));   // work queues and the clamp counter
    // -1.0f everywhere first: skipped blocks report "no HSS" rows
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->maxPtr), 0xBF800000u, static_cast<size_t>(n) * sampleN, cs));
    FitArgs fa{};
    fa.dblocks = R.dblocks; fa.blocks = R.okList; fa.maxima = b->maxPtr; fa.hss = b->dhss.as<DevHss>();
    fa.hssCount = b->dhssCount.as<int>(); fa.hssCap = b->hssCap; fa.out = b->dfit.as<FitOut>(); fa.flags = b->dflags.as<uint32_t>();
    fa.sampleN = sampleN; fa.stopEarly = b->par.stopEarly;
    fa.stopCutoff = static_cast<int>(b->par.cutoff * b->par.sampleN);   // score.c:992
    // --stop-early: a first round of sample groups for every block, then -- doubling the samples each round -- only the blocks
    // that are still undecided (k_stop_mark between the rounds)
    // (only when the batch fills the chip several times over: two rounds of a small batch are two under-filled launches in a
    // row -- genomic-preprocessed.maf, 33 blocks: 3.1 ms in one round, 5.0 ms in two)
    int g1 = groups;
    if (b->par.stopEarly && fa.stopCutoff >= 0 && fa.stopCutoff < sampleN && static_cast<long long>(b->okBlocks.size()) * groups >= c->stopTwoRoundsMinItems)
      g1 = std::min(groups, (fa.stopCutoff + 1 + kWave + kWave - 1) / kWave);
    {   // one staging scratch for all rounds
      size_t need = 0, most = 0;
      int r = 0;
      for (int lo = 0, hi = g1; lo < groups; r++) {
        if (r == c->stopRounds - 1) hi = groups;
        RC_TRY(launch_null_groups(R, lo, hi, r, 0, &need));
        most = std::max(most, need);
        lo = hi; hi = std::min(groups, hi * 2);
      }
      HIP_TRY(b->dscratch.ensure(most * sizeof(uint32_t)));
      if (c->tailSharing) {   // claim words of every (round, class) launch, zeroed: "nothing to take"
        const size_t bytes = static_cast<size_t>(r) * b->classes.size() * steal_words(c) * sizeof(uint32_t);
        HIP_TRY(b->dsteal.ensure(bytes));
        HIP_TRY(hipMemsetAsync(b->dsteal.p, 0, bytes, cs));
      }
    }
    // the native block's kernels go first: queued behind k_null's persistent grid they would find no free wavefront slot until its
    // tail, and the fit waits for them (long blocks: 1000 blocks of 12 x 300 took 628 ms per step that way instead of 600)
    bool nativeQueued = true;
    if (R.nativeFirst) RC_TRY(launch_native_block(R));   // on this stream, in front of k_null (fat_class): not part of the sampling interval
    HIP_TRY(b->evS0.record(cs));   // the class launches of the first round fork from here: every memset above is behind it
    if (!R.nativeFirst) RC_TRY(launch_native_block(R));
    int round = 0;
    for (int lo = 0, hi = g1; lo < groups; round++) {
      if (round == c->stopRounds - 1) hi = groups;
      if (lo > 0 && !nativeQueued) { RC_TRY(launch_native_block(R)); nativeQueued = true; }   // the stop marks need the best native scores (evN1)
      if (lo > 0) {
        if (!R.nativeFirst && lo == g1) HIP_TRY(hipStreamWaitEvent(cs, b->evN1, 0));   // best native scores
        fa.firstN = std::min(sampleN, lo * kWave);
        launch_stop_mark(fa, static_cast<int>(b->okBlocks.size()), cs);
        HIP_TRY(b->evMid.record(cs));
      }
      RC_TRY(launch_null_groups(R, lo, hi, round, lo > 0 ? kFlagStopped : 0));
      lo = hi;
      hi = std::min(groups, hi * 2);
    }
    HIP_TRY(b->evS1.record(cs));
    if (!nativeQueued) RC_TRY(launch_native_block(R));
    // ---- EVD fit
    if (!R.nativeFirst) HIP_TRY(hipStreamWaitEvent(cs, b->evN1, 0));   // native HSS of the native stream
    // (a batch of a stream: the next sub-batch's k_null will be holding the LDS by the time this fit runs)
    // (a small input -- a few dozen blocks -- is one sub-batch by itself and keeps the four-wavefront fit, whose latency is its run time)
    launch_evd_fit(fa, static_cast<int>(b->okBlocks.size()), c->inflight.load() == 0 && (!streaming || b->okBlocks.size() <= 64), cs); b->nl[4]++;
  }
  HIP_TRY(b->evFit.record(cs));
  // small results back to the host: four copies into one pinned staging area behind the fit
  {
    const int slots = R.slots;
    const size_t szFit = sizeof(FitOut) * n, szCnt = sizeof(int) * static_cast<size_t>(slots), szOff = sizeof(int) * (static_cast<size_t>(slots) + 1);
    const size_t oFit = 0, oCnt = al256(oFit + szFit), oOff = al256(oCnt + szCnt), oClamp = al256(oOff + szOff), total = oClamp + 16;
    HIP_TRY(b->hstage.ensure(total));
    if (!b->okBlocks.empty()) {
      uint8_t *hs = b->hstage.as<uint8_t>();
      // the packed HSS records: their number is only known on the device, so a generous prefix travels with the
      // counts (16 per block; 9 on average for random blocks) and rc_batch_wait fetches the rest only if there are more
      b->packedAhead = std::min(static_cast<size_t>(slots) * b->hssCap, static_cast<size_t>(n) * 16);
      HIP_TRY(b->hpacked.ensure(sizeof(DevHss) * std::max<size_t>(b->packedAhead, 1)));
      // written by a kernel of ours into the mapped pinned buffers (k_results_out explains why not hipMemcpyAsync)
      void *dStage = nullptr, *dPacked = nullptr;
      static_assert(sizeof(FitOut) % 4 == 0 && sizeof(DevHss) % 4 == 0, "results travel as 32-bit words");
      if (!c->copyResults && hipHostGetDevicePointer(&dStage, b->hstage.p, 0) == hipSuccess &&
          hipHostGetDevicePointer(&dPacked, b->hpacked.p, 0) == hipSuccess) {
        uint8_t *ds = static_cast<uint8_t *>(dStage);
        ResultsOutArgs ro{};
        auto part = [&](int x, void *dst, const void *src, size_t bytes) {
          ro.dst[x] = static_cast<uint32_t *>(dst); ro.src[x] = static_cast<const uint32_t *>(src); ro.words[x] = bytes / 4;
        };
        part(0, ds + oFit, b->dfit.p, szFit);
        part(1, ds + oCnt, b->dhssCount.p, szCnt);
        part(2, ds + oOff, b->dhssOffsets.p, szOff);
        part(3, ds + oClamp, b->dcounters.as<uint32_t>() + kCntClamp, sizeof(unsigned long long));
        part(4, dPacked, b->dhssPacked.p, sizeof(DevHss) * b->packedAhead);
        launch_results_out(ro, cs);
      } else {
        HIP_TRY(hipMemcpyAsync(hs + oFit, b->dfit.p, szFit, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(hs + oCnt, b->dhssCount.p, szCnt, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(hs + oOff, b->dhssOffsets.p, szOff, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(hs + oClamp, b->dcounters.as<uint32_t>() + kCntClamp, sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipMemcpyAsync(b->hpacked.p, b->dhssPacked.p, sizeof(DevHss) * b->packedAhead, hipMemcpyDeviceToHost, cs));
      }
    }
  }
  HIP_TRY(b->evDone.record(cs));
  HIP_TRY(hipGetLastError());
  b->state = rc_batch::RUNNING;
  c->inflight.fetch_add(1);
  return RC_OK;
}

static int batch_wait(rc_batch *b) {
  if (b->state == rc_batch::DONE) return RC_OK;
  if (b->state != rc_batch::RUNNING) return fail(RC_ERR_ARG, "batch is not running");
  rc_ctx *c = b->ctx;
  HIP_TRY(hipSetDevice(c->device));
  hipError_t e = timed_call("wait hipEventSynchronize", 0, [&] { return hipEventSynchronize(b->evDone); });
  // nothing in flight: releasing outgrown buffers stalls nobody.  A stream kept full never gets there: once the parked buffers
  // exceed a sixteenth of the device memory they are released anyway (one stall, instead of holding every generation of a growing buffer)
  if (c->inflight.fetch_sub(1) == 1 || c->retired->held() > c->retiredCapBytes) { if (!c->retired->empty()) c->retired->drain(); }
  b->state = rc_batch::READY;   // a failed run may be repeated
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
  HIP_TRY(hipGetLastError());
  const int n = b->n, slots = n * 6;
  b->fit.assign(n, FitOut{0, 0, -1, 0});
  b->hssCount.assign(static_cast<size_t>(slots), 0);
  b->hssOff.assign(static_cast<size_t>(slots) + 1, 0);
  b->hssRec.clear();
  if (!b->okBlocks.empty()) {
    float ms = 0;
    if (b->mtLaunched) { HIP_TRY(hipEventElapsedTime(&ms, b->evMt0, b->evMt1)); b->t[1] = ms; b->nl[1] = 1; }
    HIP_TRY(hipEventElapsedTime(&ms, b->evN0, b->evN1)); b->t[3] = ms;
    HIP_TRY(hipEventElapsedTime(&ms, b->evS0, b->evS1)); b->t[2] = ms;
    HIP_TRY(hipEventElapsedTime(&ms, b->evS1, b->evFit)); b->t[4] = ms;
    HIP_TRY(hipEventElapsedTime(&ms, b->evStart, b->evFit)); b->t[0] = ms;
    if (b->evPrep0.e && b->evPrep.e) { HIP_TRY(hipEventElapsedTime(&ms, b->evPrep0, b->evPrep)); b->t[5] = ms; b->nl[5] = 2; }
    const size_t szFit = sizeof(FitOut) * n, szCnt = sizeof(int) * static_cast<size_t>(slots), szOff = sizeof(int) * (static_cast<size_t>(slots) + 1);
    const size_t oFit = 0, oCnt = al256(oFit + szFit), oOff = al256(oCnt + szCnt), oClamp = al256(oOff + szOff);
    const uint8_t *hs = b->hstage.as<uint8_t>();
    const FitOut *f = reinterpret_cast<const FitOut *>(hs + oFit);
    for (int bi : b->okBlocks) b->fit[bi] = f[bi];
    std::memcpy(b->hssCount.data(), hs + oCnt, szCnt);
    std::memcpy(b->hssOff.data(), hs + oOff, szOff);
    std::memcpy(&b->clamped, hs + oClamp, sizeof(unsigned long long));
    // the used HSS records only (k_hss_pack); normally all of them came with the first copy
    const size_t nrec = static_cast<size_t>(std::max(b->hssOff[slots], 0));
    if (nrec > b->packedAhead) {
      // on the helper stream: the compute streams may already hold the kernels of later batches
      PinBuf more;
      HIP_TRY(more.ensure(sizeof(DevHss) * nrec));
      RC_STREAM_TRY(aux, stream_aux(c));
      HIP_TRY(hipMemcpyAsync(more.p, b->dhssPacked.p, sizeof(DevHss) * nrec, hipMemcpyDeviceToHost, aux));
      HIP_TRY(hipStreamSynchronize(aux));
      b->hssRec.assign(more.as<DevHss>(), more.as<DevHss>() + nrec);
    } else if (nrec) {
      b->hssRec.assign(b->hpacked.as<DevHss>(), b->hpacked.as<DevHss>() + nrec);
    }
  }
  b->state = rc_batch::DONE;
  return RC_OK;
}

int rc_batch_run_async(rc_batch *b) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  if (g_no_stream_cache && b->ctx->inflight.load() == 0) b->ctx->U_valid = false;
  return batch_run_async(b);
}

int rc_batch_wait(rc_batch *b) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  return batch_wait(b);
}

int rc_batch_run(rc_batch *b) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  trace("run", b);
  RC_TRY(rc_batch_run_async(b));
  trace("queued", b);
  const int r = batch_wait(b);
  trace("done", b);
  return r;
}

int rc_batch_bind_maxima(rc_batch *b, void *device_ptr) {
  if (!b || !device_ptr) return fail(RC_ERR_ARG, "bad argument");
  if (b->state == rc_batch::RUNNING) return fail(RC_ERR_ARG, "batch is running");
  b->dmaxima.release();
  b->maxPtr = static_cast<float *>(device_ptr);
  b->maximaExternal = true;
  if (b->state == rc_batch::DONE) b->state = rc_batch::READY;
  return RC_OK;
}

// RC_TRACE=1: where a finished batch's device events lie on the host's clock (a reference event is recorded and synchronised
// once; later events are placed by their elapsed time from it) -- one line per batch on stderr
static void trace_device(rc_batch *b) {
  if (!trace_on() || b->okBlocks.empty()) return;
  static Event ref;
  static double refHostUs = 0.0;
  if (!ref.e) {
    hipStream_t aux = stream_aux(b->ctx);
    if (!aux || ref.record(aux) != hipSuccess || hipEventSynchronize(ref) != hipSuccess) return;
    refHostUs = trace_now_us();
    std::fprintf(stderr, "[rc device clock: reference recorded now; a batch finished before it shows negative offsets]\n");
  }
  auto at = [&](const Event &e) { float ms = 0; return (e.e && hipEventElapsedTime(&ms, ref, e) == hipSuccess) ? refHostUs + ms * 1e3 : -1.0; };
  std::fprintf(stderr, "[rc device %p n=%d] now %.1f | prep0 %.1f prep %.1f start %.1f null0 %.1f null1 %.1f native0 %.1f native1 %.1f fit %.1f done %.1f (us on the device-reference clock)\n",
               static_cast<void *>(b), b->n, trace_now_us(), at(b->evPrep0), at(b->evPrep), at(b->evStart), at(b->evS0), at(b->evS1), at(b->evN0), at(b->evN1),
               at(b->evFit), at(b->evDone));
}

// ------------------------------------------------------------------------------------------ streams of batches

int rc_stream_create(rc_ctx *ctx, const rc_params *par, int32_t depth, rc_stream **out) {
  if (!ctx || !par || !out || depth < 1 || depth > 16) return fail(RC_ERR_ARG, "bad argument");
  if (par->blosum != 62 && par->blosum != 90) return fail(RC_ERR_ARG, "blosum must be 62 or 90");
  if (par->sampleN < 1) return fail(RC_ERR_ARG, "sampleN must be >= 1");
  rc_stream *s = new rc_stream();
  s->ctx = ctx; s->par = *par; s->depth = depth;
  // the HIP streams its sub-batches will use, now rather than inside the first submits (10 ms each): a driver creates its rc_stream
  // while something else is still on the way (the stand-alone driver: the first chunk of tree fits)
  (void)stream_copy(ctx);
  for (int x = 0; x < std::min(depth, ctx->computeStreams); x++) (void)ctx_stream(ctx, x);
  *out = s;
  return RC_OK;
}

void rc_stream_destroy(rc_stream *s) {
  if (!s) return;
  for (rc_batch *b : s->inflight) { b->owner = nullptr; rc_batch_destroy(b); }
  for (rc_batch *b : s->spare) { b->owner = nullptr; rc_batch_destroy(b); }
  delete s;
}

// Sub-batch sizes for a stream of n_blocks blocks.  What it balances: the first kernels should start early (a small first
// sub-batch), every later sub-batch should be queued before its predecessor drains (sizes grow geometrically: the host prepares a
// block several times faster than the GPU scores it), and a launch should fill the chip's wavefront slots a whole number of times
// (r = slots / sample groups blocks do so once; a launch of 1.2 r takes as long as one of 2 r), so the odd remainder goes FIRST.
int rc_stream_plan(const rc_ctx *c, const rc_params *par, int32_t n_blocks, int32_t row_classes, int32_t *sizes, int32_t cap) {
  if (!c || !par || !sizes || cap < 1 || n_blocks < 0 || par->sampleN < 1) return fail(RC_ERR_ARG, "bad argument");
  const long long groups = (par->sampleN + kWave - 1) / kWave;
  const long long slots = static_cast<long long>(c->numCU) * 16;                 // k_null with its codes staged in LDS: 16 wavefronts per CU
  // several row-count classes are several launches side by side: each gets its share of the slots
  const long long r = std::max<long long>(1, slots / groups) * std::max(1, (row_classes + 1) / 2);
  const long long most = 8 * r;                                                   // a long stream: bounded memory, results keep flowing
  int n = 0;
  long long left = n_blocks;
  auto put = [&](long long v) { if (v > 0 && n < cap) { sizes[n++] = static_cast<int32_t>(v); left -= v; } };
  if (left <= r + r / 2) { put(left); return n; }
  long long first = left % r;
  if (first < r / 8) first += r / 2;
  put(first);
  for (long long next = r; left > 0; next = std::min(most, next * 2)) {
    if (n == cap - 1 || left - next < next / 2) { put(left); break; }             // a short tail goes with the sub-batch before it
    put(next);
  }
  return n;
}

int rc_stream_pending(const rc_stream *s) { return s ? static_cast<int>(s->inflight.size()) : 0; }

int rc_stream_submit_bound(rc_stream *s, const rc_block *blocks, int32_t n_blocks, void *maxima_device_ptr);

int rc_stream_submit(rc_stream *s, const rc_block *blocks, int32_t n_blocks) {
  return rc_stream_submit_bound(s, blocks, n_blocks, nullptr);
}

int rc_stream_submit_bound(rc_stream *s, const rc_block *blocks, int32_t n_blocks, void *maxima_device_ptr) {
  if (!s || (!blocks && n_blocks > 0) || n_blocks < 0) return fail(RC_ERR_ARG, "bad argument");
  if (static_cast<int>(s->inflight.size()) >= s->depth) return fail(RC_ERR_ARG, "stream is full: take a finished batch with rc_stream_next first");
  rc_batch *b = nullptr;
  if (!s->spare.empty()) { b = s->spare.back(); s->spare.pop_back(); }
  else { b = new rc_batch(); b->adopt(s->ctx); b->owner = s; }
  if (g_no_stream_cache && s->inflight.empty() && s->ctx->inflight.load() == 0) s->ctx->U_valid = false;   // a new pass over the stream
  if (maxima_device_ptr) { b->dmaxima.release(); b->maxPtr = static_cast<float *>(maxima_device_ptr); b->maximaExternal = true; }
  else if (b->maximaExternal) { b->maxPtr = nullptr; b->maximaExternal = false; }
  trace("submit", b);
  int r = batch_prepare(b, blocks, n_blocks, &s->par);
  trace("prepared", b);
  if (r == RC_OK) r = batch_upload(b);
  trace("uploaded", b);
  if (r == RC_OK) r = batch_run_async(b, true);
  trace("queued", b);
  if (r != RC_OK) {
    // the batch may hold a queued upload of its pinned blob (upload succeeded, run failed): let it finish before the batch is re-used
    if (b->state == rc_batch::READY && b->evPrep.e) (void)hipEventSynchronize(b->evPrep);
    b->state = rc_batch::EMPTY;
    s->spare.push_back(b);
    return r;
  }
  s->inflight.push_back(b);
  return RC_OK;
}

int rc_stream_next(rc_stream *s, rc_batch **out) {
  if (!s || !out) return fail(RC_ERR_ARG, "bad argument");
  *out = nullptr;
  if (s->inflight.empty()) return fail(RC_ERR_ARG, "no batch in flight");
  rc_batch *b = s->inflight.front();
  s->inflight.pop_front();
  trace("wait", b);
  const int r = batch_wait(b);
  trace("done", b);
  if (r == RC_OK) trace_device(b);
  if (r != RC_OK) { s->spare.push_back(b); return r; }
  *out = b;
  return RC_OK;
}

void rc_stream_recycle(rc_stream *s, rc_batch *b) {
  if (!b) return;
  if (!s || b->owner != s || b->state == rc_batch::RUNNING) { rc_batch_destroy(b); return; }
  s->spare.push_back(b);
}

// ------------------------------------------------------------------------------------------ results

int rc_batch_work(const rc_batch *b, int64_t *sa, int64_t *cs) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  int64_t a = 0, c = 0;
  for (int bi : b->okBlocks) { a += b->par.sampleN; c += static_cast<int64_t>(b->meta[bi].cols) * b->par.sampleN; }
  if (sa) *sa = a;
  if (cs) *cs = c;
  return RC_OK;
}

int rc_batch_size(const rc_batch *b) { return b ? b->n : 0; }

int rc_batch_timing(const rc_batch *b, float t[5], int32_t nl[5]) {
  if (!b || b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  for (int i = 0; i < 5; i++) { if (t) t[i] = b->t[i]; if (nl) nl[i] = b->nl[i]; }
  return RC_OK;
}

const char *rc_batch_null_kernel(const rc_batch *b) { return b ? b->nullKernel.c_str() : ""; }

int rc_batch_prep_timing(const rc_batch *b, double *host_ms, float *table_kernels_ms, int64_t *uploaded_bytes) {
  if (!b || b->state == rc_batch::EMPTY) return fail(RC_ERR_ARG, "batch has not been prepared");
  if (host_ms) *host_ms = b->prepHostMs;
  if (table_kernels_ms) *table_kernels_ms = b->t[5];
  if (uploaded_bytes) *uploaded_bytes = static_cast<int64_t>(b->hostUsed);
  return RC_OK;
}

static int check_blk(const rc_batch *b, int blk, bool needRun) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  if (blk < 0 || blk >= b->n) return fail(RC_ERR_ARG, "block index out of range");
  if (needRun && b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  return RC_OK;
}

int rc_batch_status(const rc_batch *b, int32_t blk) {
  int r = check_blk(b, blk, false);
  return r ? r : b->meta[blk].status;
}

const char *rc_batch_block_error(const rc_batch *b, int32_t blk) {
  if (!b || blk < 0 || blk >= b->n) return "";
  auto it = b->errs.find(blk);
  return it == b->errs.end() ? "" : it->second.c_str();
}

int rc_batch_models(const rc_batch *b, int32_t blk, rc_model *fwd, rc_model *rev) {
  int r = check_blk(b, blk, false);
  if (r) return r;
  const BlockMeta &m = b->meta[blk];
  if (m.status != RC_OK) return m.status;
  if (b->state == rc_batch::EMPTY) return fail(RC_ERR_ARG, "batch has not been uploaded");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipEventSynchronize(b->evPrep));   // scores and probs are computed on the device (k_prep_models)
  std::vector<ModelRec> rec(2 * static_cast<size_t>(m.N));
  HIP_TRY(hipMemcpy(rec.data(), b->dblob.as<uint8_t>() + b->db[blk].off_models, rec.size() * sizeof(ModelRec), hipMemcpyDeviceToHost));
  for (int s = 0; s < 2; s++) {
    rc_model *dst = s ? rev : fwd;
    if (!dst) continue;
    for (int j = 0; j < m.N; j++) {
      const ModelRec &q = rec[static_cast<size_t>(s) * m.N + j];
      std::memcpy(dst[j].scores, q.scores, sizeof q.scores);
      std::memcpy(dst[j].probs, q.probs, sizeof q.probs);
      std::memcpy(dst[j].freqs, q.freqs, sizeof q.freqs);
      dst[j].kappa = q.kappa; dst[j].dist = q.dist;
    }
  }
  return RC_OK;
}

static int fetch_maxima(rc_batch *b) {
  if (b->maximaFetched) return RC_OK;
  HIP_TRY(hipSetDevice(b->device));
  b->maxima.resize(static_cast<size_t>(b->n) * b->par.sampleN);
  if (!b->maxima.empty()) {
    if (b->okBlocks.empty()) std::fill(b->maxima.begin(), b->maxima.end(), -1.0f);
    else HIP_TRY(hipMemcpy(b->maxima.data(), b->maxPtr, b->maxima.size() * sizeof(float), hipMemcpyDeviceToHost));
  }
  b->maximaFetched = true;
  return RC_OK;
}

int rc_batch_maxima(const rc_batch *bc, int32_t blk, float *out) {
  rc_batch *b = const_cast<rc_batch *>(bc);
  int r = check_blk(b, blk, true);
  if (r) return r;
  if ((r = fetch_maxima(b))) return r;
  std::memcpy(out, b->maxima.data() + static_cast<size_t>(blk) * b->par.sampleN, sizeof(float) * b->par.sampleN);
  return RC_OK;
}

int rc_batch_maxima_all(const rc_batch *bc, float *out) {
  rc_batch *b = const_cast<rc_batch *>(bc);
  if (!b || b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  int r = fetch_maxima(b);
  if (r) return r;
  std::memcpy(out, b->maxima.data(), b->maxima.size() * sizeof(float));
  return RC_OK;
}

int rc_batch_fit(const rc_batch *b, int32_t blk, int32_t *evd_rc, float *mu, float *lambda) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  if (b->meta[blk].status != RC_OK) return b->meta[blk].status;
  const FitOut &f = b->fit[blk];
  if (evd_rc) *evd_rc = f.rc;
  if (mu) *mu = static_cast<float>(f.mu);          // *parMu = mu (double -> float), score.c:1051
  if (lambda) *lambda = static_cast<float>(f.lambda);
  return RC_OK;
}

int rc_batch_fit_all(const rc_batch *b, float *out) {
  if (!b || !out || b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  for (int i = 0; i < b->n; i++) {
    float *o = out + 4 * static_cast<size_t>(i);
    if (b->meta[i].status != RC_OK) { o[0] = static_cast<float>(b->meta[i].status); o[1] = o[2] = o[3] = 0.0f; continue; }
    const FitOut &f = b->fit[i];
    o[0] = static_cast<float>(f.rc); o[1] = static_cast<float>(f.mu); o[2] = static_cast<float>(f.lambda);
    o[3] = static_cast<float>(f.better);
  }
  return RC_OK;
}

// number of HSS records of a block (all six strand x frame lists)
static int block_hss_count(const rc_batch *b, int blk) {
  int n = 0;
  for (int combo = 0; combo < 6; combo++) n += std::min(b->hssCount[static_cast<size_t>(blk) * 6 + combo], b->hssCap);
  return n;
}

int rc_batch_hss(const rc_batch *b, int32_t blk, rc_hss *out, int32_t cap) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const BlockMeta &h = b->meta[blk];
  if (h.status != RC_OK) return h.status;
  const FitOut &f = b->fit[blk];
  const float mu = static_cast<float>(f.mu), lambda = static_cast<float>(f.lambda);
  std::vector<rc_hss> all;
  for (int combo = 0; combo < 6; combo++) {   // '+' hits then '-' hits, frames ascending (score.c:1107-1127)
    const size_t slot = static_cast<size_t>(blk) * 6 + combo;
    const int cnt = b->hssCount[slot];
    if (cnt > b->hssCap) return fail(RC_ERR_UNSUPPORTED, "HSS buffer overflow");
    for (int i = 0; i < cnt; i++) {
      const DevHss &d = b->hssRec[static_cast<size_t>(b->hssOff[slot]) + i];
      if (!(d.score > 0.0f)) break;           // lists end at the first non-positive score (score.c:1112,1121)
      rc_hss o{};
      o.strand = d.strand ? '-' : '+';
      o.frame = d.frame; o.startSite = d.startSite; o.endSite = d.endSite; o.score = d.score;
      o.start = d.startSite * 3 + d.frame + 1;                       // score.c:921-922
      o.end = d.endSite * 3 + d.frame + 3;
      if (h.ref_start == 0 && h.ref_length == 0) { o.startGenomic = o.start; o.endGenomic = o.end; }   // :925-928
      else if (!d.strand) {
        o.startGenomic = h.ref_start + d.startSite * 3 + d.frame;    // :932-933
        o.endGenomic = h.ref_start + d.endSite * 3 + d.frame + 2;
      } else {
        o.endGenomic = (h.ref_start + h.ref_length - 1) - d.startSite * 3 - d.frame;       // :935-936
        o.startGenomic = (h.ref_start + h.ref_length - 1) - d.endSite * 3 - d.frame - 2;
      }
      o.pvalue = (f.rc == 1) ? pvalue_of(d.score, mu, lambda) : 99.0f;   // RNAcode.c:180-188
      all.push_back(o);
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const rc_hss &a, const rc_hss &c) { return a.score > c.score; });
  for (int i = 0; i < static_cast<int>(all.size()) && i < cap; i++) out[i] = all[i];
  return static_cast<int>(all.size());
}

int rc_batch_hss_all(const rc_batch *b, rc_hss *out, int64_t cap, int64_t *offsets) {
  if (!b || !offsets || (!out && cap > 0) || b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  int64_t total = 0;
  std::vector<rc_hss> tmp;
  for (int blk = 0; blk < b->n; blk++) {
    offsets[blk] = total;
    if (b->meta[blk].status != RC_OK) continue;
    const int n = block_hss_count(b, blk);
    if (n == 0) continue;
    tmp.resize(n);
    const int got = rc_batch_hss(b, blk, tmp.data(), n);
    if (got < 0) return got;
    for (int i = 0; i < got && i < n; i++) if (total + i < cap) out[total + i] = tmp[i];
    total += std::min(got, n);
  }
  offsets[b->n] = total;
  return RC_OK;
}

int rc_batch_clamped(const rc_batch *b, int64_t *count) {
  if (!b || b->state != rc_batch::DONE) return fail(RC_ERR_ARG, "batch has not been run");
  *count = static_cast<int64_t>(b->clamped);
  return RC_OK;
}

int rc_batch_native_S(const rc_batch *b, int32_t blk, int32_t strand, int32_t frame, float *out, int32_t cap) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const BlockMeta &h = b->meta[blk];
  if (h.status != RC_OK) return h.status;
  if (strand < 0 || strand > 1 || frame < 0 || frame > 2) return fail(RC_ERR_ARG, "bad strand/frame");
  const int sites = (h.L - frame) / 3, smax = h.L / 3;
  if (cap < sites * sites) return fail(RC_ERR_ARG, "output too small");
  HIP_TRY(hipSetDevice(b->device));
  if (sites > 0) {
    const size_t at = static_cast<size_t>(strand * 3 + frame) * smax * smax;
    // the scoring pass never materialises S: recompute this block's six matrices with the same kernel (fullS set: no records written)
    rc_ctx *c = b->ctx;
    DevBuf full, tile, idx;
    const bool generic = h.N >= c->genericMinRows;
    const size_t tileStride = generic ? static_cast<size_t>(3) * h.NK * kWave + static_cast<size_t>(kWave) * smax : static_cast<size_t>(kWave) * smax;
    HIP_TRY(full.ensure(static_cast<size_t>(6) * smax * smax * sizeof(float)));
    HIP_TRY(tile.ensure(6 * tileStride * sizeof(float)));
    HIP_TRY(idx.ensure(sizeof(int)));
    const int bi = blk;
    HIP_TRY(hipMemcpy(idx.p, &bi, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(full.p, 0, static_cast<size_t>(6) * smax * smax * sizeof(float)));
    NativeArgs na{};
    const uint8_t *blob = b->dblob.as<uint8_t>();
    na.blob = b->dblob.as<uint8_t>(); na.dblocks = reinterpret_cast<const DevBlock *>(blob + b->oDblocks); na.blocks = idx.as<int>();
    na.hssCap = b->hssCap; na.tieThr = c->tieThr; na.tile = tile.as<float>(); na.tileStride = tileStride; na.nItems = 6;
    na.fullS = full.as<float>();
    if (generic) launch_native_dp_generic(na, 1, tile.as<float>(), tileStride, nullptr);
    else if (!launch_native_dp(h.NK, na, 6, nullptr)) return fail(RC_ERR_UNSUPPORTED, "no native DP kernel for this number of rows");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, full.as<float>() + at, sizeof(float) * sites * sites, hipMemcpyDeviceToHost));
  }
  for (int a = 0; a < sites; a++)
    for (int j = 0; j < a; j++) out[a * sites + j] = 0.0f;
  return sites;
}

int rc_batch_backtrack(const rc_batch *b, int32_t blk, int32_t strand, int32_t opt_b, int32_t opt_i,
                       int32_t *states, int32_t *zout, int32_t *transitions) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const BlockMeta &h = b->meta[blk];
  if (h.status != RC_OK) return h.status;
  if (strand < 0 || strand > 1 || opt_b < 1 || opt_i > h.L) return fail(RC_ERR_ARG, "bad backtrack range");
  if (opt_i < opt_b + 2) {   // the reference's loop (score.c:629) does not run: nothing is filled in (postscript.c:264-266 asks for such ranges)
    for (int i = 0; i < h.N * (h.cols + 1); i++) states[i] = zout[i] = transitions[i] = -9;
    return RC_OK;
  }
  if ((opt_i - opt_b - 2) % 3 != 0) return fail(RC_ERR_ARG, "bad backtrack range");
  rc_ctx *c = b->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int steps = (opt_i - (opt_b + 2)) / 3 + 1, NK = h.NK;
  const DevBlock &d = b->db[blk];
  const int L1 = h.L + 1, zww = d.zw_words;
  // the z table of this strand is made on the device (k_prep_gaps): fetch it
  std::vector<uint64_t> zwv(static_cast<size_t>(L1) * zww);
  HIP_TRY(hipEventSynchronize(b->evPrep));
  HIP_TRY(hipMemcpy(zwv.data(), b->dblob.as<uint8_t>() + d.off_zw + static_cast<size_t>(strand) * L1 * zww * sizeof(uint64_t),
                    zwv.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  const uint64_t *zw = zwv.data();
  DevBuf d_out;
  HIP_TRY(d_out.ensure(sizeof(float) * NK * 3 * steps));
  RC_STREAM_TRY(st, stream_aux(c));
  launch_sk_row(b->dblob.as<uint8_t>(), reinterpret_cast<const DevBlock *>(b->dblob.as<uint8_t>() + b->oDblocks), blk, strand, opt_b, opt_i,
                d_out.as<float>(), steps, st);
  std::vector<float> sk(static_cast<size_t>(NK) * 3 * steps);
  hipError_t e = hipStreamSynchronize(st);   // the streams are non-blocking: a default-stream copy would not wait
  if (e == hipSuccess) e = hipMemcpy(sk.data(), d_out.p, sk.size() * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, hipGetErrorString(e));
  const int pitch = h.cols + 1;
  for (int i = 0; i < h.N * pitch; i++) states[i] = zout[i] = transitions[i] = -9;
  const float Delta = b->par.Delta, Omega = b->par.Omega;
  auto near = [](float x, float y) { return ((x > y) ? (x - y) : (y - x)) < 0.00001; };   // CMP, score.h:30
  for (int k = 1; k <= NK; k++) {
    auto SK = [&](int x, int t) { return t < 0 ? 0.0f : sk[(static_cast<size_t>(k - 1) * 3 + x) * steps + t]; };
    float opt = -99.0f;                                     // MINUS_INF, score.h:27
    int curr = -1, prev = -1, tr = -9;
    for (int x = 0; x < 3; x++) if (SK(x, steps - 1) > opt) { opt = SK(x, steps - 1); curr = x; }
    for (int t = steps - 1, i = opt_i; t >= 0; t--, i -= 3) {
      const int zc = static_cast<int>((zw[static_cast<size_t>(i) * zww + ((k - 1) >> 5)] >> (2 * ((k - 1) & 31))) & 3);
      const int z = zc == 0 ? 0 : (zc == 1 ? 1 : -1);
      const float p0 = SK(0, t - 1), p1 = SK(1, t - 1), p2 = SK(2, t - 1);
      if (z == 0) { prev = curr; tr = 0; }
      if (z == 1) {                                           // score.c:647-682
        if (curr == 0) { if (near(SK(0, t), p0 + Delta)) { tr = 2; prev = 0; } if (near(SK(0, t), p2 + Omega)) { tr = 1; prev = 2; } }
        if (curr == 1) { if (near(SK(1, t), p0 + Omega)) { tr = 1; prev = 0; } if (near(SK(1, t), p1 + Delta)) { tr = 1; prev = 1; } }
        if (curr == 2) { if (near(SK(2, t), p1 + Omega)) { tr = 1; prev = 1; } if (near(SK(2, t), p2 + Delta)) { tr = 2; prev = 2; } }
      }
      if (z == -1) {                                          // score.c:685-718
        if (curr == 0) { if (near(SK(0, t), p0 + Delta)) { tr = 2; prev = 0; } if (near(SK(0, t), p1 + Omega)) { tr = 1; prev = 1; } }
        if (curr == 1) { if (near(SK(1, t), p1 + Delta)) { tr = 2; prev = 1; } if (near(SK(1, t), p2 + Omega)) { tr = 1; prev = 2; } }
        if (curr == 2) { if (near(SK(2, t), p2 + Delta)) { tr = 2; prev = 2; } if (near(SK(2, t), p0 + Omega)) { tr = 1; prev = 0; } }
      }
      states[k * pitch + i] = curr;
      transitions[k * pitch + i] = tr;
      zout[k * pitch + i] = z;
      curr = prev;
    }
  }
  return RC_OK;
}

int rc_code_tables(int32_t blosum, int32_t pep_out[64], int32_t matrix_out[400]) {
  if ((blosum != 62 && blosum != 90) || !pep_out || !matrix_out) return fail(RC_ERR_ARG, "bad argument");
  const CodeTables ct(blosum);
  for (int c = 0; c < 64; c++) pep_out[c] = ct.pep[c];
  for (int p = 0; p < 20; p++) for (int q = 0; q < 20; q++) matrix_out[20 * p + q] = ct.blosum[p][q];
  return RC_OK;
}

int rc_fit_tree(const rc_block *blk, char *newick_out, int32_t cap, float *kappa_out) {
  if (!blk || !newick_out || cap < 8 || !kappa_out || !blk->rows || !blk->names) return fail(RC_ERR_ARG, "bad argument");
  std::vector<std::string> rows(blk->n_rows), names(blk->n_rows);
  for (int r = 0; r < blk->n_rows; r++) { rows[r] = blk->rows[r]; names[r] = blk->names[r]; }
  std::string nwk, err;
  float kappa = 0;
  if (!fit_tree(rows, names, nwk, kappa, nullptr, err)) return fail(RC_ERR_ARG, err);
  if (static_cast<int>(nwk.size()) + 1 > cap) return fail(RC_ERR_ARG, "newick buffer too small");
  std::memcpy(newick_out, nwk.c_str(), nwk.size() + 1);
  *kappa_out = kappa;
  return RC_OK;
}

int rc_fit_trees(const rc_block *blocks, int32_t n_blocks, char *newick_out, int32_t cap, float *kappa_out, int32_t threads) {
  if (!blocks || !newick_out || !kappa_out || n_blocks < 0 || cap < 8) return fail(RC_ERR_ARG, "bad argument");
  std::atomic<int> next{0}, done{0};
  unsigned nt = threads > 0 ? static_cast<unsigned>(threads) : static_cast<unsigned>(std::min(effective_cpus(), 32));   // the CPUs this process may use, not the host's
  nt = std::min<unsigned>(nt, static_cast<unsigned>(std::max(1, n_blocks)));
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n_blocks) break;
      char *dst = newick_out + static_cast<size_t>(i) * cap;
      dst[0] = 0; kappa_out[i] = 0.0f;
      const rc_block &b = blocks[i];
      if (!b.rows || !b.names || b.n_rows <= 2) continue;
      int L = 0;
      for (const char *p = b.rows[0]; *p; p++) L += (*p != '-');
      if (L < 3) continue;
      std::vector<std::string> rows(b.n_rows), names(b.n_rows);
      for (int r = 0; r < b.n_rows; r++) { rows[r] = b.rows[r]; names[r] = b.names[r]; }
      std::string nwk, err;
      float kappa = 0;
      if (!fit_tree(rows, names, nwk, kappa, nullptr, err) || static_cast<int>(nwk.size()) + 1 > cap) continue;
      std::memcpy(dst, nwk.c_str(), nwk.size() + 1);
      kappa_out[i] = kappa;
      done.fetch_add(1);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
  return done.load();
}

// The same fits on the GPU, one wavefront per block (rc_tree_kernel.hip).  Host work: pattern
// compression (threads) and writing the Newick text.
static constexpr int kTreeDeviceTips = 64;   // treefit::kMaxTipsDevice (rc_tree_core.h)

int rc_fit_trees_device(rc_ctx *c, const rc_block *blocks, int32_t n_blocks, char *newick_out, int32_t cap, float *kappa_out,
                        double *lnl_out) {
  if (!c || !blocks || !newick_out || !kappa_out || n_blocks < 0 || cap < 8) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  std::vector<PatternSet> ps(n_blocks);
  std::vector<char> ok(n_blocks, 0);
  {
    std::atomic<int> next{0};
    const unsigned nt = std::min<unsigned>(static_cast<unsigned>(c->hostThreads), static_cast<unsigned>(std::max(1, n_blocks)));
    auto work = [&]() {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n_blocks) break;
        const rc_block &b = blocks[i];
        if (!b.rows || !b.names || b.n_rows <= 2) continue;
        int L = 0;
        for (const char *p = b.rows[0]; *p; p++) L += (*p != '-');
        if (L < 3) continue;
        if (b.n_rows > kTreeDeviceTips) {   // more tips than the kernel's per-lane tables hold: the host estimator, here on this thread
          std::vector<std::string> rows(b.n_rows), names(b.n_rows);
          for (int r = 0; r < b.n_rows; r++) { rows[r] = b.rows[r]; names[r] = b.names[r]; }
          std::string nwk, err;
          float kappa = 0;
          double lnl = 0;
          char *dst = newick_out + static_cast<size_t>(i) * cap;
          if (fit_tree(rows, names, nwk, kappa, &lnl, err) && static_cast<int>(nwk.size()) + 1 <= cap) {
            std::memcpy(dst, nwk.c_str(), nwk.size() + 1);
            kappa_out[i] = kappa;
            if (lnl_out) lnl_out[i] = lnl;
            ok[i] = 2;
          }
          continue;
        }
        bool lengths = b.n_cols > 0;
        for (int r = 0; r < b.n_rows && lengths; r++)
          lengths = b.rows[r] && static_cast<int>(strnlen(b.rows[r], static_cast<size_t>(b.n_cols) + 1)) == b.n_cols;
        std::string err;
        ok[i] = (lengths && compress_patterns(b.rows, b.n_rows, b.n_cols, ps[i], err)) ? 1 : 0;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
  }
  // Jobs, the longest fits first (a launch hands its workgroups out in order, and a fit of 12 rows x 200 patterns takes a hundred
  // times one of 3 x 60), then grouped into launches by the LDS a fit needs: a block's whole working set -- tree, distance matrices,
  // masks, conditional-likelihood columns -- lives in its workgroup's LDS (rc_tree_kernel.hip), so the blocks of a launch are the ones
  // that fit the same number of times into a CU's 160 KB; blocks whose columns would leave fewer than two workgroups per CU keep them
  // in global memory (RC_TREE_LDS_MAX: the most LDS a fit may take, bytes).
  int hostDone = 0;
  std::vector<int> order(n_blocks);
  for (int i = 0; i < n_blocks; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    const auto cost = [&](int i) { return ok[i] == 1 ? static_cast<double>(ps[i].N) * ps[i].N * ps[i].P : -1.0; };
    return cost(x) > cost(y);
  });
  size_t ldsMax = std::min<size_t>(static_cast<size_t>(tree_fit_max_lds()), c->ldsPerCU / 2);
  if (const char *e = std::getenv("RC_TREE_LDS_MAX")) ldsMax = std::min<size_t>(static_cast<size_t>(tree_fit_max_lds()), static_cast<size_t>(std::max(0ll, std::atoll(e))));
  struct Launch { bool big; int occ; size_t lds; std::vector<int> blocks; };
  std::vector<Launch> launches;   // [0]: the big blocks; then by workgroups per CU, ascending
  launches.push_back(Launch{true, 0, 0, {}});
  for (int i : order) {
    if (ok[i] == 2) { hostDone++; continue; }   // fitted on the host above
    newick_out[static_cast<size_t>(i) * cap] = 0; kappa_out[i] = 0.0f;
    if (lnl_out) lnl_out[i] = 0.0;
    if (!ok[i]) continue;
    const size_t need = tree_fit_lds_bytes(ps[i].N, ps[i].P, false);
    if (need > ldsMax) {
      launches[0].blocks.push_back(i);
      launches[0].lds = std::max(launches[0].lds, tree_fit_lds_bytes(ps[i].N, ps[i].P, true));
      continue;
    }
    const int occ = static_cast<int>(std::min<size_t>(32, c->ldsPerCU / std::max<size_t>(need, 1)));
    size_t at = 1;
    while (at < launches.size() && launches[at].occ != occ) at++;
    if (at == launches.size()) launches.push_back(Launch{false, occ, 0, {}});
    launches[at].blocks.push_back(i);
    launches[at].lds = std::max(launches[at].lds, need);
  }
  std::sort(launches.begin() + 1, launches.end(), [](const Launch &a, const Launch &b) { return a.occ < b.occ; });
  std::vector<TreeJob> jobs;
  std::vector<int> owner;
  size_t work_doubles = 0, in_bytes = 0, res_doubles = 0;
  for (const Launch &L : launches)
    for (int i : L.blocks) {
      TreeJob j{};
      j.N = ps[i].N; j.P = ps[i].P;
      j.off_mask = in_bytes;
      in_bytes = (in_bytes + ps[i].mask.size() + 7) & ~static_cast<size_t>(7);
      j.off_w = in_bytes;
      in_bytes += sizeof(double) * ps[i].P;
      j.off_work = work_doubles;
      if (L.big) work_doubles += tree_work_doubles(j.N, j.P);
      j.off_out = res_doubles;
      res_doubles += tree_result_doubles(j.N);
      jobs.push_back(j);
      owner.push_back(i);
    }
  const int nj = static_cast<int>(jobs.size()), nBig = static_cast<int>(launches[0].blocks.size());
  if (nj == 0) return hostDone;
  if (nBig && launches[0].lds > static_cast<size_t>(tree_fit_max_lds())) return fail(RC_ERR_UNSUPPORTED, "a block's tree does not fit the device's LDS");
  // the column areas of the big blocks of one launch share a scratch of at most 8 GiB (RC_TREE_SCRATCH_BYTES): long batches
  // go in several launches on the one stream, which re-use it
  size_t capDoubles = (static_cast<size_t>(8) << 30) / sizeof(double);
  if (const char *e = std::getenv("RC_TREE_SCRATCH_BYTES")) capDoubles = std::max<size_t>(1, static_cast<size_t>(std::atoll(e)) / sizeof(double));
  std::vector<int> chunkStart{0};
  size_t maxChunk = 0;
  {
    size_t base = 0;
    for (int q = 0; q < nBig; q++) {
      const size_t end = (q + 1 < nBig) ? static_cast<size_t>(jobs[q + 1].off_work) : work_doubles;
      if (end - base > capDoubles && q > chunkStart.back()) { chunkStart.push_back(q); base = static_cast<size_t>(jobs[q].off_work); }
      maxChunk = std::max(maxChunk, end - base);
    }
    chunkStart.push_back(nBig);
    for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
      if (chunkStart[ch] >= nBig) break;
      const size_t b0 = static_cast<size_t>(jobs[chunkStart[ch]].off_work);
      for (int q = chunkStart[ch]; q < chunkStart[ch + 1]; q++) jobs[q].off_work -= b0;
    }
  }
  // device and pinned buffers live in the context: a driver fits its blocks in several calls
  std::lock_guard<std::mutex> treeLock(c->treeMutex);
  DevBuf &d_jobs = c->treeJobs, &d_in = c->treeIn, &d_work = c->treeWork, &d_res = c->treeRes;
  HIP_TRY(d_jobs.ensure(sizeof(TreeJob) * nj));
  HIP_TRY(d_in.ensure(in_bytes));
  HIP_TRY(d_work.ensure(sizeof(double) * std::max<size_t>(maxChunk, 1)));
  HIP_TRY(d_res.ensure(sizeof(double) * res_doubles));
  HIP_TRY(c->treeInPin.ensure(in_bytes));
  HIP_TRY(c->treeResPin.ensure(sizeof(double) * res_doubles));
  {   // masks and weights straight into pinned memory (threads), one copy
    uint8_t *in = c->treeInPin.as<uint8_t>();
    std::atomic<int> next{0};
    const unsigned nt = std::min<unsigned>(static_cast<unsigned>(c->hostThreads), static_cast<unsigned>(std::max(1, nj / 256)));
    auto fill = [&]() {
      for (;;) {
        const int q = next.fetch_add(1);
        if (q >= nj) break;
        const PatternSet &p = ps[owner[q]];
        std::memcpy(in + jobs[q].off_mask, p.mask.data(), p.mask.size());
        std::memcpy(in + jobs[q].off_w, p.w.data(), sizeof(double) * p.P);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::max(1u, nt); t++) th.emplace_back(fill);
    fill();
    for (auto &x : th) x.join();
  }
  RC_STREAM_TRY(ts, stream_tree(c));
  HIP_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), sizeof(TreeJob) * nj, hipMemcpyHostToDevice, ts));
  HIP_TRY(hipMemcpyAsync(d_in.p, c->treeInPin.p, in_bytes, hipMemcpyHostToDevice, ts));
  const TreeJob *dj = static_cast<const TreeJob *>(d_jobs.p);
  const uint8_t *din = static_cast<const uint8_t *>(d_in.p);
  for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
    const int q0 = chunkStart[ch], q1 = chunkStart[ch + 1];
    if (q1 > q0 && !launch_tree_fit(dj + q0, q1 - q0, true, launches[0].lds, din, static_cast<double *>(d_work.p), static_cast<double *>(d_res.p), ts))
      return fail(RC_ERR_DEVICE, "k_tree_fit: the device refused the launch's LDS request");
  }
  {
    int q0 = nBig;
    for (size_t l = 1; l < launches.size(); l++) {
      const int cnt = static_cast<int>(launches[l].blocks.size());
      if (cnt && !launch_tree_fit(dj + q0, cnt, false, launches[l].lds, din, static_cast<double *>(d_work.p), static_cast<double *>(d_res.p), ts))
        return fail(RC_ERR_DEVICE, "k_tree_fit: the device refused the launch's LDS request");
      q0 += cnt;
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->treeResPin.p, d_res.p, sizeof(double) * res_doubles, hipMemcpyDeviceToHost, ts));
  HIP_TRY(hipStreamSynchronize(ts));
  const double *res = c->treeResPin.as<double>();
  std::atomic<int> done{0};
  {
    std::atomic<int> next{0};
    const unsigned nt = std::min<unsigned>(static_cast<unsigned>(c->hostThreads), static_cast<unsigned>(std::max(1, nj / 64)));
    auto write = [&]() {
      std::vector<std::string> names;
      for (;;) {
        const int q = next.fetch_add(1);
        if (q >= nj) break;
        const int i = owner[q];
        const rc_block &b = blocks[i];
        names.resize(b.n_rows);
        for (int r = 0; r < b.n_rows; r++) names[r] = b.names[r];
        const int nn = 2 * jobs[q].N - 2;
        const double *rd = res + jobs[q].off_out;
        const int *ri = reinterpret_cast<const int *>(rd + 2 + nn);
        const std::string nwk = newick_of(b.n_rows, ri[0], ri + 1, ri + 1 + nn, rd + 2, names);
        if (static_cast<int>(nwk.size()) + 1 > cap) continue;
        std::memcpy(newick_out + static_cast<size_t>(i) * cap, nwk.c_str(), nwk.size() + 1);
        kappa_out[i] = static_cast<float>(rd[0]);
        if (lnl_out) lnl_out[i] = rd[1];
        done.fetch_add(1);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::max(1u, nt); t++) th.emplace_back(write);
    write();
    for (auto &x : th) x.join();
  }
  return done.load() + hostDone;
}

int rc_tree_lnl(const rc_block *blk, double *lnl_out) {
  if (!blk || !lnl_out || !blk->rows || !blk->names || !blk->newick) return fail(RC_ERR_ARG, "bad argument");
  std::vector<std::string> rows(blk->n_rows), names(blk->n_rows);
  for (int r = 0; r < blk->n_rows; r++) { rows[r] = blk->rows[r]; names[r] = blk->names[r]; }
  std::string err;
  if (!tree_lnl(rows, names, blk->newick, blk->kappa, lnl_out, err)) return fail(RC_ERR_ARG, err);
  return RC_OK;
}

float rc_pvalue(float score, float mu, float lambda) { return pvalue_of(score, mu, lambda); }

int rc_evd_fit(rc_ctx *c, const double *x, int32_t n, double *mu, double *lambda) {
  if (!c || !x || n < 1) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  RC_STREAM_TRY(aux, stream_aux(c));
  double *d_x = nullptr;
  FitOut *d_o = nullptr;
  HIP_TRY(hipMalloc(&d_x, sizeof(double) * n));
  HIP_TRY(hipMalloc(&d_o, sizeof(FitOut)));
  HIP_TRY(hipMemcpy(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice));
  launch_evd_fit_f64(d_x, n, d_o, aux);
  FitOut o;
  hipError_t e = hipStreamSynchronize(aux);
  if (e == hipSuccess) e = hipMemcpy(&o, d_o, sizeof o, hipMemcpyDeviceToHost);
  (void)hipFree(d_x); (void)hipFree(d_o);
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, hipGetErrorString(e));
  if (mu) *mu = o.mu;
  if (lambda) *lambda = o.lambda;
  return o.rc;
}

int rc_mt_stream(rc_ctx *c, uint32_t seed, uint32_t *out, int32_t n) {
  if (!c || !out || n < 1) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  RC_STREAM_TRY(aux, stream_aux(c));
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(uint32_t) * n));
  launch_mt_stream(seed, 1, n, d, aux);
  hipError_t e = hipStreamSynchronize(aux);
  if (e == hipSuccess) e = hipMemcpy(out, d, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, hipGetErrorString(e));
  return RC_OK;
}

}  // extern "C"
