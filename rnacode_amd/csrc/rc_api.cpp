// rc_api.cpp -- the C-ABI of include/rnacode_hip.h: contexts, batches resident in HBM, launches.
//
// No CPU implementation of the hot path lives here: scoring, sampling, scan and fit are
// kernels in rc_kernels.hip; this file prepares block constants (rc_host.cpp), moves bytes,
// launches, and formats results.  Every compute entry point fails with RC_ERR_DEVICE when no
// HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
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

// hipEvent that is destroyed on every exit path
struct Event {
  hipEvent_t e = nullptr;
  ~Event() { if (e) (void)hipEventDestroy(e); }
  hipError_t create() { return hipEventCreate(&e); }
  operator hipEvent_t() const { return e; }
};

struct rc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // native-block kernels: latency-bound, they run beside the null sampling
  std::vector<hipStream_t> classStreams;   // k_null launches of different row-count classes, when a batch has several
  int numCU = 0;
  CodeTables ct62{62}, ct90{90};
  PairTable pt62, pt90;
  uint8_t *d_pair62 = nullptr, *d_pair90 = nullptr;
  // MT19937 stream cache
  uint32_t *d_U = nullptr;
  size_t U_cap = 0;
  uint32_t U_seed = 0;
  int U_Spad = 0, U_D = 0;
  bool U_valid = false;
  // sigma-code staging for the null kernel
  uint32_t *d_scratch = nullptr;
  size_t scratch_cap = 0;
  unsigned long long *d_clamp = nullptr;
  unsigned int *d_work = nullptr;   // eight item counters (one per XCD queue) per row-count class: k_null pulls work items from them
  float tieThr = 0.0f;
  std::map<std::pair<int, size_t>, int> occ;   // (N-1, LDS bytes) -> resident workgroups per CU of k_null
  size_t ldsMaxBytes = 13312;   // per wavefront: >= 12 wavefronts per CU; measured crossover against reading the codes from L2 (cols 150: 52.0 vs 54.0 ms, cols 180: 41.5 vs 39.3 ms)
  int debugSkip = 0;
};

struct rc_batch {
  rc_ctx *ctx = nullptr;
  rc_params par{};
  int n = 0;
  std::vector<HostBlock> hb;
  std::vector<DevBlock> db;
  std::vector<int> okBlocks;                 // batch indices with status RC_OK
  std::map<int, std::vector<int>> classes;   // N-1 -> batch indices
  int maxL = 0, maxDraws = 0, hssCap = 8;
  // device
  uint8_t *d_blob = nullptr;
  size_t blobBytes = 0;
  DevBlock *d_dblocks = nullptr;
  int *d_okBlocks = nullptr;
  std::map<int, int *> d_classBlocks;
  float *d_maxima = nullptr;
  bool maximaExternal = false;
  float *d_S = nullptr;
  size_t *d_offS = nullptr;
  std::vector<size_t> offS;
  size_t S_floats = 0;
  DevHss *d_hss = nullptr;
  int *d_hssCount = nullptr;
  uint8_t *h_stage = nullptr;      // pinned: fit rows, HSS counts and offsets, clamp counter of one run
  size_t stageBytes = 0;
  DevHss *h_packed = nullptr;      // pinned: the packed HSS records (grown on demand)
  size_t packedCap = 0;
  DevHss *d_hssPacked = nullptr;   // the used records, packed for the copy to the host
  int *d_hssOffsets = nullptr;     // [n * 6] + 1: start of each slot in d_hssPacked; last entry = total
  FitOut *d_fit = nullptr;
  // host results
  bool ran = false;
  std::vector<float> maxima;
  bool maximaFetched = false;
  std::vector<FitOut> fit;
  std::vector<DevHss> hss;
  std::vector<int> hssCount;
  unsigned long long clamped = 0;
  float t[5] = {0, 0, 0, 0, 0};
  int nl[5] = {0, 0, 0, 0, 0};
};

extern "C" {

const char *rc_last_error(void) { return g_err.c_str(); }

void rc_default_params(rc_params *p) {
  p->Delta = -10.0f; p->Omega = -4.0f; p->omega = -2.0f;
  p->stopPenalty_0 = -9999.0f; p->stopPenalty_k = -8.0f;
  p->blosum = 62; p->sampleN = 100; p->cutoff = 1.0f; p->stopEarly = 0; p->seed_base = 42;
}

int rc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rc_ctx_create(int device, rc_ctx **out) {
  if (!out) return fail(RC_ERR_ARG, "null out pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(RC_ERR_DEVICE, "no HIP device available: the scoring path has no CPU fallback");
  if (device < 0 || device >= n) return fail(RC_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  rc_ctx *c = new rc_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->numCU = prop.multiProcessorCount;
  // non-blocking streams: no implicit ordering against the legacy default stream that a host framework in
  // the same process (torch, RCCL) may be using; the second stream gets the higher priority so that the
  // runtime puts it on a hardware queue of its own (two plain streams can end up sharing one, which
  // serialises the native-block kernels behind k_null again -- seen under torch.distributed)
  int prioLow = 0, prioHigh = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
  HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prioLow));
  if (std::getenv("RC_SERIAL_NATIVE")) c->stream2 = c->stream;   // profiling: one kernel at a time, clean per-kernel intervals
  else HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prioHigh));
  c->pt62.build(c->ct62);
  c->pt90.build(c->ct90);
  HIP_TRY(hipMalloc(&c->d_pair62, 4096));
  HIP_TRY(hipMalloc(&c->d_pair90, 4096));
  HIP_TRY(hipMemcpy(c->d_pair62, c->pt62.pair, 4096, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_pair90, c->pt90.pair, 4096, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&c->d_clamp, sizeof(unsigned long long)));
  HIP_TRY(hipMalloc(&c->d_work, sizeof(unsigned int) * (kMaxRows + 1) * 8));
  c->tieThr = float_threshold_lt(0.0001);
  if (const char *e = std::getenv("RC_LDS_MAX_BYTES")) c->ldsMaxBytes = static_cast<size_t>(std::atol(e));
  if (const char *e = std::getenv("RC_DEBUG_SKIP")) c->debugSkip = std::atoi(e);   // profiling only: results are wrong
  *out = c;
  return RC_OK;
}

void rc_ctx_destroy(rc_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->d_pair62) hipFree(c->d_pair62);
  if (c->d_pair90) hipFree(c->d_pair90);
  if (c->d_U) hipFree(c->d_U);
  if (c->d_scratch) hipFree(c->d_scratch);
  if (c->d_clamp) hipFree(c->d_clamp);
  if (c->d_work) hipFree(c->d_work);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->stream2 && c->stream2 != c->stream) hipStreamDestroy(c->stream2);
  for (hipStream_t st : c->classStreams) hipStreamDestroy(st);
  delete c;
}

void rc_batch_destroy(rc_batch *b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  if (b->d_blob) hipFree(b->d_blob);
  if (b->d_dblocks) hipFree(b->d_dblocks);
  if (b->d_okBlocks) hipFree(b->d_okBlocks);
  for (auto &kv : b->d_classBlocks) hipFree(kv.second);
  if (b->d_maxima && !b->maximaExternal) hipFree(b->d_maxima);
  if (b->d_S) hipFree(b->d_S);
  if (b->d_offS) hipFree(b->d_offS);
  if (b->d_hss) hipFree(b->d_hss);
  if (b->d_hssCount) hipFree(b->d_hssCount);
  if (b->d_hssPacked) hipFree(b->d_hssPacked);
  if (b->h_stage) hipHostFree(b->h_stage);
  if (b->h_packed) hipHostFree(b->h_packed);
  if (b->d_hssOffsets) hipFree(b->d_hssOffsets);
  if (b->d_fit) hipFree(b->d_fit);
  delete b;
}

static size_t align16(size_t x) { return (x + 15) & ~static_cast<size_t>(15); }

int rc_batch_create(rc_ctx *ctx, const rc_block *blocks, int32_t n_blocks, const rc_params *par, rc_batch **out) {
  if (!ctx || !blocks || !par || !out || n_blocks < 0) return fail(RC_ERR_ARG, "bad argument");
  if (par->blosum != 62 && par->blosum != 90) return fail(RC_ERR_ARG, "blosum must be 62 or 90");
  if (par->sampleN < 1) return fail(RC_ERR_ARG, "sampleN must be >= 1");
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  rc_batch *b = new rc_batch();
  b->ctx = ctx; b->par = *par; b->n = n_blocks;
  std::unique_ptr<rc_batch, void (*)(rc_batch *)> guard(b, rc_batch_destroy);   // released on success only
  b->hb.resize(n_blocks);
  const CodeTables &ct = (par->blosum == 90) ? ctx->ct90 : ctx->ct62;
  const PairTable &pt = (par->blosum == 90) ? ctx->pt90 : ctx->pt62;

  // block-constant preparation, blocks are independent -> host threads
  {
    std::atomic<int> next{0};
    std::vector<std::string> errs(n_blocks);
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n_blocks < 8) nt = 1;
    auto work = [&]() {
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n_blocks) break;
        prepare_block(blocks[i], *par, ct, pt, b->hb[i], errs[i]);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    for (int i = 0; i < n_blocks; i++)
      if (b->hb[i].status != RC_OK && b->hb[i].status != RC_ERR_SKIP) {
        int st = b->hb[i].status;
        std::string msg = "block " + std::to_string(i) + ": " + errs[i];
        return fail(st, msg);
      }
  }

  // pack the blob
  b->db.assign(n_blocks, DevBlock());
  std::vector<uint8_t> blob;
  {   // one allocation for the blob instead of repeated doubling (10k blocks: ~150 MB)
    size_t est = 0;
    for (int i = 0; i < n_blocks; i++) {
      const HostBlock &h = b->hb[i];
      if (h.status != RC_OK) continue;
      const size_t L1 = static_cast<size_t>(h.L) + 1, ncw = (static_cast<size_t>(h.NK) + 4) / 5;
      est += h.nodes.size() * sizeof(NodeRec) + kMaxRows + 2 * L1 * (2 + 8 + ncw * 4 + static_cast<size_t>(h.NK) * 4) +
             2 * static_cast<size_t>(h.NK) * kLutSize * 4 + 2 * static_cast<size_t>(h.N) * h.cols + 6 * ((L1 / 3 + 63) / 64) * 8 + 12 * 16;
    }
    blob.reserve(est + est / 16 + 4096);
  }
  b->offS.assign(n_blocks, 0);
  size_t sOff = 0;
  for (int i = 0; i < n_blocks; i++) {
    const HostBlock &h = b->hb[i];
    DevBlock &d = b->db[i];
    std::memset(&d, 0, sizeof d);
    d.status = h.status;
    d.out_index = static_cast<uint32_t>(i);
    if (h.status != RC_OK) continue;
    b->okBlocks.push_back(i);
    b->classes[h.NK].push_back(i);
    b->maxL = std::max(b->maxL, h.L);
    b->maxDraws = std::max(b->maxDraws, static_cast<int>(h.nodes.size()) * h.cols);
    d.N = h.N; d.NK = h.NK; d.cols = h.cols; d.L = h.L;
    d.nnodes = static_cast<int>(h.nodes.size());
    d.ref_start = h.ref_start; d.ref_length = h.ref_length;
    d.Delta = par->Delta; d.Omega = par->Omega; d.omega = par->omega;
    d.nkf = static_cast<float>(h.N - 1);
    d.stop0 = par->stopPenalty_0; d.stopk = par->stopPenalty_k;
    d.code_zero = pt.code_zero; d.code_stop0 = pt.code_stop0; d.code_stopk = pt.code_stopk;
    d.exact_div = h.exact_div;
    const int L1 = h.L + 1;
    auto put = [&](const void *p, size_t bytes) {
      size_t off = align16(blob.size());
      blob.resize(off + bytes);
      if (p) std::memcpy(blob.data() + off, p, bytes); else std::memset(blob.data() + off, 0, bytes);
      return static_cast<uint32_t>(off);
    };
    d.off_nodes = put(h.nodes.data(), h.nodes.size() * sizeof(NodeRec));
    {
      uint8_t qt[kMaxRows] = {0};   // k_null reads it as 32-bit words
      std::memcpy(qt, h.qtip.data(), h.qtip.size());
      d.off_qtip = put(qt, sizeof qt);
    }
    {
      std::vector<uint16_t> rc2(2 * L1);
      std::memcpy(rc2.data(), h.refcol[0].data(), L1 * 2);
      std::memcpy(rc2.data() + L1, h.refcol[1].data(), L1 * 2);
      d.off_refcol = put(rc2.data(), rc2.size() * 2);
      std::vector<uint64_t> z2(2 * L1);
      std::memcpy(z2.data(), h.zw[0].data(), L1 * 8);
      std::memcpy(z2.data() + L1, h.zw[1].data(), L1 * 8);
      d.off_zw = put(z2.data(), z2.size() * 8);
      {   // gap masks, five 6-bit fields per word like the sigma codes: [strand][i][word]
        const int ncw = (h.NK + 4) / 5;
        std::vector<uint32_t> mw(2 * static_cast<size_t>(L1) * ncw, 0);
        for (int s = 0; s < 2; s++)
          for (int k = 0; k < h.NK; k++)
            for (int p = 0; p < L1; p++)
              mw[(static_cast<size_t>(s) * L1 + p) * ncw + k / 5] |= static_cast<uint32_t>(h.mask[s][static_cast<size_t>(k) * L1 + p]) << (6 * (k % 5));
        d.off_mask = put(mw.data(), mw.size() * 4);
      }
      std::vector<float> l2(2 * static_cast<size_t>(h.NK) * kLutSize);
      std::memcpy(l2.data(), h.lut[0].data(), static_cast<size_t>(h.NK) * kLutSize * 4);
      std::memcpy(l2.data() + static_cast<size_t>(h.NK) * kLutSize, h.lut[1].data(), static_cast<size_t>(h.NK) * kLutSize * 4);
      d.off_lut = put(l2.data(), l2.size() * 4);
      std::vector<uint8_t> ch(2 * static_cast<size_t>(h.N) * h.cols);
      for (int s = 0; s < 2; s++)
        for (int r = 0; r < h.N; r++) std::memcpy(ch.data() + (static_cast<size_t>(s) * h.N + r) * h.cols, h.rows[s][r].data(), h.cols);
      d.off_chars = put(ch.data(), ch.size());
      d.off_sigma = put(nullptr, 2 * static_cast<size_t>(h.NK) * L1 * 4);
      {
        const int smax = h.L / 3, zmw = (smax + 63) / 64;
        std::vector<uint64_t> zm(static_cast<size_t>(6) * zmw, 0);
        for (int s = 0; s < 2; s++)
          for (int f = 0; f < 3; f++)
            for (int j = 0; j < (h.L - f) / 3; j++)
              if (h.zw[s][3 * j + 3 + f] != 0) zm[static_cast<size_t>(s * 3 + f) * zmw + (j >> 6)] |= 1ull << (j & 63);
        d.off_zmask = put(zm.data(), zm.size() * 8);
        d.zm_words = zmw;
      }
    }
    if (blob.size() > 0xFFFFFFF0ull) return fail(RC_ERR_UNSUPPORTED, "batch too large: split it (blob > 4 GiB)");
    const size_t smax = h.L / 3;
    b->offS[i] = sOff;
    sOff += 6 * smax * smax;
    b->hssCap = std::max(b->hssCap, static_cast<int>(smax) / 3 + 2);
  }
  b->S_floats = sOff;
  b->blobBytes = blob.size();

  // upload
  auto up = [&](void **dst, const void *src, size_t bytes) -> int {
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(dst, bytes));
    if (src) HIP_TRY(hipMemcpy(*dst, src, std::max<size_t>(bytes, 1), hipMemcpyHostToDevice));
    return RC_OK;
  };
  int rcv;
  std::vector<uint8_t> pad(16, 0);
  if ((rcv = up(reinterpret_cast<void **>(&b->d_blob), blob.empty() ? pad.data() : blob.data(), blob.empty() ? 16 : blob.size()))) return rcv;
  if (n_blocks > 0) {
    if ((rcv = up(reinterpret_cast<void **>(&b->d_dblocks), b->db.data(), sizeof(DevBlock) * n_blocks))) return rcv;
    if ((rcv = up(reinterpret_cast<void **>(&b->d_offS), b->offS.data(), sizeof(size_t) * n_blocks))) return rcv;
  }
  if (!b->okBlocks.empty()) {
    if ((rcv = up(reinterpret_cast<void **>(&b->d_okBlocks), b->okBlocks.data(), sizeof(int) * b->okBlocks.size()))) return rcv;
    for (auto &kv : b->classes) {
      // longest blocks first: k_null's workgroups pull items from a shared counter, so the order of the
      // list is the order of dispatch (longest-processing-time-first keeps the tail short)
      std::stable_sort(kv.second.begin(), kv.second.end(), [&](int x, int y) { return b->hb[x].L > b->hb[y].L; });
      int *p = nullptr;
      if ((rcv = up(reinterpret_cast<void **>(&p), kv.second.data(), sizeof(int) * kv.second.size()))) return rcv;
      b->d_classBlocks[kv.first] = p;
    }
  }
  const size_t nm = static_cast<size_t>(std::max(n_blocks, 1)) * par->sampleN;
  HIP_TRY(hipMalloc(&b->d_maxima, nm * sizeof(float)));
  HIP_TRY(hipMalloc(&b->d_S, std::max<size_t>(b->S_floats, 4) * sizeof(float)));
  HIP_TRY(hipMalloc(&b->d_hss, static_cast<size_t>(std::max(n_blocks, 1)) * 6 * b->hssCap * sizeof(DevHss)));
  HIP_TRY(hipMalloc(&b->d_hssCount, static_cast<size_t>(std::max(n_blocks, 1)) * 6 * sizeof(int)));
  HIP_TRY(hipMalloc(&b->d_hssPacked, static_cast<size_t>(std::max(n_blocks, 1)) * 6 * b->hssCap * sizeof(DevHss)));
  HIP_TRY(hipMalloc(&b->d_hssOffsets, (static_cast<size_t>(std::max(n_blocks, 1)) * 6 + 1) * sizeof(int)));
  HIP_TRY(hipMalloc(&b->d_fit, static_cast<size_t>(std::max(n_blocks, 1)) * sizeof(FitOut)));
  *out = guard.release();
  return RC_OK;
}

static int ensure_stream(rc_ctx *c, uint32_t seed, int Spad, int D, float *ms, int *launches) {
  *ms = 0; *launches = 0;
  if (c->U_valid && c->U_seed == seed && c->U_Spad == Spad && c->U_D >= D) return RC_OK;
  const size_t need = static_cast<size_t>(Spad) * D;
  if (need > c->U_cap) {
    if (c->d_U) hipFree(c->d_U);
    c->d_U = nullptr; c->U_cap = 0;
    HIP_TRY(hipMalloc(&c->d_U, need * sizeof(uint32_t)));
    c->U_cap = need;
  }
  Event e0, e1;
  HIP_TRY(e0.create()); HIP_TRY(e1.create());
  HIP_TRY(hipEventRecord(e0, c->stream));
  launch_mt_stream(seed, Spad, D, c->d_U, c->stream);
  HIP_TRY(hipEventRecord(e1, c->stream));
  HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  *launches = 1;
  c->U_seed = seed; c->U_Spad = Spad; c->U_D = D; c->U_valid = true;
  return RC_OK;
}

// When set, the MT19937 stream is regenerated on every run (bench: the stream belongs to a step).
static bool g_no_stream_cache = false;
void rc_set_stream_cache(int enabled) { g_no_stream_cache = !enabled; }

// One run's launches, in the order rc_batch_run issues them.  ev[] are the run's timing events:
// 0 start, 1..2 native block (second stream), 3..4 null sampling, 5 after the fit.
struct RunEnv {
  rc_batch *b;
  rc_ctx *c;
  int n, sampleN, groups, Spad, slots;
  const uint8_t *d_pair;
  Event *ev;
};

static int launch_native_block(const RunEnv &R) {
  rc_batch *b = R.b; rc_ctx *c = R.c; Event *ev = R.ev;
  const int n = R.n, slots = R.slots;
  const uint8_t *d_pair = R.d_pair;
  // ---- native block: sigma -> DP per N-1 class -> scan, on the second stream: these kernels are small and
  // latency-bound, the null sampling below fills the chip beside them; the fit waits for both
  HIP_TRY(hipEventRecord(ev[1], c->stream2));
  NativeArgs na{};
  na.blob = b->d_blob; na.dblocks = b->d_dblocks; na.blocks = b->d_okBlocks; na.pair = d_pair;
  na.S = b->d_S; na.offS = b->d_offS; na.hss = b->d_hss; na.hssCount = b->d_hssCount; na.hssCap = b->hssCap; na.tieThr = c->tieThr;
  HIP_TRY(hipMemsetAsync(b->d_hssCount, 0, static_cast<size_t>(n) * 6 * sizeof(int), c->stream2));
  launch_native_sigma(na, static_cast<int>(b->okBlocks.size()), c->stream2); b->nl[3]++;
  for (auto &kv : b->classes) {
    NativeArgs nc = na;
    nc.blocks = b->d_classBlocks[kv.first];
    if (!launch_native_dp(kv.first, nc, static_cast<int>(kv.second.size()), c->stream2))
      return fail(RC_ERR_UNSUPPORTED, "no native DP kernel for this number of rows");
    b->nl[3]++;
  }
  launch_native_scan(na, static_cast<int>(b->okBlocks.size()), c->stream2); b->nl[3]++;
  HIP_TRY(hipMemsetAsync(b->d_hssOffsets + slots, 0, sizeof(int), c->stream2));
  launch_hss_pack(b->d_hss, b->d_hssCount, b->hssCap, slots, b->d_hssPacked, b->d_hssOffsets, b->d_hssOffsets + slots, c->stream2);
  HIP_TRY(hipEventRecord(ev[2], c->stream2));

  return RC_OK;
}

static int launch_null_sampling(const RunEnv &R) {
  rc_batch *b = R.b; rc_ctx *c = R.c; Event *ev = R.ev;
  const int n = R.n, sampleN = R.sampleN, groups = R.groups, Spad = R.Spad;
  const uint8_t *d_pair = R.d_pair;
  // ---- null sampling per N-1 class
  HIP_TRY(hipMemsetAsync(c->d_clamp, 0, sizeof(unsigned long long), c->stream));
  HIP_TRY(hipMemsetAsync(c->d_work, 0, sizeof(unsigned int) * (kMaxRows + 1) * 8, c->stream));
  // -1.0f everywhere first: skipped blocks report "no HSS" rows
  HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->d_maxima), 0xBF800000u, static_cast<size_t>(n) * sampleN, c->stream));
  HIP_TRY(hipEventRecord(ev[3], c->stream));
  struct Plan { int NK, grid, comboSplit; size_t stride, ldsBytes, need; bool exact; };
  std::vector<Plan> plans;
  size_t needSum = 0, needMax = 0;
  for (auto &kv : b->classes) {
    Plan p{};
    p.NK = kv.first;
    int maxLc = 0;
    for (int bi : kv.second) maxLc = std::max(maxLc, b->hb[bi].L);
    const int NCW = (p.NK + 4) / 5;   // 5 six-bit sigma codes per word
    p.stride = static_cast<size_t>(2) * (maxLc + 1) * NCW * kWave;
    // sigma codes of one strand x frame are staged in LDS when that still leaves >= 12 wavefronts per
    // CU (RC_LDS_MAX_BYTES overrides the per-wavefront budget); otherwise they are read from the
    // per-workgroup scratch in global memory.
    p.ldsBytes = static_cast<size_t>(maxLc / 3) * NCW * kWave * sizeof(uint32_t);
    for (int bi : kv.second) p.exact = p.exact || b->hb[bi].exact_div;
    if (p.ldsBytes > c->ldsMaxBytes || p.exact) p.ldsBytes = 0;
    else p.ldsBytes = std::max<size_t>(p.ldsBytes, kPhaseALds);   // phase A keeps the tree's threshold table and the pair table there
    const std::pair<int, size_t> key(p.NK, p.ldsBytes);
    if (!c->occ.count(key)) c->occ[key] = null_occupancy(p.NK, p.ldsBytes);
    if (c->occ[key] <= 0 && p.ldsBytes) { p.ldsBytes = 0; const std::pair<int, size_t> k2(p.NK, 0); if (!c->occ.count(k2)) c->occ[k2] = null_occupancy(p.NK, 0); }
    const int occ = std::max(1, c->occ[std::pair<int, size_t>(p.NK, p.ldsBytes)]);
    int items = static_cast<int>(kv.second.size()) * groups;
    // small batches: split every item into its six strand x frame parts to fill the chip
    p.comboSplit = (items * 2 <= c->numCU * occ) ? 1 : 0;
    if (p.comboSplit) items *= 6;
    p.grid = std::min(items, c->numCU * occ);
    // very long blocks: bound the sigma-code staging area (4 GiB of uint32) by running fewer workgroups
    p.grid = static_cast<int>(std::max<size_t>(1, std::min<size_t>(p.grid, (static_cast<size_t>(1) << 30) / std::max<size_t>(p.stride, 1))));
    p.need = p.stride * p.grid;
    needSum += p.need; needMax = std::max(needMax, p.need);
    plans.push_back(p);
  }
  // Several row-count classes: their launches are independent (own work queues, own part of the staging
  // scratch), so they go on separate streams and share the chip -- a small class no longer waits for the tail
  // of the previous one.  One class, a scratch that would exceed 4 GiB, or RC_SERIAL_NATIVE: main stream only.
  const bool concurrent = plans.size() > 1 && needSum <= (static_cast<size_t>(1) << 30) && c->stream2 != c->stream;
  const size_t need = concurrent ? needSum : needMax;
  if (need > c->scratch_cap) {
    HIP_TRY(hipDeviceSynchronize());
    if (c->d_scratch) hipFree(c->d_scratch);
    c->d_scratch = nullptr; c->scratch_cap = 0;
    HIP_TRY(hipMalloc(&c->d_scratch, need * sizeof(uint32_t)));
    c->scratch_cap = need;
  }
  std::vector<Event> classDone(concurrent ? plans.size() : 0);
  size_t scratchOff = 0;
  for (size_t pi = 0; pi < plans.size(); pi++) {
    const Plan &p = plans[pi];
    const auto &members = b->classes[p.NK];
    hipStream_t st = c->stream;
    if (concurrent) {
      while (c->classStreams.size() < std::min<size_t>(plans.size(), 8)) {
        hipStream_t ns = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        c->classStreams.push_back(ns);
      }
      st = c->classStreams[pi % c->classStreams.size()];
      HIP_TRY(hipStreamWaitEvent(st, ev[3], 0));   // the memsets above
    }
    NullArgs a{};
    a.blob = b->d_blob; a.dblocks = b->d_dblocks; a.classBlocks = b->d_classBlocks[p.NK];
    a.nClassBlocks = static_cast<int>(members.size()); a.groups = groups; a.sampleN = sampleN; a.Spad = Spad;
    a.U = c->d_U; a.pair = d_pair; a.scratch = c->d_scratch + scratchOff; a.scratchStride = p.stride;
    a.maxima = b->d_maxima; a.clampCount = c->d_clamp; a.tieThr = c->tieThr; a.debugSkip = c->debugSkip; a.comboSplit = p.comboSplit;
    a.workCounter = c->d_work + p.NK * 8;
    if (!launch_null(p.NK, a, p.grid, p.ldsBytes, p.exact, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
    b->nl[2]++;
    if (concurrent) {
      scratchOff += p.need;
      HIP_TRY(classDone[pi].create());
      HIP_TRY(hipEventRecord(classDone[pi], st));
      HIP_TRY(hipStreamWaitEvent(c->stream, classDone[pi], 0));
    }
  }
  HIP_TRY(hipEventRecord(ev[4], c->stream));
  return RC_OK;
}

static int fetch_results(const RunEnv &R) {
  rc_batch *b = R.b; rc_ctx *c = R.c; Event *ev = R.ev;
  const int n = R.n, slots = R.slots;
  // small results back to the host: four copies into one pinned staging area behind the fit, one wait
  const size_t szFit = sizeof(FitOut) * n, szCnt = sizeof(int) * static_cast<size_t>(slots), szOff = sizeof(int) * (static_cast<size_t>(slots) + 1);
  const size_t oFit = 0, oCnt = (oFit + szFit + 15) & ~static_cast<size_t>(15), oOff = (oCnt + szCnt + 15) & ~static_cast<size_t>(15),
               oClamp = (oOff + szOff + 15) & ~static_cast<size_t>(15), total = oClamp + 16;
  if (total > b->stageBytes) {
    if (b->h_stage) hipHostFree(b->h_stage);
    b->h_stage = nullptr; b->stageBytes = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_stage), total, hipHostMallocDefault));
    b->stageBytes = total;
  }
  HIP_TRY(hipMemcpyAsync(b->h_stage + oFit, b->d_fit, szFit, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(b->h_stage + oCnt, b->d_hssCount, szCnt, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(b->h_stage + oOff, b->d_hssOffsets, szOff, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(b->h_stage + oClamp, c->d_clamp, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipGetLastError());
  float ms;
  HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); b->t[3] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ev[3], ev[4])); b->t[2] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ev[4], ev[5])); b->t[4] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[5])); b->t[0] = ms;
  {
    const FitOut *f = reinterpret_cast<const FitOut *>(b->h_stage + oFit);
    for (int bi : b->okBlocks) b->fit[bi] = f[bi];
    std::memcpy(b->hssCount.data(), b->h_stage + oCnt, szCnt);
    const int *off = reinterpret_cast<const int *>(b->h_stage + oOff);
    std::memcpy(&b->clamped, b->h_stage + oClamp, sizeof(unsigned long long));
    // the used HSS records only (k_hss_pack): that many records, second and last wait
    const size_t nrec = static_cast<size_t>(std::max(off[slots], 0));
    if (nrec > b->packedCap) {
      if (b->h_packed) hipHostFree(b->h_packed);
      b->h_packed = nullptr; b->packedCap = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_packed), sizeof(DevHss) * (nrec + nrec / 4 + 64), hipHostMallocDefault));
      b->packedCap = nrec + nrec / 4 + 64;
    }
    if (nrec) {
      HIP_TRY(hipMemcpyAsync(b->h_packed, b->d_hssPacked, sizeof(DevHss) * nrec, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int t = 0; t < slots; t++) {
      const int cnt = std::min(b->hssCount[t], b->hssCap);
      for (int i = 0; i < cnt; i++) b->hss[static_cast<size_t>(t) * b->hssCap + i] = b->h_packed[static_cast<size_t>(off[t]) + i];
    }
  }
  return RC_OK;
}

int rc_batch_run(rc_batch *b) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  rc_ctx *c = b->ctx;
  HIP_TRY(hipSetDevice(c->device));
  std::fill(b->t, b->t + 5, 0.0f);
  std::fill(b->nl, b->nl + 5, 0);
  b->ran = false; b->maximaFetched = false;
  const int n = b->n, sampleN = b->par.sampleN;
  const int groups = (sampleN + kWave - 1) / kWave, Spad = groups * kWave;
  b->fit.assign(n, FitOut{0, 0, -1, 0});
  b->hss.resize(static_cast<size_t>(n) * 6 * b->hssCap);   // entries beyond hssCount[slot] are never read
  b->hssCount.assign(static_cast<size_t>(n) * 6, 0);
  b->clamped = 0;
  if (b->okBlocks.empty()) { b->ran = true; return RC_OK; }
  const uint8_t *d_pair = (b->par.blosum == 90) ? c->d_pair90 : c->d_pair62;

  Event ev[6];
  for (auto &e : ev) HIP_TRY(e.create());
  HIP_TRY(hipEventRecord(ev[0], c->stream));

  if (g_no_stream_cache) c->U_valid = false;
  int rcv = ensure_stream(c, b->par.seed_base, Spad, b->maxDraws, &b->t[1], &b->nl[1]);
  if (rcv) return rcv;

  const RunEnv R{b, c, n, sampleN, groups, Spad, n * 6, d_pair, ev};
  if ((rcv = launch_native_block(R))) return rcv;
  if ((rcv = launch_null_sampling(R))) return rcv;

  // ---- EVD fit
  HIP_TRY(hipStreamWaitEvent(c->stream, ev[2], 0));   // native HSS of the second stream
  FitArgs fa{};
  fa.dblocks = b->d_dblocks; fa.blocks = b->d_okBlocks; fa.maxima = b->d_maxima; fa.hss = b->d_hss;
  fa.hssCount = b->d_hssCount; fa.hssCap = b->hssCap; fa.out = b->d_fit; fa.sampleN = sampleN; fa.stopEarly = b->par.stopEarly;
  fa.stopCutoff = static_cast<int>(b->par.cutoff * b->par.sampleN);   // score.c:992
  launch_evd_fit(fa, static_cast<int>(b->okBlocks.size()), c->stream); b->nl[4]++;
  HIP_TRY(hipEventRecord(ev[5], c->stream));

  if ((rcv = fetch_results(R))) return rcv;
  b->ran = true;
  return RC_OK;
}

int rc_batch_bind_maxima(rc_batch *b, void *device_ptr) {
  if (!b || !device_ptr) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(b->ctx->device));
  if (b->d_maxima && !b->maximaExternal) hipFree(b->d_maxima);
  b->d_maxima = static_cast<float *>(device_ptr);
  b->maximaExternal = true;
  b->ran = false;
  return RC_OK;
}

int rc_batch_work(const rc_batch *b, int64_t *sa, int64_t *cs) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  int64_t a = 0, c = 0;
  for (int bi : b->okBlocks) { a += b->par.sampleN; c += static_cast<int64_t>(b->hb[bi].cols) * b->par.sampleN; }
  if (sa) *sa = a;
  if (cs) *cs = c;
  return RC_OK;
}

int rc_batch_timing(const rc_batch *b, float t[5], int32_t nl[5]) {
  if (!b || !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  for (int i = 0; i < 5; i++) { if (t) t[i] = b->t[i]; if (nl) nl[i] = b->nl[i]; }
  return RC_OK;
}

static int check_blk(const rc_batch *b, int blk, bool needRun) {
  if (!b) return fail(RC_ERR_ARG, "null batch");
  if (blk < 0 || blk >= b->n) return fail(RC_ERR_ARG, "block index out of range");
  if (needRun && !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  return RC_OK;
}

int rc_batch_status(const rc_batch *b, int32_t blk) {
  int r = check_blk(b, blk, false);
  return r ? r : b->hb[blk].status;
}

int rc_batch_models(const rc_batch *b, int32_t blk, rc_model *fwd, rc_model *rev) {
  int r = check_blk(b, blk, false);
  if (r) return r;
  const HostBlock &h = b->hb[blk];
  if (h.status != RC_OK) return h.status;
  if (fwd) std::memcpy(fwd, h.models[0].data(), sizeof(rc_model) * h.N);
  if (rev) std::memcpy(rev, h.models[1].data(), sizeof(rc_model) * h.N);
  return RC_OK;
}

static int fetch_maxima(rc_batch *b) {
  if (b->maximaFetched) return RC_OK;
  b->maxima.resize(static_cast<size_t>(b->n) * b->par.sampleN);
  if (!b->maxima.empty()) HIP_TRY(hipMemcpy(b->maxima.data(), b->d_maxima, b->maxima.size() * sizeof(float), hipMemcpyDeviceToHost));
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
  if (!b || !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  int r = fetch_maxima(b);
  if (r) return r;
  std::memcpy(out, b->maxima.data(), b->maxima.size() * sizeof(float));
  return RC_OK;
}

int rc_batch_fit(const rc_batch *b, int32_t blk, int32_t *evd_rc, float *mu, float *lambda) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  if (b->hb[blk].status != RC_OK) return b->hb[blk].status;
  const FitOut &f = b->fit[blk];
  if (evd_rc) *evd_rc = f.rc;
  if (mu) *mu = static_cast<float>(f.mu);          // *parMu = mu (double -> float), score.c:1051
  if (lambda) *lambda = static_cast<float>(f.lambda);
  return RC_OK;
}

int rc_batch_fit_all(const rc_batch *b, float *out) {
  if (!b || !out || !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  for (size_t i = 0; i < b->hb.size(); i++) {
    float *o = out + 4 * i;
    if (b->hb[i].status != RC_OK) { o[0] = static_cast<float>(b->hb[i].status); o[1] = o[2] = o[3] = 0.0f; continue; }
    const FitOut &f = b->fit[i];
    o[0] = static_cast<float>(f.rc); o[1] = static_cast<float>(f.mu); o[2] = static_cast<float>(f.lambda);
    o[3] = static_cast<float>(f.better);
  }
  return RC_OK;
}

int rc_batch_hss(const rc_batch *b, int32_t blk, rc_hss *out, int32_t cap) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const HostBlock &h = b->hb[blk];
  if (h.status != RC_OK) return h.status;
  const FitOut &f = b->fit[blk];
  const float mu = static_cast<float>(f.mu), lambda = static_cast<float>(f.lambda);
  std::vector<rc_hss> all;
  for (int combo = 0; combo < 6; combo++) {   // '+' hits then '-' hits, frames ascending (score.c:1107-1127)
    const int cnt = b->hssCount[static_cast<size_t>(blk) * 6 + combo];
    if (cnt > b->hssCap) return fail(RC_ERR_UNSUPPORTED, "HSS buffer overflow");
    for (int i = 0; i < cnt; i++) {
      const DevHss &d = b->hss[(static_cast<size_t>(blk) * 6 + combo) * b->hssCap + i];
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
  if (!b || !offsets || (!out && cap > 0) || !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  int64_t total = 0;
  std::vector<rc_hss> tmp;
  for (int blk = 0; blk < b->n; blk++) {
    offsets[blk] = total;
    if (b->hb[blk].status != RC_OK) continue;
    int n = 0;
    for (int combo = 0; combo < 6; combo++) n += std::min(b->hssCount[static_cast<size_t>(blk) * 6 + combo], b->hssCap);
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
  if (!b || !b->ran) return fail(RC_ERR_ARG, "batch has not been run");
  *count = static_cast<int64_t>(b->clamped);
  return RC_OK;
}

int rc_batch_native_S(const rc_batch *b, int32_t blk, int32_t strand, int32_t frame, float *out, int32_t cap) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const HostBlock &h = b->hb[blk];
  if (h.status != RC_OK) return h.status;
  if (strand < 0 || strand > 1 || frame < 0 || frame > 2) return fail(RC_ERR_ARG, "bad strand/frame");
  const int sites = (h.L - frame) / 3, smax = h.L / 3;
  if (cap < sites * sites) return fail(RC_ERR_ARG, "output too small");
  HIP_TRY(hipSetDevice(b->ctx->device));
  if (sites > 0)
    HIP_TRY(hipMemcpy(out, b->d_S + b->offS[blk] + static_cast<size_t>(strand * 3 + frame) * smax * smax,
                      sizeof(float) * sites * sites, hipMemcpyDeviceToHost));
  for (int a = 0; a < sites; a++)
    for (int j = 0; j < a; j++) out[a * sites + j] = 0.0f;
  return sites;
}

int rc_batch_backtrack(const rc_batch *b, int32_t blk, int32_t strand, int32_t opt_b, int32_t opt_i,
                       int32_t *states, int32_t *zout, int32_t *transitions) {
  int r = check_blk(b, blk, true);
  if (r) return r;
  const HostBlock &h = b->hb[blk];
  if (h.status != RC_OK) return h.status;
  if (strand < 0 || strand > 1 || opt_b < 1 || opt_i > h.L || opt_i < opt_b + 2 || (opt_i - opt_b - 2) % 3 != 0)
    return fail(RC_ERR_ARG, "bad backtrack range");
  rc_ctx *c = b->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int steps = (opt_i - (opt_b + 2)) / 3 + 1, NK = h.NK;
  float *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, sizeof(float) * NK * 3 * steps));
  launch_sk_row(b->d_blob, b->d_dblocks, blk, strand, opt_b, opt_i, d_out, steps, c->stream);
  std::vector<float> sk(static_cast<size_t>(NK) * 3 * steps);
  hipError_t e = hipStreamSynchronize(c->stream);   // the streams are non-blocking: a default-stream copy would not wait
  if (e == hipSuccess) e = hipMemcpy(sk.data(), d_out, sk.size() * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(d_out);
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
      const int zc = static_cast<int>((h.zw[strand][i] >> (2 * (k - 1))) & 3);
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
  unsigned nt = threads > 0 ? static_cast<unsigned>(threads) : std::max(1u, std::thread::hardware_concurrency());
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
int rc_fit_trees_device(rc_ctx *c, const rc_block *blocks, int32_t n_blocks, char *newick_out, int32_t cap, float *kappa_out,
                        double *lnl_out) {
  if (!c || !blocks || !newick_out || !kappa_out || n_blocks < 0 || cap < 8) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  std::vector<PatternSet> ps(n_blocks);
  std::vector<char> ok(n_blocks, 0);
  {
    std::atomic<int> next{0};
    const unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), static_cast<unsigned>(std::max(1, n_blocks)));
    auto work = [&]() {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n_blocks) break;
        const rc_block &b = blocks[i];
        if (!b.rows || !b.names || b.n_rows <= 2) continue;
        int L = 0;
        for (const char *p = b.rows[0]; *p; p++) L += (*p != '-');
        if (L < 3) continue;
        std::vector<std::string> rows(b.n_rows);
        for (int r = 0; r < b.n_rows; r++) rows[r] = b.rows[r];
        std::string err;
        ok[i] = compress_patterns(rows, ps[i], err) ? 1 : 0;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
  }
  std::vector<TreeJob> jobs;
  std::vector<int> owner;
  std::vector<uint8_t> in;
  size_t work_doubles = 0;
  for (int i = 0; i < n_blocks; i++) {
    newick_out[static_cast<size_t>(i) * cap] = 0; kappa_out[i] = 0.0f;
    if (lnl_out) lnl_out[i] = 0.0;
    if (!ok[i]) continue;
    TreeJob j{};
    j.N = ps[i].N; j.P = ps[i].P;
    j.off_mask = in.size();
    in.insert(in.end(), ps[i].mask.begin(), ps[i].mask.end());
    in.resize((in.size() + 7) & ~static_cast<size_t>(7));
    j.off_w = in.size();
    const uint8_t *wp = reinterpret_cast<const uint8_t *>(ps[i].w.data());
    in.insert(in.end(), wp, wp + sizeof(double) * ps[i].P);
    j.off_work = work_doubles;
    work_doubles += 2 * static_cast<size_t>(j.N) * j.N + 3 * static_cast<size_t>(2 * j.N - 2) * 4 * j.P;
    jobs.push_back(j);
    owner.push_back(i);
  }
  const int nj = static_cast<int>(jobs.size());
  if (nj == 0) return 0;
  // the per-block work areas of one launch share a scratch of at most 8 GiB (RC_TREE_SCRATCH_BYTES): long batches
  // go in several launches on the one stream, which re-use it
  size_t capDoubles = (static_cast<size_t>(8) << 30) / sizeof(double);
  if (const char *e = std::getenv("RC_TREE_SCRATCH_BYTES")) capDoubles = std::max<size_t>(1, static_cast<size_t>(std::atoll(e)) / sizeof(double));
  std::vector<int> chunkStart{0};
  size_t maxChunk = 0;
  {
    size_t base = 0;
    for (int q = 0; q < nj; q++) {
      const size_t end = (q + 1 < nj) ? static_cast<size_t>(jobs[q + 1].off_work) : work_doubles;
      if (end - base > capDoubles && q > chunkStart.back()) { chunkStart.push_back(q); base = static_cast<size_t>(jobs[q].off_work); }
      maxChunk = std::max(maxChunk, end - base);
    }
    chunkStart.push_back(nj);
    for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
      const size_t b0 = static_cast<size_t>(jobs[chunkStart[ch]].off_work);
      for (int q = chunkStart[ch]; q < chunkStart[ch + 1]; q++) jobs[q].off_work -= b0;
    }
  }
  struct DevMem { void *p = nullptr; ~DevMem() { if (p) hipFree(p); } } d_jobs, d_in, d_work, d_out;
  HIP_TRY(hipMalloc(&d_jobs.p, sizeof(TreeJob) * nj));
  HIP_TRY(hipMalloc(&d_in.p, in.size()));
  HIP_TRY(hipMalloc(&d_work.p, sizeof(double) * maxChunk));
  HIP_TRY(hipMalloc(&d_out.p, sizeof(TreeOut) * nj));
  HIP_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), sizeof(TreeJob) * nj, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_in.p, in.data(), in.size(), hipMemcpyHostToDevice, c->stream));
  for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
    const int q0 = chunkStart[ch], q1 = chunkStart[ch + 1];
    launch_tree_fit(static_cast<const TreeJob *>(d_jobs.p) + q0, q1 - q0, static_cast<const uint8_t *>(d_in.p), static_cast<double *>(d_work.p),
                    static_cast<TreeOut *>(d_out.p) + q0, c->stream);
  }
  HIP_TRY(hipGetLastError());
  std::vector<TreeOut> out(nj);
  HIP_TRY(hipMemcpyAsync(out.data(), d_out.p, sizeof(TreeOut) * nj, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int done = 0;
  for (int q = 0; q < nj; q++) {
    const int i = owner[q];
    const rc_block &b = blocks[i];
    std::vector<std::string> names(b.n_rows);
    for (int r = 0; r < b.n_rows; r++) names[r] = b.names[r];
    const TreeOut &o = out[q];
    const std::string nwk = newick_of(b.n_rows, o.root, o.nchild, o.child, o.len, names);
    if (static_cast<int>(nwk.size()) + 1 > cap) continue;
    std::memcpy(newick_out + static_cast<size_t>(i) * cap, nwk.c_str(), nwk.size() + 1);
    kappa_out[i] = static_cast<float>(o.kappa);
    if (lnl_out) lnl_out[i] = o.lnl;
    done++;
  }
  return done;
}

int rc_tree_lnl(const rc_block *blk, double *lnl_out) {
  if (!blk || !lnl_out || !blk->rows || !blk->names || !blk->newick) return fail(RC_ERR_ARG, "bad argument");
  std::vector<std::string> rows(blk->n_rows), names(blk->n_rows);
  for (int r = 0; r < blk->n_rows; r++) { rows[r] = blk->rows[r]; names[r] = blk->names[r]; }
  std::string err;
  if (!tree_lnl(rows, names, blk->newick, blk->kappa, lnl_out, err)) return fail(RC_ERR_ARG, err);
  return RC_OK;
}

int rc_evd_fit(rc_ctx *c, const double *x, int32_t n, double *mu, double *lambda) {
  if (!c || !x || n < 1) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  double *d_x = nullptr;
  FitOut *d_o = nullptr;
  HIP_TRY(hipMalloc(&d_x, sizeof(double) * n));
  HIP_TRY(hipMalloc(&d_o, sizeof(FitOut)));
  HIP_TRY(hipMemcpy(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice));
  launch_evd_fit_f64(d_x, n, d_o, c->stream);
  FitOut o;
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(&o, d_o, sizeof o, hipMemcpyDeviceToHost);
  hipFree(d_x); hipFree(d_o);
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, hipGetErrorString(e));
  if (mu) *mu = o.mu;
  if (lambda) *lambda = o.lambda;
  return o.rc;
}

int rc_mt_stream(rc_ctx *c, uint32_t seed, uint32_t *out, int32_t n) {
  if (!c || !out || n < 1) return fail(RC_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(uint32_t) * n));
  launch_mt_stream(seed, 1, n, d, c->stream);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return fail(RC_ERR_DEVICE, hipGetErrorString(e));
  return RC_OK;
}

}  // extern "C"
