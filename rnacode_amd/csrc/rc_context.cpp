// rc_context.cpp -- contexts of the C-ABI (include/rnacode_hip.h): device, streams, constant tables, host threads.
// No CPU implementation of the hot path lives in this library: without a HIP device rc_ctx_create fails with RC_ERR_DEVICE.
#include "rc_runtime.h"

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
}  // extern "C"

void want_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" {

int rc_device_count(void) {
  want_hw_queues();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rc_host_cpus(void) { return effective_cpus(); }

void rc_ctx_trim(rc_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // the retired list holds outgrown buffers that queued kernels and copies may still use: wait for the device first (alloc_with_retry does the same
  // before it drains).  To be called between batches, like rc_ctx_set_host_threads: nothing here is guarded against another thread using the context.
  (void)hipDeviceSynchronize();
  c->retired->drain();
  c->bufPool->drain(false);
}

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
  if (c->tree2) (void)hipStreamDestroy(c->tree2);
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
  c->bufPool->device = device;
  static const int hostExp = exp_mode_of_host();   // (once per process: 20 000 exp calls, 2 ms)
  c->expMode = hostExp;
  // RC_FIT_EXP: which exp the EVD fit computes with, whatever this host's C library does -- to reproduce listings made on another machine
  // (the reference's fit follows ITS host's exp), and for the tests that run both glibc variants.  rc_ctx_fit_exp_mode() reports what is in use.
  if (const char *e = std::getenv("RC_FIT_EXP")) {
    const std::string v(e);
    if (v == "generic" || v == "1") c->expMode = 1;
    else if (v == "fused" || v == "fma" || v == "2") c->expMode = 2;
    else if (v == "device" || v == "0") c->expMode = 0;
    else return fail(RC_ERR_ARG, "RC_FIT_EXP must be generic, fused or device");
  }
  for (DevBuf *d : {&c->treeJobs, &c->treeIn, &c->treeWork, &c->treeRes, &c->d_tables, &c->d_cellStats}) d->retired = c->retired.get();
  for (PinBuf *h : {&c->treeInPin, &c->treeResPin}) h->retired = c->retired.get();
  c->genericScratchWords = std::max<size_t>(static_cast<size_t>(1) << 30, prop.totalGlobalMem / 12 / sizeof(uint32_t));
  c->togetherWords = std::max<size_t>(static_cast<size_t>(1) << 30, prop.totalGlobalMem / 36 / sizeof(uint32_t));   // 8 GB of 288: allocating more costs a short run more than it gains (tools/cli_mixed_sweep.sh)
  if (const char *e = std::getenv("RC_TOGETHER_MAX_MB")) c->togetherWords = static_cast<size_t>(std::max(0ll, std::atoll(e))) * (1u << 20) / sizeof(uint32_t);
  if (const char *e = std::getenv("RC_CLASS_STREAMS")) c->classStreamCount = std::max(1, std::min(16, std::atoi(e)));
  if (const char *e = std::getenv("RC_GRID_CAP_PER_CU")) c->gridCapPerCU = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("RC_SPLIT_FACTOR")) c->splitFactor = std::max(0.0, std::atof(e));
  if (const char *e = std::getenv("RC_NATIVE_WAVES_PER_CU")) c->nativeWavesPerCU = std::max(1, std::min(32, std::atoi(e)));
  if (const char *e = std::getenv("RC_TAIL_SHARING")) c->tailSharing = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_ROW_SPLIT")) c->rowSplit = std::atoi(e) != 0;
  if (const char *e = std::getenv("RC_HIGH_OCCUPANCY")) c->highOccupancy = std::max(0, std::min(2, std::atoi(e)));
  if (prop.maxSharedMemoryPerMultiProcessor > 0) c->ldsPerCU = prop.maxSharedMemoryPerMultiProcessor;
  if (const char *e = std::getenv("RC_FAT_STREAM_MIN_ITEMS")) c->fatStreamMinItems = std::max(0ll, std::atoll(e));
  if (const char *e = std::getenv("RC_GENERIC_SCRATCH_MB")) c->genericScratchWords = std::max<size_t>(1, static_cast<size_t>(std::atoll(e))) * (1u << 20) / sizeof(uint32_t);
  // non-blocking streams: no implicit ordering against the legacy default stream that a host framework in
  // the same process (torch, RCCL) may be using; the native stream gets the higher priority so that the
  // runtime puts it on a hardware queue of its own (two plain streams can end up sharing one, which
  // serialises the native-block kernels behind k_null again -- seen under torch.distributed)
  int prioLow = 0, prioHigh = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
  // The streams are created when they are first used (ctx_stream): a HIP stream with a hardware queue of its own costs 10 ms to create,
  // six of them were 60 ms of every context -- a fifth of the stand-alone driver's run on a 10 000-block input -- and a run of fat batches
  // (rc_schedule.cpp, fat_class) never touches the native-block stream, a single batch never the second compute stream.
  // TWO priority levels, not three.  The HIP runtime keeps a pool of hardware queues per stream priority (GPU_MAX_HW_QUEUES each).
  // Until round 5 the compute streams were created low, the small latency-bound ones (native block, copies, helpers, tree fits) high
  // and the class streams at the default priority: a process that had used all of them held queues at three levels, and the device's
  // scheduler then parked whole queues for 50 ms .. 5 s at a time -- 7 % of the steps of a small multi-class batch, the 1.6 s step of
  // round 4's bench line (tools/stall_hunt.py; profiles/r05/stall_hunt.txt: three levels stall with 6 or 8 queues per level, two
  // levels or one never did with 4 .. 24).  Now: compute and class streams low, everything else at the default priority, which is
  // also where a host framework's own streams (torch, RCCL) live.
  c->prioLow = prioLow; c->prioHigh = 0; c->classPrio = prioLow;
  trace("ctx: streams", nullptr);
  c->serialNative = std::getenv("RC_SERIAL_NATIVE") ? 1 : 0;   // profiling: one kernel at a time, clean per-kernel intervals
  c->stopTwoRoundsMinItems = 4LL * c->numCU * 16;   // the chip filled four times over
  if (const char *e = std::getenv("RC_STOP_MIN_ITEMS")) c->stopTwoRoundsMinItems = std::atoll(e);
  if (const char *e = std::getenv("RC_GENERIC_MIN_ROWS")) c->rule.genericMinRows = std::max(3, std::min(kTemplRows + 1, std::atoi(e)));
  if (const char *e = std::getenv("RC_TILED_MAX_L")) c->rule.tiledMaxL = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("RC_TILED_MIN_ROWS")) c->rule.tiledMinRows = std::max(kTiledMaxKT + 2, std::min(kTemplRows + 1, std::atoi(e)));
  if (const char *e = std::getenv("RC_TILED_MAX_ROWS")) c->rule.tiledMaxRows = std::max(kTemplRows, std::min(kMaxRows, std::atoi(e)));
  if (const char *e = std::getenv("RC_TILED_WIDE_MAX_L")) c->rule.tiledWideMaxL = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("RC_TILED_ANY_L_MIN_ROWS")) c->rule.tiledAnyLMinRows = std::max(kTemplRows + 1, std::atoi(e));
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
#ifdef RC_PROFILING   // these two switches exist in profiling builds only (tools/mk_ab.sh): RC_DEBUG_SKIP gives wrong results
  if (const char *e = std::getenv("RC_DEBUG_SKIP")) c->debugSkip = std::atoi(e);
  if (std::getenv("RC_CELL_STATS")) {
    HIP_TRY(c->d_cellStats.ensure(6 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_cellStats.p, 0, 6 * sizeof(unsigned long long)));
  }
#endif
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

int rc_ctx_fit_exp_mode(const rc_ctx *c) { return c ? c->expMode : RC_ERR_ARG; }

}  // extern "C"
