// rc_batch.cpp -- a batch's life: prepare -> upload -> run -> wait (the rc_batch_* entry points of include/rnacode_hip.h).
#include "rc_runtime.h"

// A batch goes through: prepare (host threads write the block tables straight into one pinned blob) -> upload (ONE asynchronous
// copy + the kernels that derive the expected-score tables on the device) -> run (all launches asynchronous, results staged into
// pinned memory) -> wait.  rc_batch_create / rc_batch_run are the synchronous forms; rc_stream_* (rc_stream.cpp) keeps several
// batches in flight so that the host prepares batch i + 1 while the GPU scores batch i (RNAcode.c:115-221 is a stream of blocks).

extern "C" {

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

}  // extern "C"

// ------------------------------------------------------------------------------------------ prepare + upload

extern "C" void rc_batch_destroy(rc_batch *b);

int batch_prepare(rc_batch *b, const rc_block *blocks, int32_t n_blocks, const rc_params *par) {
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
    auto &cls = b->classes[block_class(m.N, m.L, b->db[i].omega, ctx->rule)];
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
  // a tiled class holds several row counts: most rows first (the costlier items), each row count one contiguous piece of the list -- the native
  // block's kernels are launched per row count (rc_schedule.cpp)
  for (auto &kv : b->classes)
    if (kv.first > kGenericClass) std::stable_sort(kv.second.begin(), kv.second.end(), [&](int x, int y) { return b->meta[x].NK > b->meta[y].NK; });

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
int batch_upload(rc_batch *b) {
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
    pa.rule = c->rule;
    launch_prep(pa, ps);
  }
  HIP_TRY(b->evPrep.record(ps));
  HIP_TRY(hipGetLastError());
  b->state = rc_batch::READY;
  b->maximaFetched = false;
  return RC_OK;
}


extern "C" {

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
}  // extern "C"

bool g_no_stream_cache = false;
extern "C" void rc_set_stream_cache(int enabled) { g_no_stream_cache = !enabled; }


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


int batch_run_async(rc_batch *b, bool streaming) {
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
    allFat = allFat && kv.first < kGenericClass && fat_class(b, c, kv.first, maxLc);
    wide = wide || kv.first >= kGenericClass;   // (the generic class and the tiled ones: persistent grids that take every register of the chip)
  }
  bool fat = allFat && (!streaming || static_cast<long long>(b->okBlocks.size()) * groups >= c->fatStreamMinItems);
  if (fat) nativeFirst = true;
  // k_generic_dp's persistent grid takes every register of the chip as well (128 VGPRs, four wavefronts per SIMD): native-block
  // kernels queued beside it either wait for its tail or, worse, get in first and keep part of its workgroups out for their whole run
  if (wide) nativeFirst = true;
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
    fa.sampleN = sampleN; fa.stopEarly = b->par.stopEarly; fa.expMode = c->expMode;
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
      b->sbufNeed = 0;
      for (int lo = 0, hi = g1; lo < groups; r++) {
        if (r == c->stopRounds - 1) hi = groups;
        RC_TRY(launch_null_groups(R, lo, hi, r, 0, &need));
        most = std::max(most, need);
        lo = hi; hi = std::min(groups, hi * 2);
      }
      HIP_TRY(b->dscratch.ensure(most * sizeof(uint32_t)));
      if (b->sbufNeed) HIP_TRY(b->dsbuf.ensure(b->sbufNeed * sizeof(float)));   // rows split over workgroups: the S values between k_null and k_null_rowscan
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
      if (hipHostGetDevicePointer(&dStage, b->hstage.p, 0) == hipSuccess &&
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
      } else {   // (no device view of the pinned buffers: plain copies)
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

int batch_wait(rc_batch *b) {
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

extern "C" {

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
}  // extern "C"

void trace_device(rc_batch *b) {
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

