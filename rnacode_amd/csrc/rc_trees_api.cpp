// rc_trees_api.cpp -- the tree estimator's entry points (rc_fit_tree, rc_fit_trees, rc_fit_trees_device, rc_tree_lnl), the stand-alone
// EVD fit, p-values and the MT19937 accessor.
#include "rc_runtime.h"

extern "C" {

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
  trace("trees: call", blocks);
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
  trace("trees: patterns", blocks);
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
  // (workgroups per CU beyond what the kernel's registers allow make no class of their own: round 5's first version launched
  // sixteen under-filled grids one after the other for 10 000 blocks of 6 x 120, each as long as one fit)
  const int occReg = std::max(1, tree_fit_register_occupancy());
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
    const int occ = static_cast<int>(std::min<size_t>(static_cast<size_t>(occReg), c->ldsPerCU / std::max<size_t>(need, 1)));
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
  trace("trees: jobs", blocks);
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
  trace("trees: pinned", blocks);
  RC_STREAM_TRY(ts, stream_tree(c));
  HIP_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), sizeof(TreeJob) * nj, hipMemcpyHostToDevice, ts));
  HIP_TRY(hipMemcpyAsync(d_in.p, c->treeInPin.p, in_bytes, hipMemcpyHostToDevice, ts));
  const TreeJob *dj = static_cast<const TreeJob *>(d_jobs.p);
  const uint8_t *din = static_cast<const uint8_t *>(d_in.p);
  // The launches of the LDS classes alternate between two streams: a launch ends with the tail of its slowest fits, and the next
  // class's workgroups fill the chip meanwhile.  (The big blocks' launches share one scratch and stay in order on the first.)
  size_t ldsPad = 0;
#ifdef RC_TREE_PROFILE   // occupancy experiment: a padded LDS request leaves fewer fits per CU (tools/tree_phases.sh)
  if (const char *e = std::getenv("RC_TREE_LDS_PAD")) ldsPad = static_cast<size_t>(std::atoll(e));
#endif
  int nSmall = 0;
  for (size_t l = 1; l < launches.size(); l++) nSmall += launches[l].blocks.empty() ? 0 : 1;
  // (a second stream costs 10 ms to create: not for a call whose launches are over before that -- a driver's first chunk)
  hipStream_t ts2 = ((nSmall + (nBig ? 1 : 0)) > 1 && (nj >= 4096 || c->tree2)) ? stream_tree2(c) : nullptr;
  if (ts2) { HIP_TRY(c->treeFork.record(ts)); HIP_TRY(hipStreamWaitEvent(ts2, c->treeFork, 0)); }
  for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
    const int q0 = chunkStart[ch], q1 = chunkStart[ch + 1];
    if (q1 > q0 && !launch_tree_fit(dj + q0, q1 - q0, true, launches[0].lds, din, static_cast<double *>(d_work.p), static_cast<double *>(d_res.p), ts))
      return fail(RC_ERR_DEVICE, "k_tree_fit: the device refused the launch's LDS request");
  }
  {
    int q0 = nBig, turn = nBig ? 1 : 0;
    for (size_t l = 1; l < launches.size(); l++) {
      const int cnt = static_cast<int>(launches[l].blocks.size());
      hipStream_t st = (ts2 && (turn & 1)) ? ts2 : ts;
      if (cnt && !launch_tree_fit(dj + q0, cnt, false, launches[l].lds + ldsPad, din, static_cast<double *>(d_work.p), static_cast<double *>(d_res.p), st))
        return fail(RC_ERR_DEVICE, "k_tree_fit: the device refused the launch's LDS request");
      if (cnt) turn++;
      q0 += cnt;
    }
  }
  if (ts2) { HIP_TRY(c->treeJoin.record(ts2)); HIP_TRY(hipStreamWaitEvent(ts, c->treeJoin, 0)); }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->treeResPin.p, d_res.p, sizeof(double) * res_doubles, hipMemcpyDeviceToHost, ts));
  trace("trees: queued", blocks);
  HIP_TRY(hipStreamSynchronize(ts));
  trace("trees: fitted", blocks);
  const double *res = c->treeResPin.as<double>();
  if (kTreeProfDoubles) {   // -DRC_TREE_PROFILE builds: where the wavefronts' cycles went, summed over the jobs
    double sum[16] = {0};
    for (int q = 0; q < nj; q++) {
      const double *pd = res + jobs[q].off_out + tree_result_doubles(jobs[q].N) - kTreeProfDoubles;
      for (int x = 0; x < kTreeProfDoubles; x++) sum[x] += pd[x];
    }
    static const char *names[] = {"load", "base_freqs", "distances", "bionj", "likelihood passes", "branch constants", "newton", "subtree refresh + constants", "optimiser logic"};
    double tot = 0;
    for (int x = 0; x < 9; x++) tot += sum[x];
    std::fprintf(stderr, "[rc tree profile] %d fits, %.0f cycles per fit; ", nj, tot / nj);
    for (int x = 0; x < 9; x++) std::fprintf(stderr, "%s %.1f %%, ", names[x], 100.0 * sum[x] / tot);
    std::fprintf(stderr, "per fit: %.1f likelihood passes, %.1f Newton iterations, %.2f rounds\n", sum[9] / nj, sum[10] / nj, sum[11] / nj);
  }
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
  trace("trees: newick", blocks);
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
  launch_evd_fit_f64(d_x, n, d_o, c->expMode, aux);
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
