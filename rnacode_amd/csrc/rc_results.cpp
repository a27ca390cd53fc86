// rc_results.cpp -- what a finished batch holds: statuses, models, maxima, fits, HSS tables, native S matrices, backtrack paths.
#include "rc_runtime.h"

extern "C" {

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
    const bool generic = h.N > kTemplRows || block_class(h.N, h.L, b->db[blk].omega, c->rule) == kGenericClass;   // (the tiled classes' native block takes k_native_dp<N-1> up to 64 rows)
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
    na.flags = b->dflags.as<uint32_t>();
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
  launch_sk_row(b->dblob.as<uint8_t>(), reinterpret_cast<const DevBlock *>(b->dblob.as<uint8_t>() + b->oDblocks), b->dflags.as<uint32_t>(), blk, strand, opt_b, opt_i,
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

}  // extern "C"
