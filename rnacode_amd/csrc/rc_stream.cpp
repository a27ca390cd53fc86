// rc_stream.cpp -- streams of sub-batches (rc_stream_*): up to `depth` batches in flight, batch objects recycled.
#include "rc_runtime.h"

extern "C" {

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
  // a long stream: bounded memory, results keep flowing.  (Up to thirty-two rounds per sub-batch in inputs long enough for a dozen of those -- fewer
  // launches, fewer tails -- were measured on the native driver with 100 000 blocks of 6 x 120: 0.72..0.75 s from main() to the last byte,
  // against 0.62 with eight: the trees of a sub-batch, its results and its part of the listing are all the longer in front of and behind the
  // GPU's work.  profiles/r06/e2e100k_plan_ab.txt)
  const long long most = 8 * r;
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

}  // extern "C"
