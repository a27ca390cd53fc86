/* example_stream.c -- the stream API of include/rnacode_hip.h from plain C: the shape of
 * RNAcode's driver loop (src/RNAcode.c:115-221: read a block, fit its tree, score it, print) as a pipeline in which the
 * host prepares sub-batch i+1 while the GPU scores sub-batch i.
 *
 *   gcc -std=c99 -Iinclude integration/example_stream.c -Lrnacode_amd -lrnacode_hip -Wl,-rpath,$PWD/rnacode_amd -o example_stream
 *   ./example_stream          (needs an MI355X; without a GPU rc_ctx_create fails and the program says so)
 *
 * 24 copies of three small blocks (one of them with two rows only: the driver skips it, RNAcode.c:142) go through a
 * stream of sub-batches of 5 blocks, two in flight; the same blocks scored as one batch must give the same tables. */
#include <stdio.h>
#include <string.h>

#include "rnacode_hip.h"

#define NBLOCKS 72
#define SUB 5

int main(void) {
  const char *rows_a[] = {"ATGGCTAAAGGGGCATGCGATCGATCGTTTATGCCGTAA", "ATGGCTAAAGGAGCATGCGATCGTTCGTTTATGCCGTAA",
                          "ATGGCAAAAGGGGCATGTGATCGATCGTTTATGCCTTAA", "ATGGCTAAGGGGGCTTGCGATCGATCCTTTATGCCGTAA"};
  const char *names_a[] = {"a.chr1", "b.chr1", "c.chr1", "d.chr1"};
  const char *rows_b[] = {"ATGAAACCCGGGTTTACGATG", "ATGAAGCCCGGGTTTACGATG", "ATGAAACCAGGGTTCACGATG"};
  const char *names_b[] = {"a.chr2", "b.chr2", "c.chr2"};
  const char *rows_c[] = {"ATGAAACCCGGG", "ATGAAGCCCGGG"};
  const char *names_c[] = {"a.chr3", "b.chr3"};
  static rc_block blocks[NBLOCKS];
  static char newick[NBLOCKS][4096];
  static float kappa[NBLOCKS];
  static float want[NBLOCKS][4], got[NBLOCKS][4];
  rc_ctx *ctx = NULL;
  rc_stream *stream = NULL;
  rc_batch *b = NULL;
  rc_params par;
  int i, next = 0, done = 0, scored = 0, skipped = 0;

  memset(blocks, 0, sizeof blocks);
  for (i = 0; i < NBLOCKS; i++) {
    const int kind = i % 3;
    blocks[i].n_rows = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    blocks[i].rows = kind == 0 ? rows_a : (kind == 1 ? rows_b : rows_c);
    blocks[i].names = kind == 0 ? names_a : (kind == 1 ? names_b : names_c);
    blocks[i].n_cols = (int)strlen(blocks[i].rows[0]);
    blocks[i].ref_start = 1000 * i; blocks[i].ref_length = blocks[i].n_cols;
  }
  if (rc_ctx_create(0, &ctx) != RC_OK) {
    fprintf(stderr, "no HIP device: %s\n", rc_last_error());
    return 2;
  }
  if (rc_fit_trees_device(ctx, blocks, NBLOCKS, &newick[0][0], (int)sizeof newick[0], kappa, NULL) != 2 * (NBLOCKS / 3)) {
    fprintf(stderr, "tree fit failed: %s\n", rc_last_error());
    return 1;
  }
  for (i = 0; i < NBLOCKS; i++) { blocks[i].newick = newick[i][0] ? newick[i] : NULL; blocks[i].kappa = kappa[i]; }
  rc_default_params(&par);
  par.sampleN = 128;

  /* reference result: everything as one batch */
  if (rc_batch_create(ctx, blocks, NBLOCKS, &par, &b) != RC_OK || rc_batch_run(b) != RC_OK || rc_batch_fit_all(b, &want[0][0]) != RC_OK) {
    fprintf(stderr, "batch failed: %s\n", rc_last_error());
    return 1;
  }
  rc_batch_destroy(b);

  /* the same blocks as a stream of sub-batches, two in flight */
  if (rc_stream_create(ctx, &par, 2, &stream) != RC_OK) { fprintf(stderr, "%s\n", rc_last_error()); return 1; }
  while (next < NBLOCKS || rc_stream_pending(stream) > 0) {
    while (next < NBLOCKS && rc_stream_pending(stream) < 2) {
      const int n = NBLOCKS - next < SUB ? NBLOCKS - next : SUB;
      if (rc_stream_submit(stream, blocks + next, n) != RC_OK) { fprintf(stderr, "submit: %s\n", rc_last_error()); return 1; }
      next += n;
    }
    if (rc_stream_next(stream, &b) != RC_OK) { fprintf(stderr, "next: %s\n", rc_last_error()); return 1; }
    if (rc_batch_fit_all(b, &got[done][0]) != RC_OK) { fprintf(stderr, "fits: %s\n", rc_last_error()); return 1; }
    for (i = 0; i < rc_batch_size(b); i++) {
      const int st = rc_batch_status(b, i);
      if (st == RC_OK) scored++; else if (st == RC_ERR_SKIP) skipped++;
    }
    done += rc_batch_size(b);
    rc_stream_recycle(stream, b);
  }
  rc_stream_destroy(stream);
  rc_ctx_destroy(ctx);
  if (done != NBLOCKS || memcmp(want, got, sizeof want) != 0) {
    fprintf(stderr, "stream and batch results differ\n");
    return 1;
  }
  printf("stream ok: %d blocks in sub-batches of %d, %d scored, %d skipped, fits equal to the single batch\n", done, SUB, scored, skipped);
  return 0;
}
