/* example_batch.c -- the batch API of include/rnacode_hip.h from plain C.
 *
 *   gcc -std=c99 -Iinclude integration/example_batch.c -Lrnacode_amd -lrnacode_hip -Wl,-rpath,$PWD/rnacode_amd -o example_batch
 *   ./example_batch          (needs an MI355X; without a GPU rc_ctx_create fails and the program says so)
 *
 * Scores two small blocks with trees fitted on the GPU and prints their HSS tables. */
#include <stdio.h>
#include <string.h>

#include "rnacode_hip.h"

int main(void) {
  const char *rows_a[] = {"ATGGCTAAAGGGGCATGCGATCGATCGTTTATGCCGTAA", "ATGGCTAAAGGAGCATGCGATCGTTCGTTTATGCCGTAA",
                          "ATGGCAAAAGGGGCATGTGATCGATCGTTTATGCCTTAA", "ATGGCTAAGGGGGCTTGCGATCGATCCTTTATGCCGTAA"};
  const char *names_a[] = {"a.chr1", "b.chr1", "c.chr1", "d.chr1"};
  const char *rows_b[] = {"ATGAAACCCGGGTTTACGATG", "ATGAAGCCCGGGTTTACGATG", "ATGAAACCAGGGTTCACGATG"};
  const char *names_b[] = {"a.chr2", "b.chr2", "c.chr2"};
  rc_block blocks[2];
  char newick[2][4096];
  float kappa[2];
  rc_ctx *ctx = NULL;
  rc_batch *batch = NULL;
  rc_params par;
  int i, n, k;

  memset(blocks, 0, sizeof blocks);
  blocks[0].n_rows = 4; blocks[0].n_cols = (int)strlen(rows_a[0]); blocks[0].rows = rows_a; blocks[0].names = names_a;
  blocks[1].n_rows = 3; blocks[1].n_cols = (int)strlen(rows_b[0]); blocks[1].rows = rows_b; blocks[1].names = names_b;

  if (rc_ctx_create(0, &ctx) != RC_OK) {
    fprintf(stderr, "no HIP device: %s\n", rc_last_error());
    return 2;
  }
  /* what treeML() would hand over (RNAcode.c:153): tree + kappa per block, fitted on the GPU */
  if (rc_fit_trees_device(ctx, blocks, 2, &newick[0][0], (int)sizeof newick[0], kappa, NULL) != 2) {
    fprintf(stderr, "tree fit failed: %s\n", rc_last_error());
    return 1;
  }
  for (i = 0; i < 2; i++) { blocks[i].newick = newick[i]; blocks[i].kappa = kappa[i]; }

  rc_default_params(&par);
  par.sampleN = 200;
  par.seed_base = 42;
  if (rc_batch_create(ctx, blocks, 2, &par, &batch) != RC_OK || rc_batch_run(batch) != RC_OK) {
    fprintf(stderr, "scoring failed: %s\n", rc_last_error());
    return 1;
  }
  for (i = 0; i < 2; i++) {
    rc_hss hss[64];
    if (rc_batch_status(batch, i) != RC_OK) continue;
    n = rc_batch_hss(batch, i, hss, 64);
    printf("block %d: tree %s kappa %.3f, %d HSS\n", i, newick[i], kappa[i], n);
    for (k = 0; k < n && k < 64; k++)
      printf("  %c%d  sites %d-%d  score %.2f  p %.3g\n", hss[k].strand, hss[k].frame + 1, hss[k].startSite + 1,
             hss[k].endSite + 1, hss[k].score, hss[k].pvalue);
  }
  rc_batch_destroy(batch);
  rc_ctx_destroy(ctx);
  return 0;
}
