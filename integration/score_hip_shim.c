/* score_hip_shim.c -- the reference-side binding: RNAcode's scoring object (src/score.c) replaced
 * by calls into librnacode_hip.so.
 *
 * This file is what a maintainer of the reference adds to its build INSTEAD of src/score.c.  It
 * defines exactly the symbols the rest of the program takes from score.o:
 *   getModels, freeModels, scoreAln, getExtremeValuePars     used by src/RNAcode.c:164-216
 *   backtrack, getScoringMatrix, freeScoringMatrix, ntMap    used by src/postscript.c:303-305,362,677
 * and honours their ownership rules (malloc'ed segmentStats list terminated by score = -1 with
 * strdup'ed names; Sk / Sk_native / Sk_native_rev left as free()-able jagged arrays because
 * main() frees them row by row, RNAcode.c:192-209).
 *
 * It is compiled against the reference's own headers (struct aln, TTree, bgModel, segmentStats,
 * parameters) and therefore only builds where the reference tree is present; oracle/Makefile links
 * it with the caller-side harness into oracle/_ref/ref_harness_hip to prove the link-level drop-in
 * (tests/test_gpu_dropin.py runs that binary on the GPU box and compares with the goldens).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rnaz_utils.h"
#include "tree.h"
#include "treeSimulate.h"
#include "score.h"
#include "RNAcode.h"

#include "rnacode_hip.h"

extern parameters pars;
extern bgModel *models, *modelsRev;
extern float ****Sk, ****Sk_native, ****Sk_native_rev;

int ntMap[256];   /* defined by score.c in the reference (score.c:41), filled by main */

/* optional hooks of the test harness: seed base for the null samples, capture of the maxima */
extern unsigned long rc_harness_seed_base __attribute__((weak));
extern void rc_harness_capture(const float *maxima, int n) __attribute__((weak));

static rc_ctx *g_ctx = NULL;
static rc_batch *g_batch = NULL;
static int g_N = 0, g_cols = 0;
static char *g_fwd_row0 = NULL;

static void die(const char *what) {
  fprintf(stderr, "score_hip_shim: %s: %s\n", what, rc_last_error());
  exit(1);
}

/* Newick text of a Seq-Gen tree (seqgen/treefile.c:548-579 prints %.6f; %.17g keeps the doubles) */
static void write_node(TTree *tree, TNode *node, char **buf, size_t *len, size_t *cap) {
  char tmp[512];
  if (*len + 600 > *cap) { *cap = *cap * 2 + 1024; *buf = (char *)realloc(*buf, *cap); }
  if (node->tipNo == -1) {
    (*buf)[(*len)++] = '(';
    write_node(tree, node->branch1, buf, len, cap);
    (*buf)[(*len)++] = ',';
    write_node(tree, node->branch2, buf, len, cap);
    if (*len + 600 > *cap) { *cap = *cap * 2 + 1024; *buf = (char *)realloc(*buf, *cap); }
    (*buf)[(*len)++] = ')';
  } else {
    size_t n = strlen(tree->names[node->tipNo]);
    memcpy(*buf + *len, tree->names[node->tipNo], n);
    *len += n;
  }
  snprintf(tmp, sizeof tmp, ":%.17g", node->length0);
  if (*len + strlen(tmp) + 8 > *cap) { *cap = *cap * 2 + 1024; *buf = (char *)realloc(*buf, *cap); }
  memcpy(*buf + *len, tmp, strlen(tmp));
  *len += strlen(tmp);
}

static char *tree_to_newick(TTree *tree) {
  size_t len = 0, cap = 4096;
  char *buf = (char *)malloc(cap);
  buf[len++] = '(';
  write_node(tree, tree->root->branch1, &buf, &len, &cap);
  buf[len++] = ',';
  write_node(tree, tree->root->branch2, &buf, &len, &cap);
  if (!tree->rooted) {
    buf[len++] = ',';
    write_node(tree, tree->root->branch0, &buf, &len, &cap);
  }
  if (len + 4 > cap) buf = (char *)realloc(buf, cap + 8);
  buf[len++] = ')'; buf[len++] = ';'; buf[len] = 0;
  return buf;
}

static void drop_batch(void) {
  if (g_batch) { rc_batch_destroy(g_batch); g_batch = NULL; }
  free(g_fwd_row0); g_fwd_row0 = NULL;
}

static void make_batch(TTree *tree, struct aln *alignment[], float kappa) {
  int N, i;
  rc_block blk;
  rc_params par;
  const char **rows, **names;
  char *nwk;
  if (!g_ctx && rc_ctx_create(0, &g_ctx) != RC_OK) die("rc_ctx_create");
  drop_batch();
  for (N = 0; alignment[N] != NULL; N++);
  rows = (const char **)malloc(sizeof(char *) * N);
  names = (const char **)malloc(sizeof(char *) * N);
  for (i = 0; i < N; i++) { rows[i] = alignment[i]->seq; names[i] = alignment[i]->name; }
  nwk = tree_to_newick(tree);
  memset(&blk, 0, sizeof blk);
  blk.n_rows = N; blk.n_cols = (int)strlen(alignment[0]->seq);
  blk.rows = rows; blk.names = names;
  blk.ref_start = alignment[0]->start; blk.ref_length = alignment[0]->length;
  blk.newick = nwk; blk.kappa = kappa;
  rc_default_params(&par);
  par.Delta = pars.Delta; par.Omega = pars.Omega; par.omega = pars.omega;
  par.stopPenalty_0 = pars.stopPenalty_0; par.stopPenalty_k = pars.stopPenalty_k;
  par.blosum = pars.blosum; par.sampleN = pars.sampleN; par.cutoff = pars.cutoff; par.stopEarly = pars.stopEarly;
  par.seed_base = (&rc_harness_seed_base != NULL) ? (uint32_t)rc_harness_seed_base
                  : (getenv("RC_SEED_BASE") ? (uint32_t)strtoul(getenv("RC_SEED_BASE"), NULL, 10) : 42u);
  if (rc_batch_create(g_ctx, &blk, 1, &par, &g_batch) != RC_OK) die("rc_batch_create");
  if (rc_batch_status(g_batch, 0) != RC_OK) die("block rejected");
  g_N = N; g_cols = blk.n_cols;
  g_fwd_row0 = strdup(alignment[0]->seq);
  free(rows); free(names); free(nwk);
}

/* revAln's mapping (rnaz_utils.c:316-348): is s the reverse complement of the batch's forward row 0? */
static int is_revcomp_of(const char *s, const char *fwd) {
  size_t n = strlen(fwd), j;
  if (strlen(s) != n) return 0;
  for (j = 0; j < n; j++) {
    char c = fwd[n - 1 - j];
    switch (c) { case 'T': case 'U': c = 'A'; break; case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; default: break; }
    if (s[j] != c) return 0;
  }
  return strcmp(s, fwd) != 0;   /* a row equal to its own reverse complement starts a new block */
}

/* score.c:50-76: a malloc'ed 24x24 copy of BLOSUM62/90 in the amino-acid order of transcode (code.c:28-39).
 * src/postscript.c:437,527 index it with scoringMatrix[pepA][pepB], pepA/pepB = transcode values (0..19, stops
 * excluded before the look-up), to pick the colour of a codon pair.  Rows and columns 20..23 (B, Z, X, *) cannot
 * be reached through transcode; they are zero here. */
int **getScoringMatrix(void) {
  int i, j, **m = (int **)malloc(sizeof(int *) * 24);
  int32_t pep[64], mat[400];
  if (rc_code_tables(pars.blosum == 90 ? 90 : 62, pep, mat) != RC_OK) die("rc_code_tables");
  for (i = 0; i < 24; i++) {
    m[i] = (int *)calloc(24, sizeof(int));
    if (i < 20) for (j = 0; j < 20; j++) m[i][j] = mat[20 * i + j];
  }
  return m;
}
void freeScoringMatrix(int **m) { int i; for (i = 0; i < 24; i++) free(m[i]); free(m); }

/* main calls getModels(tree, aln, kappa) and then getModels(tree, revAln, kappa) (RNAcode.c:164-165) */
bgModel *getModels(TTree *tree, struct aln *alignment[], float kappa) {
  int N, j, is_fwd;
  rc_model *f, *r;
  bgModel *out;
  for (N = 0; alignment[N] != NULL; N++);
  is_fwd = !(g_batch && g_N == N && g_fwd_row0 && is_revcomp_of(alignment[0]->seq, g_fwd_row0));
  if (is_fwd) make_batch(tree, alignment, kappa);
  f = (rc_model *)malloc(sizeof(rc_model) * N);
  r = (rc_model *)malloc(sizeof(rc_model) * N);
  if (rc_batch_models(g_batch, 0, f, r) != RC_OK) die("rc_batch_models");
  out = (bgModel *)malloc(sizeof(bgModel) * N);
  for (j = 0; j < N; j++) {
    const rc_model *m = is_fwd ? &f[j] : &r[j];
    memcpy(out[j].scores, m->scores, sizeof m->scores);
    memcpy(out[j].probs, m->probs, sizeof m->probs);
    memcpy(out[j].freqs, m->freqs, sizeof m->freqs);
    out[j].kappa = m->kappa; out[j].dist = m->dist; out[j].weight = 0;
    out[j].matrix = getScoringMatrix();
  }
  free(f); free(r);
  return out;
}

void freeModels(bgModel *m, int N) {
  int j;
  for (j = 0; j < N; j++) freeScoringMatrix(m[j].matrix);
  free(m);
}

/* jagged [N][3][L+1] of NULL leaves: what main() frees at RNAcode.c:192-209 */
static float ****jagged(int N, int L) {
  int k, x;
  float ****S = (float ****)malloc(sizeof(float ***) * (N + 1));
  for (k = 0; k < N; k++) {
    S[k] = (float ***)malloc(sizeof(float **) * 3);
    for (x = 0; x < 3; x++) S[k][x] = (float **)calloc(L + 1, sizeof(float *));
  }
  return S;
}

segmentStats *scoreAln(const struct aln *inputAln[], TTree *tree, float kappa, int backtrack) {
  rc_hss *h;
  segmentStats *out;
  int n, i, L = 0;
  const char *p;
  (void)tree; (void)kappa;
  if (!backtrack) { fprintf(stderr, "score_hip_shim: scoreAln is only entered for the native block\n"); exit(1); }
  if (rc_batch_run(g_batch) != RC_OK) die("rc_batch_run");
  for (p = inputAln[0]->seq; *p; p++) L += (*p != '-');
  if (Sk == NULL) Sk = jagged(g_N, L);
  if (Sk_native == NULL) { Sk_native = jagged(g_N, L); Sk_native_rev = jagged(g_N, L); }
  n = rc_batch_hss(g_batch, 0, NULL, 0);
  if (n < 0) die("rc_batch_hss");
  h = (rc_hss *)malloc(sizeof(rc_hss) * (n + 1));
  rc_batch_hss(g_batch, 0, h, n);
  out = (segmentStats *)malloc(sizeof(segmentStats) * (n + 1));
  for (i = 0; i < n; i++) {
    out[i].start = h[i].start; out[i].end = h[i].end;
    out[i].startGenomic = h[i].startGenomic; out[i].endGenomic = h[i].endGenomic;
    out[i].startSite = h[i].startSite; out[i].endSite = h[i].endSite;
    out[i].strand = h[i].strand; out[i].frame = h[i].frame;
    out[i].name = strdup(inputAln[0]->name);
    out[i].score = h[i].score; out[i].pvalue = h[i].pvalue; out[i].hide = 0;
  }
  if (n == 0) out[0].pvalue = 1.0;
  out[n].score = -1.0;
  free(h);
  return out;
}

int getExtremeValuePars(TTree *tree, const struct aln *alignment[], int sampleN, float maxNativeScore,
                        float *parMu, float *parLambda) {
  int32_t rc = -1;
  (void)tree; (void)alignment; (void)maxNativeScore;
  if (rc_batch_fit(g_batch, 0, &rc, parMu, parLambda) != RC_OK) die("rc_batch_fit");
  if (&rc_harness_capture != NULL) {
    float *mx = (float *)malloc(sizeof(float) * sampleN);
    if (rc_batch_maxima(g_batch, 0, mx) == RC_OK) rc_harness_capture(mx, sampleN);
    free(mx);
  }
  return rc;
}

backtrackData *backtrack(int opt_b, int opt_i, float ****SSk, const struct aln *alignment[]) {
  int N, k, cols = (int)strlen(alignment[0]->seq);
  int32_t *st, *z, *tr;
  backtrackData *out;
  for (N = 0; alignment[N] != NULL; N++);
  st = (int32_t *)malloc(sizeof(int32_t) * N * (cols + 1));
  z = (int32_t *)malloc(sizeof(int32_t) * N * (cols + 1));
  tr = (int32_t *)malloc(sizeof(int32_t) * N * (cols + 1));
  if (rc_batch_backtrack(g_batch, 0, SSk == Sk_native_rev ? 1 : 0, opt_b, opt_i, st, z, tr) != RC_OK) die("rc_batch_backtrack");
  out = (backtrackData *)malloc(sizeof(backtrackData) * N);
  for (k = 1; k < N; k++) {
    out[k].states = (int *)malloc(sizeof(int) * (cols + 1));
    out[k].z = (int *)malloc(sizeof(int) * (cols + 1));
    out[k].transitions = (int *)malloc(sizeof(int) * (cols + 1));
    out[k].scores = (float *)malloc(sizeof(float) * (cols + 1));
    memcpy(out[k].states, st + k * (cols + 1), sizeof(int) * (cols + 1));
    memcpy(out[k].z, z + k * (cols + 1), sizeof(int) * (cols + 1));
    memcpy(out[k].transitions, tr + k * (cols + 1), sizeof(int) * (cols + 1));
  }
  free(st); free(z); free(tr);
  return out;
}
