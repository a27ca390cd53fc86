/* ref_harness.c -- TEST INFRASTRUCTURE, development container only.
 *
 * Caller-side driver for the *unmodified* reference objects of the scoring path
 * (src/score.c, src/misc.c, src/code.c, src/treeSimulate.c, src/extreme_fit.c,
 * src/treeML.c, seqgen/, phyml/), compiled in place from /root/reference by
 * oracle/Makefile into oracle/_ref/.  Nothing of the reference is copied into
 * this repository; this file only *calls* it, the way src/RNAcode.c:115-221
 * does, and prints what it computed as one JSON object per alignment block.
 *
 * What this file supplies instead of reference code, and why:
 *  - the globals that the reference's main file owns (src/RNAcode.c:40-46);
 *  - __wrap_CreateSeed: the reference seeds MT19937 from time()/clock() per
 *    sample (seqgen/twister.c:195-224); linking with -Wl,--wrap=CreateSeed makes
 *    seed(sample s) = seedBase + s, reset per block;
 *  - __wrap_EVDMaxLikelyFit: records the per-sample maxima handed to the fit
 *    (src/score.c:1050) and forwards to the real function;
 *  - createAlnEntry / freeAln / revAln / space: the alignment container helpers
 *    live in src/rnaz_utils.c and src/utils.c, which unconditionally include the
 *    autoconf-generated config.h and are therefore not buildable here.  They are
 *    re-stated below from their documented behaviour (rnaz_utils.c:316-348,
 *    388-420).  Consequence: revAln is NOT pinned by the reference build; it is
 *    pinned by the known-answer '-' strand HSS recorded from a full reference
 *    build in SURVEY.md section 8c (coding.aln: "- 3 32 12 43 ... 36 131 19.782").
 *  - consensus: src/postscript.c (the --eps plots, compiled in place like the other path files) draws
 *    its conservation bars from rnaz_utils.c:249-264's consensus(), which is in the unbuildable file;
 *    re-stated below together with librna/pair_mat.h:26-38's encode_char.  Consequence: the bar heights
 *    of the EPS goldens are pinned by this re-statement, everything else in them by the reference.
 *
 * Input format (written by oracle/make_goldens.py from MAF/ClustalW files):
 *   B <id>
 *   S <name> <start> <length> <strand> <fullLength> <seq>     (one per row)
 *   T <newick>        (optional: use this tree instead of treeML)
 *   K <kappa>         (optional, with T)
 *   E
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>

#include "rnaz_utils.h"
#include "tree.h"
#include "treeSimulate.h"
#include "score.h"
#include "RNAcode.h"
#include "misc.h"
#include "treeML.h"
#include "extreme_fit.h"

/* ---- globals owned by the reference's main file (src/RNAcode.c:40-46) ---- */
parameters pars;
bgModel *models, *modelsRev;
float ****Sk;
float ****Sk_native;
float ****Sk_native_rev;
long int hitCounter;

void freeModels(bgModel *models, int N);

/* ---- deterministic seeding ---- */
unsigned long rc_harness_seed_base = 42;   /* also read by integration/score_hip_shim.c */
#define g_seed_base rc_harness_seed_base
static unsigned long g_seed_counter = 0;
unsigned long __wrap_CreateSeed(void) { return g_seed_base + g_seed_counter++; }

/* ---- capture of the maxima handed to the EVD fit ---- */
static double *g_max = NULL;
static int g_max_n = 0;
static double g_mu_d = 0, g_lambda_d = 0;
static int g_fit_rc = -2;
int __real_EVDMaxLikelyFit(double *x, int *c, int n, double *ret_mu, double *ret_lambda);
int __wrap_EVDMaxLikelyFit(double *x, int *c, int n, double *ret_mu, double *ret_lambda) {
  int rc;
  free(g_max);
  g_max = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  memcpy(g_max, x, sizeof(double) * n);
  g_max_n = n;
  rc = __real_EVDMaxLikelyFit(x, c, n, ret_mu, ret_lambda);
  g_fit_rc = rc;
  if (rc == 1) { g_mu_d = *ret_mu; g_lambda_d = *ret_lambda; }
  return rc;
}

/* hook for integration/score_hip_shim.c: the drop-in scorer hands over its per-sample maxima */
void rc_harness_capture(const float *maxima, int n) {
  int i;
  free(g_max);
  g_max = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  for (i = 0; i < n; i++) g_max[i] = maxima[i];
  g_max_n = n;
}

/* ---- alignment container helpers (see header comment) ---- */
void *space(unsigned size) {
  void *p = calloc(1, size ? size : 1);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(1); }
  return p;
}
struct aln *createAlnEntry(char *name, char *seq, int start, int length, int fullLength, char strand) {
  struct aln *e = (struct aln *)space(sizeof(struct aln));
  e->name = name; e->seq = seq; e->start = start; e->length = length;
  e->fullLength = fullLength; e->strand = strand;
  return e;
}
void freeAlnEntry(struct aln *e) {
  free(e->name); free(e->seq);
  if (e->fullSeq != NULL) free(e->fullSeq);
  free(e);
}
void freeAln(struct aln *AS[]) {
  int i;
  for (i = 0; AS[i] != NULL; i++) freeAlnEntry(AS[i]);
}
void revAln(struct aln *AS[]) {
  int i, j, n = (int)strlen(AS[0]->seq);
  for (i = 0; AS[i] != NULL; i++) {
    char *s = AS[i]->seq;
    char *t = (char *)space((unsigned)n + 1);
    for (j = 0; j < n; j++) {
      char c = s[n - 1 - j];
      switch (c) {
        case 'T': c = 'A'; break;
        case 'U': c = 'A'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        case 'A': c = 'T'; break;
        default: break;
      }
      t[j] = c;
    }
    t[n] = '\0';
    strcpy(s, t);
    free(t);
    if (AS[i]->strand == '+') AS[i]->strand = '-';
    else if (AS[i]->strand == '-') AS[i]->strand = '+';
  }
}
/* rnaz_utils.c:249-264 with encode_char of librna/pair_mat.h:26-38 (energy_set == 0): most frequent
 * character class per column, first maximum wins; the class-to-letter table is indexed with the
 * T/U-merged code, so class 5 ('X') prints 'T' -- kept as it is. */
char *consensus(const struct aln *AS[]) {
  static const char order[] = "_ACGUTXKI";
  int i, n = (int)strlen(AS[0]->seq);
  char *string = (char *)calloc((size_t)n + 1, 1);
  for (i = 0; i < n; i++) {
    int s, c, fm, freq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (s = 0; AS[s] != NULL; s++) {
      const char *pos = strchr(order, AS[s]->seq[i]);
      int code = (pos == NULL || AS[s]->seq[i] == '\0') ? 0 : (int)(pos - order);
      if (code > 4) code--;
      freq[code]++;
    }
    for (s = c = fm = 0; s < 8; s++)
      if (freq[s] > fm) { c = s; fm = freq[c]; }
    string[i] = order[c];
  }
  return string;
}

/* ---- JSON helpers ---- */
static void jstr(const char *s) {
  putchar('"');
  for (; *s; s++) {
    if (*s == '"' || *s == '\\') putchar('\\');
    putchar(*s);
  }
  putchar('"');
}
static void jfarr(const char *key, const float *v, int n) {
  int i;
  printf("\"%s\":[", key);
  for (i = 0; i < n; i++) printf("%s%.9g", i ? "," : "", (double)v[i]);
  printf("]");
}

static void dump_models(const char *key, bgModel *m, int N) {
  int k;
  printf("\"%s\":[", key);
  for (k = 0; k < N; k++) {
    printf("%s{\"dist\":%.9g,\"kappa\":%.9g,", k ? "," : "", (double)m[k].dist, (double)m[k].kappa);
    jfarr("freqs", m[k].freqs, 4); putchar(',');
    jfarr("scores", m[k].scores, 4); putchar(',');
    jfarr("probs", m[k].probs, 4);
    printf("}");
  }
  printf("]");
}

static void dump_backtrack(const char *key, int b, int i, float ****SSk, const struct aln *aln[], int N) {
  int k, x;
  backtrackData *bt = backtrack(b, i, SSk, aln);
  printf("\"%s\":{\"b\":%d,\"i\":%d,\"k\":[", key, b, i);
  for (k = 1; k < N; k++) {
    printf("%s{\"states\":[", k > 1 ? "," : "");
    for (x = i; x >= b + 2; x -= 3) printf("%s%d", x == i ? "" : ",", bt[k].states[x]);
    printf("],\"z\":[");
    for (x = i; x >= b + 2; x -= 3) printf("%s%d", x == i ? "" : ",", bt[k].z[x]);
    printf("],\"transitions\":[");
    for (x = i; x >= b + 2; x -= 3) printf("%s%d", x == i ? "" : ",", bt[k].transitions[x]);
    printf("]}");
    free(bt[k].states); free(bt[k].z); free(bt[k].transitions); free(bt[k].scores);
  }
  printf("]}");
  free(bt);
}

/* order-independent digest of one strand's Sk (used to pin the DP itself) */
static void dump_sk_digest(const char *key, float ****SSk, int N, int L) {
  int k, x, b, i;
  double sum = 0.0, wsum = 0.0;
  for (k = 1; k < N; k++)
    for (x = 0; x < 3; x++)
      for (b = 1; b <= L; b++)
        for (i = b + 2; i <= L; i += 3) {
          double v = SSk[k][x][b][i];
          sum += v;
          wsum += v * (double)((k * 3 + x) * 131 + b * 7 + i);
        }
  printf("\"%s\":[%.17g,%.17g]", key, sum, wsum);
}

#define MAXROWS 500

int main(int argc, char **argv) {
  const char *path = NULL;
  int a;
  FILE *fp;
  char *line = NULL;
  size_t cap = 0;
  struct aln *inputAln[MAXROWS + 1];
  struct aln *inputAlnRev[MAXROWS + 1];
  int nrows = 0;
  char *userTree = NULL;
  int haveKappa = 0;
  float userKappa = 0;
  char blockId[256] = "";
  int want_bt = 1, want_digest = 1, dump_sample = -1;
  FILE *printTo = NULL;   /* --print-to: also run the reference's printResults (misc.c:392-552) into this file */

  pars.Delta = -10.0; pars.Omega = -4.0; pars.omega = -2.0;
  pars.stopPenalty_k = -8.0; pars.stopPenalty_0 = -9999.0;
  pars.inputFile = stdin; pars.outputFile = stdout; pars.debugFile = stdout;
  pars.bestOnly = 0; pars.bestRegion = 0; pars.stopEarly = 0; pars.postscript = 0;
  pars.postscript_cutoff = 0.05; pars.sampleN = 100; pars.blosum = 62;
  pars.cutoff = 1.0; pars.outputFormat = 0;

  for (a = 1; a < argc; a++) {
    if (!strcmp(argv[a], "--samples") && a + 1 < argc) pars.sampleN = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--seed") && a + 1 < argc) g_seed_base = strtoul(argv[++a], NULL, 10);
    else if (!strcmp(argv[a], "--blosum") && a + 1 < argc) pars.blosum = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--stop-early")) pars.stopEarly = 1;
    else if (!strcmp(argv[a], "--cutoff") && a + 1 < argc) pars.cutoff = (float)atof(argv[++a]);
    else if (!strcmp(argv[a], "--no-backtrack")) want_bt = 0;
    else if (!strcmp(argv[a], "--dump-sample") && a + 1 < argc) dump_sample = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--no-digest")) want_digest = 0;
    else if (!strcmp(argv[a], "--print-to") && a + 1 < argc) printTo = fopen(argv[++a], "w");
    else if (!strcmp(argv[a], "--format") && a + 1 < argc) pars.outputFormat = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--eps-dir") && a + 1 < argc) { pars.postscript = 1; strcpy(pars.postscriptDir, argv[++a]); }
    else if (!strcmp(argv[a], "--eps-cutoff") && a + 1 < argc) pars.postscript_cutoff = (float)atof(argv[++a]);
    else if (!strcmp(argv[a], "--best-only")) pars.bestOnly = 1;
    else if (!strcmp(argv[a], "--best-region")) pars.bestRegion = 1;
    else if (!strcmp(argv[a], "--pars") && a + 4 < argc) {
      pars.Delta = (float)atof(argv[++a]); pars.Omega = (float)atof(argv[++a]);
      pars.omega = (float)atof(argv[++a]); pars.stopPenalty_k = (float)atof(argv[++a]);
    } else path = argv[a];
  }
  if (!path) { fprintf(stderr, "usage: ref_harness blocks.txt [--samples n] [--seed s]\n"); return 2; }
  fp = fopen(path, "r");
  if (!fp) { perror(path); return 2; }

  ntMap['A'] = ntMap['a'] = 0; ntMap['C'] = ntMap['c'] = 1;
  ntMap['G'] = ntMap['g'] = 2; ntMap['T'] = ntMap['t'] = 3; ntMap['U'] = ntMap['u'] = 3;

  while (getline(&line, &cap, fp) > 0) {
    size_t n = strlen(line);
    while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    if (line[0] == 'B' && line[1] == ' ') {
      strncpy(blockId, line + 2, sizeof blockId - 1);
      nrows = 0; free(userTree); userTree = NULL; haveKappa = 0;
    } else if (line[0] == 'S' && line[1] == ' ') {
      char *name = (char *)malloc(n + 1), *seq = (char *)malloc(n + 1);
      int start, length, fullLength; char strand;
      if (sscanf(line + 2, "%s %d %d %c %d %s", name, &start, &length, &strand, &fullLength, seq) != 6) {
        fprintf(stderr, "bad S line\n"); return 2;
      }
      inputAln[nrows++] = createAlnEntry(name, seq, start, length, fullLength, strand);
      inputAln[nrows] = NULL;
    } else if (line[0] == 'T' && line[1] == ' ') {
      userTree = strdup(line + 2);
    } else if (line[0] == 'K' && line[1] == ' ') {
      userKappa = strtof(line + 2, NULL); haveKappa = 1;
    } else if (line[0] == 'E') {
      /* ---- one block, following src/RNAcode.c:115-221 ---- */
      int i, j, k, L, N, hssCount, rc;
      char *treeString; float kappa, maxScore, parMu = 0, parLambda = 0;
      TTree *tree; segmentStats *results;

      for (i = 0; inputAln[i] != NULL; i++)
        for (j = 0; inputAln[i]->seq[j]; j++) inputAln[i]->seq[j] = toupper(inputAln[i]->seq[j]);
      L = getSeqLength(inputAln[0]->seq);
      for (N = 0; inputAln[N] != NULL; N++);
      printf("{\"id\":"); jstr(blockId);
      printf(",\"N\":%d,\"L\":%d,\"cols\":%d", N, L, (int)strlen(inputAln[0]->seq));
      if (N <= 2) { printf(",\"skipped\":\"too few sequences\"}\n"); freeAln(inputAln); nrows = 0; continue; }
      if (L < 3) { printf(",\"skipped\":\"too short\"}\n"); freeAln(inputAln); nrows = 0; continue; }

      if (userTree) {
        treeString = strdup(userTree);
        kappa = haveKappa ? userKappa : 2.0f;
        printf(",\"tree_source\":\"given\"");
      } else {
        if (treeML((const struct aln **)inputAln, &treeString, &kappa) == 0) {
          printf(",\"skipped\":\"tree failed\"}\n"); freeAln(inputAln); nrows = 0; continue;
        }
        printf(",\"tree_source\":\"treeML\"");
      }
      printf(",\"tree\":"); jstr(treeString);
      printf(",\"kappa\":%.9g", (double)kappa);

      tree = string2tree(treeString);
      free(treeString);

      copyAln((struct aln **)inputAln, (struct aln **)inputAlnRev);
      revAln((struct aln **)inputAlnRev);
      models = getModels(tree, (struct aln **)inputAln, kappa);
      modelsRev = getModels(tree, inputAlnRev, kappa);
      putchar(','); dump_models("models", models, N);
      putchar(','); dump_models("modelsRev", modelsRev, N);

      Sk = NULL; Sk_native = NULL; Sk_native_rev = NULL;
      results = scoreAln((const struct aln **)inputAln, tree, kappa, 1);
      hssCount = 0;
      while (results[hssCount++].score > 0.0);
      qsort((segmentStats *)results, hssCount, sizeof(segmentStats), compareScores);
      maxScore = results[0].score;
      if (want_digest) {
        putchar(','); dump_sk_digest("sk_digest_fwd", Sk_native, N, L);
        putchar(','); dump_sk_digest("sk_digest_rev", Sk_native_rev, N, L);
      }

      if (want_bt && results[0].score > 0.0) {
        putchar(',');
        if (results[0].strand == '+')
          dump_backtrack("backtrack", results[0].start, results[0].end, Sk_native, (const struct aln **)inputAln, N);
        else
          dump_backtrack("backtrack", results[0].start, results[0].end, Sk_native_rev, (const struct aln **)inputAlnRev, N);
      }

      if (dump_sample >= 0) {   /* debugging aid: one null sample as getExtremeValuePars draws and scores it (score.c:1005-1031) */
        struct aln *sampledAln[MAX_NUM_NAMES];
        segmentStats *sr;
        int x;
        g_seed_counter = (unsigned long)dump_sample;
        simulateTree(tree, models[0].freqs, models[0].kappa, (int)strlen(inputAln[0]->seq));
        tree2aln(tree, sampledAln);
        sortAln((const struct aln **)inputAln, sampledAln);
        reintroduceGaps((const struct aln **)inputAln, sampledAln);
        fprintf(stderr, "sample %d\n", dump_sample);
        for (x = 0; sampledAln[x] != NULL; x++) fprintf(stderr, "R %s %s\n", sampledAln[x]->name, sampledAln[x]->seq);
        sr = scoreAln((const struct aln **)sampledAln, tree, kappa, 0);
        for (x = 0; sr[x].score >= 0; x++)
          fprintf(stderr, "H %c %d %d %d %.9g\n", sr[x].strand, sr[x].frame, sr[x].startSite, sr[x].endSite, (double)sr[x].score);
      }
      g_seed_counter = 0; g_max_n = 0; g_fit_rc = -2;
      rc = getExtremeValuePars(tree, (const struct aln **)inputAln, pars.sampleN, maxScore, &parMu, &parLambda);
      for (i = 0; i < hssCount; i++) {
        if (rc == 1) results[i].pvalue = 1 - exp((-1) * exp((-1) * parLambda * (results[i].score - parMu)));
        else results[i].pvalue = 99.0;
      }
      printf(",\"maxNative\":%.9g,\"evd_rc\":%d,\"fit_rc\":%d,\"mu\":%.9g,\"lambda\":%.9g,\"mu_d\":%.17g,\"lambda_d\":%.17g",
             (double)maxScore, rc, g_fit_rc, (double)parMu, (double)parLambda, g_mu_d, g_lambda_d);
      printf(",\"maxScores\":[");
      for (i = 0; i < g_max_n; i++) printf("%s%.9g", i ? "," : "", g_max[i]);
      printf("],\"hss\":[");
      for (i = 0; i < hssCount && results[i].score > 0.0; i++) {
        printf("%s{\"strand\":\"%c\",\"frame\":%d,\"startSite\":%d,\"endSite\":%d,\"start\":%d,\"end\":%d,"
               "\"startGenomic\":%d,\"endGenomic\":%d,\"score\":%.9g,\"pvalue\":%.9g,\"name\":",
               i ? "," : "", results[i].strand, results[i].frame, results[i].startSite, results[i].endSite,
               results[i].start, results[i].end, results[i].startGenomic, results[i].endGenomic,
               (double)results[i].score, (double)results[i].pvalue);
        jstr(results[i].name);
        printf("}");
      }
      printf("]}\n");
      fflush(stdout);
      if (printTo) printResults(printTo, pars.outputFormat, (const struct aln **)inputAln, results);

      for (k = 0; k < N; k++) {
        int x;
        for (x = 0; x < 3; x++) {
          for (i = 0; i < L + 1; i++) { free(Sk[k][x][i]); free(Sk_native[k][x][i]); free(Sk_native_rev[k][x][i]); }
          free(Sk[k][x]); free(Sk_native[k][x]); free(Sk_native_rev[k][x]);
        }
        free(Sk[k]); free(Sk_native[k]); free(Sk_native_rev[k]);
      }
      free(Sk); free(Sk_native); free(Sk_native_rev);
      Sk = Sk_native = Sk_native_rev = NULL;
      freeSeqgenTree(tree);
      freeResults(results);
      freeModels(models, N);
      freeModels(modelsRev, N);
      freeAln((struct aln **)inputAln);
      freeAln((struct aln **)inputAlnRev);
      nrows = 0;
    }
  }
  free(line);
  fclose(fp);
  if (printTo) fclose(printTo);
  return 0;
}
