/* rnacode_oracle.c -- TEST INFRASTRUCTURE ONLY (see rnacode_oracle.h).
 *
 * CPU restatement of the reference algorithm.  Compile with -ffp-contract=off and without
 * -ffast-math: every float/double promotion below is the one the reference's C expressions
 * produce on x86-64 (FLT_EVAL_METHOD == 0).  Citations are reference file:line.
 */
#define _POSIX_C_SOURCE 200809L
#include "rnacode_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables (code.c:26-88) */

/* amino-acid order of the reference's substitution matrices */
static const char AA_ORDER[] = "ARNDCQEGHILKMFPSTWYV";
/* standard genetic code, codon index = 16*n1 + 4*n2 + n3 with A=0 C=1 G=2 T=3 */
static const char GENETIC_CODE[] =
    "KNKNTTTTRSRSIIMI"
    "QHQHPPPPRRRRLLLL"
    "EDEDAAAAGGGGVVVV"
    "*Y*YSSSS*CWCLFLF";

/* lower triangles of BLOSUM62 / BLOSUM90 in AA_ORDER (the matrices are symmetric) */
static const signed char B62_TRI[] = {
    4,
    -1, 5,
    -2, 0, 6,
    -2, -2, 1, 6,
    0, -3, -3, -3, 9,
    -1, 1, 0, 0, -3, 5,
    -1, 0, 0, 2, -4, 2, 5,
    0, -2, 0, -1, -3, -2, -2, 6,
    -2, 0, 1, -1, -3, 0, 0, -2, 8,
    -1, -3, -3, -3, -1, -3, -3, -4, -3, 4,
    -1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4,
    -1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5,
    -1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5,
    -2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6,
    -1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7,
    1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4,
    0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5,
    -3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11,
    -2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7,
    0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4};
static const signed char B90_TRI[] = {
    5,
    -2, 6,
    -2, -1, 7,
    -3, -3, 1, 7,
    -1, -5, -4, -5, 9,
    -1, 1, 0, -1, -4, 7,
    -1, -1, -1, 1, -6, 2, 6,
    0, -3, -1, -2, -4, -3, -3, 6,
    -2, 0, 0, -2, -5, 1, -1, -3, 8,
    -2, -4, -4, -5, -2, -4, -4, -5, -4, 5,
    -2, -3, -4, -5, -2, -3, -4, -5, -4, 1, 5,
    -1, 2, 0, -1, -4, 1, 0, -2, -1, -4, -3, 6,
    -2, -2, -3, -4, -2, 0, -3, -4, -3, 1, 2, -2, 7,
    -3, -4, -4, -5, -3, -4, -5, -5, -2, -1, 0, -4, -1, 7,
    -1, -3, -3, -3, -4, -2, -2, -3, -3, -4, -4, -2, -3, -4, 8,
    1, -1, 0, -1, -2, -1, -1, -1, -2, -3, -3, -1, -2, -3, -2, 5,
    0, -2, 0, -2, -2, -1, -1, -3, -2, -1, -2, -1, -1, -3, -2, 1, 6,
    -4, -4, -5, -6, -4, -3, -5, -4, -3, -4, -3, -5, -2, 0, -5, -4, -4, 11,
    -3, -3, -3, -4, -4, -3, -4, -5, 1, -2, -2, -3, -2, 3, -4, -3, -2, 2, 8,
    -1, -3, -4, -5, -2, -3, -3, -5, -4, 3, 0, -3, 0, -2, -3, -2, -1, -3, -3, 5};

static int g_tables_ready = 0;
static int g_pep[64];       /* amino-acid index of a codon, -1 = stop (code.c transcode) */
static int g_b62[20][20], g_b90[20][20];
static int g_ntmap[256];    /* score.c:41 + RNAcode.c:94-98: everything unknown maps to 0 */

static void init_tables(void) {
  int c, i, j, t;
  if (g_tables_ready) return;
  for (c = 0; c < 64; c++) {
    const char *p = strchr(AA_ORDER, GENETIC_CODE[c]);
    g_pep[c] = (GENETIC_CODE[c] == '*' || !p) ? -1 : (int)(p - AA_ORDER);
  }
  for (i = 0, t = 0; i < 20; i++)
    for (j = 0; j <= i; j++, t++) {
      g_b62[i][j] = g_b62[j][i] = B62_TRI[t];
      g_b90[i][j] = g_b90[j][i] = B90_TRI[t];
    }
  memset(g_ntmap, 0, sizeof g_ntmap);
  g_ntmap['A'] = g_ntmap['a'] = 0;
  g_ntmap['C'] = g_ntmap['c'] = 1;
  g_ntmap['G'] = g_ntmap['g'] = 2;
  g_ntmap['T'] = g_ntmap['t'] = 3;
  g_ntmap['U'] = g_ntmap['u'] = 3;
  g_tables_ready = 1;
}

static int blosum_at(int blosum, int a, int b) { return blosum == 90 ? g_b90[a][b] : g_b62[a][b]; }
static int pep_of(int n1, int n2, int n3) { return g_pep[16 * n1 + 4 * n2 + n3]; }

void orc_default_params(orc_params *p) {
  p->Delta = -10.0f; p->Omega = -4.0f; p->omega = -2.0f;
  p->stopPenalty_0 = -9999.0f; p->stopPenalty_k = -8.0f;
  p->blosum = 62; p->sampleN = 100; p->cutoff = 1.0f; p->stopEarly = 0;
}

/* ------------------------------------------------------------------ a4: score.c:204-244 */

float orc_prob_hky(int i, int j, float d, const float freqs[4], float kappa) {
  float piA = freqs[0], piC = freqs[1], piG = freqs[2], piT = freqs[3];
  float piR, piY, r, l, k1, k2, e1, e22, e21;
  float P[4][4];
  piR = piA + piG;
  piY = piT + piC;
  /* float products and sums left to right, widened only by the double literals (score.c:217) */
  r = 1. / (2. * (piA * piC + piC * piG + piA * piT + piG * piT + kappa * (piC * piT + piA * piG)));
  l = r * d;
  k1 = kappa * piY + piR;
  k2 = kappa * piR + piY;
  e1 = exp(-l);
  e22 = exp(-k2 * l);
  e21 = exp(-k1 * l);
  /* purine rows/cols use e22, pyrimidine ones e21; transversions only e1 (score.c:225-240) */
  P[0][0] = piA * (1. + (piY / piR) * e1) + (piG / piR) * e22;
  P[0][2] = piG * (1. + (piY / piR) * e1) - (piG / piR) * e22;
  P[2][0] = piA * (1. + (piY / piR) * e1) - (piA / piR) * e22;
  P[2][2] = piG * (1. + (piY / piR) * e1) + (piA / piR) * e22;
  P[1][1] = piC * (1. + (piR / piY) * e1) + (piT / piY) * e21;
  P[1][3] = piT * (1. + (piR / piY) * e1) - (piT / piY) * e21;
  P[3][1] = piC * (1. + (piR / piY) * e1) - (piC / piY) * e21;
  P[3][3] = piT * (1. + (piR / piY) * e1) + (piC / piY) * e21;
  P[0][1] = P[2][1] = piC * (1. - e1);
  P[0][3] = P[2][3] = piT * (1. - e1);
  P[1][0] = P[3][0] = piA * (1. - e1);
  P[1][2] = P[3][2] = piG * (1. - e1);
  return P[i][j];
}

/* ------------------------------------------------------------------ a5: score.c:107-193 */

static int hdist3(int a1, int a2, int a3, int b1, int b2, int b3) { /* misc.c:303-313 */
  return (a1 != b1) + (a2 != b2) + (a3 != b3);
}

void orc_calculate_bg(float dist, const float freqs[4], float kappa, int blosum,
                      float scores[4], float probs_out[4]) {
  int a1, a2, a3, b1, b2, b3, i, j, h, pepA, pepB;
  float f, prob, score, probStop;
  float counts[4];
  float P[4][4];
  init_tables();
  for (i = 0; i < 4; i++)
    for (j = 0; j < 4; j++) P[i][j] = orc_prob_hky(i, j, dist, freqs, kappa);
  counts[0] = counts[1] = counts[2] = counts[3] = 0.0;
  scores[0] = scores[1] = scores[2] = scores[3] = 0.0;
  probStop = 0;
  /* pass 1: probability mass of codon pairs containing a stop (score.c:126-148) */
  for (a1 = 0; a1 < 4; a1++) for (a2 = 0; a2 < 4; a2++) for (a3 = 0; a3 < 4; a3++)
    for (b1 = 0; b1 < 4; b1++) for (b2 = 0; b2 < 4; b2++) for (b3 = 0; b3 < 4; b3++) {
      pepA = pep_of(a1, a2, a3);
      pepB = pep_of(b1, b2, b3);
      if (pepA != -1 && pepB != -1) continue;
      f = freqs[a1] * freqs[a2] * freqs[a3];
      prob = P[a1][b1] * P[a2][b2] * P[a3][b3];
      prob *= f;
      probStop += prob;
    }
  /* pass 2: conditional probabilities and expected scores per Hamming class (score.c:151-187) */
  for (a1 = 0; a1 < 4; a1++) for (a2 = 0; a2 < 4; a2++) for (a3 = 0; a3 < 4; a3++) {
    pepA = pep_of(a1, a2, a3);
    if (pepA == -1) continue;
    for (b1 = 0; b1 < 4; b1++) for (b2 = 0; b2 < 4; b2++) for (b3 = 0; b3 < 4; b3++) {
      pepB = pep_of(b1, b2, b3);
      if (pepB == -1) continue;
      h = hdist3(a1, a2, a3, b1, b2, b3);
      f = freqs[a1] * freqs[a2] * freqs[a3];
      prob = P[a1][b1] * P[a2][b2] * P[a3][b3];
      prob *= f;
      prob /= (1 - probStop);
      score = blosum_at(blosum, pepA, pepB);
      counts[h] += prob;
      scores[h] += score * prob;
    }
  }
  for (i = 0; i < 4; i++) {
    scores[i] /= counts[i];
    probs_out[i] = counts[i];
  }
}

/* ------------------------------------------------------------------ a2: score.c:255-280 */

void orc_count_freqs(const char *const *rows, int N, float freqs[4]) {
  int i, k;
  unsigned long counter = 0;
  char c;
  init_tables();
  for (i = 0; i < 4; i++) freqs[i] = 0.0;
  for (i = 0; i < N; i++) {
    const char *s = rows[i];
    k = 0;
    while ((c = s[k++]) != '\0') {
      if (c == '-') continue;
      freqs[g_ntmap[(unsigned char)c]]++;
      counter++;
    }
  }
  for (i = 0; i < 4; i++) freqs[i] /= (float)counter;
}

/* ------------------------------------------------------------------ Newick reader (seqgen/treefile.c:301-530) */

typedef struct onode {
  struct onode *parent;
  struct onode *child[3];
  int nchild;
  double length0;  /* branch to parent, read with %lf (treefile.c:431) */
  int tipNo;       /* -1 for internal (treefile.c:95) */
  char *seq;       /* state codes during simulation */
} onode;

typedef struct {
  onode *root;
  onode **nodes; int numNodes;
  onode **tips; char **names; int numTips;
  int rooted;
} otree;

typedef struct { const char *s; int pos; otree *t; int err; } nparser;

static void skip_ws(nparser *p) { while (p->s[p->pos] == ' ' || p->s[p->pos] == '\t' || p->s[p->pos] == '\n' || p->s[p->pos] == '\r') p->pos++; }

static onode *new_node(otree *t) {
  onode *n = (onode *)calloc(1, sizeof(onode));
  n->tipNo = -1;
  t->nodes = (onode **)realloc(t->nodes, sizeof(onode *) * (t->numNodes + 1));
  t->nodes[t->numNodes++] = n;
  return n;
}

static onode *parse_branch(nparser *p, int is_root_level);

/* ReadNode (treefile.c:363-404): exactly two children, except the root may carry a third
 * (unrooted tree, treefile.c:511-519). Labels after ')' are skipped. */
static onode *parse_internal(nparser *p, int allow_three) {
  onode *n = new_node(p->t);
  for (;;) {
    onode *c;
    if (n->nchild == 3 || (n->nchild == 2 && !allow_three)) { p->err = 1; return n; }
    c = parse_branch(p, 0);
    if (p->err) return n;
    n->child[n->nchild++] = c;
    c->parent = n;
    skip_ws(p);
    if (p->s[p->pos] == ',') { p->pos++; continue; }
    if (p->s[p->pos] == ')') { p->pos++; break; }
    p->err = 1; return n;
  }
  if (n->nchild < 2) p->err = 1;
  /* skip an internal label / support value */
  while (p->s[p->pos] && p->s[p->pos] != ':' && p->s[p->pos] != ',' && p->s[p->pos] != ')' && p->s[p->pos] != ';') p->pos++;
  return n;
}

static onode *parse_branch(nparser *p, int is_root_level) {
  onode *n;
  skip_ws(p);
  if (p->s[p->pos] == '(') {
    p->pos++;
    n = parse_internal(p, is_root_level);
    if (p->err) return n;
  } else {
    /* ReadTip (treefile.c:301-360): name runs to ':' ',' or ')', white space dropped */
    char name[257]; int len = 0;
    n = new_node(p->t);
    while (p->s[p->pos] && p->s[p->pos] != ':' && p->s[p->pos] != ',' && p->s[p->pos] != ')' && len < 256) {
      char ch = p->s[p->pos++];
      if (!(ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r')) name[len++] = ch;
    }
    name[len] = 0;
    n->tipNo = p->t->numTips;
    p->t->tips = (onode **)realloc(p->t->tips, sizeof(onode *) * (p->t->numTips + 1));
    p->t->names = (char **)realloc(p->t->names, sizeof(char *) * (p->t->numTips + 1));
    p->t->tips[p->t->numTips] = n;
    p->t->names[p->t->numTips] = strdup(name);
    p->t->numTips++;
  }
  skip_ws(p);
  if (p->s[p->pos] == ':') {
    char *end;
    p->pos++;
    n->length0 = strtod(p->s + p->pos, &end);
    if (end == p->s + p->pos) { p->err = 1; return n; }
    p->pos = (int)(end - p->s);
    skip_ws(p);
    if (p->s[p->pos] == '[') { while (p->s[p->pos] && p->s[p->pos] != ']') p->pos++; if (p->s[p->pos]) p->pos++; }
  }
  return n;
}

static void free_tree(otree *t) {
  int i;
  if (!t) return;
  for (i = 0; i < t->numNodes; i++) { free(t->nodes[i]->seq); free(t->nodes[i]); }
  for (i = 0; i < t->numTips; i++) free(t->names[i]);
  free(t->nodes); free(t->tips); free(t->names); free(t);
}

static otree *parse_newick(const char *s) {
  nparser p; otree *t = (otree *)calloc(1, sizeof(otree));
  p.s = s; p.pos = 0; p.t = t; p.err = 0;
  while (s[p.pos] && s[p.pos] != '(') p.pos++;
  if (!s[p.pos]) { free_tree(t); return NULL; }
  p.pos++;
  t->root = parse_internal(&p, 1);
  if (p.err) { free_tree(t); return NULL; }
  t->rooted = (t->root->nchild == 2);
  return t;
}

static int tip_by_name(const otree *t, const char *name) {
  int k;
  for (k = 0; k < t->numTips; k++) if (strcmp(t->names[k], name) == 0) return k;
  return -1;
}

/* ------------------------------------------------------------------ a3: treeSimulate.c:118-251 */

static onode *lca_of(onode *a, onode *b) {
  onode *x, *y;
  for (x = a; x; x = x->parent)
    for (y = b; y; y = y->parent)
      if (x == y) return x;
  return NULL;
}

static int tip_distances(const otree *t, const char *const *names, int N, float *dist) {
  int k, i0 = tip_by_name(t, names[0]);
  if (i0 < 0) return -1;
  dist[0] = 0.0;
  for (k = 1; k < N; k++) {
    int ik = tip_by_name(t, names[k]);
    onode *a, *b, *l;
    float sum = 0.0;
    if (ik < 0) return -1;
    a = t->tips[i0]; b = t->tips[ik];
    l = lca_of(a, b);
    /* float accumulator += double branch length (treeSimulate.c:164-182) */
    while (a != l) { sum += a->length0; a = a->parent; }
    while (b != l) { sum += b->length0; b = b->parent; }
    dist[k] = sum;
  }
  return 0;
}

int orc_tip_distances(const char *newick, const char *const *names, int N, float *dist) {
  otree *t = parse_newick(newick);
  int rc;
  if (!t) return -1;
  rc = tip_distances(t, names, N, dist);
  free_tree(t);
  return rc;
}

/* ------------------------------------------------------------------ a5: score.c:291-344 */

static int get_models_tree(const otree *t, const char *const *rows, const char *const *names, int N,
                           float kappa, int blosum, orc_model *models) {
  float freqs[4];
  float *dist = (float *)malloc(sizeof(float) * N);
  int j;
  if (tip_distances(t, names, N, dist) != 0) { free(dist); return -1; }
  orc_count_freqs(rows, N, freqs);
  for (j = 0; j < N; j++) {
    models[j].dist = dist[j];
    models[j].kappa = kappa;
    memcpy(models[j].freqs, freqs, sizeof freqs);
    orc_calculate_bg(models[j].dist, models[j].freqs, models[j].kappa, blosum, models[j].scores, models[j].probs);
  }
  free(dist);
  return 0;
}

int orc_get_models(const char *newick, const char *const *rows, const char *const *names, int N,
                   float kappa, int blosum, orc_model *models) {
  otree *t = parse_newick(newick);
  int rc;
  if (!t) return -1;
  rc = get_models_tree(t, rows, names, N, kappa, blosum, models);
  free_tree(t);
  return rc;
}

/* ------------------------------------------------------------------ rnaz_utils.c:316-348 */

void orc_rev_aln(const char *const *rows, int N, char **out) {
  int i, j, n = (int)strlen(rows[0]);
  for (i = 0; i < N; i++) {
    char *t = (char *)malloc(n + 1);
    for (j = 0; j < n; j++) {
      char c = rows[i][n - 1 - j];
      if (c == 'T' || c == 'U') c = 'A';
      else if (c == 'A') c = 'T';
      else if (c == 'C') c = 'G';
      else if (c == 'G') c = 'C';
      t[j] = c;
    }
    t[n] = 0;
    out[i] = t;
  }
}

/* ------------------------------------------------------------------ a6: misc.c:186-289 */

static int seq_length(const char *s) { /* misc.c:272-289 */
  int n = 0;
  for (; *s; s++) if (*s != '-') n++;
  return n;
}

/* col[pos] = 1-based column of the pos-th residue (misc.c:250-269, pos2col for all pos at once) */
static int *residue_columns(const char *s, int L) {
  int *col = (int *)calloc(L + 2, sizeof(int));
  int i, p = 0;
  for (i = 0; s[i]; i++) if (s[i] != '-') { p++; if (p <= L) col[p] = i + 1; }
  return col;
}

/* misc.c:186-246: columns (col(i-3), col(i)] of both rows (from column 1 when i == 3) and z */
static void get_block(int i, const char *seq0, const char *seqk, const int *map0, char *b0, char *bk, int *z) {
  int start = (i > 3) ? map0[i - 3] + 1 : 1;
  int end = map0[i];
  int n = end - start + 1, c, g0 = 0, gk = 0, diff;
  memcpy(b0, seq0 + start - 1, n); b0[n] = 0;
  memcpy(bk, seqk + start - 1, n); bk[n] = 0;
  for (c = 0; c < n; c++) { if (b0[c] == '-') g0++; if (bk[c] == '-') gk++; }
  diff = gk - g0;
  if (diff < 0) diff = -diff;
  *z = (diff % 3 == 0) ? 0 : (diff % 3 == 1) ? +1 : -1;
}

/* ------------------------------------------------------------------ a7: score.c:375-426 */

static float calculate_sigma(const char *b0, const char *bk, const orc_model *mk, const orc_params *par) {
  char cA[3] = {'X', 'X', 'X'}, cB[3] = {'X', 'X', 'X'};
  int i, j = 0, h, pepA, pepB, a[3], b[3];
  for (i = 0; b0[i]; i++)
    if (b0[i] != '-') { if (j < 3) { cA[j] = b0[i]; cB[j] = bk[i]; } j++; }
  if (cB[0] == 'X' && cB[1] == 'X' && cB[2] == 'X') return 0.0; /* score.c:394, unreachable */
  for (i = 0; i < 3; i++) if (cA[i] == 'N' || cB[i] == 'N') return 0.0;
  for (i = 0; i < 3; i++) { a[i] = g_ntmap[(unsigned char)cA[i]]; b[i] = g_ntmap[(unsigned char)cB[i]]; }
  h = hdist3(a[0], a[1], a[2], b[0], b[1], b[2]);
  if (h == 0) return 0.0;
  pepA = pep_of(a[0], a[1], a[2]);
  pepB = pep_of(b[0], b[1], b[2]);
  if (pepA == -1) return par->stopPenalty_0;
  if (pepB == -1) return par->stopPenalty_k;
  return (float)blosum_at(par->blosum, pepA, pepB) - mk->scores[h];
}

/* ------------------------------------------------------------------ a8: score.c:441-556 */

#define SK(sk, L, k, x, b, i) ((sk)[((((size_t)(k)) * 3 + (x)) * ((L) + 1) + (b)) * ((L) + 1) + (i)])
#define FMAX(a, b) (((a) > (b)) ? (a) : (b))

static void pairwise_matrix(const char *const *rows, int N, const orc_model *models, const orc_params *par, float *sk, int L) {
  const char *seq0 = rows[0];
  int cols = (int)strlen(seq0), k, x, b, i, z;
  int *map0 = residue_columns(seq0, L);
  char *b0 = (char *)malloc(cols + 1), *bk = (char *)malloc(cols + 1);
  int *zs = (int *)malloc(sizeof(int) * (L + 1));
  float *sig = (float *)malloc(sizeof(float) * (L + 1));
  for (k = 1; k < N; k++) {
    for (x = 3; x <= L; x++) {
      get_block(x, seq0, rows[k], map0, b0, bk, &z);
      zs[x] = z;
      sig[x] = calculate_sigma(b0, bk, &models[k], par);
    }
    for (b = 1; b <= L; b++) {
      for (i = b + 2; i <= L; i += 3) {
        float p0, p1, p2;
        if (i - 3 < b) { p0 = p1 = p2 = 0.0; }
        else { p0 = SK(sk, L, k, 0, b, i - 3); p1 = SK(sk, L, k, 1, b, i - 3); p2 = SK(sk, L, k, 2, b, i - 3); }
        z = zs[i];
        if (z == 0) {
          SK(sk, L, k, 0, b, i) = p0 + sig[i];
          SK(sk, L, k, 1, b, i) = p1 + par->omega;
          SK(sk, L, k, 2, b, i) = p2 + par->omega;
        } else if (z == +1) {
          SK(sk, L, k, 0, b, i) = FMAX(p0 + par->Delta, p2 + par->Omega);
          SK(sk, L, k, 1, b, i) = FMAX(p0 + par->Omega, p1 + par->Delta);
          SK(sk, L, k, 2, b, i) = FMAX(p1 + par->Omega, p2 + par->Delta);
        } else {
          SK(sk, L, k, 0, b, i) = FMAX(p0 + par->Delta, p1 + par->Omega);
          SK(sk, L, k, 1, b, i) = FMAX(p1 + par->Delta, p2 + par->Omega);
          SK(sk, L, k, 2, b, i) = FMAX(p2 + par->Delta, p0 + par->Omega);
        }
      }
    }
  }
  free(map0); free(b0); free(bk); free(zs); free(sig);
}

/* ------------------------------------------------------------------ a9: score.c:811-848 */

static float *multiple_matrix(const float *sk, int N, int L, const orc_params *par) {
  float *S = (float *)calloc((size_t)(L + 1) * (L + 1), sizeof(float));
  int b, i, k;
  for (b = 1; b <= L; b++)
    for (i = b + 2; i <= L; i += 3) {
      float sum = 0, m;
      for (k = 1; k < N; k++) {
        m = FMAX(FMAX(SK(sk, L, k, 0, b, i), SK(sk, L, k, 1, b, i)), SK(sk, L, k, 2, b, i));
        sum += m;
      }
      m = FMAX(FMAX(sum, S[(size_t)b * (L + 1) + i - 1] + par->Delta), S[(size_t)b * (L + 1) + i - 2] + par->Delta);
      S[(size_t)b * (L + 1) + i] = m / (N - 1);
    }
  return S;
}

/* ------------------------------------------------------------------ a10: score.c:864-974 */

/* diagnostic for the tests: how often getHSS's "equal score, segment at least as long" rule (score.c:953-954)
 * decided a replacement that "v > currMax" alone would not have made */
static unsigned long tie_replacements = 0;
unsigned long orc_tie_replacements(int reset) {
  unsigned long r = tie_replacements;
  if (reset) tie_replacements = 0;
  return r;
}

static int get_hss(const float *S, int L, int ref_start, int ref_length, char strand, orc_hss *out, int cap) {
  int frame, sites, i, j, n = 0;
  const int minSegmentLength = 2;
  for (frame = 0; frame <= 2; frame++) {
    float currMax = 0.0, v;
    int segStart = -1, segEnd = -1, last;
    sites = (L - frame) / 3;
    for (i = 0; i < sites; i++)
      for (j = i; j < sites; j++) {
        v = S[(size_t)(i * 3 + 1 + frame) * (L + 1) + (j * 3 + 3 + frame)];
        last = (i == sites - 1 && j == sites - 1);
        if (v > 0.0 || last) {
          if ((currMax > 0.0 && segEnd < i) || last) {
            if (segEnd - segStart >= minSegmentLength) {
              if (n < cap) {
                orc_hss *r = &out[n];
                r->strand = strand; r->frame = frame;
                r->startSite = segStart; r->endSite = segEnd; r->score = currMax;
                r->start = segStart * 3 + frame + 1;
                r->end = segEnd * 3 + frame + 3;
                if (ref_start == 0 && ref_length == 0) { r->startGenomic = r->start; r->endGenomic = r->end; }
                else if (strand == '+') {
                  r->startGenomic = ref_start + segStart * 3 + frame;
                  r->endGenomic = ref_start + segEnd * 3 + frame + 2;
                } else {
                  r->endGenomic = (ref_start + ref_length - 1) - segStart * 3 - frame;
                  r->startGenomic = (ref_start + ref_length - 1) - segEnd * 3 - frame - 2;
                }
                r->pvalue = 0;
              }
              n++;
            }
            currMax = v; segStart = i; segEnd = j;
          } else if (v > currMax || ((fabs(v - currMax) < 0.0001) && ((j - i) >= (segEnd - segStart)))) {
            if (!(v > currMax)) tie_replacements++;
            currMax = v; segStart = i; segEnd = j;
          }
        }
      }
  }
  return n;
}

/* ------------------------------------------------------------------ a11: score.c:1067-1147 */

float *orc_score_matrix(const char *const *rows, int N, const orc_model *models, const orc_params *par, int *L_out) {
  int L;
  float *sk, *S;
  init_tables();
  L = seq_length(rows[0]);
  sk = (float *)calloc((size_t)N * 3 * (L + 1) * (L + 1), sizeof(float));
  pairwise_matrix(rows, N, models, par, sk, L);
  S = multiple_matrix(sk, N, L, par);
  free(sk);
  if (L_out) *L_out = L;
  return S;
}

int orc_score_aln(const char *const *rows, int N, int ref_start, int ref_length,
                  const orc_model *models, const orc_model *modelsRev, const orc_params *par,
                  orc_hss *hss, int cap, float **sk_fwd, float **sk_rev) {
  int L, i, n = 0, nf, nr;
  char **rev = (char **)malloc(sizeof(char *) * N);
  float *sk, *S;
  orc_hss *tmp = (orc_hss *)malloc(sizeof(orc_hss) * (cap > 0 ? cap : 1));
  size_t skn;
  init_tables();
  L = seq_length(rows[0]);
  skn = (size_t)N * 3 * (L + 1) * (L + 1);
  orc_rev_aln(rows, N, rev);
  sk = (float *)calloc(skn, sizeof(float));

  pairwise_matrix(rows, N, models, par, sk, L);
  S = multiple_matrix(sk, N, L, par);
  if (sk_fwd) { *sk_fwd = (float *)malloc(skn * sizeof(float)); memcpy(*sk_fwd, sk, skn * sizeof(float)); }
  nf = get_hss(S, L, ref_start, ref_length, '+', tmp, cap);
  free(S);
  for (i = 0; i < nf && i < cap; i++) if (tmp[i].score > 0.0) { if (n < cap) hss[n] = tmp[i]; n++; } else break;

  /* the global Sk is reused, not cleared, between strands (score.c:1096) */
  pairwise_matrix((const char *const *)rev, N, modelsRev, par, sk, L);
  S = multiple_matrix(sk, N, L, par);
  if (sk_rev) { *sk_rev = (float *)malloc(skn * sizeof(float)); memcpy(*sk_rev, sk, skn * sizeof(float)); }
  nr = get_hss(S, L, ref_start, ref_length, '-', tmp, cap);
  free(S);
  for (i = 0; i < nr && i < cap; i++) if (tmp[i].score > 0.0) { if (n < cap) hss[n] = tmp[i]; n++; } else break;

  free(sk); free(tmp);
  for (i = 0; i < N; i++) free(rev[i]);
  free(rev);
  return n;
}

/* ------------------------------------------------------------------ MT19937 (seqgen/twister.c:73-89,118-166) */

typedef struct { unsigned int mt[624]; int mti; } mt_state;

static void mt_seed(mt_state *s, unsigned long seed) {
  int i;
  s->mt[0] = (unsigned int)(seed & 0xffffffffUL);
  for (i = 1; i < 624; i++) s->mt[i] = 1812433253U * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (unsigned int)i;
  s->mti = 624;
}

static unsigned int mt_next(mt_state *s) {
  unsigned int y;
  if (s->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624; kk++) {
      y = (s->mt[kk] & 0x80000000U) | (s->mt[(kk + 1) % 624] & 0x7fffffffU);
      s->mt[kk] = s->mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    s->mti = 0;
  }
  y = s->mt[s->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

static double mt_real1(mt_state *s) { return mt_next(s) * (1.0 / 4294967295.0); } /* twister.c:162-166 */

void orc_mt_stream(unsigned long seed, unsigned int *out, int n) {
  mt_state s; int i;
  mt_seed(&s, seed);
  for (i = 0; i < n; i++) out[i] = mt_next(&s);
}

/* ------------------------------------------------------------------ a12: Seq-Gen HKY (nucmodels.c:87-131,187-196,302-362) */

typedef struct {
  double fA, fC, fG, fT, fR, fY;
  double t1A, t2A, t3A, t1C, t2C, t3C, t1G, t2G, t3G, t1T, t2T, t3T;
  double beta, beta_A_R, beta_A_Y;
  double addFreq[4];
} hky_model;

static void hky_setup(hky_model *m, const float freqs[4], float kap) {
  double fAG, fCT, tstv, kappa;
  /* treeSimulate.c:70-80 */
  m->fA = (double)freqs[0]; m->fC = (double)freqs[1]; m->fG = (double)freqs[2]; m->fT = (double)freqs[3];
  m->fR = m->fA + m->fG;
  m->fY = m->fC + m->fT;
  fAG = m->fA * m->fG;
  fCT = m->fC * m->fT;
  tstv = (double)kap * (fAG + fCT) / (m->fR * m->fY);
  /* nucmodels.c:99-131 */
  m->t1A = m->fA * ((1 / m->fR) - 1); m->t2A = (m->fR - m->fA) / m->fR; m->t3A = m->fA / m->fR;
  m->t1C = m->fC * ((1 / m->fY) - 1); m->t2C = (m->fY - m->fC) / m->fY; m->t3C = m->fC / m->fY;
  m->t1G = m->fG * ((1 / m->fR) - 1); m->t2G = (m->fR - m->fG) / m->fR; m->t3G = m->fG / m->fR;
  m->t1T = m->fT * ((1 / m->fY) - 1); m->t2T = (m->fY - m->fT) / m->fY; m->t3T = m->fT / m->fY;
  kappa = (tstv * m->fR * m->fY) / (fAG + fCT);
  m->beta = -1.0 / (2 * (m->fR * m->fY + kappa * (fAG + fCT)));
  m->beta_A_R = m->beta * (1.0 + m->fR * (kappa - 1));
  m->beta_A_Y = m->beta * (1.0 + m->fY * (kappa - 1));
  /* model.c:116-119 */
  m->addFreq[0] = m->fA;
  m->addFreq[1] = m->addFreq[0] + m->fC;
  m->addFreq[2] = m->addFreq[1] + m->fG;
  m->addFreq[3] = m->addFreq[2] + m->fT;
}

static void hky_matrix(const hky_model *m, double len, double M[16]) {
  double aa = exp(m->beta * len), bbR = exp(m->beta_A_R * len), bbY = exp(m->beta_A_Y * len);
  int r;
  M[0] = m->fA + (m->t1A * aa) + (m->t2A * bbR);   /* A->A */
  M[1] = m->fC * (1 - aa);                         /* A->C */
  M[2] = m->fG + (m->t1G * aa) - (m->t3G * bbR);   /* A->G */
  M[3] = m->fT * (1 - aa);                         /* A->T */
  M[4] = m->fA * (1 - aa);                         /* C->A */
  M[5] = m->fC + (m->t1C * aa) + (m->t2C * bbY);   /* C->C */
  M[6] = m->fG * (1 - aa);                         /* C->G */
  M[7] = m->fT + (m->t1T * aa) - (m->t3T * bbY);   /* C->T */
  M[8] = m->fA + (m->t1A * aa) - (m->t3A * bbR);   /* G->A */
  M[9] = M[1];
  M[10] = m->fG + (m->t1G * aa) + (m->t2G * bbR);  /* G->G */
  M[11] = M[3];
  M[12] = M[4];
  M[13] = m->fC + (m->t1C * aa) - (m->t3C * bbY);  /* T->C */
  M[14] = M[6];
  M[15] = m->fT + (m->t1T * aa) + (m->t2T * bbY);  /* T->T */
  for (r = 0; r < 4; r++) { M[4 * r + 1] += M[4 * r]; M[4 * r + 2] += M[4 * r + 1]; M[4 * r + 3] += M[4 * r + 2]; }
}

/* evolve.c:167-175; the reference can return 4 here and then reads out of bounds: clamp and count */
static int set_state(mt_state *rng, const double *P, long *clamped) {
  double r = mt_real1(rng);
  int j;
  for (j = 0; j < 4 && r > P[j]; j++);
  if (j == 4) { j = 3; if (clamped) (*clamped)++; }
  return j;
}

static void evolve_node(const hky_model *m, mt_state *rng, onode *anc, onode *des, int n, long *clamped) {
  double M[16];
  int i;
  memcpy(des->seq, anc->seq, n);                       /* evolve.c:406 */
  hky_matrix(m, des->length0, M);                      /* evolve.c:292, scale 1.0 */
  for (i = 0; i < n; i++) des->seq[i] = (char)set_state(rng, M + 4 * des->seq[i], clamped);
  if (des->tipNo == -1) {                              /* evolve.c:409-412 */
    evolve_node(m, rng, des, des->child[0], n, clamped);
    evolve_node(m, rng, des, des->child[1], n, clamped);
  }
}

static void simulate_tree(otree *t, const float freqs[4], float kap, int n, unsigned long seed, long *clamped) {
  hky_model m; mt_state rng; int i;
  hky_setup(&m, freqs, kap);
  mt_seed(&rng, seed);
  for (i = 0; i < t->numNodes; i++) { free(t->nodes[i]->seq); t->nodes[i]->seq = (char *)calloc(n + 1, 1); }
  for (i = 0; i < n; i++) t->root->seq[i] = (char)set_state(&rng, m.addFreq, clamped);   /* evolve.c:189-199 */
  evolve_node(&m, &rng, t->root, t->root->child[0], n, clamped);   /* evolve.c:429-432 */
  evolve_node(&m, &rng, t->root, t->root->child[1], n, clamped);
  if (!t->rooted) evolve_node(&m, &rng, t->root, t->root->child[2], n, clamped);
}

/* tree2aln + sortAln + reintroduceGaps (treeSimulate.c:254-283, misc.c:127-171) */
static int sampled_rows(otree *t, const char *const *rows, const char *const *names, int N, int n, char **out) {
  int i, j;
  char **seqs = (char **)malloc(sizeof(char *) * t->numTips);
  char **snames = (char **)malloc(sizeof(char *) * t->numTips);
  if (t->numTips != N) { free(seqs); free(snames); return -1; }
  for (i = 0; i < t->numTips; i++) {
    seqs[i] = (char *)malloc(n + 1);
    for (j = 0; j < n; j++) seqs[i][j] = "ACGT"[(int)t->tips[i]->seq[j]];
    seqs[i][n] = 0;
    snames[i] = t->names[i];
  }
  for (i = 0; i < N; i++)          /* misc.c:159-170 swap-into-place by name */
    for (j = 0; j < N; j++)
      if (strcmp(names[i], snames[j]) == 0) {
        char *ts = seqs[j], *tn = snames[j];
        seqs[j] = seqs[i]; snames[j] = snames[i];
        seqs[i] = ts; snames[i] = tn;
      }
  for (i = 0; i < N; i++) {
    for (j = 0; j < n; j++) if (rows[i][j] == '-') seqs[i][j] = '-';
    out[i] = seqs[i];
  }
  free(seqs); free(snames);
  return 0;
}

int orc_simulate_null(const char *newick, const char *const *rows, const char *const *names, int N,
                      const float freqs[4], float kappa, unsigned long seed, char **out_rows, long *clamped) {
  otree *t = parse_newick(newick);
  int rc, n = (int)strlen(rows[0]);
  if (!t) return -1;
  simulate_tree(t, freqs, kappa, n, seed, clamped);
  rc = sampled_rows(t, rows, names, N, n, out_rows);
  free_tree(t);
  return rc;
}

/* ------------------------------------------------------------------ a14: extreme_fit.c:36-65,157-251 */

static void lawless416(const double *x, int n, double lambda, double *ret_f, double *ret_df) {
  double esum = 0., xesum = 0., xxesum = 0., xsum = 0., total = 0., mult;
  int i;
  for (i = 0; i < n; i++) {
    mult = 1.;
    xsum += mult * x[i];
    xesum += mult * x[i] * exp(-1. * lambda * x[i]);
    xxesum += mult * x[i] * x[i] * exp(-1. * lambda * x[i]);
    esum += mult * exp(-1. * lambda * x[i]);
    total += mult;
  }
  *ret_f = 1. / lambda - xsum / total + xesum / esum;
  *ret_df = ((xesum / esum) * (xesum / esum)) - (xxesum / esum) - (1. / (lambda * lambda));
}

int orc_evd_fit(const double *x, int n, double *ret_mu, double *ret_lambda) {
  double lambda = 0.2, mu, fx, dfx, esum, total, tol = 1e-5;
  int i;
  for (i = 0; i < 100; i++) {
    lawless416(x, n, lambda, &fx, &dfx);
    if (fabs(fx) < tol) break;
    lambda = lambda - fx / dfx;
    if (lambda <= 0.) lambda = 0.001;
  }
  if (i == 100) {
    double left, right, mid = 0;
    lambda = right = left = 0.2;
    lawless416(x, n, lambda, &fx, &dfx);
    if (fx < 0.) {
      do {
        left -= 0.1;
        if (left < 0.) return 0;
        lawless416(x, n, left, &fx, &dfx);
      } while (fx < 0.);
    } else {
      do {
        right += 0.1;
        lawless416(x, n, right, &fx, &dfx);
        if (right > 100.) return 0;
      } while (fx > 0.);
    }
    for (i = 0; i < 100; i++) {
      mid = (left + right) / 2.;
      lawless416(x, n, mid, &fx, &dfx);
      if (fabs(fx) < tol) break;
      if (fx > 0.) left = mid; else right = mid;
    }
    if (i == 100) return 0;
    lambda = mid;
  }
  esum = 0.; total = 0.;
  for (i = 0; i < n; i++) { esum += 1. * exp(-1 * lambda * x[i]); total += 1.; }
  mu = -1. * log(esum / total) / lambda;
  *ret_lambda = lambda;
  *ret_mu = mu;
  return 1;
}

float orc_pvalue(float score, float parMu, float parLambda) { /* RNAcode.c:182 */
  float p = 1 - exp((-1) * exp((-1) * parLambda * (score - parMu)));
  return p;
}

/* ------------------------------------------------------------------ a13: score.c:976-1064 */

static int cmp_desc(const void *a, const void *b) { /* misc.c:84-99 */
  return (((const orc_hss *)a)->score < ((const orc_hss *)b)->score) ? 1 : -1;
}

static int extreme_value_pars(otree *t, const char *const *rows, const char *const *names, int N,
                              float kappa, const orc_model *models, const orc_model *modelsRev,
                              const orc_params *par, unsigned long seedBase, float maxNativeScore,
                              double *maxScores, float *mu_out, float *lambda_out, long *clamped) {
  int cols = (int)strlen(rows[0]), i, r, n, better = 0;
  int stopCutoff = (int)(par->cutoff * par->sampleN);
  int cap = 4096;
  orc_hss *hss = (orc_hss *)malloc(sizeof(orc_hss) * cap);
  char **srows = (char **)malloc(sizeof(char *) * N);
  double mu, lambda;
  (void)kappa;
  for (i = 0; i < par->sampleN; i++) {
    float best;
    simulate_tree(t, models[0].freqs, models[0].kappa, cols, seedBase + (unsigned long)i, clamped);
    if (sampled_rows(t, rows, names, N, cols, srows) != 0) { free(hss); free(srows); return -1; }
    n = orc_score_aln((const char *const *)srows, N, 0, 0, models, modelsRev, par, hss, cap, NULL, NULL);
    if (n > cap) n = cap;
    best = -1.0f;                       /* empty list: results[0].score is the -1 sentinel */
    for (r = 0; r < n; r++) if (r == 0 || hss[r].score > best) best = hss[r].score;
    if (best > maxNativeScore) better++;
    if (par->stopEarly && better > stopCutoff) { for (r = 0; r < N; r++) free(srows[r]); free(hss); free(srows); return -1; }
    maxScores[i] = best;
    for (r = 0; r < N; r++) free(srows[r]);
  }
  free(hss); free(srows);
  if (orc_evd_fit(maxScores, par->sampleN, &mu, &lambda) == 1) { *mu_out = mu; *lambda_out = lambda; return 1; }
  return -1;
}

int orc_extreme_value_pars(const char *newick, const char *const *rows, const char *const *names, int N,
                           float kappa, const orc_model *models, const orc_model *modelsRev,
                           const orc_params *par, unsigned long seedBase, float maxNativeScore,
                           double *maxScores, float *mu, float *lambda, long *clamped) {
  otree *t = parse_newick(newick);
  int rc;
  if (!t) return -2;
  rc = extreme_value_pars(t, rows, names, N, kappa, models, modelsRev, par, seedBase, maxNativeScore, maxScores, mu, lambda, clamped);
  free_tree(t);
  return rc;
}

/* ------------------------------------------------------------------ a15: score.c:558-797 */

#define CMPF(x, y) ((((x) > (y)) ? ((x) - (y)) : ((y) - (x))) < 0.00001)

int orc_backtrack(int opt_b, int opt_i, const float *sk, const char *const *rows, int N,
                  const orc_params *par, int *states, int *zout, int *transitions) {
  const char *seq0 = rows[0];
  int L, cols = (int)strlen(seq0), k, x, i, z, b = opt_b;
  int *map0;
  char *b0 = (char *)malloc(cols + 1), *bk = (char *)malloc(cols + 1);
  init_tables();
  L = seq_length(seq0);
  map0 = residue_columns(seq0, L);
  for (i = 0; i < N * (cols + 1); i++) states[i] = zout[i] = transitions[i] = -9;
  for (k = 1; k < N; k++) {
    float opt_score = -99.0;
    int curr = -1, prev = -1, transition = -9;
    for (x = 0; x < 3; x++)
      if (SK(sk, L, k, x, opt_b, opt_i) > opt_score) { opt_score = SK(sk, L, k, x, opt_b, opt_i); curr = x; }
    for (i = opt_i; i >= opt_b + 2; i -= 3) {
      float c0, p0, p1, p2;
      get_block(i, seq0, rows[k], map0, b0, bk, &z);
      if (i - 3 < b) p0 = p1 = p2 = 0.0;   /* the reference reads Sk[..][b][i-3], zeroed at score.c:500-504 */
      else { p0 = SK(sk, L, k, 0, b, i - 3); p1 = SK(sk, L, k, 1, b, i - 3); p2 = SK(sk, L, k, 2, b, i - 3); }
      if (z == 0) { prev = curr; transition = 0; }
      if (z == +1) {
        if (curr == 0) { c0 = SK(sk, L, k, 0, b, i);
          if (CMPF(c0, p0 + par->Delta)) { transition = 2; prev = 0; }
          if (CMPF(c0, p2 + par->Omega)) { transition = 1; prev = 2; } }
        if (curr == 1) { c0 = SK(sk, L, k, 1, b, i);
          if (CMPF(c0, p0 + par->Omega)) { transition = 1; prev = 0; }
          if (CMPF(c0, p1 + par->Delta)) { transition = 1; prev = 1; } }   /* sic: 1, score.c:667 */
        if (curr == 2) { c0 = SK(sk, L, k, 2, b, i);
          if (CMPF(c0, p1 + par->Omega)) { transition = 1; prev = 1; }
          if (CMPF(c0, p2 + par->Delta)) { transition = 2; prev = 2; } }
      }
      if (z == -1) {
        if (curr == 0) { c0 = SK(sk, L, k, 0, b, i);
          if (CMPF(c0, p0 + par->Delta)) { transition = 2; prev = 0; }
          if (CMPF(c0, p1 + par->Omega)) { transition = 1; prev = 1; } }
        if (curr == 1) { c0 = SK(sk, L, k, 1, b, i);
          if (CMPF(c0, p1 + par->Delta)) { transition = 2; prev = 1; }
          if (CMPF(c0, p2 + par->Omega)) { transition = 1; prev = 2; } }
        if (curr == 2) { c0 = SK(sk, L, k, 2, b, i);
          if (CMPF(c0, p2 + par->Delta)) { transition = 2; prev = 2; }
          if (CMPF(c0, p0 + par->Omega)) { transition = 1; prev = 0; } }
      }
      states[k * (cols + 1) + i] = curr;
      transitions[k * (cols + 1) + i] = transition;
      zout[k * (cols + 1) + i] = z;
      curr = prev;
    }
  }
  free(map0); free(b0); free(bk);
  return 0;
}

/* ------------------------------------------------------------------ RNAcode.c:153-188 */

int orc_run_block(const char *const *rows, const char *const *names, int N, int ref_start, int ref_length,
                  const char *newick, float kappa, const orc_params *par, unsigned long seedBase,
                  orc_model *models, orc_model *modelsRev, orc_hss *hss, int cap,
                  double *maxScores, float *mu, float *lambda, int *evd_rc, long *clamped) {
  otree *t = parse_newick(newick);
  char **rev;
  int i, n, rc;
  float maxScore;
  if (!t) return -1;
  init_tables();
  rev = (char **)malloc(sizeof(char *) * N);
  orc_rev_aln(rows, N, rev);
  if (get_models_tree(t, rows, names, N, kappa, par->blosum, models) != 0 ||
      get_models_tree(t, (const char *const *)rev, names, N, kappa, par->blosum, modelsRev) != 0) {
    for (i = 0; i < N; i++) free(rev[i]);
    free(rev); free_tree(t); return -1;
  }
  n = orc_score_aln(rows, N, ref_start, ref_length, models, modelsRev, par, hss, cap, NULL, NULL);
  if (n > cap) n = cap;
  qsort(hss, n, sizeof(orc_hss), cmp_desc);
  maxScore = n > 0 ? hss[0].score : -1.0f;
  *mu = 0; *lambda = 0;
  rc = extreme_value_pars(t, rows, names, N, kappa, models, modelsRev, par, seedBase, maxScore, maxScores, mu, lambda, clamped);
  for (i = 0; i < n; i++) hss[i].pvalue = (rc == 1) ? orc_pvalue(hss[i].score, *mu, *lambda) : 99.0f;
  *evd_rc = rc;
  for (i = 0; i < N; i++) free(rev[i]);
  free(rev);
  free_tree(t);
  return n;
}
