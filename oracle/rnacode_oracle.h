/* rnacode_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's scoring + null-sampling path, used as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It is never
 * linked into, loaded by or called from the product (rnacode_amd/, include/).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
 * golden vectors produced by the unmodified reference objects (oracle/ref_harness.c,
 * oracle/make_goldens.py -> tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree).  Structure deliberately mirrors the reference (O(N L^2) Sk matrices,
 * string blocks) so that it is an independent implementation from the O(L)-memory
 * device design it checks.
 */
#ifndef RNACODE_ORACLE_H
#define RNACODE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ROWS 500 /* rnaz_utils.h:8 MAX_NUM_NAMES */

typedef struct {
  float Delta, Omega, omega;      /* RNAcode.c:68-70 defaults -10, -4, -2 */
  float stopPenalty_0;            /* RNAcode.c:72  -9999 */
  float stopPenalty_k;            /* RNAcode.c:71  -8 */
  int blosum;                     /* 62 or 90 */
  int sampleN;                    /* RNAcode.c:81 */
  float cutoff;                   /* RNAcode.c:84 */
  int stopEarly;                  /* RNAcode.c:78 */
} orc_params;

typedef struct {
  float scores[4];                /* score.h:36 expected score by Hamming class */
  float probs[4];
  float kappa, dist;
  float freqs[4];
} orc_model;

typedef struct {                  /* score.h:48-63 */
  int start, end, startGenomic, endGenomic, startSite, endSite;
  int strand;                     /* '+' or '-' */
  int frame;
  float score;
  float pvalue;
} orc_hss;

void orc_default_params(orc_params *p);

/* a4: score.c:204-244 */
float orc_prob_hky(int i, int j, float d, const float freqs[4], float kappa);
/* a5: score.c:107-193 */
void orc_calculate_bg(float dist, const float freqs[4], float kappa, int blosum,
                      float scores[4], float probs[4]);
/* a2: score.c:255-280 */
void orc_count_freqs(const char *const *rows, int N, float freqs[4]);
/* a3: treeSimulate.c:118-251; dist[k] = patristic distance row0 <-> row k; returns 0 ok */
int orc_tip_distances(const char *newick, const char *const *names, int N, float *dist);
/* a5: score.c:291-344; rows must already be upper-case; models has N entries */
int orc_get_models(const char *newick, const char *const *rows, const char *const *names, int N,
                   float kappa, int blosum, orc_model *models);
/* rnaz_utils.c:316-348; out rows are malloc'ed strings */
void orc_rev_aln(const char *const *rows, int N, char **out);

/* a6-a11: score.c:1067-1147 (scoreAln) for one alignment with given models.
 * ref_start/ref_length: aln[0]->start/length (0,0 for ClustalW input).
 * hss: caller array of capacity cap; returns the number of HSS ('+' hits then '-' hits), or -1.
 * If sk_fwd/sk_rev are non-NULL they receive malloc'ed copies of Sk (flat [k][x][b][i],
 * dims N x 3 x (L+1) x (L+1)) as kept for backtracking (score.c:1084-1101). */
int orc_score_aln(const char *const *rows, int N, int ref_start, int ref_length,
                  const orc_model *models, const orc_model *modelsRev, const orc_params *par,
                  orc_hss *hss, int cap, float **sk_fwd, float **sk_rev);

/* S matrix of one strand, flat (L+1)x(L+1) (score.c:811-848), for cell-level parity tests. */
float *orc_score_matrix(const char *const *rows, int N, const orc_model *models, const orc_params *par, int *L_out);

/* a12: treeSimulate.c:52-97 + seqgen; one null alignment for seed; out_rows: N malloc'ed
 * strings of length cols in NATIVE row order with the native gap mask reintroduced
 * (score.c:1006-1012).  *clamped counts draws that fell past the cumulative vector
 * (the reference reads out of bounds there). */
int orc_simulate_null(const char *newick, const char *const *rows, const char *const *names, int N,
                      const float freqs[4], float kappa, unsigned long seed, char **out_rows, long *clamped);

/* a13: score.c:976-1064; maxScores has sampleN doubles; seeds are seedBase + i.
 * Returns 1 / -1 like the reference; fills mu/lambda on success. */
int orc_extreme_value_pars(const char *newick, const char *const *rows, const char *const *names, int N,
                           float kappa, const orc_model *models, const orc_model *modelsRev,
                           const orc_params *par, unsigned long seedBase, float maxNativeScore,
                           double *maxScores, float *mu, float *lambda, long *clamped);

/* a14: extreme_fit.c:157-251 */
int orc_evd_fit(const double *x, int n, double *mu, double *lambda);
/* RNAcode.c:182 */
float orc_pvalue(float score, float mu, float lambda);

/* a15: score.c:558-797; sk = flat Sk of the strand, rows of that strand.
 * states/z/transitions: arrays [N][cols+1] (row 0 unused), filled at i = opt_i, opt_i-3, ... >= opt_b+2,
 * other entries set to -9. */
int orc_backtrack(int opt_b, int opt_i, const float *sk, const char *const *rows, int N,
                  const orc_params *par, int *states, int *z, int *transitions);

/* MT19937 as used by Seq-Gen (twister.c:73-89,118-166): first n outputs for a seed. */
/* test diagnostic: count of replacements decided by the 1e-4 tie rule of getHSS since the last reset */
unsigned long orc_tie_replacements(int reset);

void orc_mt_stream(unsigned long seed, unsigned int *out, int n);

/* whole block, the way RNAcode.c:153-188 drives it. Returns number of HSS (sorted by score
 * descending as main does), evd_rc in *evd_rc. */
int orc_run_block(const char *const *rows, const char *const *names, int N, int ref_start, int ref_length,
                  const char *newick, float kappa, const orc_params *par, unsigned long seedBase,
                  orc_model *models, orc_model *modelsRev, orc_hss *hss, int cap,
                  double *maxScores, float *mu, float *lambda, int *evd_rc, long *clamped);

#ifdef __cplusplus
}
#endif
#endif
