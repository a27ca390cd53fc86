/* rnacode_hip.h -- C-ABI of the MI355X-native RNAcode scoring path (librnacode_hip.so).
 *
 * The reference has no plugin/FFI layer: the drop-in boundary is the link-level symbol set of
 * its scoring object (src/score.h:83-118 as used by src/RNAcode.c:164-216 and
 * src/postscript.c:303-305).  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no C++ or torch types.  All functions return
 * RC_OK (0) or a negative RC_ERR_* code; rc_last_error() gives the message.  The library
 * never falls back to a CPU implementation of the hot path: without a usable HIP device
 * every compute entry point fails with RC_ERR_DEVICE.
 */
#ifndef RNACODE_HIP_H
#define RNACODE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_OK 0
#define RC_ERR_ARG (-1)         /* malformed block / tree / parameter */
#define RC_ERR_DEVICE (-2)      /* no HIP device, or a HIP call failed */
#define RC_ERR_UNSUPPORTED (-3) /* shape outside the built kernels (N > RC_MAX_ROWS, more than RC_MAX_COLS columns) */
#define RC_ERR_SKIP (-4)        /* block the reference driver skips (RNAcode.c:142-150) */

#define RC_MAX_ROWS 500         /* rows per block, the reference's own limit (MAX_NUM_NAMES, rnaz_utils.h:7).  N <= 32 runs the
                                   register-resident kernels, 33..64 a slower instantiation of the same code, 65..500 generic kernels
                                   (sequence-by-sequence over register-tiled rows, codes in a global scratch); the built-in tree estimator
                                   runs on the GPU for up to 64 rows and on host threads for wider blocks */
#define RC_MAX_COLS 65535       /* alignment columns per block (16-bit column indices); longer blocks get the status RC_ERR_UNSUPPORTED --
                                   the reference's breakMAF.pl (python -m rnacode_amd.breakmaf) splits them first, as its README asks */
/* Two further bounds on a block's size.  (1) Device memory: the MT19937 streams of a block take 4 bytes x (2 rows - 1) x columns x
 * samples (padded to 64); a block for which that exceeds a quarter of the device memory (MI355X: rows x columns x samples > ~9e9,
 * e.g. 500 rows x 18 000 columns at 1000 samples) gets RC_ERR_UNSUPPORTED with a reason, the other blocks of the batch are scored.
 * (2) Bit parity of the base frequencies: the reference counts each of the four residues in a float counter, which stops at 2^24,
 * and divides by an integer total converted to float once (score.c:255-280); the library reproduces both (counts clamped to 2^24,
 * the total unclamped), so frequencies match the reference for any size -- a block only gets there with more than 16.7 million
 * residues of one kind. */

/* Scoring parameters: the members of `parameters pars` (src/RNAcode.h:29-54) that the
 * scoring path reads (score.c:415,419,508-533,842,992,1040), plus the seed base that
 * replaces Seq-Gen's time-based CreateSeed (seqgen/twister.c:195-224):
 * sample s of every block is simulated with MT19937 seeded by seed_base + s. */
typedef struct rc_params {
  float Delta;          /* -10   */
  float Omega;          /* -4    */
  float omega;          /* -2    */
  float stopPenalty_0;  /* -9999 */
  float stopPenalty_k;  /* -8    */
  int32_t blosum;       /* 62 | 90 */
  int32_t sampleN;      /* --num-samples */
  float cutoff;         /* --cutoff, only used by stop-early (score.c:992) */
  int32_t stopEarly;    /* --stop-early */
  uint32_t seed_base;
} rc_params;

void rc_default_params(rc_params *p); /* RNAcode.c:68-85 */

/* One alignment block: `struct aln *[]` (src/rnaz_utils.h:12-20) flattened, plus what
 * treeML/string2tree hand to the scorer (RNAcode.c:153-158): Newick text and kappa. */
typedef struct rc_block {
  int32_t n_rows;            /* N, row 0 = reference sequence */
  int32_t n_cols;            /* alignment columns */
  const char *const *rows;   /* n_rows strings of n_cols chars (A C G T U N - ...); upper-cased internally */
  const char *const *names;  /* n_rows row names; must match the tree's tip labels */
  int32_t ref_start;         /* aln[0]->start  (0 for ClustalW input) */
  int32_t ref_length;        /* aln[0]->length (0 for ClustalW input) */
  const char *newick;        /* tree with branch lengths (phyml Write_Tree format) */
  float kappa;               /* ts/tv ratio from treeML */
} rc_block;

/* segmentStats (src/score.h:48-63) without the heap-allocated name. */
typedef struct rc_hss {
  int32_t start, end;                /* nt positions in the reference row, 1-based */
  int32_t startGenomic, endGenomic;
  int32_t startSite, endSite;        /* codon indices, 0-based */
  int32_t strand;                    /* '+' or '-' */
  int32_t frame;                     /* 0..2 */
  float score;
  float pvalue;                      /* filled by rc_batch_run (RNAcode.c:180-188); 99 on fit failure */
} rc_hss;

/* bgModel (src/score.h:34-44) as computed by getModels (score.c:291-344). */
typedef struct rc_model {
  float scores[4];
  float probs[4];
  float kappa, dist;
  float freqs[4];
} rc_model;

typedef struct rc_ctx rc_ctx;     /* one per process and GPU: device, stream, caches */
typedef struct rc_batch rc_batch; /* blocks resident in HBM + their results */

const char *rc_last_error(void);
int rc_device_count(void);

/* One context per process and GPU (one process per GPU; multi-GPU sharding is done by the
 * caller).  A context owns its HIP streams, the MT19937 stream cache and the host
 * threads that prepare blocks: calls on one context -- directly or through one of its batches or
 * streams -- must come from one thread at a time; different contexts are independent.
 * rc_last_error() is per thread. */
int rc_ctx_create(int device, rc_ctx **out);
void rc_ctx_destroy(rc_ctx *ctx);   /* destroy the context's batches and streams first */
/* A context keeps the device and pinned buffers of its destroyed batches and streams for the next ones (hipFree / hipHostFree wait
 * for the whole device): up to 1/16 of the device's memory and 1 GB of pinned memory, until rc_ctx_destroy.  rc_ctx_trim gives them
 * back now (between phases of a job that shares the GPU).  An allocation that fails first gives back what the pools of every
 * context of this process on that device keep, then tries again. */
void rc_ctx_trim(rc_ctx *ctx);

/* Host threads used to prepare blocks (default: the CPUs this process may use -- affinity mask and
 * cgroup quota, rc_host_cpus() -- capped at 32; RC_HOST_THREADS overrides).  With several ranks per
 * node give each rank its share. */
int rc_host_cpus(void);
int rc_ctx_set_host_threads(rc_ctx *ctx, int32_t threads);
int rc_ctx_host_threads(const rc_ctx *ctx);

/* Which exp() the EVD fit (EVDMaxLikelyFit, extreme_fit.c:157-251) computes with.  The reference's Newton iteration stops on |f| < 1e-5, and on
 * near-degenerate maxima (one or two samples, all maxima equal) the last bit of exp decides which iterate passes or whether the fit fails; the
 * reference follows ITS host's C library.  The library reproduces glibc >= 2.28's exp bit for bit in both x86-64 variants and picks the one the
 * host's exp() agrees with when a context is created: 1 = generic, 2 = compiled with fused multiply-adds; 0 = the host's exp is neither (musl,
 * aarch64, another glibc): the device library's exp (<= 1 ulp), and such fits may then differ from a reference run on this host.
 * RC_FIT_EXP=generic|fused|device overrides the probe (to reproduce a listing made on another machine). */
int rc_ctx_fit_exp_mode(const rc_ctx *ctx);

/* getModels x2 (RNAcode.c:164-165) + everything block-constant the kernels need.  The host (threads)
 * parses the tree, takes base frequencies, tip distances, the HKY85 matrices, the gap pattern and the
 * integer thresholds of the branch matrices, and writes them straight into one pinned blob; ONE
 * asynchronous copy moves it; calculateBG's accumulations (score.c:107-193) and the sigma tables are
 * then made on the device.  Returns when the copy has been queued.
 * Per-block outcome is in rc_batch_status: RC_OK; RC_ERR_SKIP for blocks the reference driver skips
 * (N <= 2, L < 3, RNAcode.c:142-150); RC_ERR_ARG / RC_ERR_UNSUPPORTED for blocks that cannot be scored
 * (no or malformed tree, names that do not match, too many rows) -- those are left out like the
 * reference's "Skipping alignment" (RNAcode.c:153-156), rc_batch_block_error() has the reason, and
 * the other blocks of the batch are scored normally. */
int rc_batch_create(rc_ctx *ctx, const rc_block *blocks, int32_t n_blocks, const rc_params *par, rc_batch **out);
void rc_batch_destroy(rc_batch *b);
const char *rc_batch_block_error(const rc_batch *b, int32_t blk);   /* "" if the block was accepted */
int rc_batch_size(const rc_batch *b);

/* Optional: make the kernels write the per-sample maxima ([n_blocks][sampleN] floats) straight
 * into a caller-owned DEVICE buffer (e.g. a torch tensor handed to an RCCL all-gather).  Must be
 * called before rc_batch_run; the buffer must outlive the batch.  The library writes it on its own
 * (non-blocking) HIP streams: whoever reads it on another stream -- a collective -- must have finished
 * before the next rc_batch_run of this batch, which starts by overwriting it. */
int rc_batch_bind_maxima(rc_batch *b, void *device_ptr);

/* The hot path for the whole batch, all on the device:
 *   scoreAln(native, backtrack=1)      score.c:1067-1147  -> HSS lists
 *   getExtremeValuePars                score.c:976-1064   -> per-sample maxima, mu, lambda
 *   p-values                           RNAcode.c:180-188
 * rc_batch_run is synchronous: it returns when the results are on the host.  rc_batch_run_async queues
 * every launch and the copies of the small results and returns; rc_batch_wait blocks until they have
 * arrived.  Several batches of one context may be between run_async and wait at the same time
 * (consecutive ones go to alternating HIP streams, so the tail of one overlaps the head of the next). */
int rc_batch_run(rc_batch *b);
int rc_batch_run_async(rc_batch *b);
int rc_batch_wait(rc_batch *b);

/* A stream of batches (the reference's unit of work is a stream of blocks, RNAcode.c:115-221): up to
 * `depth` batches in flight.  rc_stream_submit prepares the blocks on the host threads, queues the copy
 * and every launch, and returns -- while the GPU scores that batch the caller submits the next one.
 * rc_stream_next waits for the OLDEST submitted batch and hands it over; read its results with the
 * rc_batch_* accessors, then give it back with rc_stream_recycle (its pinned and device buffers are
 * reused by a later submit: the steady state allocates nothing) or keep it and rc_batch_destroy it.
 * The rc_block arrays only need to live during rc_stream_submit.  Submitting to a full stream is an error. */
typedef struct rc_stream rc_stream;
int rc_stream_create(rc_ctx *ctx, const rc_params *par, int32_t depth, rc_stream **out);
int rc_stream_submit(rc_stream *s, const rc_block *blocks, int32_t n_blocks);
/* the same, with this batch's per-sample maxima ([n_blocks][sampleN] floats) written into a caller-owned device
 * buffer (see rc_batch_bind_maxima): e.g. consecutive slices of the tensor a rank hands to the RCCL all-gather */
int rc_stream_submit_bound(rc_stream *s, const rc_block *blocks, int32_t n_blocks, void *maxima_device_ptr);
int rc_stream_next(rc_stream *s, rc_batch **out);
void rc_stream_recycle(rc_stream *s, rc_batch *b);
int rc_stream_pending(const rc_stream *s);
void rc_stream_destroy(rc_stream *s);   /* also destroys the batches it still holds */
/* How to cut n_blocks blocks into sub-batches for a stream: writes the sizes (they sum to n_blocks), returns how many.  A small
 * first sub-batch starts the GPU early; the next ones double, so that each is queued before its predecessor drains; all but the
 * first fill the chip's wavefront slots a whole number of times (slots / sample groups blocks do so once), the odd remainder
 * goes first; no sub-batch exceeds eight such rounds, except that a tail shorter than half a sub-batch goes with the last one.
 * row_classes: distinct row counts among the blocks (each is a launch of
 * its own, side by side with the others). */
int rc_stream_plan(const rc_ctx *ctx, const rc_params *par, int32_t n_blocks, int32_t row_classes, int32_t *sizes, int32_t cap);

/* By default the MT19937 output streams (one per sample index, shared by all blocks because
 * seed = seed_base + s) are cached in the context between runs with the same seed_base and
 * sampleN; rc_set_stream_cache(0) regenerates them in every rc_batch_run. */
void rc_set_stream_cache(int enabled);

/* Number of (block, sample) units and alignment-columns x samples of the last run. */
int rc_batch_work(const rc_batch *b, int64_t *sample_alignments, int64_t *column_samples);

/* HIP-event timings (ms) of the last rc_batch_run on the context's stream:
 * t[0] total, t[1] MT19937 stream kernel, t[2] null-sampling kernels (simulate+score),
 * t[3] native scoring kernels, t[4] EVD fit kernel.  n_launch[i] = launches behind t[i]. */
int rc_batch_timing(const rc_batch *b, float t[5], int32_t n_launch[5]);
/* The null-sampling kernel instantiation that did most of the last run's work, spelled as a profiler prints it
 * ("rc::k_null<5, true, false>"): lets a benchmark check that counter data it quotes belongs to the kernel it timed. */
const char *rc_batch_null_kernel(const rc_batch *b);
/* Preparation of the batch: wall time of the host part (ms), duration of the device kernels that make the
 * expected-score tables (ms, known after a run), bytes copied to the device. */
int rc_batch_prep_timing(const rc_batch *b, double *host_ms, float *table_kernels_ms, int64_t *uploaded_bytes);

/* Per-block results (host copies).  status: see rc_batch_create. */
int rc_batch_status(const rc_batch *b, int32_t blk);
int rc_batch_models(const rc_batch *b, int32_t blk, rc_model *fwd, rc_model *rev); /* n_rows entries each */
/* maxScores[] of score.c:1044: sampleN floats (-1 = sample without HSS) */
int rc_batch_maxima(const rc_batch *b, int32_t blk, float *out);
/* all blocks at once, [n_blocks][sampleN]; skipped blocks are filled with -1 */
int rc_batch_maxima_all(const rc_batch *b, float *out);
/* getExtremeValuePars' outputs: rc 1 / -1 like the reference, mu and lambda narrowed to float */
int rc_batch_fit(const rc_batch *b, int32_t blk, int32_t *evd_rc, float *mu, float *lambda);
/* The same for every block at once: out[4*blk + {0,1,2,3}] = {evd_rc or the block's status if it was
 * not scored, mu, lambda, number of samples above the best native score}.  This is what leaves a GPU
 * when blocks are sharded across ranks (the p-values follow from mu and lambda). */
int rc_batch_fit_all(const rc_batch *b, float *out);
/* HSS list sorted by score descending as main() does (RNAcode.c:173-176); returns the count
 * (may exceed cap; only cap entries are written) */
int rc_batch_hss(const rc_batch *b, int32_t blk, rc_hss *out, int32_t cap);
/* The HSS lists of every block in one call: block i's list (sorted as above) is out[offsets[i] .. offsets[i+1]);
 * offsets has n_blocks + 1 entries, offsets[n_blocks] is the total, which may exceed cap (only cap records are
 * written -- call with cap = 0 first to size the buffer).  Blocks that were not scored have empty lists. */
int rc_batch_hss_all(const rc_batch *b, rc_hss *out, int64_t cap, int64_t *offsets);
/* draws that fell past the cumulative probability vector (reference reads out of bounds
 * there, seqgen/evolve.c:173; we clamp to state 3 and count) */
int rc_batch_clamped(const rc_batch *b, int64_t *count);

/* Debug/parity access: multiple-score matrix S[a][j] of the native block for one strand and
 * frame (score.c:811-848 restricted to that frame): sites x sites floats, row a = start
 * codon, column j = end codon; entries with j < a are 0.  Returns sites.  (The scoring pass does
 * not keep these matrices; the call recomputes the block's with the scoring kernel.) */
int rc_batch_native_S(const rc_batch *b, int32_t blk, int32_t strand /*0:'+',1:'-'*/, int32_t frame, float *out, int32_t cap);

/* backtrack (score.c:558-797) for the native block: states/z/transitions for rows 1..N-1 at
 * i = opt_i, opt_i-3, ... >= opt_b+2.  Arrays are [n_rows][n_cols+1]; untouched entries -9. */
int rc_batch_backtrack(const rc_batch *b, int32_t blk, int32_t strand, int32_t opt_b, int32_t opt_i,
                       int32_t *states, int32_t *z, int32_t *transitions);

/* The substitution matrix and genetic code the scorer uses, for callers that render results
 * (getScoringMatrix() score.c:50-76 and transcode[4][4][4] code.c:28-39, which src/postscript.c:362,412
 * read): pep_out[16a+4b+c] = amino-acid index 0..19 of codon (a,b,c) in A,C,G,T order or -1 for a stop;
 * matrix_out[20p+q] = BLOSUM entry of amino acids p,q.  blosum is 62 or 90. */
int rc_code_tables(int32_t blosum, int32_t pep_out[64], int32_t matrix_out[400]);

/* Tree + kappa for one block, the inputs treeML() hands to the scorer (src/treeML.c:35-152 via the
 * bundled PhyML): BIONJ topology from pairwise ML distances, HKY85 maximum-likelihood branch lengths
 * and kappa.  Host code, re-entrant (call it from several threads); blk->newick/kappa are ignored.
 * Agreement with PhyML is to optimiser tolerance, not bitwise.
 * Writes a NUL-terminated Newick string ("%f" branch lengths) into newick_out[cap]. */
int rc_fit_tree(const rc_block *blk, char *newick_out, int32_t cap, float *kappa_out);
/* The same for n_blocks blocks on `threads` host threads (0 = hardware concurrency).  newick_out is one
 * buffer of n_blocks * cap bytes (block i at i*cap); blocks the driver skips (N <= 2, L < 3) or that
 * fail get an empty string and kappa 0.  Returns the number of fitted blocks. */
int rc_fit_trees(const rc_block *blocks, int32_t n_blocks, char *newick_out, int32_t cap, float *kappa_out, int32_t threads);
/* The same estimator on the GPU of `ctx`, one wavefront per block, all blocks concurrently (the host
 * only compresses site patterns and writes the Newick text).  Same algorithm as rc_fit_trees; the
 * per-site sums are taken in a different order, so branch lengths agree to ~1e-6, not bitwise.
 * lnl_out (may be NULL) receives the log-likelihoods.  Blocks with more than 64 rows are fitted by the host estimator
 * inside the same call (the kernel keeps BIONJ's tables per lane).  Returns the number of fitted blocks or a
 * negative error code. */
int rc_fit_trees_device(rc_ctx *ctx, const rc_block *blocks, int32_t n_blocks, char *newick_out, int32_t cap, float *kappa_out,
                        double *lnl_out);
/* HKY85 log-likelihood of blk->newick / blk->kappa on blk's rows under the same model and data
 * handling (diagnostic: compares a tree from elsewhere with rc_fit_tree's on equal terms). */
int rc_tree_lnl(const rc_block *blk, double *lnl_out);

/* EVDMaxLikelyFit (src/extreme_fit.c:157-251) on the device for n doubles; returns 1 / 0. */
int rc_evd_fit(rc_ctx *ctx, const double *x, int32_t n, double *mu, double *lambda);
/* p = 1 - exp(-exp(-lambda (score - mu))) with RNAcode.c:182's float/double promotions (host arithmetic): for callers
 * that fit gathered maxima themselves (sample-range sharding) */
float rc_pvalue(float score, float mu, float lambda);

/* First n MT19937 outputs for a seed, generated by the device stream kernel (parity hook
 * for seqgen/twister.c:73-152). */
int rc_mt_stream(rc_ctx *ctx, uint32_t seed, uint32_t *out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RNACODE_HIP_H */
