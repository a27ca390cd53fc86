// rc_launch.h -- kernel argument blocks and launcher prototypes (the .hip units <-> rc_schedule.cpp, rc_batch.cpp, rc_trees_api.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "rc_device.h"

namespace rc {

// -DRC_PROFILING builds (tools/mk_ab.sh) can switch phases of k_null off through RC_DEBUG_SKIP to time them
// separately; the product library has no such switch.
#ifdef RC_PROFILING
constexpr bool kProfiling = true;
#else
constexpr bool kProfiling = false;
#endif

struct NullArgs {
  const uint8_t *blob;
  const DevBlock *dblocks;
  const int *classBlocks;     // batch indices of the blocks of this launch
  int nClassBlocks;
  const int *nBlocksPtr;      // if set: the number of blocks is read from device memory (list built by k_prep_lut)
  const uint32_t *flags;      // per batch index: kFlagExact, kFlagStopped
  uint32_t skipMask;          // blocks with (flags & skipMask) != 0 are left out
  uint32_t onlyMask;          // k_tiled_dp: if set, only blocks with (flags & onlyMask) != 0 are taken (the launch for blocks with NaN score tables)
  int tiledKT;                // k_generic_sim<true> / k_tiled_dp: the tile size of the launch's class
  int gLo, gHi;               // sample groups [gLo, gHi) of 64 samples each
  int sampleN;
  int Spad;                   // ceil(sampleN / 64) * 64: row pitch of U
  const uint32_t *U;          // MT19937 streams [draw][Spad]
  const uint8_t *pair;        // [64][64] codon pair -> sigma code
  uint32_t *scratch;          // per-workgroup sigma-code staging
  size_t scratchStride;       // uint32 per workgroup
  float *maxima;              // [n_blocks][sampleN]
  unsigned long long *clampCount;
  float tieThr;               // float form of the 0.0001 tie tolerance (score.c:954)
  int comboSplit;             // 1: items are (block, group, strand x frame); maxima combined by atomic max
  unsigned int *workCounter;  // [8], zeroed before the launch: next unclaimed work item of each XCD queue
  unsigned int *steal;        // tail sharing (see k_null): 16 + 4 x grid words, zeroed before the launch; null = off
  uint8_t *codesAll;          // k_generic_sim / k_generic_dp: the codes of every item of the launch pair, item u at u * codesStride
  size_t codesStride;
  int cacheSites;             // codes from L2: the words of the last cacheSites sites of a strand x frame are kept in LDS (0: none)
  int plainLds;               // host only: dynamic LDS bytes of a launch that reads its codes from L2 (phase A's tables, then the suffix cache); 0: the default
  int hiOcc;                  // host only: 1 = launch k_null_occ (one more wavefront per SIMD), where this N-1 has one
  int dual;                   // 1: the instantiation that walks two rows of S per pass (N-1 <= kDualRowsMaxNK, codes staged in LDS)
  int stealWait;              // 1: a workgroup without work waits for owners that are still simulating a late item; 0: it leaves at once (a stream: the next sub-batch's workgroups want its place)
  int simParts, simGrid;      // rows split over workgroups: an item's simulation is cut into simParts site ranges, simGrid workgroups take them (k_null<.., 2>)
  unsigned int *simCounter;   // ... from queues of their own [8]
  int rowParts;               // > 1: a strand x frame part's rows are split into this many ranges, each a work item (k_null<.., ROWS>); needs comboSplit
  float *sbuf;                // ... whose S values go here, [item][6][sbufStride]: getHSS's entry order, [entry][64 lanes]; k_null_rowscan folds them
  size_t sbufStride;          // floats per (item, strand x frame): 64 x (largest sites (sites + 1) / 2 of the launch)
  int debugSkip;              // RC_PROFILING builds only: bit0 skip the DP/scan phase, bit1 skip simulation
  unsigned long long *cellStats;   // RC_PROFILING builds only: [0] wavefront-cells, [1] of those with no lane's sum > 0, [2] lanes with sum > 0, [3..5] cells before the row's first event / at events / after its last
};

struct NativeArgs {
  uint8_t *blob;
  const DevBlock *dblocks;
  const int *blocks;          // batch indices handled by this launch
  const uint8_t *pair;
  float *tile;                // k_native_dp: per-workgroup buffer for 64 rows of one matrix, [gridDim][tileStride]
  size_t tileStride;          // floats: 64 * (largest L/3 of the launch)
  int nItems;                 // k_native_dp: (block, strand x frame) items of the launch = 6 * blocks
  float *fullS;               // debugging accessor only (one block per launch): the six whole matrices, no records written
  float *sAll;                // != nullptr: k_native_dp writes every item's matrix, [item][sAllSites][sAllSites] (row a, end codon j), and leaves getHSS to
  int sAllSites;              //   k_native_scan (one LANE per item: the scan is serial per matrix, and a wavefront per matrix spent two thirds of the kernel on it)
  DevHss *hss;                // [n_blocks][6][kHssCap]
  int *hssCount;              // [n_blocks][6]
  int hssCap;                 // records per (block, strand, frame)
  float tieThr;
  const uint32_t *flags;      // per batch index: kFlagNan picks the reference's NaN-order-dependent maxima (ref_max)
};

// block-constant tables derived on the device (k_prep_models, k_prep_lut)
struct PrepArgs {
  uint8_t *blob;
  const DevBlock *dblocks;
  const int *blocks;          // batch indices with status RC_OK
  int nBlocks;
  const int *modelPrefix;     // [nBlocks + 1] running sum of 2 N over `blocks`: model t belongs to the block whose range holds t
  int nModels;
  const int *pep;             // [64] amino-acid index per codon, -1 = stop
  const int *blosum;          // [20][20]
  const CodeInfo *codeInfo;   // [64]
  uint32_t *flags;            // per batch index
  int *exactList;             // per class: batch indices flagged kFlagExact, at classOff[NK]
  int *exactCount;            // [kClassSlots] by class
  int classOff[kClassSlots];  // by class (N-1 for the instantiated kernels, kGenericClass for wider blocks)
  ClassRule rule;   // which blocks take which kernels (rc_device.h, block_class)
};

struct FitOut {
  double mu, lambda;
  int rc;                     // 1 ok, -1 failed or stopped early (score.c:1041,1061)
  int better;                 // samples scoring above the best native HSS
};

struct FitArgs {
  const DevBlock *dblocks;
  const int *blocks;
  const float *maxima;
  const DevHss *hss;
  const int *hssCount;
  int hssCap;
  FitOut *out;                // [n_blocks]
  uint32_t *flags;            // k_stop_mark sets kFlagStopped
  int sampleN;
  int firstN;                 // k_stop_mark: number of samples simulated so far
  int stopEarly;
  int stopCutoff;
  int expMode;                // which exp the fit uses: 0 the device library's, 1 / 2 the host C library's algorithm, generic / fused (rc_refexp.h)
};

// tree estimator (rc_tree_kernel.hip): one job = one alignment block
struct TreeJob {
  int N, P;
  uint64_t off_mask;          // bytes into the input blob: [N][P] state masks
  uint64_t off_w;             // bytes into the input blob (8-aligned): [P] pattern weights
  uint64_t off_work;          // doubles into the scratch: tree_work_doubles(N, P) -- only jobs of the big-block launches have one
  uint64_t off_out;           // doubles into the result buffer: the job's TreeResult record (tree_result_doubles(N) doubles)
};
#ifdef RC_TREE_PROFILE
constexpr int kTreeProfDoubles = 12;   // profiling builds: cycles of phases 0..8 and counters 9..11 behind every record
#else
constexpr int kTreeProfDoubles = 0;
#endif
// What travels back to the host per job, nn = 2N - 2 nodes: doubles {kappa, lnl, len[nn]} then ints {root, nchild[nn], child[3 nn]}
// (260 bytes for six rows)
__host__ __device__ inline size_t tree_result_doubles(int N) { const size_t nn = 2 * static_cast<size_t>(N) - 2; return 2 + nn + (1 + 4 * nn + 1) / 2 + kTreeProfDoubles; }
// A fit's working set, as offsets (in doubles) into the workgroup's LDS.  Always there: branch lengths, the five transfer constants
// and three exponentials per branch, BIONJ's D and V and its temporaries, the tree as integer arrays.  With cols: the pattern weights,
// the conditional-likelihood columns -- dn and up of the N - 3 internal nodes below the root, [4][P] each, and the four Newton
// constants per pattern -- and the masks; a block whose columns do not fit keeps those three in global memory (big).
struct TreeLdsLayout { uint32_t len, coef, e3, D, V, tmpD, opt, ints, w, cols, mask, total; };
constexpr uint32_t kTreeOptDoubles = 32 + kTreeProfDoubles;   // room for treefit::OptState (rc_tree_kernel.hip asserts it), profiling builds: and the phase counters
__host__ __device__ inline size_t tree_col_doubles(int N, int P) { return (static_cast<size_t>(2) * (N > 3 ? N - 3 : 0) * 4 + 4) * static_cast<size_t>(P); }
__host__ __device__ inline TreeLdsLayout tree_lds_layout(int N, int P, bool big) {
  const uint32_t n = static_cast<uint32_t>(N), nn = 2 * n - 2, p = static_cast<uint32_t>(P);
  TreeLdsLayout l{};
  uint32_t at = 0;
  l.len = at; at += nn;
  l.coef = at; at += 5 * nn;
  l.e3 = at; at += 3 * nn;
  l.D = at; at += n * n;
  l.V = at; at += n * n;
  l.tmpD = at; at += n;
  l.opt = at; at += kTreeOptDoubles;
  l.ints = at; at += (7 * nn + 2 * n + 1) / 2;   // parent, nchild, child[3], preorder, then bionj's 2 N + nn
  l.w = l.cols = l.mask = at;
  if (!big) {
    l.w = at; at += p;
    l.cols = at; at += static_cast<uint32_t>(tree_col_doubles(N, P));
    l.mask = at; at += (n * p + 7) / 8;
  }
  l.total = at;
  return l;
}
inline size_t tree_fit_lds_bytes(int N, int P, bool big) {   // (64-bit: a block that cannot fit must not wrap round to a small number)
  const size_t n = static_cast<size_t>(N), nn = 2 * n - 2, p = static_cast<size_t>(P);
  size_t at = 9 * nn + 2 * n * n + n + kTreeOptDoubles + (7 * nn + 2 * n + 1) / 2;
  if (!big) at += p + tree_col_doubles(N, P) + (n * p + 7) / 8;
  return at * sizeof(double);
}
// a big-block job's work area in global memory: the columns
inline size_t tree_work_doubles(int N, int P) { return tree_col_doubles(N, P); }
int tree_fit_max_lds();   // bytes of LDS one workgroup may ask for on the current device
int tree_fit_register_occupancy();   // workgroups per CU the kernel's registers allow (no LDS asked for)
// one launch: jobs that share a storage class and an LDS request; false if the device refuses that much LDS
bool launch_tree_fit(const TreeJob *jobs, int njobs, bool big, size_t ldsBytes, const uint8_t *in, double *scratch, double *results, hipStream_t stream);

void launch_mt_stream(uint32_t seedBase, int Spad, int D, uint32_t *U, hipStream_t stream);
void launch_prep(const PrepArgs &a, hipStream_t stream);   // k_prep_models, then k_prep_lut
int null_occupancy(int NK, size_t ldsBytes);   // resident workgroups per CU (0 = cannot launch)
int null_occupancy_plain(int NK, size_t dynLds, int hiOcc);   // the same for a launch that reads its codes from L2 with dynLds bytes of LDS; hiOcc: k_null_occ (0 if N-1 has none)
bool launch_null(int NK, const NullArgs &a, int grid, size_t ldsBytes, bool exact, hipStream_t stream);
void launch_null_rowscan(const NullArgs &a, int items, hipStream_t stream);   // getHSS over the buffers a ROWS launch of k_null left: one wavefront per (item, strand x frame)
void launch_native_sigma(const NativeArgs &a, int nblocks, hipStream_t stream);
bool launch_native_dp(int NK, const NativeArgs &a, int grid, hipStream_t stream);   // a.nItems items over `grid` persistent workgroups
void launch_native_scan(const NativeArgs &a, hipStream_t stream);   // getHSS over a.sAll, one lane per item
void launch_hss_pack(const DevHss *hss, const int *count, int cap, int slots, DevHss *packed, int *offsets, int *total,
                     hipStream_t stream);
// k_results_out: up to five word arrays copied from device memory into (mapped) pinned host memory
struct ResultsOutArgs {
  static constexpr int kParts = 5;
  uint32_t *dst[kParts];
  const uint32_t *src[kParts];
  size_t words[kParts];
};
void launch_results_out(const ResultsOutArgs &a, hipStream_t stream);
void launch_sk_row(const uint8_t *blob, const DevBlock *dblocks, const uint32_t *flags, int bi, int s, int b, int iMax, float *out, int stride,
                   hipStream_t stream);
// wider blocks (N > 64): generic kernels with their states in a global scratch (rc_null_generic.h)
size_t null_generic_lds_bytes(int N, int nnodes);   // packed node states + codon windows of the widest block of the launch
// the same in two launches (simulation with many light wavefronts, then the DP): bytes of an item's codes / of a DP workgroup's states
size_t null_generic_codes_bytes(int N, int L, int nnodes);
size_t null_generic_state_bytes(int N, int L, int nnodes);
int generic_sim_occupancy(size_t ldsBytes);
int generic_dp_occupancy(size_t ldsBytes);
void launch_generic_sim(const NullArgs &a, int grid, size_t ldsBytes, hipStream_t stream, bool tiled = false);   // tiled: the codes in k_tiled_dp's layout
// blocks of 28..64 rows in tiles of KT sequences (rc_null_tiled.h): bytes of an item's codes / of a DP workgroup's row buffer, occupancy, launch
size_t null_tiled_codes_bytes(int NK, int KT, int L);
size_t null_tiled_state_bytes(int L);
int tiled_dp_occupancy(int KT, size_t ldsBytes);
bool launch_tiled_dp(int KT, bool nanSem, const NullArgs &a, int grid, size_t ldsBytes, uint8_t *scratchBytes, hipStream_t stream);
void launch_generic_dp(const NullArgs &a, int grid, size_t ldsBytes, uint8_t *scratchBytes, hipStream_t stream);
void launch_native_dp_generic(const NativeArgs &a, int nblocks, float *scratch, size_t scratchStride, hipStream_t stream);
void launch_stop_mark(const FitArgs &a, int nblocks, hipStream_t stream);
void launch_evd_fit(const FitArgs &a, int nblocks, bool latency, hipStream_t stream);   // latency: no other batch is in flight
void launch_evd_fit_f64(const double *x, int n, FitOut *out, int expMode, hipStream_t stream);

}  // namespace rc
