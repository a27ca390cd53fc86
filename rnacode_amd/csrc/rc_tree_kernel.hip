// rc_tree_kernel.hip -- tree + kappa estimator on the GPU: one wavefront per alignment block
// (stands in for src/treeML.c:35-152 + the bundled PhyML).
//
// The algorithm is rc_tree_core.h, shared with the host build.  Lane l owns site patterns l, l+64, ... for the whole fit; pairwise
// distances go one pair per lane; BIONJ (O(N^3), N <= 64) runs on lane 0.  All lanes drive the 1-D searches in lockstep: the site
// sums come back as the same bits on every lane (DPP inside the rows of sixteen, v_readlane across them), so control flow never
// diverges.  FP64 throughout (the reference's PhyML is double).
//
// Round 5: the fit's whole working set lives in LDS.  The round-3/4 kernel kept the tree, the distance matrices and the
// conditional-likelihood columns in global memory and spent 74 % of its wavefront cycles waiting for them: 12 800 dependent global
// loads per fit, a microsecond each, were the 5 ms a fit took.  Now
//   * the tree (parent / child / pre-order / lengths), the per-branch transfer constants, BIONJ's matrices, the block's masks and
//     weights and the columns are LDS arrays (k_tree_fit<false>: everything; k_tree_fit<true>, for blocks whose columns do not fit
//     a CU's LDS: the columns, masks and weights stay in global memory, read and written a whole column at a time -- a handful of
//     round trips per branch, none inside Newton's iterations);
//   * a branch's image of a column needs no 4 x 4 matrix (rc_tree_core.h, transfer: HKY85's three exponentials enter through five
//     constants), and the three exponentials of a branch are computed on three lanes at once and handed round with v_readlane;
//   * Newton's iterations for a branch length run on four constants per pattern (branch_constants), not on the columns;
//   * the branches are visited in pre-order, so every column is made once per round.
// Launches are grouped by the LDS a block needs (tree_fit_lds_bytes): small blocks run many to a CU, long ones few.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rc_launch.h"
#include "rc_tree_core.h"

namespace rc {
namespace {

template <int CTRL> __device__ __forceinline__ double dpp_add(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// the same bits on every lane, no LDS round trip (see k_evd_fit in rc_kernels.hip)
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// -DRC_TREE_PROFILE (tools/tree_phases.sh builds it into tools/, never the product): cycles per phase of the fit and three counters,
// appended to every job's result record (kTreeProfDoubles doubles) and summed by rc_fit_trees_device
#ifdef RC_TREE_PROFILE
constexpr bool kTreeProfile = true;
#else
constexpr bool kTreeProfile = false;
#endif

struct WaveExec {
  double *prof = nullptr;          // [kTreeProfDoubles] in LDS
  long long last = 0;
  __device__ void mark(int phase) {
    if constexpr (kTreeProfile) {
      const long long now = clock64();
      if (threadIdx.x == 0) prof[phase] += static_cast<double>(now - last);
      last = now;
    }
  }
  __device__ void count(int c) { if constexpr (kTreeProfile) { if (threadIdx.x == 0) prof[c] += 1.0; } }
  __device__ int lane() const { return static_cast<int>(threadIdx.x); }
  __device__ int nlanes() const { return kWave; }
  __device__ double sum(double x) const { return wave_sum(x); }
  __device__ void sum2(double &a, double &b) const { a = wave_sum(a); b = wave_sum(b); }
  // three exponentials for the price of one: lane 1 and lane 2 take the second and third argument
  __device__ void exp3(double x0, double x1, double x2, double &e0, double &e1, double &e2) const {
    const int l = static_cast<int>(threadIdx.x);
    const double e = exp(l == 1 ? x1 : (l == 2 ? x2 : x0));
    e0 = lane_value(e, 0); e1 = lane_value(e, 1); e2 = lane_value(e, 2);
  }
  __device__ void sync() const { __syncthreads(); }
  __device__ int broadcast_root(int r) const { return __builtin_amdgcn_readfirstlane(r); }
};

extern __shared__ double tree_lds[];

// BIG: columns, masks and weights in global memory (the job's work area / the input blob); otherwise everything in LDS
template <bool BIG>
__device__ __forceinline__ void tree_fit_body(const TreeJob *jobs, const uint8_t *in, double *scratch, double *results) {
  const TreeJob j = jobs[blockIdx.x];
  treefit::Work k;
  const int N = j.N, P = j.P, nn = 2 * N - 2, nI = N > 3 ? N - 3 : 0;   // column slots: the internal nodes below the root (bionj numbers the root last)
  k.N = N; k.P = P; k.nn = nn; k.root = 0;
  const TreeLdsLayout lay = tree_lds_layout(N, P, BIG);
  double *d = tree_lds;
  k.len = d + lay.len; k.coef = reinterpret_cast<treefit::Coef *>(d + lay.coef); k.e3 = d + lay.e3;
  k.D = d + lay.D; k.V = d + lay.V; k.tmpD = d + lay.tmpD;
  static_assert(sizeof(treefit::OptState) <= kTreeOptDoubles * sizeof(double), "OptState outgrew its LDS slot");
  k.opt = reinterpret_cast<treefit::OptState *>(d + lay.opt);
  int *ib = reinterpret_cast<int *>(d + lay.ints);
  k.parent = ib; k.nchild = ib + nn; k.child = ib + 2 * nn; k.preorder = ib + 5 * nn; k.tmpI = ib + 6 * nn;
  const uint8_t *gmask = in + j.off_mask;
  const double *gw = reinterpret_cast<const double *>(in + j.off_w);
  if constexpr (BIG) {
    double *s = scratch + j.off_work;
    const size_t col = static_cast<size_t>(4) * P;
    k.dnI = s; k.upI = s + static_cast<size_t>(nI) * col; k.kc = s + 2 * static_cast<size_t>(nI) * col;
    k.mask = gmask; k.w = gw;
  } else {
    const size_t col = static_cast<size_t>(4) * P;
    k.dnI = d + lay.cols; k.upI = k.dnI + static_cast<size_t>(nI) * col; k.kc = k.upI + static_cast<size_t>(nI) * col;
    double *lw = d + lay.w;
    uint8_t *lm = reinterpret_cast<uint8_t *>(d + lay.mask);
    for (int p = threadIdx.x; p < P; p += kWave) lw[p] = gw[p];
    for (int q = threadIdx.x; q < N * P; q += kWave) lm[q] = gmask[q];
    k.mask = lm; k.w = lw;
    __syncthreads();
  }
  WaveExec ex;
  if constexpr (kTreeProfile) {
    ex.prof = d + lay.opt + kTreeOptDoubles - kTreeProfDoubles;   // the tail of the optimiser-state slot
    static_assert(sizeof(treefit::OptState) <= (kTreeOptDoubles - kTreeProfDoubles) * sizeof(double), "no room for the profile");
    if (threadIdx.x < kTreeProfDoubles) ex.prof[threadIdx.x] = 0.0;
    __syncthreads();
    ex.last = clock64();
  }
  const double lnl = treefit::fit(k, ex);
  if constexpr (kTreeProfile) ex.mark(8);
  // the compact result record (rc_launch.h, tree_result_doubles)
  double *rd = results + j.off_out;
  int *ri = reinterpret_cast<int *>(rd + 2 + nn);
  if (threadIdx.x == 0) { rd[0] = k.mod.kappa; rd[1] = lnl; ri[0] = k.root; }
  for (int v = threadIdx.x; v < nn; v += kWave) {
    rd[2 + v] = k.len[v];
    ri[1 + v] = k.nchild[v];
    for (int c = 0; c < 3; c++) ri[1 + nn + 3 * v + c] = k.child[3 * v + c];
  }
  if constexpr (kTreeProfile) {
    __syncthreads();
    double *pd = rd + tree_result_doubles(N) - kTreeProfDoubles;
    if (threadIdx.x < kTreeProfDoubles) pd[threadIdx.x] = ex.prof[threadIdx.x];
  }
}

// RC_TREE_WAVES (a build-time experiment: make OBJDIR=… CXXFLAGS+=-DRC_TREE_WAVES=3, profiles/r05/tree_occupancy_and_register_budgets.txt): the register budget as wavefronts per SIMD; unset = what the compiler takes
#ifdef RC_TREE_WAVES
#define RC_TREE_BUDGET __attribute__((amdgpu_waves_per_eu(RC_TREE_WAVES, RC_TREE_WAVES)))
#else
#define RC_TREE_BUDGET
#endif
template <bool BIG>
__global__ __launch_bounds__(kWave) RC_TREE_BUDGET void k_tree_fit(const TreeJob *jobs, const uint8_t *in, double *scratch, double *results) {
  tree_fit_body<BIG>(jobs, in, scratch, results);
}

}  // namespace

int tree_fit_max_lds() {
  static const int cap = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 65536;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 65536;
    return v;
  }();
  return cap;
}

int tree_fit_register_occupancy() {
  static const int occ = [] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tree_fit<false>, kWave, 0) != hipSuccess || nb <= 0) { (void)hipGetLastError(); nb = 8; }
    return nb;
  }();
  return occ;
}

bool launch_tree_fit(const TreeJob *jobs, int njobs, bool big, size_t ldsBytes, const uint8_t *in, double *scratch, double *results, hipStream_t stream) {
  if (njobs <= 0) return true;
  auto *fn = big ? k_tree_fit<true> : k_tree_fit<false>;
  if (ldsBytes > 65536 && hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  hipLaunchKernelGGL(fn, dim3(njobs), dim3(kWave), ldsBytes, stream, jobs, in, scratch, results);
  return true;
}

}  // namespace rc
