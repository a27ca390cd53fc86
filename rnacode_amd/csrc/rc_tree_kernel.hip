// rc_tree_kernel.hip -- tree + kappa estimator on the GPU: one wavefront per alignment block
// (stands in for src/treeML.c:35-152 + the bundled PhyML).
//
// The algorithm is rc_tree_core.h, shared with the host build.  Lane l owns site patterns l, l+64, ...
// for the whole fit (their conditional-likelihood columns live in global scratch, pattern-contiguous,
// so a wave's accesses coalesce); pairwise distances are spread one pair per lane; BIONJ (O(N^3) on
// N <= 32) runs on lane 0.  All lanes drive the 1-D minimisers in lockstep: the site sums come out
// of an xor-butterfly, which leaves the same bits on every lane, so control flow never diverges.
// FP64 throughout (the reference's PhyML is double); latency-bound, not a roofline kernel -- what
// matters is that 10^4 blocks are fitted concurrently instead of ~10^3 per second per host core.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rc_launch.h"
#include "rc_tree_core.h"

namespace rc {
namespace {

struct WaveExec {
  __device__ int lane() const { return static_cast<int>(threadIdx.x); }
  __device__ int nlanes() const { return kWave; }
  __device__ double sum(double x) const {
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
    return x;
  }
  __device__ void sync() const { __syncthreads(); }
  __device__ int broadcast_root(int r) const { return __shfl(r, 0, kWave); }
};

__device__ __forceinline__ void tree_fit_body(const TreeJob *jobs, const uint8_t *in, double *scratch, TreeOut *out, double *results) {
  const TreeJob j = jobs[blockIdx.x];
  TreeOut &o = out[blockIdx.x];
  treefit::Work k;
  k.N = j.N; k.P = j.P; k.nn = 2 * j.N - 2; k.root = 0;
  k.mask = in + j.off_mask;
  k.w = reinterpret_cast<const double *>(in + j.off_w);
  k.parent = o.parent; k.nchild = o.nchild; k.child = o.child; k.preorder = o.preorder; k.len = o.len;
  double *s = scratch + j.off_work;
  const size_t nn2 = static_cast<size_t>(j.N) * j.N, part = static_cast<size_t>(k.nn) * 4 * j.P;
  k.D = s; k.V = s + nn2;
  k.down = s + 2 * nn2; k.msg = k.down + part; k.up = k.msg + part;
  k.tmpD = k.up + part;
  k.tmpI = reinterpret_cast<int *>(k.tmpD + j.N);
  WaveExec ex;
  const double lnl = treefit::fit(k, ex);
  // the compact result record (rc_launch.h, tree_result_doubles)
  double *rd = results + j.off_out;
  int *ri = reinterpret_cast<int *>(rd + 2 + k.nn);
  if (threadIdx.x == 0) { rd[0] = k.mod.kappa; rd[1] = lnl; ri[0] = k.root; }
  for (int v = threadIdx.x; v < k.nn; v += kWave) {
    rd[2 + v] = k.len[v];
    ri[1 + v] = k.nchild[v];
    for (int c = 0; c < 3; c++) ri[1 + k.nn + 3 * v + c] = k.child[3 * v + c];
  }
}


// The same body under three register budgets (wavefronts per SIMD the compiler must leave room for: 1, 2 or 3): the fit is latency-bound,
// so more resident wavefronts hide more of it, until the spills a smaller budget forces cost more than they hide
// (RC_TREE_WAVES picks one at run time; the default is the measured best).
#define RC_TREE_KERNEL(W)                                                                                                        \
  __global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(W, W))) void k_tree_fit_w##W(                            \
      const TreeJob *jobs, const uint8_t *in, double *scratch, TreeOut *out, double *results) {                                    \
    tree_fit_body(jobs, in, scratch, out, results);                                                                                \
  }
RC_TREE_KERNEL(1)
RC_TREE_KERNEL(2)
RC_TREE_KERNEL(3)
#undef RC_TREE_KERNEL

}  // namespace

void launch_tree_fit(const TreeJob *jobs, int njobs, const uint8_t *in, double *scratch, TreeOut *work, double *results, hipStream_t stream) {
  if (njobs <= 0) return;
  static const int waves = [] { const char *e = std::getenv("RC_TREE_WAVES"); const int w = e ? std::atoi(e) : 2; return (w >= 1 && w <= 3) ? w : 2; }();
  if (waves == 1) hipLaunchKernelGGL(k_tree_fit_w1, dim3(njobs), dim3(kWave), 0, stream, jobs, in, scratch, work, results);
  else if (waves == 3) hipLaunchKernelGGL(k_tree_fit_w3, dim3(njobs), dim3(kWave), 0, stream, jobs, in, scratch, work, results);
  else hipLaunchKernelGGL(k_tree_fit_w2, dim3(njobs), dim3(kWave), 0, stream, jobs, in, scratch, work, results);
}

}  // namespace rc
