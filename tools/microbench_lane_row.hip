// microbench_lane_row.hip -- what the long-block shape would cost in the OTHER decomposition: one wavefront per sample, lanes over
// start codons (rows of S), the sigma values of a sample wave-resident (registers, broadcast with v_readlane), rows scanned 64 at
// a time with the ballot walk of the native kernel.  No code words are re-read from memory at all in this form; the question is
// what it costs in instructions: the triangle leaves lanes idle (rows a > j), and getHSS's fold over the positive entries is
// serial per sample (here: per wavefront).
//
// One unit = one (sample, strand x frame) of 12 sequences x 300 columns (100 codon sites): the DP over all (a, j), a <= j, with the
// recurrence of the event-free cell (score.c:506-510: s0 += sigma, s1 += omega, s2 += omega, max3, sum, divide), S kept for 64 rows
// in LDS, then getHSS's state machine (scan_step of rc_null_kernel.h) over the positive entries in the reference's order.
// sigma values are pseudo-random with the drift that gives the positive-entry rate measured in k_null at this shape (10.7 %).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rnacode_amd/csrc tools/microbench_lane_row.hip -o /tmp/mb_lane_row && /tmp/mb_lane_row
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "rc_null_kernel.h"

using namespace rc;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int NK, bool WITH_SCAN>
__global__ __launch_bounds__(64) void lane_row(float *out, int iters, int sites, float omega, float drift, unsigned long long *stats) {
  extern __shared__ float tile[];   // [64 rows][stride]
  const int lane = threadIdx.x;
  const int stride = sites | 1;     // odd: rows land in different banks
  float acc = 0.0f;
  unsigned long long positives = 0, entries = 0;
  for (int it = 0; it < iters; it++) {
    // sigma of this sample: lane l holds sites l and 64 + l for every sequence
    float sig[2][NK];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int k = 0; k < NK; k++) {
        const uint32_t r = mix(static_cast<uint32_t>(blockIdx.x) * 7919u + static_cast<uint32_t>(it) * 104729u + (h * 64 + lane) * 131u + k);
        sig[h][k] = static_cast<float>(static_cast<int>(r & 15u) - 8) * 0.75f + drift;   // -6 .. +5.25, mean drift - 0.375
      }
    ScanState st{0.0f, -1, -1};
    float best = -1.0f;
    for (int c0 = 0; c0 < sites; c0 += kWave) {
      const int a = c0 + lane;
      float s0[NK], s1[NK], s2[NK];
#pragma unroll
      for (int k = 0; k < NK; k++) s0[k] = s1[k] = s2[k] = 0.0f;
      for (int j = c0; j < sites; j++) {
        float sg[NK];
        const int src = j & 63;
        if (j < 64) {
#pragma unroll
          for (int k = 0; k < NK; k++) sg[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sig[0][k]), src));
        } else {
#pragma unroll
          for (int k = 0; k < NK; k++) sg[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sig[1][k]), src));
        }
        if (a <= j && a < sites) {
          float sum = 0.0f;
#pragma unroll
          for (int k = 0; k < NK; k++) {
            s0[k] = s0[k] + sg[k];
            s1[k] = s1[k] + omega;
            s2[k] = s2[k] + omega;
            sum = sum + fmaxf(fmaxf(s0[k], s1[k]), s2[k]);
          }
          tile[lane * stride + j] = div_by_nk<NK, false>(sum, static_cast<float>(NK));
        }
      }
      __syncthreads();
      if (WITH_SCAN) {
        const int rows = (sites - c0 < kWave) ? sites - c0 : kWave;
        for (int r = 0; r < rows; r++) {
          const int arow = c0 + r;
          for (int j0 = arow; j0 < sites; j0 += kWave) {
            const int j = j0 + lane;
            const float v = (j < sites) ? tile[r * stride + j] : 0.0f;
            const bool last = (arow == sites - 1) && (j == sites - 1);
            unsigned long long m = __ballot((j < sites) && (v > 0.0f || last));
            entries += __builtin_popcountll(__ballot(j < sites));
            positives += __builtin_popcountll(m);
            while (m) {
              const int t = __builtin_ctzll(m);
              m &= m - 1ull;
              const float vt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t));
              scan_step(st, vt, arow, j0 + t, (arow == sites - 1) && (j0 + t == sites - 1), 1e-4f,
                        [&](const ScanState &e) { best = e.cm > best ? e.cm : best; });
            }
          }
        }
      } else {
        acc += tile[lane * stride + (sites - 1)];
      }
      __syncthreads();
    }
    acc += best;
  }
  out[blockIdx.x * kWave + lane] = acc;
  if (WITH_SCAN && lane == 0) { atomicAdd(stats, positives); atomicAdd(stats + 1, entries); }
}

template <bool WITH_SCAN>
int run(const char *name, int sites, float drift) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  constexpr int NK = 11;
  const size_t lds = static_cast<size_t>(64) * (sites | 1) * sizeof(float);
  int occ = 0;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lane_row<NK, WITH_SCAN>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lane_row<NK, WITH_SCAN>, 64, lds));
  const int grid = prop.multiProcessorCount * occ, iters = 64;
  float *out;
  unsigned long long *stats;
  CHECK(hipMalloc(&out, sizeof(float) * grid * 64));
  CHECK(hipMalloc(&stats, 16));
  CHECK(hipMemset(stats, 0, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((lane_row<NK, WITH_SCAN>), dim3(grid), dim3(64), lds, 0, out, 4, sites, -2.0f, drift, stats);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(stats, 0, 16));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((lane_row<NK, WITH_SCAN>), dim3(grid), dim3(64), lds, 0, out, iters, sites, -2.0f, drift, stats);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long st[2] = {0, 0};
  CHECK(hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost));
  const double units = static_cast<double>(grid) * iters;
  printf("%-34s sites=%d  %d wavefronts per CU (LDS %zu B each)  %.3f ms for %.0f (sample, strand x frame) units = %.3e units/s", name, sites, occ, lds, ms, units,
         units / (ms * 1e-3));
  if (WITH_SCAN && st[1]) printf("  positive entries %.1f %%", 100.0 * st[0] / st[1]);
  printf("\n");
  (void)hipFree(out); (void)hipFree(stats);
  return 0;
}

int main() {
  printf("lane = start codon, one wavefront per sample; 12 sequences x 300 columns (100 codon sites per frame)\n");
  printf("reference point: k_null<11,false,false> (lane = sample) scores 1000 blocks x 10 000 samples x 6 strand x frame units in ~600 ms = 1.0e8 units/s\n");
  for (float drift : {0.3f, 0.1f, -0.1f, -0.3f, -0.6f}) {
    printf("sigma drift %.1f:\n", drift);
    run<false>("DP only (no getHSS fold)", 100, drift);
    run<true>("DP + serial fold over positives", 100, drift);
  }
  return 0;
}
