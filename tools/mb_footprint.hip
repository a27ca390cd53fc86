// mb_footprint.hip -- read bandwidth of k_generic_dp's access pattern against the footprint: every wavefront owns a region of R bytes
// and reads 2 KB runs (eight 256-byte rows, one dword per lane) at pseudo-random places in it, eight runs in flight.  Is the
// 3.4 TB/s the DP of the wide blocks gets a property of the pattern (then R does not matter) or of the address translation
// (then it falls as 4096 x R outgrows the TLBs)?
//   hipcc --offload-arch=gfx950 -O2 -o tools/mb_footprint tools/mb_footprint.hip && tools/mb_footprint
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int RUN>   // rows of 256 B per run
__global__ __launch_bounds__(64) void k(const uint32_t *__restrict__ buf, size_t regionWords, int iters, uint32_t *out) {
  const int lane = threadIdx.x;
  const uint32_t *mine = buf + static_cast<size_t>(blockIdx.x) * regionWords + lane;
  const uint32_t runs = static_cast<uint32_t>(regionWords / (64 * RUN));
  uint32_t x = blockIdx.x * 2654435761u + 12345u, acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t v[4][RUN];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      x = x * 1664525u + 1013904223u;
      const uint32_t r = __builtin_amdgcn_readfirstlane((x >> 8) % runs);
      const uint32_t *p = mine + static_cast<size_t>(r) * (64 * RUN);
#pragma unroll
      for (int w = 0; w < RUN; w++) v[q][w] = p[w * 64];
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int w = 0; w < RUN; w++) acc += v[q][w];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int RUN> void run(const uint32_t *d, size_t total, int waves, double regionMB, uint32_t *out) {
  const size_t regionWords = static_cast<size_t>(regionMB * (1 << 20)) / 4;
  if (regionWords * 4 * waves > total) return;
  const int iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<RUN>, dim3(waves), dim3(64), 0, 0, d, regionWords, 50, out);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<RUN>, dim3(waves), dim3(64), 0, 0, d, regionWords, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = static_cast<double>(waves) * iters * 4 * RUN * 256;
  std::printf("run %4d B  waves %5d  region %6.2f MB (footprint %6.1f GB): %7.1f ms  %6.2f TB/s\n", RUN * 256, waves, regionMB, regionMB * waves / 1024.0, ms,
              bytes / ms / 1e9);
}

int main() {
  const size_t total = static_cast<size_t>(48) << 30;
  uint32_t *d, *out;
  CHECK(hipMalloc(&d, total));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(d, 1, total));
  for (int waves : {4096, 8192})
    for (double mb : {0.125, 0.5, 1.0, 2.0, 5.0}) {
      run<8>(d, total, waves, mb, out);
      run<1>(d, total, waves, mb, out);
    }
  return 0;
}
