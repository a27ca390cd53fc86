// mb_lds_lookup.hip -- a 64-entry table look-up per lane: ds_bpermute_b32 (table spread over the lanes of a register; what k_null does)
// against ds_read_b32 from a table in LDS, at the address patterns the sigma codes make: every lane another entry (uniform random),
// most lanes the same entry (the samples of a column mostly agree), all lanes the same.  Rates per CU, 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mb_lds_lookup tools/mb_lds_lookup.hip && tools/mb_lds_lookup
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int MODE>   // 0 bpermute, 1 ds_read_b32
__global__ __launch_bounds__(64) void k(const uint32_t *codes, int iters, float *out) {
  __shared__ float table[64 * 12];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 12; i += 64) table[i] = static_cast<float>(i) * 0.5f;
  __syncthreads();
  const float mine = static_cast<float>(lane) * 0.5f;
  uint32_t a[8];
  for (int t = 0; t < 8; t++) a[t] = codes[t * 64 + lane];   // code * 4, per lane
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    float g[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if (MODE == 0) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(g[t]) : "v"(a[t]), "v"(mine));
      else asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(g[t]) : "v"(a[t]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]));
#pragma unroll
    for (int t = 0; t < 8; t++) acc += g[t];
    a[0] ^= (__float_as_uint(acc) & 0u);   // keep the loop honest without changing the addresses
  }
  if (acc == 12345.0f) out[0] = acc + table[1];
}

int main() {
  const int waves = 256 * 4 * 4;   // 4 per SIMD
  uint32_t *d; float *out;
  CHECK(hipMalloc(&d, 8 * 64 * 4)); CHECK(hipMalloc(&out, 64));
  uint32_t h[8 * 64];
  const char *names[] = {"every lane another entry (uniform random)", "3 of 4 lanes the same entry, the rest random", "15 of 16 lanes the same entry", "all lanes the same entry",
                         "lane i reads entry i"};
  for (int pat = 0; pat < 5; pat++) {
    uint32_t x = 12345u;
    for (int t = 0; t < 8; t++) {
      x = x * 1664525u + 1013904223u;
      const uint32_t common = (x >> 10) & 63u;
      for (int l = 0; l < 64; l++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = (x >> 10) & 63u, u = (x >> 20) & 15u;
        uint32_t c = pat == 0 ? r : pat == 1 ? (u < 12 ? common : r) : pat == 2 ? (u < 15 ? common : r) : pat == 3 ? common : static_cast<uint32_t>(l);
        h[t * 64 + l] = c * 4u;
      }
    }
    CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; mode++) {
      const int iters = 20000;
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, 100, out); else hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, 100, out);
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, iters, out); else hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, iters, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%-50s %-16s %7.2f ms  %.3f look-ups per ns and CU\n", names[pat], mode == 0 ? "ds_bpermute_b32" : "ds_read_b32", ms,
                  static_cast<double>(waves) * iters * 8 / 256 / (ms * 1e6));
    }
  }
  return 0;
}
