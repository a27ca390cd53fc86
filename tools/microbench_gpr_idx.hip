// microbench_gpr_idx.hip -- what an indexed VGPR move (s_set_gpr_idx_on / v_mov_b32 / s_set_gpr_idx_off) costs on gfx950 next to a plain
// v_mov_b32, inside a loop of ordinary VALU work, at 1..8 wavefronts per SIMD.  (k_null's two-rows-per-pass variant keeps row a + 1's
// S values in a 32-register buffer indexed with the wave-uniform site number.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_gpr_idx.hip -o tools/mb_gpr_idx && tools/mb_gpr_idx
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v32f __attribute__((ext_vector_type(32)));

template <int MODE>   // 0: 16 v_add only; 1: + plain v_mov; 2: + indexed store; 3: + indexed store and indexed load
__global__ __launch_bounds__(64) void k(float *out, int iters, int n) {
  v32f buf = {};
  float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, s = 0.5f, extra = 0.0f;
  asm volatile("v_mov_b32 %0, %0" : "+v"(s));
  for (int it = 0; it < iters; it++) {
    for (int j = 0; j < n; j++) {   // n <= 32, wave-uniform
#pragma unroll
      for (int r = 0; r < 4; r++) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(s));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(s));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(s));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(s));
      }
      if (MODE == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(extra) : "v"(a0));
      if (MODE >= 2) buf[j] = a0;
      if (MODE == 3) extra += buf[(j + 7) & 31];
    }
  }
  float acc = a0 + a1 + a2 + a3 + extra;
  if (MODE >= 2) for (int j = 0; j < 32; j++) acc += buf[j];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE> int run(const char *name, int wavesPerSimd) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * 4 * wavesPerSimd, iters = 2000, n = 32;
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * grid * 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 10, n);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, n);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bodies = static_cast<double>(iters) * n;   // per wavefront
  printf("%-44s waves/SIMD=%d  %.3f ms  %.1f ns per loop body and wavefront (%.1f cycles at 2.4 GHz / waves)\n", name, wavesPerSimd, ms, ms * 1e6 / bodies,
         ms * 1e6 / bodies * 2.4 / wavesPerSimd);
  (void)hipFree(out);
  return 0;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("16 v_add_f32", w);
    run<1>("16 v_add_f32 + v_mov_b32", w);
    run<2>("16 v_add_f32 + indexed store", w);
    run<3>("16 v_add_f32 + indexed store + indexed load", w);
  }
  return 0;
}
