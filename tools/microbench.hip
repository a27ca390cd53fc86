// microbench.hip -- issue-rate probes on gfx950 that calibrate the VALU / LDS-permute roofs used in
// DESIGN.md (run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb && /tmp/mb)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void probe(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  int p0 = threadIdx.x * 4;
  const double ones = __hiloint2double(0x3f800000, 0x3f800000);   // (1.0f, 1.0f)
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 8 independent v_add_f32 chains
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 1) {  // v_max3_f32
      asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                   "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 2) {  // ds_bpermute_b32, 8 independent
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_bpermute_b32 %0, %8, %9\n ds_bpermute_b32 %1, %8, %10\n ds_bpermute_b32 %2, %8, %11\n ds_bpermute_b32 %3, %8, %12\n"
                   "ds_bpermute_b32 %4, %8, %13\n ds_bpermute_b32 %5, %8, %14\n ds_bpermute_b32 %6, %8, %15\n ds_bpermute_b32 %7, %8, %16\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                   : "v"(p0), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
      x0 = __int_as_float(r0); x1 = __int_as_float(r1); x2 = __int_as_float(r2); x3 = __int_as_float(r3);
      x4 = __int_as_float(r4); x5 = __int_as_float(r5); x6 = __int_as_float(r6); x7 = __int_as_float(r7);
    } else if (MODE == 3) {  // v_pk_add_f32 on 4 register pairs (8 float adds per 4 instructions)
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(*(double *)&x0), "+v"(*(double *)&x2), "+v"(*(double *)&x4), "+v"(*(double *)&x6) : "v"(*(double *)&a));
    } else if (MODE == 4) {  // v_cndmask_b32 with vcc
      asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
    } else if (MODE == 5) {  // dependent v_add_f32 chain (latency)
      asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   : "+v"(x0) : "v"(a));
    } else if (MODE == 7) {  // v_pk_fma_f32 with multiplier (1.0, 1.0): fma(x, 1, y) = RN(x + y), two adds per instruction
      asm volatile("v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %2, %2, %5, %4\n v_pk_fma_f32 %3, %3, %5, %4\n"
                   "v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %2, %2, %5, %4\n v_pk_fma_f32 %3, %3, %5, %4\n"
                   : "+v"(*(double *)&x0), "+v"(*(double *)&x2), "+v"(*(double *)&x4), "+v"(*(double *)&x6) : "v"(*(double *)&a), "v"(ones));
    } else if (MODE == 8) {  // v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(*(double *)&x0), "+v"(*(double *)&x2), "+v"(*(double *)&x4), "+v"(*(double *)&x6) : "v"(ones));
    } else if (MODE == 9) {  // v_pk_add_f32 with op_sel broadcasting the low half of the second operand (one omega register for both halves)
      asm volatile("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                   : "+v"(*(double *)&x0), "+v"(*(double *)&x2), "+v"(*(double *)&x4), "+v"(*(double *)&x6) : "v"(*(double *)&a));
    } else if (MODE == 10) {  // v_max_f32 with both operands distinct VGPRs, result to a third (no read-write operand)
      asm volatile("v_max_f32 %0, %1, %8\n v_max_f32 %1, %2, %8\n v_max_f32 %2, %3, %8\n v_max_f32 %3, %4, %8\n"
                   "v_max_f32 %4, %5, %8\n v_max_f32 %5, %6, %8\n v_max_f32 %6, %7, %8\n v_max_f32 %7, %0, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 11) {  // v_pk_max_f32? (not in the gfx950 ISA: kept out) -- v_max_f32 e64 encoding (VOP3)
      asm volatile("v_max_f32_e64 %0, %0, %8\n v_max_f32_e64 %1, %1, %8\n v_max_f32_e64 %2, %2, %8\n v_max_f32_e64 %3, %3, %8\n"
                   "v_max_f32_e64 %4, %4, %8\n v_max_f32_e64 %5, %5, %8\n v_max_f32_e64 %6, %6, %8\n v_max_f32_e64 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 12) {  // the common arm of a cheaper scan step: one v_cmp to an SGPR pair + two v_cndmask, against the full step's 4 + 2
      asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %1, %1, %0, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n"
                   "v_cmp_gt_f32 s[22:23], %3, %4\n v_cndmask_b32 %4, %4, %3, s[22:23]\n v_cndmask_b32 %5, %5, %8, s[22:23]\n"
                   "v_add_f32 %0, %0, %8\n v_add_f32 %3, %3, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "s20", "s21", "s22", "s23");
    } else if (MODE == 6) {  // mixed: 8 v_add + 8 s_add (SALU co-issue)
      int s = i;
      asm volatile("v_add_f32 %0, %0, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %1, %1, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %2, %2, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %3, %3, %9\n s_add_u32 %8, %8, 1\n"
                   "v_add_f32 %4, %4, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %5, %5, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %6, %6, %9\n s_add_u32 %8, %8, 1\n v_add_f32 %7, %7, %9\n s_add_u32 %8, %8, 1\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s) : "v"(a));
      if (s == 12345) x0 += 1;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
int run(const char *name, int opsPerIter) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 32 * 64 * 2));
  const int iters = 100000;
  for (int wavesPerSimd : {1, 2, 4, 8}) {
    const int blocks = cus * 4 * wavesPerSimd;  // 64-thread blocks
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0f, 2.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f, 2.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = double(blocks) * iters * opsPerIter;            // wave-instructions
    const double perSimdPerNs = winst / (cus * 4.0) / (ms * 1e6);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Gwave-inst/s chip  %.3f wave-inst/ns/SIMD (= %.2f cycles/inst at 2.4 GHz)\n", name,
           wavesPerSimd, ms, winst / (ms * 1e6), perSimdPerNs, 2.4 / perSimdPerNs);
  }
  CHECK(hipFree(out));
  return 0;
}


#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int MODE>
__global__ void probe2(float *out, int iters, float a, float b) {
  float x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
  unsigned long long m = 0x5555555555555555ull + (unsigned long long)iters;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {        // v_cndmask only, mask in an SGPR pair prepared outside the loop
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "s"(m));
      REP8(OP)
#undef OP
    } else if (MODE == 1) { // v_cmp only (to vcc)
#define OP(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[k]), "v"(a) : "vcc");
      REP8(OP)
#undef OP
    } else if (MODE == 2) { // v_max_f32 (VOP2)
#define OP(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      REP8(OP)
#undef OP
    } else if (MODE == 3) { // v_bfe_u32
#define OP(k) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x[k]));
      REP8(OP)
#undef OP
    } else if (MODE == 4) { // v_and_b32
#define OP(k) asm volatile("v_and_b32 %0, 0xff, %0" : "+v"(x[k]));
      REP8(OP)
#undef OP
    } else if (MODE == 5) { // v_fma_f32
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      REP8(OP)
#undef OP
    } else if (MODE == 6) { // v_cmp to SGPR pair + v_cndmask reading it (pairs)
#define OP(k) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x[k]) : "v"(a) : "s20", "s21");
      REP8(OP)
#undef OP
    } else if (MODE == 7) { // ds_read_b32 conflict-free (lane*4)
      int addr = threadIdx.x * 4;
#define OP(k) asm volatile("ds_read_b32 %0, %1 offset:" #k "*256" : "=v"(x[k]) : "v"(addr));
      REP8(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (MODE == 8) { // v_mov_b32 dpp row_shr (cross-lane within row)
#define OP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[k]));
      REP8(OP)
#undef OP
    } else if (MODE == 9) { // v_add_f32 with SGPR operand
#define OP(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[k]) : "s"(a));
      REP8(OP)
#undef OP
    }
  }
  float r = 0;
  for (int k = 0; k < 8; k++) r += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
int run2(const char *name, int opsPerIter) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 32 * 64 * 2));
  const int iters = 100000;
  for (int wavesPerSimd : {1, 4, 8}) {
    const int blocks = cus * 4 * wavesPerSimd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe2<MODE>, dim3(blocks), dim3(64), 4096, 0, out, 100, 1.0f, 2.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe2<MODE>, dim3(blocks), dim3(64), 4096, 0, out, iters, 1.0f, 2.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = double(blocks) * iters * opsPerIter;
    const double perSimdPerNs = winst / (cus * 4.0) / (ms * 1e6);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Gwave-inst/s chip  %.3f wave-inst/ns/SIMD\n", name, wavesPerSimd, ms, winst / (ms * 1e6), perSimdPerNs);
  }
  CHECK(hipFree(out));
  return 0;
}

// The shader clock while every SIMD issues VALU work: s_memtime (shader cycles) against s_memrealtime (100 MHz) over a window in the
// middle of a long launch of MODE (0: v_add chains, 2: v_max).
template <int MODE>
__global__ void clock_probe(float *out, unsigned *cyc, int iters, float a) {
  float x[8];
  for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
  unsigned long long c0 = 0, c1 = 0, w0 = 0, w1 = 0;
  for (int rep = 0; rep < 40; rep++) {
    if (rep == 20) { c0 = clock64(); w0 = wall_clock64(); }   // s_memtime: shader cycles (MI355X_MICROARCH.md); s_memrealtime: 100 MHz
    if (rep == 21) { c1 = clock64(); w1 = wall_clock64(); }
    for (int i = 0; i < iters; i++) {
      if (MODE == 0) {
#define OP(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
        REP8(OP)
#undef OP
      } else {
#define OP(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
        REP8(OP)
#undef OP
      }
    }
  }
  float r = 0;
  for (int k = 0; k < 8; k++) r += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = static_cast<unsigned>(c1 - c0); cyc[1] = static_cast<unsigned>(w1 - w0); }
}
template <int MODE>
int run_clock(const char *name) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  unsigned *cyc, h[2];
  CHECK(hipMalloc(&out, sizeof(float) * cus * 32 * 64 * 2));
  CHECK(hipMalloc(&cyc, 8));
  for (int wavesPerSimd : {1, 4, 8}) {
    const int blocks = cus * 4 * wavesPerSimd;
    const int iters = 8000 / wavesPerSimd;
    hipLaunchKernelGGL(clock_probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost));
    const double instr = double(iters) * 8 * wavesPerSimd;   // wave-instructions the SIMD issued inside the window (all its wavefronts)
    printf("%-28s waves/SIMD=%d  window %u shader cycles in %u ticks of 10 ns = %.3f GHz; %.2f shader cycles per wave-instruction and SIMD\n", name, wavesPerSimd,
           h[0], h[1], h[0] / (h[1] * 10.0), h[0] / instr);
  }
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
  return 0;
}

int main() {
  run_clock<0>("clock under v_add_f32");
  run_clock<2>("clock under v_max_f32");
  run<0>("v_add_f32 x8 indep", 8);
  run<1>("v_max3_f32 x8 indep", 8);
  run<2>("ds_bpermute_b32 x8", 8);
  run<3>("v_pk_add_f32 x8 (16 adds)", 8);
  run<4>("v_cmp + 7 v_cndmask", 8);
  run<5>("v_add_f32 dependent x8", 8);
  run<6>("8 v_add + 8 s_add", 16);
  run<7>("v_pk_fma_f32 x1.0 x8 (16 adds)", 8);
  run<8>("v_pk_mul_f32 x8 (16 muls)", 8);
  run<9>("v_pk_add_f32 op_sel_hi x8", 8);
  run<10>("v_max_f32 3 distinct regs", 8);
  run<11>("v_max_f32_e64", 8);
  run<12>("2x(cmp+2cndmask)+2 add", 8);
  run2<0>("v_cndmask (sgpr mask)", 8);
  run2<1>("v_cmp_gt_f32 -> vcc", 8);
  run2<2>("v_max_f32 vop2", 8);
  run2<3>("v_bfe_u32", 8);
  run2<4>("v_and_b32", 8);
  run2<5>("v_fma_f32", 8);
  run2<6>("v_cmp->sgpr + v_cndmask", 16);
  run2<7>("ds_read_b32", 8);
  run2<8>("v_mov_b32_dpp row_shr", 8);
  run2<9>("v_add_f32 sgpr operand", 8);
  return 0;
}
