// mb_ldsdma.hip -- what an LDS-DMA load (global_load_lds_*) writes where, on gfx950: the semantics k_null's code-word ring relies on.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mb_ldsdma tools/mb_ldsdma.hip && tools/mb_ldsdma
// Checks: (1) saddr form with an immediate offset: the offset moves BOTH the global source and the LDS destination;
//         (2) global_load_lds_ubyte / _ushort: one DWORD per lane in LDS (lane x 4), the value zero-extended;
//         (3) a counted vmcnt orders the issuing wave's own ds_read behind the DMA.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k(const uint32_t *__restrict__ src, const uint8_t *__restrict__ srcb, uint32_t *__restrict__ out) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t lo = lane * 4u, lb = lane * 1u, lh = lane * 2u;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  const uint32_t *p = src + 100;          // wave-uniform
  const uint8_t *pb = srcb + 64;
  // (1) dwords: M0 = base + 512 (bytes); offset:256 on the instruction
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 offset:256\n\t" : : "v"(lo), "s"(p), "s"(base + 512u) : "memory");
  // (2) bytes and halfwords
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ubyte %0, %1\n\t" : : "v"(lb), "s"(pb), "s"(base + 2048u) : "memory");
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %0, %1 offset:128\n\t" : : "v"(lh), "s"(pb), "s"(base + 3072u) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<uint32_t> h(4096);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0x1000u + static_cast<uint32_t>(i);
  std::vector<uint8_t> hb(4096);
  for (size_t i = 0; i < hb.size(); i++) hb[i] = static_cast<uint8_t>(i * 7 + 3);
  uint32_t *d, *o; uint8_t *db;
  hipMalloc(&d, h.size() * 4); hipMalloc(&db, hb.size()); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, db, o);
  std::vector<uint32_t> r(1024);
  if (hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("launch failed\n"); return 1; }
  // where did the dwords land?
  int first = -1;
  for (int i = 0; i < 1024; i++) if (r[i] != 0xdeadbeefu && first < 0) first = i;
  std::printf("first changed dword index %d (bytes %d)\n", first, first * 4);
  auto show = [&](const char *what, int at) { std::printf("%s: lds[%d..] = %08x %08x %08x %08x ... [63] %08x\n", what, at, r[at], r[at + 1], r[at + 2], r[at + 3], r[at + 63]); };
  show("dword, M0 = +512 B, offset:256 (expect at dword 192 if the offset moves the LDS side too, values 0x1000 + 100 + 64 + lane)", 192);
  show("same, if the offset did NOT move the LDS side", 128);
  show("ubyte, M0 = +2048 B (expect dword 512 + lane = byte 64 + lane)", 512);
  show("ushort, M0 = +3072 B, offset:128 (expect at dword 768 + 32, halfword (128 + 2 lane) / 2)", 800);
  show("ushort if the offset did not move the LDS side", 768);
  int ok = 1;
  for (int l = 0; l < 64; l++) {
    ok &= r[192 + l] == 0x1000u + 100 + 64 + l;
    ok &= r[512 + l] == hb[64 + l];
    ok &= r[800 + l] == (static_cast<uint32_t>(hb[64 + 128 + 2 * l]) | (static_cast<uint32_t>(hb[64 + 128 + 2 * l + 1]) << 8));
  }
  std::printf(ok ? "SEMANTICS AS ASSUMED: offset moves both sides, sub-dword loads land zero-extended at lane x 4\n" : "SEMANTICS DIFFER\n");
  return 0;
}
