// mb_bpermute_addr.hip -- does ds_bpermute_b32 ignore the address bits above the lane index (addr[7:2]) on gfx950?
// k_generic_dp keeps its sigma codes pre-multiplied by four, one per byte, and uses the code WORD (byte 0) or the word shifted
// right by 8, 16, 24 as the address: the bytes above the selected one must not matter.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mb_bpermute_addr tools/mb_bpermute_addr.hip && tools/mb_bpermute_addr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64) void k(uint32_t *out) {
  const int lane = threadIdx.x;
  const int src = lane * 1000 + 7;                                  // what lane `lane` offers
  const uint32_t want = static_cast<uint32_t>((lane * 37 + 5) & 63);   // the lane to read from
  const uint32_t hi = 0xa5c3e100u * static_cast<uint32_t>(lane + 1);   // garbage in bits 8..31
  const uint32_t addr = (want << 2) | (hi & 0xffffff00u);
  out[lane] = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(addr), src));
  out[64 + lane] = want * 1000 + 7;
  out[128 + lane] = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((want << 2) | 3u), src));   // bits 0..1 set
}

int main() {
  uint32_t *d, h[192];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("launch failed\n"); return 1; }
  int okHi = 1, okLo = 1;
  for (int l = 0; l < 64; l++) { okHi &= h[l] == h[64 + l]; okLo &= h[128 + l] == h[64 + l]; }
  std::printf("address bits 8..31 %s, bits 0..1 %s\n", okHi ? "IGNORED" : "NOT ignored", okLo ? "IGNORED" : "NOT ignored");
  return okHi ? 0 : 2;
}
