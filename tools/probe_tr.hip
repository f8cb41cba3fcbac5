// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds lds[i] = i (as f16 bit pattern = index), lane l reads at a
// chosen byte address; dump which 4 elements every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short u16;
__global__ void k(u16* out, int mode) {
  __shared__ __attribute__((aligned(16))) u16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (u16)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;                                     // contiguous 8 B per lane
  else if (mode == 1) addr = ((l & 15) >> 2) * 96 + (l & 3) * 8 + (l >> 4) * 4 * 96;   // 4x16 block rows with pitch 96 B, group g -> rows 4g..
  else addr = (l & 15) * 64 + (l >> 4) * 8;                        // lane = row with pitch 64 B
  addr += (unsigned)(size_t)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  u16* d; hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    u16 h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (element indices; /2 = byte offset)\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 1) ? "\n" : "   |   ");
  }
  return 0;
}
