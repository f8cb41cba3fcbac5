// Probe: do out-of-range lanes of buffer_load_dwordx4 ... lds write ZEROS to LDS (gfx950)?  LDS is pre-filled with 7.0.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 7.0f;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lanes 0..39 in range, lanes 40..47 past num_records, lanes 48..63 offset 0xFFFFFFF0
  const unsigned voff = threadIdx.x < 48 ? threadIdx.x * 16u : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 40 * 16);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
