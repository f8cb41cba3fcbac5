// Probe: operand layout of v_mfma_f32_16x16x16_f16 (the K = 16 form).  Assumption under test: lane l holds A[i = l % 16][k = 4 * (l / 16) + j],
// B[k = 4 * (l / 16) + j][n = l % 16], and D[m = 4 * (l / 16) + r][n = l % 16] like the 16x16x32 form.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const f16* A, const f16* B, float* D) {   // A [16][16] row-major (i, k), B [16][16] (k, n)
  const int l = threadIdx.x, i = l % 16, kg = l / 16;
  f16x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = A[i * 16 + 4 * kg + j]; b[j] = B[(4 * kg + j) * 16 + i]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * kg + r) * 16 + i] = c[r];
}
int main() {
  f16 hA[256], hB[256]; float hD[256], ref[256];
  for (int i = 0; i < 256; ++i) { hA[i] = (f16)((i * 7 % 13) - 6); hB[i] = (f16)((i * 5 % 11) - 5); }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += (float)hA[m * 16 + kk] * (float)hB[kk * 16 + n]; ref[m * 16 + n] = s; }
  f16 *dA, *dB; float* dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x16f16 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  return 0;
}
