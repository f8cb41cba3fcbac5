// Store-path micro-benchmark (gfx950): how fast can ONE 512-thread block per CU push an fp16 [rows, 320] output tile to HBM,
// as a function of how a wave's 16-byte stores are spread over rows?  Each wave issues 20 global_store_dwordx4 per 256 x 320 tile
// (the GEMM epilogue's volume) for `tiles` tiles; patterns:
//   0: 16 rows x 64 contiguous bytes per instruction  (the MFMA-fragment layout after the permlane16 pairing)
//   1:  8 rows x 128 B     2: 4 rows x 256 B     3: 2 rows x 512 B     4: 1 KB contiguous (part of one row's 640 B / next row)
//   5: 64 rows x 16 B (row-per-lane)
// Output: GB/s and B/clk/CU at the measured time (clock assumed 2.1 GHz for the B/clk figure).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PAT>
__global__ __launch_bounds__(512) void k(uint4* out, long rows, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  constexpr int LDROW = 40;   // uint4 per row (320 halves)
  for (int t = 0; t < tiles; ++t) {
    const long tile = (long)blockIdx.x + (long)t * gridDim.x;
    const long r0 = tile * 256 + (wave >> 1) * 64;      // this wave's 64 rows x 160 columns (20 uint4 per row)
    const int c0 = (wave & 1) * 20;
    if (r0 + 64 > rows) return;
#pragma unroll
    for (int s = 0; s < 20; ++s) {
      long row; int col;
      if (PAT == 0) { const int i = s / 5, jp = s % 5; row = r0 + i * 16 + (lane & 15); col = c0 + jp * 4 + (lane >> 4); }
      else if (PAT == 1) { const int i = s / 5 * 2, jp = s % 5; const int half = (s / 5) >= 2; row = r0 + (i % 4) * 16 + half * 8 + (lane & 7); col = c0 + (jp * 8 + (lane >> 3)) % 20; }
      else if (PAT == 2) { row = r0 + (s / 5) * 16 + (s % 5) * 3 % 16 + (lane & 3) * 4 % 16; col = c0 + (lane >> 2) % 20; row = r0 + ((s * 4 + (lane & 3)) & 63); }
      else if (PAT == 3) { row = r0 + ((s * 2 + (lane & 1)) & 63) ; col = c0 + (lane >> 1) % 20; }
      else if (PAT == 4) { const int lin = s * 64 + lane; row = r0 + lin / 20; col = c0 + lin % 20; }
      else { row = r0 + lane; col = c0 + s; }
      out[row * LDROW + col] = v;
    }
  }
}

template <int PAT>
void run(uint4* d, long rows) {
  const int tiles = (int)(rows / 256 / 256);
  hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, d, rows, tiles);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, d, rows, tiles);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)tiles * 256 * 256 * 320 * 2;
  printf("pattern %d: %.3f ms  %.0f GB/s  %.1f B/clk/CU\n", PAT, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.1);
}

int main() {
  const long rows = 393216;
  uint4* d; hipMalloc(&d, rows * 320 * 2 + 4096);
  run<0>(d, rows); run<1>(d, rows); run<3>(d, rows); run<4>(d, rows); run<5>(d, rows);
  return 0;
}
