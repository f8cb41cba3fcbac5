// Operand-fill micro-benchmark (gfx950): what bounds the global -> LDS fill of the K = 320 GEMM tile -- LATENCY (bytes in flight per CU) or THROUGHPUT?
//
// Background (DESIGN.md sections 3.1, 9.1): a 256 x 320 output tile at K = 320 stages 5 slabs of (256 A rows + 320 W rows) x 128 bytes through LDS by
// `buffer_load_dwordx4 ... lds`; the "loads only" ablation of the real kernel took 83 % of the full kernel's time (16 GB/s per CU) although the same DMA
// path feeds the attention kernel's K | V tiles at 37 GB/s per CU.  The real kernel has two LDS buffers, i.e. ONE slab (73.7 KB) in flight per CU while the
// other is consumed.  This program issues exactly that traffic with nothing else in the way -- one 512-thread block per CU, 9 DMA instructions per wave and
// slab, a barrier per slab -- and varies
//   D     slabs in flight (ring of D LDS slots; the wait is a counted vmcnt),
//   SEG   bytes per row and slab: 128 (BK = 64, the kernel's) or 64 (BK = 32: half-size slabs, so that up to 4 fit the 160 KB of LDS),
//   SRC   0: A streams through a [393216, 320] matrix once per pass (first touch: HBM), W = one [320, 320] matrix (L2);  1: A confined to 4 MB (L2 hits),
//   ST    1: every tile ends with the epilogue's stores (20 global_store_dwordx4 per wave: 16 rows x 64 B each) -- do fills and stores share a bottleneck?
//         2: the same, with the stores credited in the counted waits (they drain under the next fills)
// If GB/s grows with D at equal SEG the fill is latency-bound and a deeper ring (4 x BK = 32 instead of 2 x BK = 64: 110 KB in flight instead of 74) pays;
// if it is flat the path is throughput-bound and only fewer bytes per FLOP help.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_fill tools/ubench_fill.hip && tools/_bin/ubench_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0x80000000u;   // an offset past the buffer: the lane fetches nothing (zero-fill in LDS)
constexpr int ROWS_A = 256, ROWS_W = 320, KBYTES = 640;   // tile rows; bytes of one K = 320 fp16 row

template <int D, int SEG, int ST>
__global__ __launch_bounds__(512) void fill_kernel(const char* A, const char* W, uint4* out, long rows_total, int tiles, int src_window_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPR = SEG / 16;                                 // lanes per row segment
  constexpr int NA = (ROWS_A * LPR + 511) / 512, NW = (ROWS_W * LPR + 511) / 512;   // DMA instructions per wave and slab: A part, W part (4 + 5 at SEG = 128)
  constexpr int NI = NA + NW;
  constexpr int SLABS = KBYTES / SEG;                           // slabs per tile (5 at SEG = 128)
  constexpr int SLOT = (NA + NW) * 8192;                        // LDS bytes per slab slot (8 waves x 1 KB per instruction)
  const int tid = threadIdx.x, wave = tid >> 6;
  const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W), 0, (unsigned)(ROWS_W * KBYTES), 0x00020000);
  // per-lane 16-byte piece -> (row, byte in segment) of the A part and of the W part; pieces past the part are off.  Every instruction reads through ONE
  // descriptor (a per-lane choice of descriptor would make hipcc wrap the load in a waterfall loop)
  unsigned aoff[NA], woff[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int piece = i * 512 + tid, row = piece / LPR;
    aoff[i] = row < ROWS_A ? (unsigned)(row * KBYTES + (piece % LPR) * 16) : OOB;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int piece = i * 512 + tid, row = piece / LPR;
    woff[i] = row < ROWS_W ? (unsigned)(row * KBYTES + (piece % LPR) * 16) : OOB;
  }
  long issued = 0, total = (long)tiles * SLABS;
  int store_credit = 0;
  auto issue = [&](long s) {      // slab s of this block's sequence -> ring slot s % D
    const long tile_seq = s / SLABS;
    const int kc = (int)(s % SLABS);
    long tile = (long)blockIdx.x + tile_seq * gridDim.x;
    tile %= rows_total / ROWS_A;                              // the three column passes of the N = 960 projection re-read every row block
    if (src_window_tiles > 0) tile = (long)blockIdx.x % src_window_tiles + (tile_seq % 2) * src_window_tiles;   // SRC 1: a window that stays in L2
    const auto asrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A + tile * ROWS_A * KBYTES), 0, (unsigned)(ROWS_A * KBYTES), 0x00020000);
    char* dst = smem + (s % D) * SLOT + wave * 1024;
    // one instruction moves 64 lanes x 16 B = 1 KB into LDS
#pragma unroll
    for (int i = 0; i < NA; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(asrd, (lptr_t)(dst + i * 8192), 16, (int)aoff[i], kc * SEG, 0, 0);
#pragma unroll
    for (int i = 0; i < NW; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(dst + (NA + i) * 8192), 16, (int)woff[i], kc * SEG, 0, 0);
  };
  for (int d = 0; d < D && issued < total; ++d) issue(issued++);
  const uint4 v = make_uint4(tid, blockIdx.x, 3, 7);
  for (long s = 0; s < total; ++s) {
    // wait for the OLDEST slab in flight: at most (D - 1) * NI younger DMA instructions of this wave may remain
    // (vmcnt counts stores too and retires in issue order: ST == 2 adds the epilogue's 20 stores to the allowance for the D slabs that were already in flight
    //  when they were issued, so that the stores drain under the following fills instead of ahead of them; ST == 1 is the naive counted wait)
    if (ST == 2 && store_credit > 0) {
      --store_credit;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI + 20 < 63 ? (D - 1) * NI + 20 : 63));
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI < 63 ? (D - 1) * NI : 63));
    }
    __builtin_amdgcn_s_barrier();                          // (no fence: __syncthreads would drain vmcnt to 0 and with it the slabs in flight) every wave's pieces of the slab have landed (the consumer would read it here)
    if (issued < total) issue(issued++);                   // refill the slot
    if (ST && (s % SLABS) == SLABS - 1) {                  // the tile's epilogue: this wave's 64 rows x 160 columns, 16 rows x 64 B per instruction
      const long tile = (long)blockIdx.x + (s / SLABS) * gridDim.x, nrb = rows_total / ROWS_A;
      const long r0 = (tile % nrb) * 256 + (wave >> 1) * 64;
      const int c0 = (int)(tile / nrb) * 40 + (wave & 1) * 20, lane = tid & 63;      // column pass -> its 320 of the 960 output columns (120 uint4 per row)
      if (r0 + 64 <= rows_total && tile / nrb < 3) {
#pragma unroll
        for (int q = 0; q < 20; ++q) out[(r0 + (q / 5) * 16 + (lane & 15)) * 120 + c0 + (q % 5) * 4 + (lane >> 4)] = v;
        store_credit = D;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);
}

template <int D, int SEG, int ST>
void run(const char* A, const char* W, uint4* out, long rows, int src) {
  const int tiles = (int)(rows / ROWS_A / 256) * 3;            // 3 column passes, as the N = 960 projection makes them
  constexpr int LPR_ = SEG / 16;
  const size_t lds = (size_t)D * ((ROWS_A * LPR_ + 511) / 512 + (ROWS_W * LPR_ + 511) / 512) * 8192;
  if (lds > 160 * 1024) { printf("D %d SEG %3d: %zu KB of LDS -- skipped\n", D, SEG, lds >> 10); return; }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<D, SEG, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int window = src ? 64 : 0;                             // 64 tiles x 160 KB = 10 MB over 8 XCDs: L2-resident per XCD
  hipLaunchKernelGGL((fill_kernel<D, SEG, ST>), dim3(256), dim3(512), lds, 0, A, W, out, rows, tiles, window);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((fill_kernel<D, SEG, ST>), dim3(256), dim3(512), lds, 0, A, W, out, rows, tiles, window);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double bytes = (double)tiles * 256 * (ROWS_A + ROWS_W) * KBYTES;   // fills only
  printf("in flight %d x %5.1f KB  SEG %3d  A from %-3s  stores %d : %7.3f ms  fill %6.2f TB/s  %5.1f GB/s per CU  (%.1f us per tile)\n", D,
         (ROWS_A + ROWS_W) * SEG / 1024.0, SEG, src ? "L2" : "HBM", ST, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e3 / tiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  HIP error: %s\n", hipGetErrorString(e));
}

int main() {
  const long rows = 393216;
  char *A, *W;
  uint4* out;
  hipMalloc(&A, rows * KBYTES + (1 << 20));
  hipMalloc(&W, ROWS_W * KBYTES + 4096);
  hipMalloc(&out, rows * 3 * KBYTES + 4096);
  hipMemset(A, 1, rows * KBYTES);
  hipMemset(W, 2, ROWS_W * KBYTES);
  for (int src = 0; src < 2; ++src) {
    run<1, 128, 0>(A, W, out, rows, src);
    run<2, 128, 0>(A, W, out, rows, src);
    run<1, 64, 0>(A, W, out, rows, src);
    run<2, 64, 0>(A, W, out, rows, src);
    run<3, 64, 0>(A, W, out, rows, src);
    run<4, 64, 0>(A, W, out, rows, src);
  }
  run<1, 128, 1>(A, W, out, rows, 0);   // the real kernel's in-flight depth, with the epilogue's stores
  run<2, 128, 1>(A, W, out, rows, 0);
  run<2, 128, 2>(A, W, out, rows, 0);
  run<3, 64, 1>(A, W, out, rows, 0);
  run<3, 64, 2>(A, W, out, rows, 0);
  run<4, 64, 2>(A, W, out, rows, 0);
  return 0;
}
