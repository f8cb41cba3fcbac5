// Epilogue experiment on the ring-GEMM model (tools/ubench_ring_gemm.hip, D-slot ring, persistent walk): WHY do the 256 x 320 tile's output stores
// (164 KB per tile) not overlap with the next tile's main loop?  Round 5, first GPU call: ring depth 2 / 3 / 4 is flat (0.49 / 0.46 / 0.46 ms for
// [393216 x 960] K = 320), stores off 0.24 ms -- the fill is NOT latency-bound, the stores cost as much as the whole main loop and add serially.
// Variants (EPI):
//   0  no stores                      1  plain 8-byte stores (16 rows x 32 B per instruction), the model's baseline
//   2  the same stores aliased into an L2-resident window (no HBM write traffic): HBM-write-bound or issue-bound?
//   3  non-temporal stores            4  plain stores, every second CU's block starts half a tile late (are the CUs' store bursts in lockstep?)
//   5  wave-private LDS transposition, 16-byte row-contiguous stores (160 B per row and wave: 6.4 rows per instruction), ring depth 3 + 40 KB scratch
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_bin/ubench_epi tools/ubench_epi.hip && tools/_bin/ubench_epi
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 320, BK = 32;
constexpr int MT = 8, NT = 5;                         // MFMA tiles per wave: 128 rows x 80 columns
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;                             // 16 KB + 20 KB
constexpr int NIA = A_BYTES / 8192, NIW = (W_BYTES + 8191) / 8192, NI = NIA + NIW;    // DMA instructions per wave and slab: 2 + 3 (the last one half masked)
// every wave issues all NI instructions (one vmcnt arithmetic for all waves); the masked lanes of the last W instruction still WRITE zeros to LDS, so a slot
// is padded to whole instructions: 16 + 24 = 40 KB, four slots = the CU's 160 KB exactly
constexpr int SLOT = A_BYTES + NIW * 8192;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ int xcd_remap(int bid, int total) {   // as me_common.h: each XCD (block id % 8) gets a contiguous run of work items
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// vmcnt counts loads AND stores on gfx9-family parts and retires them in issue order: "slab q has landed" = at most `younger` DMA slabs plus -- for the
// D - 1 slabs that were already in flight when a tile's epilogue issued its MT * NT stores -- those stores may still be outstanding.  Without the store
// credit the first waits of the next tile would drain the whole epilogue before the MFMAs restart (what a persistent walk is supposed to avoid).
template <int REM, int STORES>
__device__ __forceinline__ void wait_younger() {
  static_assert(REM * NI + STORES <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(REM * NI + STORES));
}

template <int D, int PERSIST, int STORE>
__global__ __launch_bounds__(512) void ring_gemm(const f16* __restrict__ X, const f16* __restrict__ W, f16* __restrict__ Y, int M, int N, int K, int epi, int ywin, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbn = N / BN, ntiles = (M / BM) * nbn, S = K / BK;
  const int my_tiles = PERSIST ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 1;
  const long total = (long)my_tiles * S;

  // DMA source offsets (bytes inside the tile's A / W row block): LDS piece p = j * 64 + lane of a part holds (row p >> 2, k-chunk (p & 3) ^ ((row >> 1) & 2))
  if (stagger && ((blockIdx.x >> 3) & 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  unsigned aoff[NIA], woff[NIW];
#pragma unroll
  for (int t = 0; t < NIA; ++t) {
    const int p = (wave + 8 * t) * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 1) & 2);
    aoff[t] = (unsigned)(row * K * 2 + c * 16);
  }
#pragma unroll
  for (int t = 0; t < NIW; ++t) {
    const int p = (wave + 8 * t) * 64 + lane, row = p >> 2, c = (p & 3) ^ ((row >> 1) & 2);
    woff[t] = row < BN ? (unsigned)(row * K * 2 + c * 16) : OOB;
  }
  auto tile_of = [&](int seq) {    // the seq-th tile of this block (wave-uniform: forced into an SGPR so that the descriptors below stay scalar)
    return __builtin_amdgcn_readfirstlane(PERSIST ? xcd_remap((int)blockIdx.x + seq * (int)gridDim.x, ntiles) : xcd_remap((int)blockIdx.x, ntiles));
  };
  int iss_seq = 0, iss_kc = 0, iss_slot = 0;   // the next slab to issue: tile of the sequence, slab of the tile, ring slot (no divisions in the loop)
  auto issue = [&]() {
    const int w = tile_of(iss_seq);
    const int tile_n = w % nbn, tile_m = w / nbn;
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(X + (long)tile_m * BM * K), 0, (unsigned)(BM * K * 2), 0x00020000);
    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W + (long)tile_n * BN * K), 0, (unsigned)(BN * K * 2), 0x00020000);
    char* dst = smem + iss_slot * SLOT + wave * 1024;
#pragma unroll
    for (int t = 0; t < NIA; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(dst + t * 8192), 16, (int)aoff[t], iss_kc * BK * 2, 0, 0);
#pragma unroll
    for (int t = 0; t < NIW; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(dst + A_BYTES + t * 8192), 16, (int)woff[t], iss_kc * BK * 2, 0, 0);
    if (++iss_kc == S) { iss_kc = 0; ++iss_seq; }
    if (++iss_slot == D) iss_slot = 0;
  };

  // fragment addresses inside a slot: row (lane & 15) of the 16-row block, k-chunk (lane >> 4) at slot (lane >> 4) ^ ((row >> 1) & 2); block offsets are
  // multiples of 16 rows, so the swizzle term depends on the lane alone
  const int frow = lane & 15, fslot = (lane >> 4) ^ ((frow >> 1) & 2);
  const int xbase = ((wm * 128 + frow) * 4 + fslot) * 16, wbase = A_BYTES + ((wn * 80 + frow) * 4 + fslot) * 16;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NST = STORE == 2 ? 20 : MT * NT;   // store instructions per wave and tile (vmcnt credit)
  int issued = 0, q_seq = 0, q_kc = 0, q_slot = 0, store_credit = 0;
  const int total_i = (int)total;
  for (int d = 0; d < D - 1 && issued < total_i; ++d, ++issued) issue();
  for (int q = 0; q < total_i; ++q) {
    // slab q must have landed; slabs q + 1 .. issued - 1 (at most D - 2 of them) may stay in flight
    const int younger = issued - 1 - q;
    if (STORE && store_credit > 0) {       // slab q was issued BEFORE the last epilogue's stores: they are younger than it, too
      --store_credit;
      if (D >= 4 && younger >= 2) wait_younger<2, NST>();
      else if (D >= 3 && younger >= 1) wait_younger<1, NST>();
      else wait_younger<0, NST>();
    } else {
      if (D >= 4 && younger >= 2) wait_younger<2, 0>();
      else if (D >= 3 && younger >= 1) wait_younger<1, 0>();
      else wait_younger<0, 0>();
    }
    __builtin_amdgcn_s_barrier();          // all waves' pieces of slab q are in LDS, and every wave is done reading the slot of slab q - 1
    if (issued < total_i) { issue(); ++issued; }   // ... which slab q + D - 1 now refills
    const char* sl = smem + q_slot * SLOT;
    f16x8 fx[MT], fw[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fx[i] = *reinterpret_cast<const f16x8*>(sl + xbase + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < NT; ++j) fw[j] = *reinterpret_cast<const f16x8*>(sl + wbase + j * 16 * 64);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fx[i], acc[j][i], 0, 0, 0);
    if (++q_slot == D) q_slot = 0;
    const bool last = q_kc == S - 1;
    if (++q_kc == S) q_kc = 0;
    if (last) {                            // the tile is complete: lane holds Y[m0 + (lane & 15)][n0 + (lane >> 4) * 4 + 0..3] of every (j, i) MFMA tile
      const int w = tile_of(q_seq++);
      const int tile_n = w % nbn, tile_m = w / nbn;
      long m0 = (long)tile_m * BM + wm * 128 + (lane & 15);
      if (ywin) m0 = ((long)tile_m * BM) % ywin + wm * 128 + (lane & 15);      // EPI 2: the stores stay inside an L2-resident window of ywin rows
      const int n0 = tile_n * BN + wn * 80 + (lane >> 4) * 4;
      if (epi == 5) {
        // wave-private scratch behind the ring: 32 rows x 160 B (two 16-row MFMA row blocks) -> 5 row-contiguous 16-byte store instructions
        char* scr = smem + D * SLOT + wave * 5120;
        const long mw = (ywin ? ((long)tile_m * BM) % ywin : (long)tile_m * BM) + wm * 128;
#pragma unroll
        for (int i = 0; i < MT; i += 2) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const f32x4 a = acc[j][i + ii];
              const f16x4 h = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
              *reinterpret_cast<f16x4*>(scr + (ii * 16 + (lane & 15)) * 160 + (j * 16 + (lane >> 4) * 4) * 2) = h;
              acc[j][i + ii] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const int p = t * 64 + lane, r = p / 10, c = p % 10;       // 320 pieces of 16 B = 32 rows x 10
            const uint4 v = *reinterpret_cast<const uint4*>(scr + r * 160 + c * 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Y + (mw + i * 16 + r) * N + tile_n * BN + wn * 80) + c * 16) = v;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
        }
      } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 a = acc[j][i];
          const f16x4 h = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
          f16x4* dst = reinterpret_cast<f16x4*>(Y + (m0 + i * 16) * N + n0 + j * 16);
          if (epi == 3) __builtin_nontemporal_store(h, dst);
          else if (STORE || a[0] == 1.2345e30f) *dst = h;
          acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      store_credit = issued - 1 - q;       // the slabs in flight right now were issued before these stores: their waits carry the credit
    }
  }
}

static float h2f(f16 h) { return (float)h; }

template <int D, int PERSIST, int STORE>
void run(const f16* X, const f16* W, f16* Y, int M, int N, int K, const std::vector<f16>& hx, const std::vector<f16>& hw, int epi, int ywin = 0, int stagger = 0) {
  const size_t lds = (size_t)D * SLOT + (epi == 5 ? 8 * 5120 : 0);
  const int ntiles = (M / BM) * (N / BN);
  const int grid = PERSIST ? 256 : ntiles;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_gemm<D, PERSIST, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipMemset(Y, 0, (size_t)M * N * 2);
  hipLaunchKernelGGL((ring_gemm<D, PERSIST, STORE>), dim3(grid), dim3(512), lds, 0, X, W, Y, M, N, K, epi, ywin, stagger);
  (void)hipDeviceSynchronize();
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("D %d persist %d: HIP error %s\n", D, PERSIST, hipGetErrorString(e)); return; }
  double worst = 0.0;
  if (STORE && !ywin) {   // sample check against a host dot product of the same fp16 inputs
    std::vector<f16> row(N);
    for (int s = 0; s < 24; ++s) {
      const long m = ((long)s * 1000003L + 17) % M;
      (void)hipMemcpy(row.data(), Y + m * N, (size_t)N * 2, hipMemcpyDeviceToHost);
      for (int n = (s * 7) % 13; n < N; n += 29) {
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)h2f(hx[(size_t)m * K + k]) * (double)h2f(hw[(size_t)n * K + k]);
        const double err = fabs((double)h2f(row[n]) - ref) / (fabs(ref) + 1.0);
        if (err > worst) worst = err;
      }
    }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((ring_gemm<D, PERSIST, STORE>), dim3(grid), dim3(512), lds, 0, X, W, Y, M, N, K, epi, ywin, stagger);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  static const char* names[] = {"no stores", "plain 8 B", "L2 window", "nt 8 B", "staggered", "LDS 16 B rows"};
  printf("M %d N %4d K %d  ring %d x %.1f KB  %s  epi %d %-13s stagger %d : %7.3f ms  %7.1f TF/s  %5.1f us per tile%s\n", M, N, K, D, SLOT / 1024.0,
         PERSIST ? "persistent " : "one tile/blk", epi, names[epi], stagger, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / ((double)ntiles / 256.0),
         (STORE && !ywin) ? (worst < 2e-2 ? "  [check ok]" : "  [CHECK FAILED]") : "");
  if (STORE && worst >= 2e-2) printf("   worst relative error of the sample: %.3e\n", worst);
}

int main() {
  const int M = 393216, K = 320, NMAX = 2560;
  std::vector<f16> hx((size_t)M * K), hw((size_t)NMAX * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0x7fff) / 32768.0f - 0.5f; };
  for (auto& v : hx) v = (f16)rnd();
  for (auto& v : hw) v = (f16)(rnd() * 0.2f);
  f16 *X, *W, *Y;
  (void)hipMalloc(&X, hx.size() * 2);
  (void)hipMalloc(&W, hw.size() * 2);
  (void)hipMalloc(&Y, (size_t)M * NMAX * 2);
  (void)hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  for (int N : {960, 320}) {
    run<4, 1, 0>(X, W, Y, M, N, K, hx, hw, 0);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 1);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 2, 4096);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 2, 32768);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 3);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 1, 0, 1);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 1, 0, 2);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 1, 0, 3);
    run<4, 1, 1>(X, W, Y, M, N, K, hx, hw, 1, 0, 5);
    run<3, 1, 1>(X, W, Y, M, N, K, hx, hw, 1);
    run<3, 1, 2>(X, W, Y, M, N, K, hx, hw, 5);
    run<3, 1, 2>(X, W, Y, M, N, K, hx, hw, 5, 4096);
  }
  return 0;
}
