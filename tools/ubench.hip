// Instruction-rate micro-benchmarks for gfx950 (standalone: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/_bin/ubench).
// Every test is a loop of 32 asm-volatile instructions per iteration; each wave brackets its loop with s_memtime and the
// host prints cycles per instruction per WAVE (min over waves) for 1, 2 and 3 waves per SIMD.  Used to budget the
// attention / GEMM inner loops; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R4(X) X(0) X(1) X(2) X(3)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

enum { T_MFMA16x32 = 0, T_MFMA16x16, T_MFMA32x16, T_MFMA32x8, T_EXP, T_FMA, T_PKFMA, T_MAX, T_MAX3, T_CVTPK, T_PERM32, T_PKMAXH, T_PKFMAH, T_EXPH,
       T_MIX1, T_MIX2, T_MIX4, T_MIX6, T_MIXE2, T_MIXE4, T_ROLE, T_DSR128, T_MIX32_8, T_MIX32_E8, T_COUNT };
static const char* NAMES[] = {"mfma_16x16x32_f16", "mfma_16x16x16_f16", "mfma_32x32x16_f16", "mfma_32x32x8_f16", "v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max_f32", "v_max3_f32",
                              "v_cvt_pk_f16_f32", "v_permlane32_swap", "v_pk_max_f16", "v_pk_fma_f16", "v_exp_f16",
                              "mix 1mfma16+1fma", "mix 1mfma16+2fma", "mix 1mfma16+4fma", "mix 1mfma16+6fma", "mix 1mfma16+2exp", "mix 1mfma16+4exp",
                              "roles: even waves mfma16, odd waves exp", "ds_read_b128", "mix 1mfma32+8fma", "mix 1mfma32+8exp"};

template <int T>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * (lane + i)); b[i] = (f16)(0.002f * (lane - i)); }
  f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f32x16 d0, d1, d2, d3;
  for (int i = 0; i < 16; ++i) d0[i] = d1[i] = d2[i] = d3[i] = 0.f;
  float x0 = 0.001f * lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
  unsigned u0 = lane, u1 = lane + 1, u2 = lane + 2, u3 = lane + 3, u4 = lane + 4, u5 = lane + 5, u6 = lane + 6, u7 = lane + 7;
  f32x4 l0, l1, l2, l3, l4, l5, l6, l7;
  const unsigned laddr = (unsigned)(size_t)(lds) + lane * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

#define MF16(c) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define MF16L(c) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a4), "v"(b4));
#define MF32(d) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
#define MF32L(d) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a4), "v"(b4));
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
#define PKFMA(p) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
#define MAXF(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(x7));
#define MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(x6), "v"(x7));
#define CVT(u, x) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(x7));
#define PERM(u, v) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
#define PKMAXH(u) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u) : "v"(u7));
#define PKFMAH(u) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(u));
#define EXPH(u) asm volatile("v_exp_f16 %0, %0" : "+v"(u));
#define DSR(l) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(laddr));

  const unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    if constexpr (T == T_MFMA16x32) { for (int r = 0; r < 4; ++r) { MF16(c0) MF16(c1) MF16(c2) MF16(c3) MF16(c4) MF16(c5) MF16(c6) MF16(c7) } }
    if constexpr (T == T_MFMA16x16) { for (int r = 0; r < 4; ++r) { MF16L(c0) MF16L(c1) MF16L(c2) MF16L(c3) MF16L(c4) MF16L(c5) MF16L(c6) MF16L(c7) } }
    if constexpr (T == T_MFMA32x16) { for (int r = 0; r < 8; ++r) { MF32(d0) MF32(d1) MF32(d2) MF32(d3) } }
    if constexpr (T == T_MFMA32x8) { for (int r = 0; r < 8; ++r) { MF32L(d0) MF32L(d1) MF32L(d2) MF32L(d3) } }
    if constexpr (T == T_EXP) { for (int r = 0; r < 4; ++r) { EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7) } }
    if constexpr (T == T_FMA) { for (int r = 0; r < 4; ++r) { FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) } }
    if constexpr (T == T_PKFMA) { for (int r = 0; r < 4; ++r) { PKFMA(p0) PKFMA(p1) PKFMA(p2) PKFMA(p3) PKFMA(p4) PKFMA(p5) PKFMA(p6) PKFMA(p7) } }
    if constexpr (T == T_MAX) { for (int r = 0; r < 4; ++r) { MAXF(x0) MAXF(x1) MAXF(x2) MAXF(x3) MAXF(x4) MAXF(x5) MAXF(x0) MAXF(x1) } }
    if constexpr (T == T_MAX3) { for (int r = 0; r < 4; ++r) { MAX3(x0) MAX3(x1) MAX3(x2) MAX3(x3) MAX3(x4) MAX3(x5) MAX3(x0) MAX3(x1) } }
    if constexpr (T == T_CVTPK) { for (int r = 0; r < 4; ++r) { CVT(u0, x0) CVT(u1, x1) CVT(u2, x2) CVT(u3, x3) CVT(u4, x4) CVT(u5, x5) CVT(u6, x6) CVT(u7, x0) } }
    if constexpr (T == T_PERM32) { for (int r = 0; r < 8; ++r) { PERM(u0, u1) PERM(u2, u3) PERM(u4, u5) PERM(u6, u7) } }
    if constexpr (T == T_PKMAXH) { for (int r = 0; r < 4; ++r) { PKMAXH(u0) PKMAXH(u1) PKMAXH(u2) PKMAXH(u3) PKMAXH(u4) PKMAXH(u5) PKMAXH(u6) PKMAXH(u0) } }
    if constexpr (T == T_PKFMAH) { for (int r = 0; r < 4; ++r) { PKFMAH(u0) PKFMAH(u1) PKFMAH(u2) PKFMAH(u3) PKFMAH(u4) PKFMAH(u5) PKFMAH(u6) PKFMAH(u7) } }
    if constexpr (T == T_EXPH) { for (int r = 0; r < 4; ++r) { EXPH(u0) EXPH(u1) EXPH(u2) EXPH(u3) EXPH(u4) EXPH(u5) EXPH(u6) EXPH(u7) } }
    // in-wave mixes: 8 MFMAs per iteration, each followed by n VALU ops (independent registers)
    if constexpr (T == T_MIX1) { MF16(c0) FMA(x0) MF16(c1) FMA(x1) MF16(c2) FMA(x2) MF16(c3) FMA(x3) MF16(c4) FMA(x4) MF16(c5) FMA(x5) MF16(c6) FMA(x6) MF16(c7) FMA(x7) }
    if constexpr (T == T_MIX2) { MF16(c0) FMA(x0) FMA(x1) MF16(c1) FMA(x2) FMA(x3) MF16(c2) FMA(x4) FMA(x5) MF16(c3) FMA(x6) FMA(x7) MF16(c4) FMA(x0) FMA(x1) MF16(c5) FMA(x2) FMA(x3) MF16(c6) FMA(x4) FMA(x5) MF16(c7) FMA(x6) FMA(x7) }
    if constexpr (T == T_MIX4) { MF16(c0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c1) FMA(x4) FMA(x5) FMA(x6) FMA(x7) MF16(c2) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c3) FMA(x4) FMA(x5) FMA(x6) FMA(x7)
                                 MF16(c4) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c5) FMA(x4) FMA(x5) FMA(x6) FMA(x7) MF16(c6) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c7) FMA(x4) FMA(x5) FMA(x6) FMA(x7) }
    if constexpr (T == T_MIX6) { MF16(c0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) MF16(c1) FMA(x6) FMA(x7) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c2) FMA(x4) FMA(x5) FMA(x6) FMA(x7) FMA(x0) FMA(x1) MF16(c3) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7)
                                 MF16(c4) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) MF16(c5) FMA(x6) FMA(x7) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF16(c6) FMA(x4) FMA(x5) FMA(x6) FMA(x7) FMA(x0) FMA(x1) MF16(c7) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) }
    if constexpr (T == T_MIXE2) { MF16(c0) EXP(x0) EXP(x1) MF16(c1) EXP(x2) EXP(x3) MF16(c2) EXP(x4) EXP(x5) MF16(c3) EXP(x6) EXP(x7) MF16(c4) EXP(x0) EXP(x1) MF16(c5) EXP(x2) EXP(x3) MF16(c6) EXP(x4) EXP(x5) MF16(c7) EXP(x6) EXP(x7) }
    if constexpr (T == T_MIXE4) { MF16(c0) EXP(x0) EXP(x1) EXP(x2) EXP(x3) MF16(c1) EXP(x4) EXP(x5) EXP(x6) EXP(x7) MF16(c2) EXP(x0) EXP(x1) EXP(x2) EXP(x3) MF16(c3) EXP(x4) EXP(x5) EXP(x6) EXP(x7)
                                  MF16(c4) EXP(x0) EXP(x1) EXP(x2) EXP(x3) MF16(c5) EXP(x4) EXP(x5) EXP(x6) EXP(x7) MF16(c6) EXP(x0) EXP(x1) EXP(x2) EXP(x3) MF16(c7) EXP(x4) EXP(x5) EXP(x6) EXP(x7) }
    if constexpr (T == T_ROLE) {
      if (wave & 1) { for (int r = 0; r < 4; ++r) { EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7) } }
      else { for (int r = 0; r < 4; ++r) { MF16(c0) MF16(c1) MF16(c2) MF16(c3) MF16(c4) MF16(c5) MF16(c6) MF16(c7) } }
    }
    if constexpr (T == T_DSR128) { for (int r = 0; r < 4; ++r) { DSR(l0) DSR(l1) DSR(l2) DSR(l3) DSR(l4) DSR(l5) DSR(l6) DSR(l7) } asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if constexpr (T == T_MIX32_8) { MF32(d0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) MF32(d1) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7)
                                    MF32(d2) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) MF32(d3) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) }
    if constexpr (T == T_MIX32_E8) { MF32(d0) EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7) MF32(d1) EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7)
                                     MF32(d2) EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7) MF32(d3) EXP(x0) EXP(x1) EXP(x2) EXP(x3) EXP(x4) EXP(x5) EXP(x6) EXP(x7) }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = now();
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1] + c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] +
            d0[0] + d1[5] + d2[9] + d3[15] + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
  if constexpr (T == T_DSR128) s += l0[0] + l1[1] + l2[2] + l3[3] + l4[0] + l5[1] + l6[2] + l7[3];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
void run(int wps, unsigned long long* dout, float* sink) {
  const int iters = 2000, blocks = 256, thr = 256 * wps;
  const int nw = blocks * wps * 4;
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(thr), 0, 0, dout, sink, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(thr), 0, 0, dout, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nw);
  hipMemcpy(h.data(), dout, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  // instructions per iteration (per wave)
  int per = 32;
  if (T == T_MIX1) per = 16; if (T == T_MIX2) per = 24; if (T == T_MIX4) per = 40; if (T == T_MIX6) per = 56; if (T == T_MIXE2) per = 24; if (T == T_MIXE4) per = 40;
  if (T == T_MIX32_8 || T == T_MIX32_E8) per = 36;
  const double tot = (double)iters * per;
  // s_memtime ticks at a constant 100 MHz on gfx9: convert with the measured wall time of the kernel instead
  printf("%-42s wps=%d  memtime ticks/instr: min %.3f med %.3f   wall %.3f ms -> %.2f ns/instr/wave, %.2f ns/instr/SIMD\n", NAMES[T], wps, h[0] / tot, h[nw / 2] / tot, ms,
         ms * 1e6 / tot, ms * 1e6 / tot / wps);
}

template <int T>
void run_all(unsigned long long* dout, float* sink) { run<T>(1, dout, sink); run<T>(2, dout, sink); run<T>(3, dout, sink); }

int main() {
  unsigned long long* dout; float* sink;
  hipMalloc(&dout, 1 << 20); hipMalloc(&sink, 64);
  run_all<T_MFMA16x32>(dout, sink); run_all<T_MFMA16x16>(dout, sink); run_all<T_MFMA32x16>(dout, sink); run_all<T_MFMA32x8>(dout, sink);
  run_all<T_EXP>(dout, sink); run_all<T_FMA>(dout, sink); run_all<T_PKFMA>(dout, sink); run_all<T_MAX>(dout, sink); run_all<T_MAX3>(dout, sink); run_all<T_CVTPK>(dout, sink);
  run_all<T_PERM32>(dout, sink); run_all<T_PKMAXH>(dout, sink); run_all<T_PKFMAH>(dout, sink); run_all<T_EXPH>(dout, sink);
  run_all<T_MIX1>(dout, sink); run_all<T_MIX2>(dout, sink); run_all<T_MIX4>(dout, sink); run_all<T_MIX6>(dout, sink); run_all<T_MIXE2>(dout, sink); run_all<T_MIXE4>(dout, sink);
  run<T_ROLE>(2, dout, sink); run<T_DSR128>(1, dout, sink); run<T_DSR128>(2, dout, sink);
  run_all<T_MIX32_8>(dout, sink); run_all<T_MIX32_E8>(dout, sink);
  return 0;
}
