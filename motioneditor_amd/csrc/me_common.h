// Shared device-side helpers for libmotioned (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- launch hook (csrc/plan.hip) -----------------------------------------------------------------------------------
// Every kernel of the library is launched through me_launch(): it converts the call's arguments to the kernel's own
// parameter types, launches with hipLaunchKernel (what `kernel<<<...>>>(...)` compiles to) and -- while this thread is
// recording a plan (me_plan_begin ... me_plan_end, include/motioned.h) -- appends {kernel, grid, block, LDS bytes, stream,
// a copy of the argument bytes} to that plan, so that me_denoise_step() can re-issue the whole step from C.
#ifndef __HIP_DEVICE_COMPILE__
#include <tuple>
#include <utility>
#endif
extern "C" int me_plan_recording(void);
extern "C" void me_plan_append_launch(const void* fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned lds_bytes, void* stream,
                                      void* const* args, const unsigned* arg_bytes, int n_args);
#ifndef __HIP_DEVICE_COMPILE__
template <typename... P, size_t... I>
inline void me_launch_impl(void (*k)(P...), dim3 g, dim3 b, unsigned lds, hipStream_t st, std::tuple<P...>& v, std::index_sequence<I...>) {
  void* ptrs[sizeof...(P) + 1] = {static_cast<void*>(&std::get<I>(v))..., nullptr};
  (void)hipLaunchKernel(reinterpret_cast<const void*>(k), g, b, ptrs, lds, st);   // an error stays in hipGetLastError() for the entry point's check
  if (me_plan_recording()) {
    static const unsigned sizes[sizeof...(P) + 1] = {(unsigned)sizeof(P)..., 0u};
    me_plan_append_launch(reinterpret_cast<const void*>(k), g.x, g.y, g.z, b.x, b.y, b.z, lds, st, ptrs, sizes, (int)sizeof...(P));
  }
}
template <typename... P, typename... A>
inline void me_launch(void (*k)(P...), dim3 g, dim3 b, size_t lds, hipStream_t st, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "me_launch: argument count differs from the kernel's parameter count");
  std::tuple<P...> v(std::forward<A>(a)...);
  me_launch_impl(k, g, b, (unsigned)lds, st, v, std::index_sequence_for<P...>{});
}
#else
template <typename K, typename... A>
inline void me_launch(K, dim3, dim3, size_t, hipStream_t, A&&...) {}
#endif
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) me_launch(kernel, grid, block, lds, stream, __VA_ARGS__)

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ME_WAVE 64

// v_mfma_f32_16x16x32_f16: D(16x16) += A(16x32) * B(32x16).
//   A operand: lane l holds A[i = l & 15][k = (l >> 4) * 8 + 0..7]
//   B operand: lane l holds B[k = (l >> 4) * 8 + 0..7][n = l & 15]
//   C/D      : lane l, reg r holds D[i = (l >> 4) * 4 + r][n = l & 15]
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

union U128 {
  uint4 u;
  f16x8 h;
  f16 e[8];
};

union U64 {
  uint2 u;
  f16x4 h;
  f16 e[4];
};

__device__ __forceinline__ uint4 ldg128(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 zero128() { return make_uint4(0u, 0u, 0u, 0u); }

// x * sigmoid(x) with hardware exp2 / rcp (1 ulp each, far below fp16 resolution) instead of the IEEE division sequence
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// exact-erf GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below fp16 resolution):
// ~14 VALU ops instead of libm erff's ~40 -- the GEGLU epilogue runs it on every feed-forward activation.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erf_abs = 1.0f - p * t * e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// GELU(x) = x Phi(x) with the normal CDF written as a logistic of an odd polynomial: Phi(x) = 1 / (1 + 2^(x Q(x^2))), Q of
// degree 4 fitted (iteratively re-weighted least squares ~ minimax over |x| <= 10) to x Phi(x) with erf from libm:
// |error| <= 3.5e-6 absolute everywhere (evaluated in fp32), <= 2.1e-4 relative where |gelu| > 0.01 -- below fp16's half-ulp
// (2.4e-4), and the result is rounded to fp16 anyway.  12 VALU slots (4 fma, 3 mul, 1 add, exp2, rcp) instead of ~19 for
// the erf form above: the GEGLU epilogue runs it on every feed-forward activation (1/3 of those GEMMs' time at K = 320).
// Large |x|: 2^(+big) = inf -> rcp -> 0 -> x * 0 = 0 (x << 0);  2^(-big) = 0 -> x (x >> 0).
__device__ __forceinline__ float gelu_sigpoly(float x) {
  const float x2 = x * x;
  float q = __builtin_fmaf(-3.2291018214891665e-06f, x2, 8.824012184049934e-05f);
  q = __builtin_fmaf(q, x2, 0.00036026412271894515f);
  q = __builtin_fmaf(q, x2, -0.10522667318582535f);
  q = __builtin_fmaf(q, x2, -2.3020453453063965f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q * x));
}

// a * gelu_sigpoly(g) on two values per lane with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: the polynomial's 4 FMAs, 3 multiplications and the
// addition, and the gate's product, at half the issue slots; the two exp2 / rcp stay scalar).  Element for element the same operations in the same order
// as the scalar form -- IEEE fma / mul per element -- so the results are bitwise those of a * gelu_sigpoly(g).  ME_GELU_PK=0 builds the scalar form (A/B).
#ifndef ME_GELU_PK
#define ME_GELU_PK 1
#endif
__device__ __forceinline__ f32x2 geglu2(f32x2 av, f32x2 x) {
#if ME_GELU_PK
  const f32x2 x2 = x * x;
  f32x2 q = __builtin_elementwise_fma(f32x2{-3.2291018214891665e-06f, -3.2291018214891665e-06f}, x2, f32x2{8.824012184049934e-05f, 8.824012184049934e-05f});
  q = __builtin_elementwise_fma(q, x2, f32x2{0.00036026412271894515f, 0.00036026412271894515f});
  q = __builtin_elementwise_fma(q, x2, f32x2{-0.10522667318582535f, -0.10522667318582535f});
  q = __builtin_elementwise_fma(q, x2, f32x2{-2.3020453453063965f, -2.3020453453063965f});
  const f32x2 t = q * x;
  const f32x2 d = f32x2{1.0f, 1.0f} + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  const f32x2 g = x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return av * g;
#else
  return f32x2{av[0] * gelu_sigpoly(x[0]), av[1] * gelu_sigpoly(x[1])};
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Cross-lane exchange with lane ^ 32 / lane ^ 16 as VALU permlane swaps (gfx950) instead of ds_bpermute
// (an LDS round trip): v_permlane32_swap(x, x) leaves {x[l], x[l^32]} in the two results for every lane,
// v_permlane16_swap(x, x) the same for 16-lane rows.
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Bijective XCD-aware remap of a 1-D block id: hardware places block b on XCD b % 8; give each
// XCD a contiguous run of work items so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
