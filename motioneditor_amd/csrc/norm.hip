// GroupNorm(+SiLU) and LayerNorm on channels-last fp16 activations (HBM-bound kernels, gfx950).
//
// GroupNorm: pass 1 accumulates (sum, sum of squares) per (row-group, channel-group) with 16-byte
// loads and register accumulation down the rows, reduces them in a fixed order (no atomics: bitwise
// reproducible) into fp64 statistics; pass 2 re-reads X, normalises, applies gamma/beta (+SiLU), writes Y.
// Algorithmic bytes per call: rows*C*2 (read) + rows*C*2 (write); the statistics pass re-reads X.
// LayerNorm: one wave per row, two-pass mean/variance in registers.
#include "me_common.h"
#include "../../include/motioned.h"

#ifndef ME_GN_APPLY_DEPTH
#define ME_GN_APPLY_DEPTH 4   // 16-byte loads in flight per thread of the apply pass
#endif
#ifndef ME_GN_UNROLL
#define ME_GN_UNROLL 4   // independent 16-byte loads in flight per thread of the statistics pass (A/B: tools/build_abl.sh)
#endif

namespace {

// Pass 1, deterministic and cancellation-free: every block reduces its row chunk to one (S1, S2) pair per channel group,
// S1 = sum(x - K), S2 = sum((x - K)^2) with the shift K = the group's first element of the chunk (so the sums stay small
// however large the group mean is: fp32 partial sums of raw x^2 lose the variance of SD-like activations whose mean is 1e2 -
// 1e3 times their spread), in a FIXED summation order -- per-thread register sums down the rows, a fixed-order sum over the
// block's row lanes per channel, a fixed-order sum over the group's channels -- and writes (S1, S2, K) to
// part[chunk][sample-group][group].  gn_finalize_kernel merges the chunks in index order in fp64 (Chan's update of
// (count, mean, M2)).  Round 1 used LDS / global float atomics on raw sums: run-to-run differences of 1.2e-3 rel-L2.
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ X, float4* __restrict__ part, int rows_per_group,
                                                       int chunk_rows, int C, int ldx, int groups) {
  extern __shared__ __attribute__((aligned(16))) float2 red[];   // [row lane][channel of the pass]
  __shared__ float2 chan[2560];                                    // per-channel sums of the block (C <= 2560: the widest skip concat)
  const int tid = threadIdx.x;
  const int sg = blockIdx.y;
  const int r0 = blockIdx.x * chunk_rows;
  const int r1 = min(r0 + chunk_rows, rows_per_group);
  const int tpr = C / 8;                       // 16-byte vectors per row
  const int tprc = tpr < 256 ? tpr : 256;      // vector columns handled per pass
  const int rl = tid / tprc, vc0 = tid - rl * tprc;
  const int RL = 256 / tprc;                   // row lanes
  const int cg = C / groups;
  const f16* base = X + (long)sg * rows_per_group * ldx;
  const f16* first = base + (long)r0 * ldx;    // the chunk's first row: source of the shifts

  for (int pass0 = 0; pass0 < tpr; pass0 += tprc) {
    const int vc = pass0 + vc0;
    if (rl < RL && vc < tpr) {
      float s[8], q[8], K[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] = 0.f;
        q[e] = 0.f;
        K[e] = (float)first[((vc * 8 + e) / cg) * cg];
      }
      // ME_GN_UNROLL independent 16-byte loads in flight per thread, issued explicitly ahead of their use (left to `#pragma unroll`, hipcc guards every
      // unrolled iteration with its own exit test and waits for each load by itself: one load in flight per thread); the sums stay in row order
      auto acc = [&](const U128& u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (float)u.e[e] - K[e];
          s[e] += v;
          q[e] += v * v;
        }
      };
      int r = r0 + rl;
      for (; r + (ME_GN_UNROLL - 1) * RL < r1; r += ME_GN_UNROLL * RL) {
        U128 u[ME_GN_UNROLL];
#pragma unroll
        for (int k = 0; k < ME_GN_UNROLL; ++k) u[k].u = ldg128(base + (long)(r + k * RL) * ldx + vc * 8);
        __builtin_amdgcn_sched_barrier(0);   // keep every load of the batch ahead of the first use (the scheduler otherwise sinks some of them between the sums)
#pragma unroll
        for (int k = 0; k < ME_GN_UNROLL; ++k) acc(u[k]);
      }
      for (; r < r1; r += RL) {
        U128 u;
        u.u = ldg128(base + (long)r * ldx + vc * 8);
        acc(u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rl * (tprc * 8) + vc0 * 8 + e] = make_float2(s[e], q[e]);
    }
    __syncthreads();
    for (int cidx = tid; cidx < tprc * 8 && pass0 * 8 + cidx < C; cidx += 256) {   // fixed order over the row lanes
      float2 a = red[cidx];
      for (int l = 1; l < RL; ++l) {
        const float2 b = red[l * (tprc * 8) + cidx];
        a.x += b.x;
        a.y += b.y;
      }
      chan[pass0 * 8 + cidx] = a;
    }
    __syncthreads();
  }
  if (tid < groups) {                                                               // fixed order over the group's channels
    float2 a = chan[tid * cg];
    for (int cc = 1; cc < cg; ++cc) {
      const float2 b = chan[tid * cg + cc];
      a.x += b.x;
      a.y += b.y;
    }
    part[((long)blockIdx.x * gridDim.y + sg) * groups + tid] = make_float4(a.x, a.y, (float)first[tid * cg], 0.f);
  }
}

// stats[sg][g] = (sum, sum of squares) in fp64 from the chunks' shifted partial sums: chunk mean = K + S1 / n, chunk
// M2 = S2 - S1^2 / n, merged with Chan's update of (n, mean, M2) in a FIXED tree (bitwise reproducible).
struct Moments { double n, mean, M2; };
__device__ __forceinline__ void merge(Moments& a, double nb, double mb, double M2b) {
  const double delta = mb - a.mean, nt = a.n + nb;
  a.mean += delta * (nb / nt);
  a.M2 += M2b + delta * delta * (a.n * nb / nt);
  a.n = nt;
}
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float4* __restrict__ part, double* __restrict__ stats, int chunks, int chunk_rows,
                                                          int rows_per_group, int nsg, int groups, int cg) {
  // one block per (sample-group, channel group): thread t folds chunks t, t + 256, ... in index order, then a fixed binary tree over the 256
  // threads (the serial walk of the first version -- 8 slices x chunks / 8 dependent fp64 updates with two divisions each -- took 10 us per
  // launch, 88 launches per step)
  __shared__ Moments sm[256];
  const int sg = blockIdx.x / groups, g = blockIdx.x - sg * groups, t = threadIdx.x;
  Moments m{0.0, 0.0, 0.0};
  for (int c = t; c < chunks; c += 256) {
    const float4 p = part[((long)c * nsg + sg) * groups + g];
    const int rows_c = min((c + 1) * chunk_rows, rows_per_group) - c * chunk_rows;
    const double nb = (double)rows_c * (double)cg;
    const double mb = (double)p.z + (double)p.x / nb, M2b = fmax((double)p.y - (double)p.x * (double)p.x / nb, 0.0);
    if (m.n == 0.0) m = Moments{nb, mb, M2b};
    else merge(m, nb, mb, M2b);
  }
  sm[t] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      const Moments o = sm[t + off];
      if (o.n > 0.0) {
        if (m.n == 0.0) m = o;
        else merge(m, o.n, o.mean, o.M2);
      }
      sm[t] = m;
    }
    __syncthreads();
  }
  if (t == 0) {
    stats[((long)sg * groups + g) * 2 + 0] = m.n * m.mean;
    stats[((long)sg * groups + g) * 2 + 1] = m.M2 + m.n * m.mean * m.mean;
  }
}

// y = x * A[c] + B[c] (+ SiLU) with A = rstd * gamma, B = beta - mean * rstd * gamma.  A thread owns ONE 16-byte
// channel vector and walks down the rows of its block's chunk, so the per-channel scale / shift live in 16
// registers and the inner loop is load - 8 FMA - store (the first version re-derived row, group, mean and rstd with
// integer divisions and an rsqrt for every vector and ran at 2 TB/s).  blockDim.x = vectors per row handled by the
// block (a divisor of C/8, <= 256), blockDim.y rows in flight: a wave covers whole contiguous row segments.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* X, f16* Y, const double* __restrict__ stats,
                                                       const f16* __restrict__ gamma, const f16* __restrict__ beta, long rows,
                                                       int rows_per_group, long rows_per_group_total, int C, int ldx, int ldy, int groups,
                                                       float eps, int silu, int chunk) {
  const int vc = blockIdx.y * blockDim.x + threadIdx.x;   // 16-byte vector column
  const int cg = C / groups;
  const double inv_cnt = 1.0 / ((double)rows_per_group_total * (double)cg);   // global count when frame-sharded
  U128 gm, bt;
  gm.u = ldg128(gamma + vc * 8);
  bt.u = ldg128(beta + vc * 8);
  float A[8], B[8];
  const long r0 = (long)blockIdx.x * chunk;
  const long r1 = r0 + chunk < rows ? r0 + chunk : rows;
  long row = r0 + threadIdx.y;
  while (row < r1) {
    // rows of one sample-group share the scale / shift: derive them once, then stream (4 loads in flight per thread)
    const int sg = (int)(row / rows_per_group);
    const long seg_end = (long)(sg + 1) * rows_per_group < r1 ? (long)(sg + 1) * rows_per_group : r1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ge = (vc * 8 + e) / cg;
      // E[x^2] - mean^2 in fp64: exact to ~1e-16 * mean^2, so a group whose mean dwarfs its spread keeps its variance
      const double mean_d = stats[((long)sg * groups + ge) * 2 + 0] * inv_cnt;
      const float var = fmaxf((float)(stats[((long)sg * groups + ge) * 2 + 1] * inv_cnt - mean_d * mean_d), 0.f);
      const float mean = (float)mean_d;
      A[e] = rsqrtf(var + eps) * (float)gm.e[e];
      B[e] = (float)bt.e[e] - mean * A[e];
    }
    auto emit = [&](const U128& u, long r) {
      U128 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = __builtin_fmaf((float)u.e[e], A[e], B[e]);
        if (silu) v = silu_f(v);
        o.e[e] = (f16)v;
      }
      *reinterpret_cast<uint4*>(Y + r * ldy + vc * 8) = o.u;
    };
    const long st = blockDim.y;
    for (; row + (ME_GN_APPLY_DEPTH - 1) * st < seg_end; row += ME_GN_APPLY_DEPTH * st) {   // X may alias Y: the loads are issued explicitly ahead of the stores
      U128 u[ME_GN_APPLY_DEPTH];
#pragma unroll
      for (int k = 0; k < ME_GN_APPLY_DEPTH; ++k) u[k].u = ldg128(X + (row + k * st) * ldx + vc * 8);
#pragma unroll
      for (int k = 0; k < ME_GN_APPLY_DEPTH; ++k) emit(u[k], row + k * st);
    }
    for (; row < seg_end; row += st) {
      U128 u;
      u.u = ldg128(X + row * ldx + vc * 8);
      emit(u, row);
    }
  }
}

// One wave per row; NV = 16-byte vectors per lane.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ X, f16* __restrict__ Y, const f16* __restrict__ gamma,
                                                        const f16* __restrict__ beta, long rows, int C, int ldx, int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int tpr = C / 8;
  U128 u[NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    u[k].u = vc < tpr ? ldg128(X + row * ldx + vc * 8) : zero128();
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)u[k].e[e];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    if (vc < tpr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)u[k].e[e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    if (vc < tpr) {
      U128 gm, bt, o;
      gm.u = ldg128(gamma + vc * 8);
      bt.u = ldg128(beta + vc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (f16)(((float)u[k].e[e] - mean) * rstd * (float)gm.e[e] + (float)bt.e[e]);
      *reinterpret_cast<uint4*>(Y + row * ldy + vc * 8) = o.u;
    }
  }
}


// C = 320 (level 0: two thirds of all LayerNorm rows of a step): 40 of a wave's 64 lanes would hold a row, and a wave that
// lives for one 640-byte row spends its life being launched.  Here HALF a wave holds a row -- lane l of the half loads
// columns [8l, 8l+8) and [256 + 2l, 256 + 2l + 2) -- every lane is busy, a wave walks U = 2 row pairs per trip with all four
// rows' loads in flight before the first reduction, and gamma / beta stay in registers for the whole walk.
__global__ __launch_bounds__(256) void layernorm320_kernel(const f16* __restrict__ X, f16* __restrict__ Y, const f16* __restrict__ gamma,
                                                           const f16* __restrict__ beta, long rows, int ldx, int ldy, float eps) {
  constexpr int U = 2;
  const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  U128 g8, b8;
  g8.u = ldg128(gamma + 8 * l32);
  b8.u = ldg128(beta + 8 * l32);
  const f16x2 g2 = *reinterpret_cast<const f16x2*>(gamma + 256 + 2 * l32), b2 = *reinterpret_cast<const f16x2*>(beta + 256 + 2 * l32);
  auto half_sum = [](float v) {   // over the 32 lanes of this half
    v = xor16_sum(v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (long r0 = wave * 2 * U; r0 < rows; r0 += nwaves * 2 * U) {
    U128 a[U];
    f16x2 c[U];
    long row[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      row[k] = r0 + 2 * k + half;
      const bool ok = row[k] < rows;
      a[k].u = ok ? ldg128(X + row[k] * ldx + 8 * l32) : zero128();
      c[k] = ok ? *reinterpret_cast<const f16x2*>(X + row[k] * ldx + 256 + 2 * l32) : f16x2{(f16)0.f, (f16)0.f};
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      float s = (float)c[k][0] + (float)c[k][1];
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)a[k].e[e];
      const float mean = half_sum(s) * (1.0f / 320.0f);
      float d0 = (float)c[k][0] - mean, d1 = (float)c[k][1] - mean;
      float q = d0 * d0 + d1 * d1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)a[k].e[e] - mean;
        q += d * d;
      }
      const float rstd = rsqrtf(half_sum(q) * (1.0f / 320.0f) + eps);
      if (row[k] < rows) {
        U128 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (f16)(((float)a[k].e[e] - mean) * rstd * (float)g8.e[e] + (float)b8.e[e]);
        *reinterpret_cast<uint4*>(Y + row[k] * ldy + 8 * l32) = o.u;
        const f16x2 o2 = {(f16)(d0 * rstd * (float)g2[0] + (float)b2[0]), (f16)(d1 * rstd * (float)g2[1] + (float)b2[1])};
        *reinterpret_cast<f16x2*>(Y + row[k] * ldy + 256 + 2 * l32) = o2;
      }
    }
  }
}


// ---- partial row sums for the LayerNorm-folded projections (ABI 9, me_gemm_args.ln_stats) ----
// stats[p * stride + 2 m] = (sum, sum of squares) of row m over the columns [320 p, 320 p + 320) -- the format the row-contiguous GEMM epilogue writes for
// the rows it produces (gemm.hip, ln_out); this kernel serves the rows that come from anywhere else (small grids, exchanged rows).  One wave per row.
template <int NV>
__global__ __launch_bounds__(256) void ln_stats_kernel(const f16* __restrict__ X, float* __restrict__ stats, long rows, int C, int ldx, long stride, int parts) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int tpr = C / 8;
  const f16x2 one2 = {(f16)1.f, (f16)1.f};
  float s1[NV], s2[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    U128 u;
    u.u = vc < tpr ? ldg128(X + row * ldx + vc * 8) : zero128();
    s1[k] = s2[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f16x2 h = {u.e[2 * q], u.e[2 * q + 1]};
      s1[k] = __builtin_amdgcn_fdot2(h, one2, s1[k], false);
      s2[k] = __builtin_amdgcn_fdot2(h, h, s2[k], false);
    }
  }
  for (int p = 0; p < parts; ++p) {    // a 320-column part = 40 vectors
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int vc = lane + 64 * k;
      const bool mine = parts == 1 || vc / 40 == p;
      a1 += mine ? s1[k] : 0.f;
      a2 += mine ? s2[k] : 0.f;
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) *reinterpret_cast<f32x2*>(stats + (long)p * stride + 2 * row) = f32x2{a1, a2};
  }
}

// C = 320: half a wave per row as in layernorm320_kernel (lane l of the half: columns [8 l, 8 l + 8) and [256 + 2 l, 256 + 2 l + 2)), U row pairs per trip
__global__ __launch_bounds__(256) void ln_stats320_kernel(const f16* __restrict__ X, float* __restrict__ stats, long rows, int ldx) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const f16x2 one2 = {(f16)1.f, (f16)1.f};
  auto half_sum = [](float v) {   // over the 32 lanes of this half
    v = xor16_sum(v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (long r0 = wave * 2 * U; r0 < rows; r0 += nwaves * 2 * U) {
    U128 a[U];
    f16x2 c[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const long row = r0 + 2 * k + half;
      const bool ok = row < rows;
      a[k].u = ok ? ldg128(X + row * ldx + 8 * l32) : zero128();
      c[k] = ok ? *reinterpret_cast<const f16x2*>(X + row * ldx + 256 + 2 * l32) : f16x2{(f16)0.f, (f16)0.f};
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      float s1 = __builtin_amdgcn_fdot2(c[k], one2, 0.f, false), s2 = __builtin_amdgcn_fdot2(c[k], c[k], 0.f, false);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f16x2 h = {a[k].e[2 * q], a[k].e[2 * q + 1]};
        s1 = __builtin_amdgcn_fdot2(h, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(h, h, s2, false);
      }
      s1 = half_sum(s1);
      s2 = half_sum(s2);
      const long row = r0 + 2 * k + half;
      if (l32 == 0 && row < rows) *reinterpret_cast<f32x2*>(stats + 2 * row) = f32x2{s1, s2};
    }
  }
}

// Row softmax, one wave per row, the row held in registers (cols <= 8192 -> <= 16 vectors of 8 per lane).
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* X, f16* Y, long rows, int cols, int ldx, int ldy) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int tpr = cols / 8;
  U128 u[NV];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    if (vc < tpr) {
      u[k].u = ldg128(X + row * ldx + vc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)u[k].e[e]);
    }
  }
  mx = wave_max(mx);
  float p[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < tpr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[k][e] = __builtin_amdgcn_exp2f(((float)u[k].e[e] - mx) * 1.4426950408889634f);
        sum += p[k][e];
      }
    }
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int vc = lane + 64 * k;
    if (vc < tpr) {
      U128 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (f16)(p[k][e] * inv);
      *reinterpret_cast<uint4*>(Y + row * ldy + vc * 8) = o.u;
    }
  }
}

}  // namespace

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_kernel(const char* name);

static int gn_validate(const me_groupnorm_args* a) {
  if (!a || !a->X || !a->Y || !a->gamma || !a->beta || !a->stats) { me_set_error("me_groupnorm: null pointer"); return ME_EINVAL; }
  if (a->rows <= 0 || a->rows_per_group <= 0 || a->rows % a->rows_per_group) { me_set_error("me_groupnorm: rows must be a multiple of rows_per_group"); return ME_EINVAL; }
  if (a->groups <= 0 || a->groups > 64 || a->C % a->groups || a->C % 8 || a->ldx % 8 || a->ldy % 8) { me_set_error("me_groupnorm: bad channel geometry"); return ME_EINVAL; }
  if (((uintptr_t)a->X | (uintptr_t)a->Y | (uintptr_t)a->gamma | (uintptr_t)a->beta) & 15) { me_set_error("me_groupnorm: misaligned pointer"); return ME_EINVAL; }
  return ME_OK;
}

// chunk geometry of the statistics pass (shared by me_groupnorm_scratch_bytes).  A function of rows_per_group ALONE: the statistics of a
// sample must not depend on how many samples share the launch (the UNet graph runs the first blocks on half the batch when the
// classifier-free-guidance halves are copies of each other, and the step has to stay bitwise the same).
static void gn_chunks(const me_groupnorm_args* a, int* chunks_out, int* chunk_rows_out) {
  // large groups (the five-dimensional GroupNorm of the upper levels, few samples per launch): ~256 chunks per sample;
  // small groups (per-frame statistics, many samples per launch): ~16
  const int big = a->rows_per_group > 4096;
  int chunk_rows = (a->rows_per_group + (big ? 255 : 15)) / (big ? 256 : 16);
  if (chunk_rows < 24) chunk_rows = 24;
  if (chunk_rows > (big ? 384 : 256)) chunk_rows = big ? 384 : 256;
  const int chunks = (a->rows_per_group + chunk_rows - 1) / chunk_rows;
  *chunks_out = chunks;
  *chunk_rows_out = chunk_rows;
}

extern "C" int64_t me_groupnorm_scratch_bytes(int32_t rows, int32_t rows_per_group, int32_t groups) {
  if (rows <= 0 || rows_per_group <= 0 || rows % rows_per_group || groups <= 0) return 0;
  me_groupnorm_args a{};
  a.rows = rows;
  a.rows_per_group = rows_per_group;
  int chunks, chunk_rows;
  gn_chunks(&a, &chunks, &chunk_rows);
  const int64_t nsg = rows / rows_per_group;
  return nsg * groups * 2 * (int64_t)sizeof(double) + (int64_t)chunks * nsg * groups * (int64_t)sizeof(float4);
}

extern "C" int me_groupnorm_stats(const me_groupnorm_args* a, void* stream) {
  if (int rc = gn_validate(a)) return rc;
  if (a->C > 2560) { me_set_error("me_groupnorm: C must be <= 2560"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsg = a->rows / a->rows_per_group;
  int chunks, chunk_rows;
  gn_chunks(a, &chunks, &chunk_rows);
  double* stats = reinterpret_cast<double*>(a->stats);
  float4* part = reinterpret_cast<float4*>(stats + (size_t)nsg * a->groups * 2);
  const int tpr = a->C / 8, tprc = tpr < 256 ? tpr : 256;
  const size_t lds = (size_t)(256 / tprc) * tprc * 8 * sizeof(float2);
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, nsg), dim3(256), lds, st, reinterpret_cast<const f16*>(a->X), part, a->rows_per_group,
                     chunk_rows, a->C, a->ldx, a->groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(nsg * a->groups), dim3(256), 0, st, part, stats, chunks, chunk_rows, a->rows_per_group, nsg, a->groups, a->C / a->groups);
  if (hipGetLastError() != hipSuccess) { me_set_error("me_groupnorm_stats: kernel launch failed"); return ME_EHIP; }
  return ME_OK;
}

extern "C" int me_groupnorm_apply(const me_groupnorm_args* a, int64_t rows_per_group_total, void* stream) {
  if (int rc = gn_validate(a)) return rc;
  if (rows_per_group_total < a->rows_per_group) { me_set_error("me_groupnorm_apply: total rows per group smaller than the local count"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // block = bx vectors of a row x by rows in flight; bx = the largest divisor of C/8 that fits 256 threads
  const int tpr = a->C / 8;
  int ny = 1;
  while (tpr / ny > 256 || tpr % ny) ++ny;
  const int bx = tpr / ny, by = 256 / bx > 0 ? 256 / bx : 1;
  // rows per block: ~16 rows per thread, but at least ~2048 blocks' worth of parallelism on big inputs
  long chunk = (long)by * 16;
  while (chunk > by && (a->rows + chunk - 1) / chunk * ny < 2048) chunk /= 2;
  if (chunk < by) chunk = by;
  const long nbx = (a->rows + chunk - 1) / chunk;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)nbx, (unsigned)ny), dim3(bx, by), 0, st, reinterpret_cast<const f16*>(a->X), reinterpret_cast<f16*>(a->Y),
                     reinterpret_cast<const double*>(a->stats), reinterpret_cast<const f16*>(a->gamma), reinterpret_cast<const f16*>(a->beta), (long)a->rows, a->rows_per_group,
                     (long)rows_per_group_total, a->C, a->ldx, a->ldy, a->groups, a->eps, a->silu, (int)chunk);
  if (hipGetLastError() != hipSuccess) { me_set_error("me_groupnorm_apply: kernel launch failed"); return ME_EHIP; }
  return ME_OK;
}

extern "C" int me_groupnorm(const me_groupnorm_args* a, void* stream) {
  if (int rc = me_groupnorm_stats(a, stream)) return rc;
  return me_groupnorm_apply(a, a->rows_per_group, stream);
}

extern "C" int me_layernorm(const me_layernorm_args* a, void* stream) {
  if (!a || !a->X || !a->Y || !a->gamma || !a->beta) { me_set_error("me_layernorm: null pointer"); return ME_EINVAL; }
  if (a->rows <= 0 || a->C <= 0 || a->C % 8 || a->C > 1536 || a->ldx % 8 || a->ldy % 8) { me_set_error("me_layernorm: C must be a multiple of 8 and <= 1536"); return ME_EINVAL; }
  if (((uintptr_t)a->X | (uintptr_t)a->Y | (uintptr_t)a->gamma | (uintptr_t)a->beta) & 15) { me_set_error("me_layernorm: misaligned pointer"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((a->rows + 3) / 4);
  const int nv = (a->C / 8 + 63) / 64;
  const f16* X = reinterpret_cast<const f16*>(a->X);
  f16* Y = reinterpret_cast<f16*>(a->Y);
  const f16* gm = reinterpret_cast<const f16*>(a->gamma);
  const f16* bt = reinterpret_cast<const f16*>(a->beta);
  (void)hipGetLastError();
  if (a->C == 320 && a->rows >= 4096) {
    const unsigned nb = (unsigned)min((long)((a->rows + 15) / 16), 256L * 24);   // 16 rows per block and trip; <= 24 blocks per CU, the rest by grid stride
    hipLaunchKernelGGL(layernorm320_kernel, dim3(nb), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->ldx, a->ldy, a->eps);
  } else if (nv == 1) hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  else if (nv == 2) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  else hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  if (hipGetLastError() != hipSuccess) { me_set_error("me_layernorm: kernel launch failed"); return ME_EHIP; }
  return ME_OK;
}


extern "C" int me_ln_stats(const void* X, int32_t ldx, int64_t rows, int32_t C, void* stats, int64_t stride, void* stream) {
  if (!X || !stats) { me_set_error("me_ln_stats: null pointer"); return ME_EINVAL; }
  if (rows <= 0 || C <= 0 || C % 8 || C > 1536 || ldx % 8 || ((uintptr_t)X & 15) || ((uintptr_t)stats & 7) || (stride & 1) || (C % 320 == 0 && C > 320 && stride < 2 * rows)) {
    me_set_error("me_ln_stats: C must be a multiple of 8 and <= 1536, X 16-byte aligned with ldx % 8 == 0, stats 8-byte aligned, parts >= 2 * rows floats apart");
    return ME_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const f16* x = reinterpret_cast<const f16*>(X);
  float* s = reinterpret_cast<float*>(stats);
  const int parts = C % 320 == 0 ? C / 320 : 1;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  const int nv = (C / 8 + 63) / 64;
  (void)hipGetLastError();
  if (C == 320 && rows >= 4096) {
    const unsigned nb = (unsigned)min((long)((rows + 31) / 32), 256L * 24);   // 32 rows per block and trip
    hipLaunchKernelGGL(ln_stats320_kernel, dim3(nb), dim3(256), 0, st, x, s, (long)rows, ldx);
  } else if (nv == 1) hipLaunchKernelGGL(ln_stats_kernel<1>, dim3(blocks), dim3(256), 0, st, x, s, (long)rows, C, ldx, (long)stride, parts);
  else if (nv == 2) hipLaunchKernelGGL(ln_stats_kernel<2>, dim3(blocks), dim3(256), 0, st, x, s, (long)rows, C, ldx, (long)stride, parts);
  else hipLaunchKernelGGL(ln_stats_kernel<3>, dim3(blocks), dim3(256), 0, st, x, s, (long)rows, C, ldx, (long)stride, parts);
  me_set_kernel("ln_stats");
  if (hipGetLastError() != hipSuccess) { me_set_error("me_ln_stats: kernel launch failed"); return ME_EHIP; }
  return ME_OK;
}

extern "C" int me_softmax_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, int64_t rows, int32_t cols, void* stream) {
  if (!Y || !X) { me_set_error("me_softmax_rows: null pointer"); return ME_EINVAL; }
  if (rows <= 0 || cols <= 0 || cols % 8 || cols > 8192 || ldx % 8 || ldy % 8) { me_set_error("me_softmax_rows: cols must be a multiple of 8 and <= 8192"); return ME_EINVAL; }
  if (((uintptr_t)X | (uintptr_t)Y) & 15) { me_set_error("me_softmax_rows: misaligned pointer"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  const int nv = (cols / 8 + 63) / 64;
  const f16* x = reinterpret_cast<const f16*>(X);
  f16* y = reinterpret_cast<f16*>(Y);
  (void)hipGetLastError();
  if (nv <= 2) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3(blocks), dim3(256), 0, st, x, y, (long)rows, cols, ldx, ldy);
  else if (nv <= 8) hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3(blocks), dim3(256), 0, st, x, y, (long)rows, cols, ldx, ldy);
  else hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3(blocks), dim3(256), 0, st, x, y, (long)rows, cols, ldx, ldy);
  if (hipGetLastError() != hipSuccess) { me_set_error("me_softmax_rows: kernel launch failed"); return ME_EHIP; }
  return ME_OK;
}
