// GroupNorm(+SiLU) and LayerNorm on channels-last fp16 activations (HBM-bound kernels, gfx950).
//
// GroupNorm: pass 1 accumulates (sum, sum of squares) per (row-group, channel-group) with 16-byte
// loads, register accumulation down the rows, LDS float atomics across the block and one global
// atomic per (block, group); pass 2 re-reads X, normalises, applies gamma/beta (+SiLU), writes Y.
// Algorithmic bytes per call: rows*C*2 (read) + rows*C*2 (write); the statistics pass re-reads X.
// LayerNorm: one wave per row, two-pass mean/variance in registers.
#include "me_common.h"
#include "../../include/motioned.h"

namespace {

__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ X, float* __restrict__ stats, int rows_per_group,
                                                       int chunk_rows, int C, int ldx, int groups) {
  __shared__ float sacc[64][2];
  const int tid = threadIdx.x;
  if (tid < 64) { sacc[tid][0] = 0.f; sacc[tid][1] = 0.f; }
  __syncthreads();

  const int sg = blockIdx.y;
  const int r0 = blockIdx.x * chunk_rows;
  const int r1 = min(r0 + chunk_rows, rows_per_group);
  const int tpr = C / 8;                       // 16-byte vectors per row
  const int tprc = tpr < 256 ? tpr : 256;      // vector columns handled per pass
  const int rl = tid / tprc, vc0 = tid - rl * tprc;
  const int RL = 256 / tprc;                   // row lanes
  const int cg = C / groups;
  const f16* base = X + (long)sg * rows_per_group * ldx;

  if (rl < RL) {
    for (int vc = vc0; vc < tpr; vc += tprc) {
      float s[8], q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll 4
      for (int r = r0 + rl; r < r1; r += RL) {   // 4 independent 16-byte loads in flight per thread
        U128 u;
        u.u = ldg128(base + (long)r * ldx + vc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (float)u.e[e];
          s[e] += v;
          q[e] += v * v;
        }
      }
      // fold the 8 channels into their (at most two when cg >= 8, else more) groups
      int gcur = (vc * 8) / cg;
      float ss = 0.f, qq = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ge = (vc * 8 + e) / cg;
        if (ge != gcur) {
          atomicAdd(&sacc[gcur][0], ss);
          atomicAdd(&sacc[gcur][1], qq);
          gcur = ge;
          ss = 0.f;
          qq = 0.f;
        }
        ss += s[e];
        qq += q[e];
      }
      atomicAdd(&sacc[gcur][0], ss);
      atomicAdd(&sacc[gcur][1], qq);
    }
  }
  __syncthreads();
  if (tid < groups) {
    atomicAdd(&stats[((long)sg * groups + tid) * 2 + 0], sacc[tid][0]);
    atomicAdd(&stats[((long)sg * groups + tid) * 2 + 1], sacc[tid][1]);
  }
}

// y = x * A[c] + B[c] (+ SiLU) with A = rstd * gamma, B = beta - mean * rstd * gamma.  A thread owns ONE 16-byte
// channel vector and walks down the rows of its block's chunk, so the per-channel scale / shift live in 16
// registers and the inner loop is load - 8 FMA - store (the first version re-derived row, group, mean and rstd with
// integer divisions and an rsqrt for every vector and ran at 2 TB/s).  blockDim.x = vectors per row handled by the
// block (a divisor of C/8, <= 256), blockDim.y rows in flight: a wave covers whole contiguous row segments.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* X, f16* Y, const float* __restrict__ stats,
                                                       const f16* __restrict__ gamma, const f16* __restrict__ beta, long rows,
                                                       int rows_per_group, long rows_per_group_total, int C, int ldx, int ldy, int groups,
                                                       float eps, int silu, int chunk) {
  const int vc = blockIdx.y * blockDim.x + threadIdx.x;   // 16-byte vector column
  const int cg = C / groups;
  const float inv_cnt = 1.0f / ((float)rows_per_group_total * (float)cg);   // global count when frame-sharded
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
      const float mean = stats[((long)sg * groups + ge) * 2 + 0] * inv_cnt;
      const float var = fmaxf(stats[((long)sg * groups + ge) * 2 + 1] * inv_cnt - mean * mean, 0.f);
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
    for (; row + 3 * st < seg_end; row += 4 * st) {   // X may alias Y: the four loads are issued explicitly ahead of the stores
      U128 u0, u1, u2, u3;
      u0.u = ldg128(X + row * ldx + vc * 8);
      u1.u = ldg128(X + (row + st) * ldx + vc * 8);
      u2.u = ldg128(X + (row + 2 * st) * ldx + vc * 8);
      u3.u = ldg128(X + (row + 3 * st) * ldx + vc * 8);
      emit(u0, row);
      emit(u1, row + st);
      emit(u2, row + 2 * st);
      emit(u3, row + 3 * st);
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

static int gn_validate(const me_groupnorm_args* a) {
  if (!a || !a->X || !a->Y || !a->gamma || !a->beta || !a->stats) { me_set_error("me_groupnorm: null pointer"); return ME_EINVAL; }
  if (a->rows <= 0 || a->rows_per_group <= 0 || a->rows % a->rows_per_group) { me_set_error("me_groupnorm: rows must be a multiple of rows_per_group"); return ME_EINVAL; }
  if (a->groups <= 0 || a->groups > 64 || a->C % a->groups || a->C % 8 || a->ldx % 8 || a->ldy % 8) { me_set_error("me_groupnorm: bad channel geometry"); return ME_EINVAL; }
  if (((uintptr_t)a->X | (uintptr_t)a->Y | (uintptr_t)a->gamma | (uintptr_t)a->beta) & 15) { me_set_error("me_groupnorm: misaligned pointer"); return ME_EINVAL; }
  return ME_OK;
}

extern "C" int me_groupnorm_stats(const me_groupnorm_args* a, void* stream) {
  if (int rc = gn_validate(a)) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsg = a->rows / a->rows_per_group;
  if (hipMemsetAsync(a->stats, 0, (size_t)nsg * a->groups * 2 * sizeof(float), st) != hipSuccess) { me_set_error("me_groupnorm: memset failed"); return ME_EHIP; }
  // ~1024 blocks in total, at most 512 per sample-group: every block ends with one global float atomic per channel
  // group on the same groups * 2 addresses of its sample-group, and with 2048 blocks on ONE sample (DDIM inversion,
  // B = 1) those atomics, not the 250 MB read, set the time (4.6 -> 3.0 ms per inversion step)
  int chunks = 1024 / nsg;
  if (chunks > 512) chunks = 512;
  if (chunks < 1) chunks = 1;
  int chunk_rows = (a->rows_per_group + chunks - 1) / chunks;
  if (chunk_rows < 8) chunk_rows = 8;
  chunks = (a->rows_per_group + chunk_rows - 1) / chunk_rows;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, nsg), dim3(256), 0, st, reinterpret_cast<const f16*>(a->X), a->stats, a->rows_per_group,
                     chunk_rows, a->C, a->ldx, a->groups);
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
                     a->stats, reinterpret_cast<const f16*>(a->gamma), reinterpret_cast<const f16*>(a->beta), (long)a->rows, a->rows_per_group,
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
  if (nv == 1) hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  else if (nv == 2) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  else hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(256), 0, st, X, Y, gm, bt, (long)a->rows, a->C, a->ldx, a->ldy, a->eps);
  if (hipGetLastError() != hipSuccess) { me_set_error("me_layernorm: kernel launch failed"); return ME_EHIP; }
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
