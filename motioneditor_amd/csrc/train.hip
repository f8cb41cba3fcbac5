// Parameter-gradient, gradient-bookkeeping and optimiser kernels (gfx950): the device side of the adapter training step
// (reference train_adaptor.py:364-385: loss.backward() into controlnet_adapter.*, accelerate's DDP gradient average,
// clip_grad_norm_, AdamW) and of the null-text optimisation's loss / Adam (p2p/null_text_optimization.py:140-160).
// Everything is deterministic: reductions over the token axis go through fixed-order partials, never atomics.
#include "me_common.h"
#include "../../include/motioned.h"

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_hip_error(const char* what, int err);
extern "C" void me_set_kernel(const char* name);

namespace {

inline unsigned grid_for(long n, long cap = 16384) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: work[s][n][k] = sum over the split's rows m of dY[m, n] * X[src(m, tap), k]   (fp32, MFMA)
// Both operands are contracted over their ROW index, so both tiles are staged transposed ([column][32 rows], rows in natural
// order: MFMA k-slot (g, j) = row g * 8 + j for A and B alike).  Block tile 64 (k) x 64 (n), 4 waves of 32 x 32; operand A = X^T
// so that a lane ends up with 4 consecutive k of one n -> 16-byte fp32 stores.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DW_MS = 32;    // rows per stage
constexpr int DW_TLD = DW_MS + 8;

__global__ __launch_bounds__(256) void gemm_dw_kernel(const me_gemm_dw_args a, int splits, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) f16 sX[64 * DW_TLD];    // [k][row]
  __shared__ __attribute__((aligned(16))) f16 sD[64 * DW_TLD];    // [n][row]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int wk = wave >> 1, wn = wave & 1;

  const int nbk = (a.K + 63) / 64, nbn = (a.N + 63) / 64;
  int w = blockIdx.x;
  const int bk = w % nbk;
  w /= nbk;
  const int bn = w % nbn;
  const int sp = w / nbn;
  const int k0 = bk * 64, n0 = bn * 64;
  const long m0 = (long)sp * rows_per_split;
  const long m1 = m0 + rows_per_split < a.M ? m0 + rows_per_split : a.M;

  const f16* __restrict__ X = reinterpret_cast<const f16*>(a.X);
  f32x4 acc[2][2];   // [k tile][n tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging map: thread -> (row r = tid / 8, 8-column chunk cc = tid % 8) of both 32 x 64 tiles
  const int r = tid >> 3, cc = tid & 7;
  for (long mb = m0; mb < m1; mb += DW_MS) {
    const long m = mb + r;
    const bool mok = m < m1;
    // source row of X for this tap (TemporalConv: the frame tap - 1 away, inside the same chunk of `chunk` frames)
    long xs = m;
    bool xok = mok;
    if (a.gather == ME_GATHER_TCONV && mok) {
      const long bf = m / a.npix;
      const int fr = (int)(bf % a.frames), fs = fr + a.tap - 1;
      xok = fs >= 0 && fs < a.frames && fs / a.chunk == fr / a.chunk;
      xs = m + (long)(a.tap - 1) * a.npix;
    }
    U128 ux, ud;
    ux.u = (xok && k0 + cc * 8 < a.K) ? ldg128(X + xs * a.ldx + k0 + cc * 8) : zero128();
    if (mok && n0 + cc * 8 < a.N) {
      if (a.dy_is_f16) {
        ud.u = ldg128(reinterpret_cast<const f16*>(a.dY) + m * a.lddy + n0 + cc * 8);
      } else {
        const float* p = reinterpret_cast<const float*>(a.dY) + m * a.lddy + n0 + cc * 8;
        const float4 f0 = *reinterpret_cast<const float4*>(p), f1 = *reinterpret_cast<const float4*>(p + 4);
        ud.e[0] = (f16)f0.x; ud.e[1] = (f16)f0.y; ud.e[2] = (f16)f0.z; ud.e[3] = (f16)f0.w;
        ud.e[4] = (f16)f1.x; ud.e[5] = (f16)f1.y; ud.e[6] = (f16)f1.z; ud.e[7] = (f16)f1.w;
      }
    } else {
      ud.u = zero128();
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sX[(cc * 8 + e) * DW_TLD + r] = ux.e[e];
      sD[(cc * 8 + e) * DW_TLD + r] = ud.e[e];
    }
    __syncthreads();
    f16x8 fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(sX + (wk * 32 + i * 16 + l15) * DW_TLD + g * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f16x8*>(sD + (wn * 32 + j * 16 + l15) * DW_TLD + g * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    __syncthreads();
  }
  // D[i = k][n' = n]: lane (n = l15, g), reg r <-> k = g * 4 + r
  float* work = reinterpret_cast<float*>(a.work) + (long)sp * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 32 + i * 16 + g * 4, n = n0 + wn * 32 + j * 16 + l15;
      if (n < a.N && k < a.K) *reinterpret_cast<float4*>(work + (long)n * a.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// dW[n][tap][k] += alpha * sum_s work[s][n][k], splits added in index order
__global__ __launch_bounds__(256) void gemm_dw_fold_kernel(const float* __restrict__ work, float* dW, int N, int K, int taps, int tap, int splits, float alpha) {
  const long total = (long)N * K;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += work[(long)sp * total + idx];
    const long n = idx / K;
    const int k = (int)(idx - n * K);
    dW[(n * taps + tap) * K + k] += alpha * s;
  }
}

void dw_geometry(int M, int N, int K, int* splits, int* rows_per_split) {
  const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
  long s = (1024 + tiles - 1) / tiles;            // ~1024 blocks
  const long smax = (M + 4 * DW_MS - 1) / (4 * DW_MS);   // at least four stages per split
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  long rps = (M + s - 1) / s;
  rps = (rps + DW_MS - 1) / DW_MS * DW_MS;
  s = (M + rps - 1) / rps;
  *splits = (int)s;
  *rows_per_split = (int)rps;
}

// ---------------------------------------------------------------------------------------------------------------------
// column sums of a gradient (bias gradients): part[rs][n] over row slices, folded in order
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CS_RS = 64;
__global__ __launch_bounds__(256) void colsum_part_f32_kernel(const void* __restrict__ dY, int lddy, int is_f16, long M, int N, float* __restrict__ part) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int rs = blockIdx.y;
  const long rows = (M + CS_RS - 1) / CS_RS, r0 = rs * rows, r1 = r0 + rows < M ? r0 + rows : M;
  if (n >= N) return;
  float s = 0.f;
  if (is_f16) {
    const f16* p = reinterpret_cast<const f16*>(dY);
    for (long r = r0; r < r1; ++r) s += (float)p[r * lddy + n];
  } else {
    const float* p = reinterpret_cast<const float*>(dY);
    for (long r = r0; r < r1; ++r) s += p[r * lddy + n];
  }
  part[(long)rs * N + n] = s;
}
__global__ __launch_bounds__(256) void colsum_fold_f32_kernel(const float* __restrict__ part, float* out, int N, int nparts, float alpha) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(long)p * N + n];
  out[n] += alpha * s;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm parameter gradients: dgamma[c] += sum_m dy[m, c] xhat[m, c], dbeta[c] += sum_m dy[m, c]
// one wave per row (statistics recomputed), per-lane column sums in registers, waves / row slices folded in fixed order
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LNP_MAXV = 24;   // C <= 1536
__global__ __launch_bounds__(256) void ln_params_part_kernel(const f16* __restrict__ X, int ldx, const float* __restrict__ dY, int lddy, long rows, int C, float eps,
                                                             float* __restrict__ part, int rows_per_block) {
  __shared__ float red[4][2][1536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float ag[LNP_MAXV], ab[LNP_MAXV];
#pragma unroll
  for (int v = 0; v < LNP_MAXV; ++v) { ag[v] = 0.f; ab[v] = 0.f; }
  for (long row = r0 + wave; row < r1; row += 4) {
    const f16* x = X + row * ldx;
    const float* dy = dY + row * lddy;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += (float)x[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float d = (float)x[c] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int v = 0; v < LNP_MAXV; ++v) {
      const int c = lane + 64 * v;
      if (c < C) {
        const float d = dy[c];
        ag[v] += d * ((float)x[c] - mean) * rstd;
        ab[v] += d;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < LNP_MAXV; ++v) {
    const int c = lane + 64 * v;
    if (c < C) {
      red[wave][0][c] = ag[v];
      red[wave][1][c] = ab[v];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part[((long)blockIdx.x * 2 + 0) * C + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
    part[((long)blockIdx.x * 2 + 1) * C + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
  }
}
__global__ __launch_bounds__(256) void ln_params_fold_kernel(const float* __restrict__ part, float* dgamma, float* dbeta, int C, int nparts, float alpha) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float sg = 0.f, sb = 0.f;
  for (int p = 0; p < nparts; ++p) {
    sg += part[((long)p * 2 + 0) * C + c];
    sb += part[((long)p * 2 + 1) * C + c];
  }
  if (dgamma) dgamma[c] += alpha * sg;
  if (dbeta) dbeta[c] += alpha * sb;
}

// ---------------------------------------------------------------------------------------------------------------------
// gradient accumulation of the tape: dst (fp32 view) += alpha * src (fp32 / fp16), optionally summing the 2 x 2 source block of
// every destination pixel (input gradient of the nearest-2x upsample in front of a convolution, resnet_2d.py:77)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_acc_kernel(float* dst, int lddst, const void* __restrict__ src, int ldsrc, int is_f16, long rows, int cols, float alpha,
                                                       int ph, int pw) {
  const int vpr = cols / 4;
  const long n = rows * vpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const long r = idx / vpr;
    const int c = (int)(idx - r * vpr) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](long sr) {
      if (is_f16 & 1) {
        U64 u;
        u.u = *reinterpret_cast<const uint2*>(reinterpret_cast<const f16*>(src) + sr * ldsrc + c);
        acc.x += (float)u.e[0]; acc.y += (float)u.e[1]; acc.z += (float)u.e[2]; acc.w += (float)u.e[3];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + sr * ldsrc + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    };
    if (ph > 0) {   // dst pixel (img, y, x) of a ph x pw grid <- src pixels (img, 2y + a, 2x + b) of the 2ph x 2pw grid
      const long img = r / ((long)ph * pw);
      const int rem = (int)(r - img * ph * pw);
      const int y = rem / pw, x = rem - y * pw;
      const long base = img * 4L * ph * pw + (long)(2 * y) * (2 * pw) + 2 * x;
      add(base);
      add(base + 1);
      add(base + 2 * pw);
      add(base + 2 * pw + 1);
    } else {
      add(r);
    }
    float4* d = reinterpret_cast<float4*>(dst + r * lddst + c);
    float4 o = (is_f16 & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : *d;   // bit 1: first touch of the buffer -- store, do not read
    o.x += alpha * acc.x; o.y += alpha * acc.y; o.z += alpha * acc.z; o.w += alpha * acc.w;
    *d = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// sum of squares and largest magnitude of an fp32 vector: out = {sum x^2, max |x|}; fixed-order two-stage reduction
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SS_BLOCKS = 1024;
__device__ __forceinline__ void block_reduce2(float& s, float& m, float (*red)[2]) {
  s = wave_sum(s);
  m = wave_max(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = m; }
  __syncthreads();
  s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  m = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
}
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ x, long n, float* __restrict__ work) {
  __shared__ float red[4][2];
  float s = 0.f, m = 0.f;
  const long per = (n + SS_BLOCKS - 1) / SS_BLOCKS, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float v = x[i];
    s += v * v;
    m = fmaxf(m, fabsf(v));
  }
  block_reduce2(s, m, red);
  if (threadIdx.x == 0) { work[2 * blockIdx.x] = s; work[2 * blockIdx.x + 1] = m; }
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ work, float* out) {
  __shared__ float red[4][2];
  float s = 0.f, m = 0.f;
  for (int i = threadIdx.x; i < SS_BLOCKS; i += 256) {   // thread t adds partials t, t + 256, ... in order
    s += work[2 * i];
    m = fmaxf(m, work[2 * i + 1]);
  }
  block_reduce2(s, m, red);
  if (threadIdx.x == 0) { out[0] = s; out[1] = m; }
}

// ---------------------------------------------------------------------------------------------------------------------
// AdamW on fp32 master parameters (torch.optim.AdamW semantics; weight_decay = 0 is torch.optim.Adam), gradient clipping by the
// global norm folded in: g_eff = g * grad_scale * min(1, max_norm / (sqrt(gnorm_sq) * grad_scale + 1e-6))
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g, long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2, const float* __restrict__ gnorm_sq,
                                                    float max_norm, float grad_scale) {
  float gs = grad_scale;
  if (gnorm_sq) {
    if (!(gnorm_sq[0] < 3.0e38f)) return;   // inf / NaN gradient norm (fp16 overflow upstream): leave the masters and the moments alone
    const float total = sqrtf(gnorm_sq[0]) * grad_scale;
    gs *= fminf(1.0f, max_norm / (total + 1e-6f));
  }
  const float step = lr / bc1, rs2 = rsqrtf(bc2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * rs2 + eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ __launch_bounds__(256) void cast_f16_kernel(f16* __restrict__ dst, const float* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (f16)src[i];
}

// dst fp16 [rows, lddst] <- src fp32 [rows, ldsrc], columns [0, cols) cast, [cols, pad_cols) zeroed (4 columns per thread)
__global__ __launch_bounds__(256) void cast_rows_f16_kernel(f16* __restrict__ dst, int lddst, const float* __restrict__ src, int ldsrc, long rows, int cols, int pad_cols) {
  const int vpr = pad_cols / 4;
  const long n = rows * vpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const long r = idx / vpr;
    const int c = (int)(idx - r * vpr) * 4;
    U64 o;
    if (c < cols) {
      const float4 v = *reinterpret_cast<const float4*>(src + r * ldsrc + c);
      o.e[0] = (f16)v.x; o.e[1] = (f16)v.y; o.e[2] = (f16)v.z; o.e[3] = (f16)v.w;
    } else {
      o.u = make_uint2(0u, 0u);
    }
    *reinterpret_cast<uint2*>(dst + r * lddst + c) = o.u;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// loss seed: rec = ca * x + cb * (eu + g (ec - eu)) (or ca * x + cb * eu without ec), diff = rec - target,
// d_eps[row, c] = coef * diff, with x / target fp32 [nb, C, frames, npix] (reference layout) and eu / ec / d_eps channels-last rows
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_seed_kernel(float* __restrict__ diff, float* __restrict__ d_eps, int ldd, const f16* __restrict__ eu, int ldu,
                                                       const f16* __restrict__ ec, int ldc, const float* __restrict__ x, const float* __restrict__ target, int nb, int C,
                                                       int frames, int npix, float guidance, float ca, float cb, float coef) {
  const long total = (long)nb * C * frames * npix;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % npix);
  long r = idx / npix;
  const int f = (int)(r % frames);
  r /= frames;
  const int c = (int)(r % C);
  const int b = (int)(r / C);
  const long row = ((long)b * frames + f) * npix + p;
  const float u = (float)eu[row * ldu + c];
  const float e = ec ? u + guidance * ((float)ec[row * ldc + c] - u) : u;
  const float rec = (x ? ca * x[idx] : 0.f) + cb * e;
  const float d = rec - target[idx];
  diff[idx] = d;
  d_eps[row * ldd + c] = coef * d;
}

}  // namespace

#define ME_TRAIN_CHECK(name)                                                        \
  {                                                                                 \
    const hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) { me_set_hip_error(name, (int)e_); return ME_EHIP; }      \
    return ME_OK;                                                                   \
  }

extern "C" int64_t me_gemm_dw_work_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int s, rps;
  dw_geometry(M, N, K, &s, &rps);
  return (int64_t)s * N * K * (int64_t)sizeof(float);
}

extern "C" int me_gemm_dw(const me_gemm_dw_args* a, void* stream) {
  if (!a || !a->dY || !a->X || !a->dW || !a->work) { me_set_error("me_gemm_dw: null pointer"); return ME_EINVAL; }
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->N % 8 || a->K % 8 || a->ldx % 8 || a->lddy % (a->dy_is_f16 ? 8 : 4)) {
    me_set_error("me_gemm_dw: N, K, ldx must be multiples of 8, lddy of 4 (fp32) / 8 (fp16)");
    return ME_EINVAL;
  }
  if (((uintptr_t)a->dY | (uintptr_t)a->X | (uintptr_t)a->dW | (uintptr_t)a->work) & 15) { me_set_error("me_gemm_dw: misaligned pointer"); return ME_EINVAL; }
  if (a->taps < 1 || a->tap < 0 || a->tap >= a->taps) { me_set_error("me_gemm_dw: bad tap"); return ME_EINVAL; }
  if (a->gather != ME_GATHER_DENSE && a->gather != ME_GATHER_TCONV) { me_set_error("me_gemm_dw: dense and TemporalConv layers only"); return ME_EINVAL; }
  if (a->gather == ME_GATHER_TCONV && (a->taps != 3 || a->frames <= 0 || a->npix <= 0 || a->chunk <= 0 || a->M % (a->frames * a->npix))) {
    me_set_error("me_gemm_dw: bad tconv geometry");
    return ME_EINVAL;
  }
  if (a->gather == ME_GATHER_DENSE && a->taps != 1) { me_set_error("me_gemm_dw: a dense layer has one tap"); return ME_EINVAL; }
  int splits, rps;
  dw_geometry(a->M, a->N, a->K, &splits, &rps);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  const long blocks = (long)((a->K + 63) / 64) * ((a->N + 63) / 64) * splits;
  hipLaunchKernelGGL(gemm_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *a, splits, rps);
  hipLaunchKernelGGL(gemm_dw_fold_kernel, dim3(grid_for((long)a->N * a->K, 4096)), dim3(256), 0, st, reinterpret_cast<const float*>(a->work),
                     reinterpret_cast<float*>(a->dW), a->N, a->K, a->taps, a->tap, splits, a->alpha);
  me_set_kernel("gemm_dw_kernel");
  ME_TRAIN_CHECK("me_gemm_dw")
}

extern "C" int64_t me_colsum_work_bytes(int32_t N) { return N > 0 ? (int64_t)CS_RS * N * (int64_t)sizeof(float) : 0; }

extern "C" int me_colsum(float* out, const void* dY, int32_t lddy, int32_t dy_is_f16, int64_t M, int32_t N, float alpha, float* work, void* stream) {
  if (!out || !dY || !work || M <= 0 || N <= 0) { me_set_error("me_colsum: bad arguments"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  hipLaunchKernelGGL(colsum_part_f32_kernel, dim3((unsigned)((N + 255) / 256), CS_RS), dim3(256), 0, st, dY, lddy, dy_is_f16, (long)M, N, work);
  hipLaunchKernelGGL(colsum_fold_f32_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, work, out, N, CS_RS, alpha);
  ME_TRAIN_CHECK("me_colsum")
}

static int ln_params_blocks(int64_t rows) {
  long b = (rows + 63) / 64;   // >= 64 rows per block
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int64_t me_layernorm_bwd_params_work_bytes(int64_t rows, int32_t C) { return rows > 0 && C > 0 ? (int64_t)ln_params_blocks(rows) * 2 * C * (int64_t)sizeof(float) : 0; }

extern "C" int me_layernorm_bwd_params(float* dgamma, float* dbeta, const void* x, int32_t ldx, const void* dy, int32_t lddy, int64_t rows, int32_t C, float eps, float alpha,
                                       float* work, void* stream) {
  if ((!dgamma && !dbeta) || !x || !dy || !work || rows <= 0 || C <= 0 || C > 1536) { me_set_error("me_layernorm_bwd_params: bad arguments (C <= 1536)"); return ME_EINVAL; }
  const int nb = ln_params_blocks(rows);
  const int rpb = (int)((rows + nb - 1) / nb);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ln_params_part_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const f16*>(x), ldx, reinterpret_cast<const float*>(dy), lddy, (long)rows, C,
                     eps, work, rpb);
  hipLaunchKernelGGL(ln_params_fold_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, work, dgamma, dbeta, C, nb, alpha);
  ME_TRAIN_CHECK("me_layernorm_bwd_params")
}

extern "C" int me_grad_acc(void* dst, int32_t lddst, const void* src, int32_t ldsrc, int32_t src_is_f16, int64_t rows, int32_t cols, float alpha, int32_t pool_h,
                           int32_t pool_w, void* stream) {
  if (!dst || !src || rows <= 0 || cols <= 0 || cols % 4 || lddst % 4 || ldsrc % 4 || (((uintptr_t)dst | (uintptr_t)src) & ((src_is_f16 & 1) ? 7 : 15)) || (src_is_f16 & ~3) || ((uintptr_t)dst & 15)) {
    me_set_error("me_grad_acc: bad arguments (cols and strides multiples of 4, aligned pointers)");
    return ME_EINVAL;
  }
  if ((pool_h > 0) != (pool_w > 0) || (pool_h > 0 && rows % ((int64_t)pool_h * pool_w))) { me_set_error("me_grad_acc: bad pooling geometry"); return ME_EINVAL; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(grad_acc_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<float*>(dst), lddst, src,
                     ldsrc, src_is_f16, (long)rows, cols, alpha, pool_h, pool_w);
  ME_TRAIN_CHECK("me_grad_acc")
}

extern "C" int64_t me_sumsq_work_bytes(void) { return (int64_t)SS_BLOCKS * 2 * (int64_t)sizeof(float); }

extern "C" int me_sumsq_absmax(float* out, const float* x, int64_t n, float* work, void* stream) {
  if (!out || !x || !work || n <= 0) { me_set_error("me_sumsq_absmax: bad arguments"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  hipLaunchKernelGGL(sumsq_part_kernel, dim3(SS_BLOCKS), dim3(256), 0, st, x, (long)n, work);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, st, work, out);
  ME_TRAIN_CHECK("me_sumsq_absmax")
}

extern "C" int me_adamw(float* p, float* m, float* v, const float* g, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_c1,
                        float bias_c2, const float* gnorm_sq, float max_grad_norm, float grad_scale, void* stream) {
  if (!p || !m || !v || !g || n <= 0 || bias_c1 <= 0.f || bias_c2 <= 0.f) { me_set_error("me_adamw: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, m, v, g, (long)n, lr, beta1, beta2, eps, weight_decay,
                     bias_c1, bias_c2, gnorm_sq, max_grad_norm, grad_scale);
  ME_TRAIN_CHECK("me_adamw")
}

extern "C" int me_cast_f16(void* dst, const float* src, int64_t n, void* stream) {
  if (!dst || !src || n <= 0) { me_set_error("me_cast_f16: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(cast_f16_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(dst), src, (long)n);
  ME_TRAIN_CHECK("me_cast_f16")
}

extern "C" int me_cast_rows_f16(void* dst, int32_t lddst, const float* src, int32_t ldsrc, int64_t rows, int32_t cols, int32_t pad_cols, void* stream) {
  if (!dst || !src || rows <= 0 || cols <= 0 || cols % 4 || pad_cols < cols || pad_cols % 4 || lddst % 4 || ldsrc % 4 || lddst < pad_cols || ((uintptr_t)dst & 7) || ((uintptr_t)src & 15)) {
    me_set_error("me_cast_rows_f16: bad arguments (columns and strides multiples of 4, aligned pointers)");
    return ME_EINVAL;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(cast_rows_f16_kernel, dim3(grid_for(rows * (pad_cols / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(dst), lddst, src,
                     ldsrc, (long)rows, cols, pad_cols);
  ME_TRAIN_CHECK("me_cast_rows_f16")
}

extern "C" int me_mse_seed(float* diff, float* d_eps, int32_t ldd, const void* eps_u, int32_t ldu, const void* eps_c, int32_t ldc, const float* x, const float* target, int32_t nb,
                           int32_t C, int32_t frames, int32_t npix, float guidance, float ca, float cb, float coef, void* stream) {
  if (!diff || !d_eps || !eps_u || !target || nb <= 0 || C <= 0 || frames <= 0 || npix <= 0 || ldd < C || ldu < C || (eps_c && ldc < C)) {
    me_set_error("me_mse_seed: bad arguments");
    return ME_EINVAL;
  }
  const long total = (long)nb * C * frames * npix;
  (void)hipGetLastError();
  hipLaunchKernelGGL(mse_seed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), diff, d_eps, ldd,
                     reinterpret_cast<const f16*>(eps_u), ldu, reinterpret_cast<const f16*>(eps_c), ldc, x, target, nb, C, frames, npix, guidance, ca, cb, coef);
  ME_TRAIN_CHECK("me_mse_seed")
}
