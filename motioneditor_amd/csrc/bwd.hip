// Backward (input-gradient) kernels of the element-wise / normalisation operators: what motioneditor_amd/autodiff.py calls for
// the null-text optimisation (reference p2p/null_text_optimization.py:133-166 differentiates through these layers with torch
// autograd).  Gradients travel in fp32 (the tape's buffers; strided views of them arrive with their row stride), activations in
// fp16 as in the forward.  First, straightforward versions: these run a few times per optimisation step on a batch-1 clip and are
// HBM-bound at worst; they have not been tuned.
#include "me_common.h"
#include <stdlib.h>
#include "../../include/motioned.h"

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_hip_error(const char* what, int err);

namespace {

__device__ __forceinline__ float gelu_erf_grad(float g) {   // d/dg [g Phi(g)] = Phi(g) + g phi(g)
  const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752f));
  return cdf + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}

// GEGLU: pre [M][N] fp16 in the packed (16 value | 16 gate) column order, dy [M][N/2] fp32 -> dpre [M][N] fp16
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const f16* __restrict__ pre, int ldp, const float* __restrict__ dy, int lddy, f16* __restrict__ dpre, int ldd,
                                                        long M, int N) {
  const long total = M * (N / 2);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long m = idx / (N / 2);
    const int c = (int)(idx - m * (N / 2));
    const int b = c >> 4, j = c & 15;
    const float val = (float)pre[m * ldp + 32 * b + j], gate = (float)pre[m * ldp + 32 * b + 16 + j];
    const float d = dy[m * lddy + c];
    dpre[m * ldd + 32 * b + j] = (f16)(d * gelu_erf_f(gate));
    dpre[m * ldd + 32 * b + 16 + j] = (f16)(d * val * gelu_erf_grad(gate));
  }
}

// LayerNorm: one wave per row.  xhat = (x - mean) rstd, g = dy gamma, dx = rstd (g - mean(g) - xhat mean(g xhat))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const f16* __restrict__ X, int ldx, const f16* __restrict__ gamma, const float* __restrict__ dY, int lddy,
                                                            float* __restrict__ dX, int lddx, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
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
  float a = 0.f, b = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[c] * (float)gamma[c], xh = ((float)x[c] - mean) * rstd;
    a += g;
    b += g * xh;
  }
  const float ma = wave_sum(a) / (float)C, mb = wave_sum(b) / (float)C;
  float* dx = dX + row * lddx;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[c] * (float)gamma[c], xh = ((float)x[c] - mean) * rstd;
    dx[c] = rstd * (g - ma - xh * mb);
  }
}

// GroupNorm (+ SiLU) backward in three row-parallel passes (the first version gave a whole (sample group, channel group) to ONE block: 32
// blocks for a batch-1 clip, 3.0 ms per call, a third of a null-text iteration):
//   statistics   me_groupnorm_stats, the forward's deterministic fp64 (sum, sum of squares)
//   sums         per row chunk and channel group: A = sum g, B = sum g xhat with g = dy (SiLU') gamma   (gn_bwd_sums_kernel, fixed order)
//   fold + apply dx = rstd (g - A / n - xhat B / n)                                                      (gn_bwd_fold_kernel, gn_bwd_apply_kernel)
// Thread layout as in norm.hip: a thread owns one 16-byte channel vector and walks down the rows of its chunk.
struct GnCoef { float mean[8], rstd[8], gm[8], bt[8]; };

__device__ __forceinline__ void gn_coef(GnCoef& k, const double* __restrict__ stats, const f16* __restrict__ gamma, const f16* __restrict__ beta, int sg, int vc, int cg,
                                        int groups, double inv_cnt, float eps) {
  U128 g8, b8;
  g8.u = ldg128(gamma + vc * 8);
  b8.u = ldg128(beta + vc * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ge = (vc * 8 + e) / cg;
    const double mean_d = stats[((long)sg * groups + ge) * 2 + 0] * inv_cnt;
    const float var = fmaxf((float)(stats[((long)sg * groups + ge) * 2 + 1] * inv_cnt - mean_d * mean_d), 0.f);
    k.mean[e] = (float)mean_d;
    k.rstd[e] = rsqrtf(var + eps);
    k.gm[e] = (float)g8.e[e];
    k.bt[e] = (float)b8.e[e];
  }
}

// g = dL/d(xhat) of one element: dy * gamma, through the SiLU of y = xhat gamma + beta when the forward applied it
__device__ __forceinline__ float gn_g(float dy, float xh, float gm, float bt, int silu) {
  if (silu) {
    const float y = __builtin_fmaf(xh, gm, bt);
    const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
    dy *= sig * (1.0f + y * (1.0f - sig));
  }
  return dy * gm;
}

__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const f16* __restrict__ X, int ldx, const float* __restrict__ dY, int lddy, const f16* __restrict__ gamma,
                                                          const f16* __restrict__ beta, const double* __restrict__ stats, float2* __restrict__ part, int rows_per_group,
                                                          int chunk_rows, int C, int groups, float eps, int silu) {
  extern __shared__ __attribute__((aligned(16))) float2 red[];   // [row lane][channel of the pass]
  __shared__ float2 chan[2560];
  const int tid = threadIdx.x;
  const int sg = blockIdx.y;
  const int r0 = blockIdx.x * chunk_rows, r1 = min(r0 + chunk_rows, rows_per_group);
  const int tpr = C / 8, tprc = tpr < 256 ? tpr : 256;
  const int rl = tid / tprc, vc0 = tid - rl * tprc, RL = 256 / tprc;
  const int cg = C / groups;
  const double inv_cnt = 1.0 / ((double)rows_per_group * (double)cg);
  const long base_row = (long)sg * rows_per_group;
  for (int pass0 = 0; pass0 < tpr; pass0 += tprc) {
    const int vc = pass0 + vc0;
    if (rl < RL && vc < tpr) {
      GnCoef k;
      gn_coef(k, stats, gamma, beta, sg, vc, cg, groups, inv_cnt, eps);
      float sa[8], sb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
#pragma unroll 2
      for (int r = r0 + rl; r < r1; r += RL) {
        U128 u;
        u.u = ldg128(X + (base_row + r) * ldx + vc * 8);
        const float* dp = dY + (base_row + r) * lddy + vc * 8;
        const float4 d0 = *reinterpret_cast<const float4*>(dp), d1 = *reinterpret_cast<const float4*>(dp + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = ((float)u.e[e] - k.mean[e]) * k.rstd[e];
          const float g = gn_g(dv[e], xh, k.gm[e], k.bt[e], silu);
          sa[e] += g;
          sb[e] += g * xh;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rl * (tprc * 8) + vc0 * 8 + e] = make_float2(sa[e], sb[e]);
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
    part[((long)blockIdx.x * gridDim.y + sg) * groups + tid] = a;
  }
}

// sums[sg][g] = (A / n, B / n): one block per (sample group, channel group); thread t adds chunks t, t + 256, ... in fp64, then a fixed binary tree
// (the first version gave each (sg, g) to ONE thread of a single block: 77 us per call for a batch-1 clip, 61 calls per null-text iteration)
__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(const float2* __restrict__ part, float2* __restrict__ sums, int chunks, int nsg, int groups, double inv_cnt) {
  __shared__ double ra[256], rb[256];
  const int idx = blockIdx.x;   // (sg, g)
  const int tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int c = tid; c < chunks; c += 256) {
    const float2 p = part[(long)c * nsg * groups + idx];
    a += (double)p.x;
    b += (double)p.y;
  }
  ra[tid] = a;
  rb[tid] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      ra[tid] += ra[tid + s];
      rb[tid] += rb[tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) sums[idx] = make_float2((float)(ra[0] * inv_cnt), (float)(rb[0] * inv_cnt));
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const f16* __restrict__ X, int ldx, const float* __restrict__ dY, int lddy, float* __restrict__ dX, int lddx,
                                                           const f16* __restrict__ gamma, const f16* __restrict__ beta, const double* __restrict__ stats,
                                                           const float2* __restrict__ sums, long rows, int rows_per_group, int C, int groups, float eps, int silu, int chunk) {
  const int vc = blockIdx.y * blockDim.x + threadIdx.x;   // 16-byte vector column
  const int cg = C / groups;
  const double inv_cnt = 1.0 / ((double)rows_per_group * (double)cg);
  const long r0 = (long)blockIdx.x * chunk;
  const long r1 = r0 + chunk < rows ? r0 + chunk : rows;
  long row = r0 + threadIdx.y;
  while (row < r1) {
    const int sg = (int)(row / rows_per_group);
    const long seg_end = (long)(sg + 1) * rows_per_group < r1 ? (long)(sg + 1) * rows_per_group : r1;
    GnCoef k;
    gn_coef(k, stats, gamma, beta, sg, vc, cg, groups, inv_cnt, eps);
    float ma[8], mb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 sm = sums[(long)sg * groups + (vc * 8 + e) / cg];
      ma[e] = sm.x;
      mb[e] = sm.y;
    }
    for (; row < seg_end; row += blockDim.y) {
      U128 u;
      u.u = ldg128(X + row * ldx + vc * 8);
      const float* dp = dY + row * lddy + vc * 8;
      const float4 d0 = *reinterpret_cast<const float4*>(dp), d1 = *reinterpret_cast<const float4*>(dp + 4);
      const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = ((float)u.e[e] - k.mean[e]) * k.rstd[e];
        const float g = gn_g(dv[e], xh, k.gm[e], k.bt[e], silu);
        o[e] = k.rstd[e] * (g - ma[e] - xh * mb[e]);
      }
      float* op = dX + row * lddx + vc * 8;
      *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}


// Temporal causal attention backward: one block per (batch row, pixel, head); F <= 64 frames, dh <= 160.  The head's q, k, v, dO rows
// (F x dh each) sit in LDS as fp32; thread i < F owns query row i for the score pass (P, dS into LDS) and dQ, and key row i for
// dK / dV.  P = softmax(causal(s q.k)), dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(P dP)), dQ = s dS K, dK = s dS^T Q.
__global__ __launch_bounds__(64) void tattn_bwd_kernel(const f16* __restrict__ Q, int ldq, const f16* __restrict__ K, int ldk, const f16* __restrict__ V, int ldv,
                                                      const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq, float* __restrict__ dK, int lddk,
                                                      float* __restrict__ dV, int lddv, int batch, int F, int npix, int heads, int dh, float scale) {
  extern __shared__ float sm[];
  float* sq = sm;                 // [F][dh]
  float* sk = sq + F * dh;
  float* sv = sk + F * dh;
  float* sdo = sv + F * dh;
  float* sp = sdo + F * dh;       // [F][F]
  float* sds = sp + F * F;        // [F][F]
  int bid = blockIdx.x;
  const int h = bid % heads;
  bid /= heads;
  const int p = bid % npix;
  const int b = bid / npix;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < F * dh; idx += 64) {
    const int j = idx / dh, d = idx - j * dh;
    const long row = ((long)b * F + j) * npix + p;
    sq[idx] = (float)Q[row * ldq + h * dh + d];
    sk[idx] = (float)K[row * ldk + h * dh + d];
    sv[idx] = (float)V[row * ldv + h * dh + d];
    sdo[idx] = dO[row * lddo + h * dh + d];
  }
  __syncthreads();
  if (tid < F) {
    const int i = tid;
    float mx = -1.0e30f;
    for (int j = 0; j <= i; ++j) {
      float acc = 0.f;
      for (int d = 0; d < dh; ++d) acc += sq[i * dh + d] * sk[j * dh + d];
      sp[i * F + j] = acc * scale;
      mx = fmaxf(mx, acc * scale);
    }
    float l = 0.f;
    for (int j = 0; j <= i; ++j) {
      const float e = __expf(sp[i * F + j] - mx);
      sp[i * F + j] = e;
      l += e;
    }
    const float inv = 1.0f / l;
    float delta = 0.f;
    for (int j = 0; j < F; ++j) {
      float pij = 0.f, dp = 0.f;
      if (j <= i) {
        pij = sp[i * F + j] * inv;
        for (int d = 0; d < dh; ++d) dp += sdo[i * dh + d] * sv[j * dh + d];
      }
      sp[i * F + j] = pij;
      sds[i * F + j] = dp;          // dP for now
      delta += pij * dp;
    }
    for (int j = 0; j < F; ++j) sds[i * F + j] = sp[i * F + j] * (sds[i * F + j] - delta) * scale;
  }
  __syncthreads();
  if (tid < F) {
    const int i = tid;
    const long row = ((long)b * F + i) * npix + p;
    for (int d = 0; d < dh; ++d) {
      float aq = 0.f, ak = 0.f, av = 0.f;
      for (int j = 0; j < F; ++j) {
        aq += sds[i * F + j] * sk[j * dh + d];      // dQ_i = sum_j dS_ij K_j
        ak += sds[j * F + i] * sq[j * dh + d];      // dK_i = sum_j dS_ji Q_j
        av += sp[j * F + i] * sdo[j * dh + d];      // dV_i = sum_j P_ji dO_j
      }
      dQ[row * lddq + h * dh + d] = aq;
      dK[row * lddk + h * dh + d] = ak;
      dV[row * lddv + h * dh + d] = av;
    }
  }
}


// ---- temporal attention backward, lane-parallel form (frames <= 32, head dim 40 / 80 / 160) ----
// The first version above runs one (pixel, head) per 64-thread block with 24 busy lanes, scalar loops and per-element strided global loads: 1.6-2.0 ms
// per call where the forward takes 0.08.  Here a block owns ONE pixel and HB heads (HB * DH = 320 columns: every frame's row segment is 640 contiguous
// bytes), stages q, k, v and the fp16-rounded dO of all frames in LDS with 16-byte loads, and each wave takes one head:
//   lane (i, half) -- frame i = lane / 2 -- keeps row i of q and dO in registers and computes S[i][j], dP[i][j] for the j of its half with v_dot2 against
//   k_j / v_j read from LDS (two distinct addresses per wave instruction: broadcasts); softmax, delta and dS across the lane pair; P and dS go to LDS
//   [i][j]; then dQ_i (this lane's half of the head dim) = sum_j dS[i][j] k_j, and with the lane now standing for KEY frame j, dK_j = sum_i dS[i][j] q_i,
//   dV_j = sum_i P[i][j] dO_i.  Causal: key frame <= query frame.  Results are written (=), as the first version does.
template <int DH, int HB>
__global__ __launch_bounds__(HB * 64) void tattn_bwd2_kernel(const f16* __restrict__ Q, int ldq, const f16* __restrict__ K, int ldk, const f16* __restrict__ V, int ldv,
                                                             const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq, float* __restrict__ dK, int lddk,
                                                             float* __restrict__ dV, int lddv, int batch, int F, int npix, int heads, float scale) {
  constexpr int CH = DH / 8, DH2 = DH / 2, C4 = DH2 / 4;
  const int FP = F | 1;   // row pitch of the [i][j] tiles (odd: column reads spread over the banks)
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  f16* sq = reinterpret_cast<f16*>(smraw);   // [HB][F][DH]
  f16* sk = sq + HB * F * DH;
  f16* sv = sk + HB * F * DH;
  f16* sd = sv + HB * F * DH;
  float* sp = reinterpret_cast<float*>(sd + HB * F * DH);   // [HB][F][FP]
  float* sds = sp + HB * F * FP;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int ng = heads / HB;
  const int hg = bid % ng;
  bid /= ng;
  const int p = bid % npix;
  const int b = bid / npix;
  const int col0 = hg * HB * DH;
  // stage: chunk (frame j, 16-byte column chunk c) of the HB * DH columns
  for (int idx = tid; idx < F * HB * CH; idx += HB * 64) {
    const int j = idx / (HB * CH), c = idx - j * (HB * CH);
    const long row = ((long)b * F + j) * npix + p;
    const int hh = c / CH, cc = c - hh * CH;
    const int o = (hh * F + j) * DH + cc * 8;
    *reinterpret_cast<uint4*>(sq + o) = ldg128(Q + row * ldq + col0 + c * 8);
    *reinterpret_cast<uint4*>(sk + o) = ldg128(K + row * ldk + col0 + c * 8);
    *reinterpret_cast<uint4*>(sv + o) = ldg128(V + row * ldv + col0 + c * 8);
    const float* g = dO + row * lddo + col0 + c * 8;
    const float4 a = *reinterpret_cast<const float4*>(g), b4 = *reinterpret_cast<const float4*>(g + 4);
    U128 u;
    u.e[0] = (f16)a.x; u.e[1] = (f16)a.y; u.e[2] = (f16)a.z; u.e[3] = (f16)a.w;
    u.e[4] = (f16)b4.x; u.e[5] = (f16)b4.y; u.e[6] = (f16)b4.z; u.e[7] = (f16)b4.w;
    *reinterpret_cast<uint4*>(sd + o) = u.u;
  }
  __syncthreads();
  const int hh = tid >> 6, lane = tid & 63;
  const int i = lane >> 1, half = lane & 1;
  const bool act = i < F;
  const int ic = act ? i : 0;
  const f16* hq = sq + hh * F * DH;
  const f16* hk = sk + hh * F * DH;
  const f16* hv = sv + hh * F * DH;
  const f16* hd = sd + hh * F * DH;
  float* hp = sp + hh * F * FP;
  float* hs = sds + hh * F * FP;
  const int JH = (F + 1) / 2;   // keys per lane of a pair: j = half * JH + jj
  {
    // ---- S, dP for this lane's keys; softmax / delta / dS across the pair ----
    U128 qi[CH], di[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      qi[c].u = *reinterpret_cast<const uint4*>(hq + ic * DH + c * 8);
      di[c].u = *reinterpret_cast<const uint4*>(hd + ic * DH + c * 8);
    }
    float sv_[16], dp_[16];
    float mx = -1.0e30f;
    const float cs = scale * 1.4426950408889634f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      sv_[jj] = -1.0e30f;
      dp_[jj] = 0.f;
      if (jj < JH) {   // uniform
        const int j = half * JH + jj;
        const int jc = j < F ? j : 0;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          U128 kk, vv;
          kk.u = *reinterpret_cast<const uint4*>(hk + jc * DH + c * 8);
          vv.u = *reinterpret_cast<const uint4*>(hv + jc * DH + c * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f16x2 q2 = {qi[c].e[2 * e], qi[c].e[2 * e + 1]}, k2 = {kk.e[2 * e], kk.e[2 * e + 1]};
            const f16x2 d2 = {di[c].e[2 * e], di[c].e[2 * e + 1]}, v2 = {vv.e[2 * e], vv.e[2 * e + 1]};
            s = __builtin_amdgcn_fdot2(q2, k2, s, false);
            dp = __builtin_amdgcn_fdot2(d2, v2, dp, false);
          }
        }
        if (act && j <= i && j < F) {
          sv_[jj] = s * cs;
          dp_[jj] = dp;
          mx = fmaxf(mx, s * cs);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float e = sv_[jj] > -1.0e29f ? __builtin_amdgcn_exp2f(sv_[jj] - mx) : 0.f;
      sv_[jj] = e;
      l += e;
    }
    l += __shfl_xor(l, 1, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      sv_[jj] *= inv;
      delta += sv_[jj] * dp_[jj];
    }
    delta += __shfl_xor(delta, 1, 64);
    if (act) {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = half * JH + jj;
        if (jj < JH && j < F) {
          hp[i * FP + j] = sv_[jj];
          hs[i * FP + j] = sv_[jj] * (dp_[jj] - delta) * scale;
        }
      }
    }
  }
  __syncthreads();   // (the tiles of a head are written and read by one wave; the barrier is the simple way to order them)
  const int d0 = half * DH2;
  if (act) {
    // ---- dQ_i[d0 .. d0 + DH2) = sum_j dS[i][j] k_j ----
    float acc[DH2];
#pragma unroll
    for (int d = 0; d < DH2; ++d) acc[d] = 0.f;
    for (int j = 0; j <= i; ++j) {
      const float ds = hs[i * FP + j];
#pragma unroll
      for (int c = 0; c < C4; ++c) {
        U64 kk;
        kk.u = *reinterpret_cast<const uint2*>(hk + j * DH + d0 + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c * 4 + e] = fmaf(ds, (float)kk.e[e], acc[c * 4 + e]);
      }
    }
    float* o = dQ + (((long)b * F + i) * npix + p) * lddq + col0 + hh * DH + d0;
#pragma unroll
    for (int c = 0; c < C4; ++c) *reinterpret_cast<float4*>(o + c * 4) = make_float4(acc[c * 4], acc[c * 4 + 1], acc[c * 4 + 2], acc[c * 4 + 3]);
  }
  if (act) {
    // ---- the lane now stands for KEY frame j = i: dK_j = sum_{i' >= j} dS[i'][j] q_i',  dV_j = sum_{i' >= j} P[i'][j] dO_i' ----
    float ak[DH2], av[DH2];
#pragma unroll
    for (int d = 0; d < DH2; ++d) { ak[d] = 0.f; av[d] = 0.f; }
    for (int r = i; r < F; ++r) {
      const float ds = hs[r * FP + i], pp = hp[r * FP + i];
#pragma unroll
      for (int c = 0; c < C4; ++c) {
        U64 qq, dd;
        qq.u = *reinterpret_cast<const uint2*>(hq + r * DH + d0 + c * 4);
        dd.u = *reinterpret_cast<const uint2*>(hd + r * DH + d0 + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ak[c * 4 + e] = fmaf(ds, (float)qq.e[e], ak[c * 4 + e]);
          av[c * 4 + e] = fmaf(pp, (float)dd.e[e], av[c * 4 + e]);
        }
      }
    }
    const long row = ((long)b * F + i) * npix + p;
    float* ok = dK + row * lddk + col0 + hh * DH + d0;
    float* ov = dV + row * lddv + col0 + hh * DH + d0;
#pragma unroll
    for (int c = 0; c < C4; ++c) {
      *reinterpret_cast<float4*>(ok + c * 4) = make_float4(ak[c * 4], ak[c * 4 + 1], ak[c * 4 + 2], ak[c * 4 + 3]);
      *reinterpret_cast<float4*>(ov + c * 4) = make_float4(av[c * 4], av[c * 4 + 1], av[c * 4 + 2], av[c * 4 + 3]);
    }
  }
}

template <int DH, int HB>
static int launch_tattn_bwd2(void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout,
                             int lddo, int batch, int frames, int npix, int heads, float scale, hipStream_t st) {
  const size_t lds = (size_t)4 * HB * frames * DH * sizeof(f16) + (size_t)2 * HB * frames * (frames | 1) * sizeof(float);
  static bool attr_set_dev[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!attr_set_dev[dev_id & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_bwd2_kernel<DH, HB>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
      me_set_error("me_tattn_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return ME_EHIP;
    }
    attr_set_dev[dev_id & 63] = true;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL((tattn_bwd2_kernel<DH, HB>), dim3((unsigned)((long)batch * npix * (heads / HB))), dim3(HB * 64), lds, st, reinterpret_cast<const f16*>(q), ldq,
                     reinterpret_cast<const f16*>(k), ldk, reinterpret_cast<const f16*>(v), ldv, reinterpret_cast<const float*>(dout), lddo, reinterpret_cast<float*>(dq), lddq,
                     reinterpret_cast<float*>(dk), lddk, reinterpret_cast<float*>(dv), lddv, batch, frames, npix, heads, scale);
  const hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) { me_set_hip_error("me_tattn_bwd", (int)e_); return ME_EHIP; }
  return ME_OK;
}

// Row softmax backward: dS = P * (dP - sum_j P_j dP_j) * scale, one wave per row (cols a multiple of 8).  Used by the first,
// matrix-materialising form of the spatial attention backward (ops.attention_bwd).
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const f16* __restrict__ P, int ldp, const f16* __restrict__ dP, int lddp, f16* __restrict__ dS, int ldds,
                                                              long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f16* p = P + row * ldp;
  const f16* dp = dP + row * lddp;
  float acc = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    U128 a, b;
    a.u = ldg128(p + c);
    b.u = ldg128(dp + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)a.e[e] * (float)b.e[e];
  }
  const float delta = wave_sum(acc);
  f16* ds = dS + row * ldds;
  for (int c = lane * 8; c < cols; c += 512) {
    U128 a, b, o;
    a.u = ldg128(p + c);
    b.u = ldg128(dp + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = (f16)((float)a.e[e] * ((float)b.e[e] - delta) * scale);
    *reinterpret_cast<uint4*>(ds + c) = o.u;
  }
}


// ReLU epilogue backward: dx = dy where the forward output was positive (fp32 gradient views, fp16 output of the forward)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, int lddy, const f16* __restrict__ out, int ldo, float* __restrict__ dx, int lddx,
                                                       long rows, int cols) {
  const long total = rows * cols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / cols;
    const int c = (int)(idx - r * cols);
    dx[r * lddx + c] = (float)out[r * ldo + c] > 0.f ? dy[r * lddy + c] : 0.f;
  }
}

}  // namespace

#define ME_BWD_LAUNCH_CHECK(name)                                                   \
  {                                                                                 \
    const hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) { me_set_hip_error(name, (int)e_); return ME_EHIP; }      \
    return ME_OK;                                                                   \
  }

extern "C" int me_geglu_bwd(void* dpre, int32_t ldd, const void* pre, int32_t ldp, const void* dy, int32_t lddy, int64_t M, int32_t N, void* stream) {
  if (!dpre || !pre || !dy || M <= 0 || N <= 0 || N % 32) { me_set_error("me_geglu_bwd: bad arguments (N must be a multiple of 32)"); return ME_EINVAL; }
  (void)hipGetLastError();
  const long total = (long)M * (N / 2);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536L * 16 ? (total + 255) / 256 : 65536L * 16);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(pre), ldp,
                     reinterpret_cast<const float*>(dy), lddy, reinterpret_cast<f16*>(dpre), ldd, (long)M, N);
  ME_BWD_LAUNCH_CHECK("me_geglu_bwd")
}

extern "C" int me_layernorm_bwd(void* dx, int32_t lddx, const void* x, int32_t ldx, const void* gamma, const void* dy, int32_t lddy, int64_t rows, int32_t C, float eps,
                                void* stream) {
  if (!dx || !x || !gamma || !dy || rows <= 0 || C <= 0) { me_set_error("me_layernorm_bwd: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(x), ldx,
                     reinterpret_cast<const f16*>(gamma), reinterpret_cast<const float*>(dy), lddy, reinterpret_cast<float*>(dx), lddx, (long)rows, C, eps);
  ME_BWD_LAUNCH_CHECK("me_layernorm_bwd")
}

// chunk geometry of the row-parallel passes (same rule as norm.hip's statistics pass: ~1024 blocks in total)
static void gn_bwd_chunks(int64_t rows, int rows_per_group, int* chunks_out, int* chunk_rows_out) {
  const int nsg = (int)(rows / rows_per_group);
  int chunks = 1024 / nsg;
  if (chunks > 512) chunks = 512;
  if (chunks < 1) chunks = 1;
  int chunk_rows = (rows_per_group + chunks - 1) / chunks;
  if (chunk_rows < 8) chunk_rows = 8;
  *chunks_out = (rows_per_group + chunk_rows - 1) / chunk_rows;
  *chunk_rows_out = chunk_rows;
}

extern "C" int64_t me_groupnorm_bwd_scratch_bytes(int32_t rows, int32_t rows_per_group, int32_t groups) {
  if (rows <= 0 || rows_per_group <= 0 || rows % rows_per_group || groups <= 0) return 0;
  int chunks, chunk_rows;
  gn_bwd_chunks(rows, rows_per_group, &chunks, &chunk_rows);
  const int64_t nsg = rows / rows_per_group;
  const int64_t fwd = (me_groupnorm_scratch_bytes(rows, rows_per_group, groups) + 15) / 16 * 16;
  return fwd + ((int64_t)chunks + 1) * nsg * groups * (int64_t)sizeof(float2);
}

extern "C" int me_groupnorm_bwd(void* dx, int32_t lddx, const void* x, int32_t ldx, const void* gamma, const void* beta, const void* dy, int32_t lddy, int64_t rows,
                                int32_t rows_per_group, int32_t C, int32_t groups, float eps, int32_t silu, void* scratch, void* stream) {
  if (!dx || !x || !gamma || !beta || !dy || !scratch || rows <= 0 || rows_per_group <= 0 || rows % rows_per_group || groups <= 0 || groups > 64 || C % groups || C % 8 ||
      C > 2560 || ldx % 8 || lddx % 4 || lddy % 4 || (((uintptr_t)dx | (uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)scratch) & 15)) {
    me_set_error("me_groupnorm_bwd: bad arguments (C % 8 == 0, C <= 2560, aligned pointers, strides multiples of 8 / 4)");
    return ME_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsg = (int)(rows / rows_per_group);
  // statistics: the forward's deterministic pass
  me_groupnorm_args fa{};
  fa.X = x;
  fa.Y = const_cast<void*>(x);   // not written by the statistics pass
  fa.gamma = gamma;
  fa.beta = beta;
  fa.stats = scratch;
  fa.rows = (int32_t)rows;
  fa.rows_per_group = rows_per_group;
  fa.C = C;
  fa.ldx = ldx;
  fa.ldy = ldx;
  fa.groups = groups;
  fa.eps = eps;
  if (int rc = me_groupnorm_stats(&fa, stream)) return rc;
  const double* stats = reinterpret_cast<const double*>(scratch);
  const int64_t fwd = (me_groupnorm_scratch_bytes((int32_t)rows, rows_per_group, groups) + 15) / 16 * 16;
  float2* part = reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + fwd);
  int chunks, chunk_rows;
  gn_bwd_chunks(rows, rows_per_group, &chunks, &chunk_rows);
  float2* sums = part + (size_t)chunks * nsg * groups;
  const int tpr = C / 8, tprc = tpr < 256 ? tpr : 256;
  const size_t lds = (size_t)(256 / tprc) * tprc * 8 * sizeof(float2);
  (void)hipGetLastError();
  hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(chunks, nsg), dim3(256), lds, st, reinterpret_cast<const f16*>(x), ldx, reinterpret_cast<const float*>(dy), lddy,
                     reinterpret_cast<const f16*>(gamma), reinterpret_cast<const f16*>(beta), stats, part, rows_per_group, chunk_rows, C, groups, eps, silu);
  hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(nsg * groups), dim3(256), 0, st, part, sums, chunks, nsg, groups,
                     1.0 / ((double)rows_per_group * (double)(C / groups)));
  int ny = 1;
  while (tpr / ny > 256 || tpr % ny) ++ny;
  const int bx = tpr / ny, by = 256 / bx > 0 ? 256 / bx : 1;
  long chunk = (long)by * 16;
  while (chunk > by && (rows + chunk - 1) / chunk * ny < 2048) chunk /= 2;
  if (chunk < by) chunk = by;
  const long nbx = (rows + chunk - 1) / chunk;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)nbx, (unsigned)ny), dim3(bx, by), 0, st, reinterpret_cast<const f16*>(x), ldx, reinterpret_cast<const float*>(dy), lddy,
                     reinterpret_cast<float*>(dx), lddx, reinterpret_cast<const f16*>(gamma), reinterpret_cast<const f16*>(beta), stats, sums, (long)rows, rows_per_group, C, groups,
                     eps, silu, (int)chunk);
  ME_BWD_LAUNCH_CHECK("me_groupnorm_bwd")
}

extern "C" int me_tattn_bwd(void* dq, int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
                            int32_t ldv, const void* dout, int32_t lddo, int32_t batch, int32_t frames, int32_t npix, int32_t heads, int32_t dh, float scale, void* stream) {
  if (!dq || !dk || !dv || !q || !k || !v || !dout || batch <= 0 || frames <= 0 || frames > 64 || npix <= 0 || heads <= 0 || dh <= 0) {
    me_set_error("me_tattn_bwd: bad arguments (frames <= 64)");
    return ME_EINVAL;
  }
  {   // the lane-parallel kernel: frames <= 32, 16-byte aligned rows, heads a multiple of the heads per block
    static const bool v2 = !(getenv("ME_TATTN_BWD2") && atoi(getenv("ME_TATTN_BWD2")) == 0);
    const bool al = !((ldq | ldk | ldv) % 8) && !((lddo | lddq | lddk | lddv) % 4) &&
                    !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15);
    hipStream_t st2 = reinterpret_cast<hipStream_t>(stream);
    if (v2 && al && frames <= 32) {
      if (dh == 40 && heads % 8 == 0) return launch_tattn_bwd2<40, 8>(dq, lddq, dk, lddk, dv, lddv, q, ldq, k, ldk, v, ldv, dout, lddo, batch, frames, npix, heads, scale, st2);
      if (dh == 80 && heads % 4 == 0) return launch_tattn_bwd2<80, 4>(dq, lddq, dk, lddk, dv, lddv, q, ldq, k, ldk, v, ldv, dout, lddo, batch, frames, npix, heads, scale, st2);
      if (dh == 160 && heads % 2 == 0) return launch_tattn_bwd2<160, 2>(dq, lddq, dk, lddk, dv, lddv, q, ldq, k, ldk, v, ldv, dout, lddo, batch, frames, npix, heads, scale, st2);
    }
  }
  const size_t lds = ((size_t)4 * frames * dh + (size_t)2 * frames * frames) * sizeof(float);
  if (lds > 150 * 1024) { me_set_error("me_tattn_bwd: frames * head dim too large for the LDS tile"); return ME_EINVAL; }
  if (lds > 48 * 1024) {   // per device: the attribute raises the dynamic LDS limit of this kernel
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (!attr_set_dev[dev_id & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        me_set_error("me_tattn_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return ME_EHIP;
      }
      attr_set_dev[dev_id & 63] = true;
    }
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(tattn_bwd_kernel, dim3((unsigned)((long)batch * npix * heads)), dim3(64), lds, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(q), ldq,
                     reinterpret_cast<const f16*>(k), ldk, reinterpret_cast<const f16*>(v), ldv, reinterpret_cast<const float*>(dout), lddo, reinterpret_cast<float*>(dq), lddq,
                     reinterpret_cast<float*>(dk), lddk, reinterpret_cast<float*>(dv), lddv, batch, frames, npix, heads, dh, scale);
  ME_BWD_LAUNCH_CHECK("me_tattn_bwd")
}

extern "C" int me_softmax_bwd_rows(void* dS, int32_t ldds, const void* P, int32_t ldp, const void* dP, int32_t lddp, int64_t rows, int32_t cols, float scale, void* stream) {
  if (!dS || !P || !dP || rows <= 0 || cols <= 0 || cols % 8 || ldds % 8 || ldp % 8 || lddp % 8 || (((uintptr_t)dS | (uintptr_t)P | (uintptr_t)dP) & 15)) {
    me_set_error("me_softmax_bwd_rows: bad arguments (cols and strides multiples of 8, 16-byte aligned)");
    return ME_EINVAL;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(P), ldp,
                     reinterpret_cast<const f16*>(dP), lddp, reinterpret_cast<f16*>(dS), ldds, (long)rows, cols, scale);
  ME_BWD_LAUNCH_CHECK("me_softmax_bwd_rows")
}

extern "C" int me_relu_bwd(void* dx, int32_t lddx, const void* dy, int32_t lddy, const void* out, int32_t ldo, int64_t rows, int32_t cols, void* stream) {
  if (!dx || !dy || !out || rows <= 0 || cols <= 0) { me_set_error("me_relu_bwd: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();
  const long total = (long)rows * cols;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536L * 16 ? (total + 255) / 256 : 65536L * 16);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float*>(dy), lddy,
                     reinterpret_cast<const f16*>(out), ldo, reinterpret_cast<float*>(dx), lddx, (long)rows, cols);
  ME_BWD_LAUNCH_CHECK("me_relu_bwd")
}
