// Fused (flash-style) backward of me_attn for PLAIN key segments ([prev | cur], self, text, [first | prev]) on MFMA (gfx950).
//
// The reference differentiates these layers with torch autograd (p2p/null_text_optimization.py:133-166 through
// attention_2d.py:705-768 / :115-201; train_adaptor.py:364-368 through controlnet_adapter.py:332-407).  Scores are never
// materialised: P is rebuilt per tile from the log-sum-exp the forward stashed (me_attn_args.lse),
//     P = 2^(s c - lse),  dP = dO V^T,  dS = P o (dP - delta),  delta_q = sum_d dO[q, d] O[q, d]
//     dV = P^T dO,  dK = scale dS^T Q,  dQ = scale dS K
// in two deterministic kernels (no atomics; every output element has exactly one owner and is ACCUMULATED into, +=):
//   * attn_bwd_dkv_kernel, key-centric: a block owns 64 * NKT keys of one (kv item, head) -- their K / V fragments live in
//     registers as MFMA operand B -- and walks over every query item that lists the kv item (an inverse segment table in
//     CSR form), 32 queries per stage.  S[q, key] = Q K^T and dP[q, key] = dO V^T leave the MFMA with the lane holding
//     4 consecutive queries of one key; packed to fp16 they ARE operand B of dV^T[d, key] += dO^T[d, q] P[q, key] and
//     dK^T[d, key] += Q^T[d, q] dS[q, key] (contraction over the 32 queries; k-slot (g, j) = query (j >> 2) * 16 + g * 4 + (j & 3)
//     for both operands), whose operand A is the transposed Q / dO tile staged in that slot order.
//   * attn_bwd_dq_kernel, query-centric: the forward's structure (a wave owns 16 queries, K / V tiles of 64 keys through LDS),
//     S^T = K Q^T and dP^T = V dO^T with the query on the lane, dS^T packed as operand B of dQ^T[d, q] += K^T[d, key] dS^T[key, q].
// Splitting costs two extra matrix products (S and dP are computed in both kernels: 7 instead of 5) and buys determinism and
// single-owner outputs.  Activations fp16, gradients fp32 in memory (loss-scaled by the caller), fp16 as MFMA operands.
#include "me_common.h"
#include <stdlib.h>
#include "../../include/motioned.h"

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_hip_error(const char* what, int err);
extern "C" void me_set_kernel(const char* name);

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// delta[row][h] = sum_d dO[row][h dh + d] * O[row][h dh + d], with dO rounded to fp16 first -- the value the MFMAs multiply: dP - delta is
// then a difference of sums over the SAME rounded factors, so a query whose softmax is a single 1 (one key: the 1x1-pixel level) gets
// dS = 0 up to fp32 summation order instead of up to the fp16 rounding of dO (its q / k gradients are exactly zero under autograd)
__global__ __launch_bounds__(256) void attn_delta_kernel(const f16* __restrict__ O, int ldo, const float* __restrict__ dO, int lddo, float* __restrict__ delta,
                                                         long rows, int heads, int dh) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // (row, head)
  if (idx >= rows * heads) return;
  const long row = idx / heads;
  const int h = (int)(idx - row * heads);
  const f16* o = O + row * ldo + h * dh;
  const float* d = dO + row * lddo + h * dh;
  float acc = 0.f;
  for (int c = 0; c < dh; c += 8) {
    U128 u;
    u.u = ldg128(o + c);
    const float4 a = *reinterpret_cast<const float4*>(d + c), b = *reinterpret_cast<const float4*>(d + c + 4);
    acc += (float)u.e[0] * (float)(f16)a.x + (float)u.e[1] * (float)(f16)a.y + (float)u.e[2] * (float)(f16)a.z + (float)u.e[3] * (float)(f16)a.w;
    acc += (float)u.e[4] * (float)(f16)b.x + (float)u.e[5] * (float)(f16)b.y + (float)u.e[6] * (float)(f16)b.z + (float)u.e[7] * (float)(f16)b.w;
  }
  delta[idx] = acc;
}

__device__ __forceinline__ uint4 f32x8_to_f16(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  U128 u;
  u.e[0] = (f16)a.x; u.e[1] = (f16)a.y; u.e[2] = (f16)a.z; u.e[3] = (f16)a.w;
  u.e[4] = (f16)b.x; u.e[5] = (f16)b.y; u.e[6] = (f16)b.z; u.e[7] = (f16)b.w;
  return u.u;
}

// ---- dK, dV: block = (kv item, head, 64 * NKT keys); wave w owns keys [w * 16 * NKT, +16 * NKT) of the block's range ----
template <int DH, int NKT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const me_attn_bwd_args a) {
  constexpr int D32 = (DH + 31) / 32, DT = (DH + 15) / 16, CH = DH / 8;
  constexpr int QB = 32;                  // queries per stage
  constexpr int RLD = D32 * 32 + 8;       // row-major tiles [query][d], halves
  constexpr int TLD = QB + 8;             // transposed tiles [d][query slot], halves (80-byte rows: 16-byte aligned fragment reads)
  __shared__ __attribute__((aligned(16))) f16 sQ[QB * RLD];
  __shared__ __attribute__((aligned(16))) f16 sdO[QB * RLD];
  __shared__ __attribute__((aligned(16))) f16 sQt[DT * 16 * TLD];
  __shared__ __attribute__((aligned(16))) f16 sdOt[DT * 16 * TLD];
  __shared__ __attribute__((aligned(16))) float slse[QB];
  __shared__ __attribute__((aligned(16))) float sdel[QB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  const int nkb = (a.nk + 64 * NKT - 1) / (64 * NKT);
  int w = blockIdx.x;
  const int kb = w % nkb;
  w /= nkb;
  const int h = w % a.heads;
  const int kit = w / a.heads;

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  const float* __restrict__ dO = reinterpret_cast<const float*>(a.dO);
  const float* __restrict__ lse = reinterpret_cast<const float*>(a.lse);
  const float* __restrict__ delta = reinterpret_cast<const float*>(a.delta);

  // zero the LDS columns / rows the staging never writes (d >= dh)
  for (int i = tid; i < QB * RLD; i += 256) { sQ[i] = (f16)0.f; sdO[i] = (f16)0.f; }
  for (int i = tid; i < DT * 16 * TLD; i += 256) { sQt[i] = (f16)0.f; sdOt[i] = (f16)0.f; }

  // this wave's keys: K and V fragments as MFMA operand B (lane (key = l15, g) holds row[key][ks * 32 + g * 8 .. + 8])
  const int key0 = kb * 64 * NKT + wave * 16 * NKT;
  f16x8 fk[NKT][D32], fv[NKT][D32];
  bool kvalid[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int key = key0 + j * 16 + l15;
    kvalid[j] = key < a.nk;
    const long row = (long)kit * a.nk + (kvalid[j] ? key : 0);
#pragma unroll
    for (int ks = 0; ks < D32; ++ks) {
      const int d = ks * 32 + g * 8;
      U128 uk, uv;
      uk.u = (kvalid[j] && d < DH) ? ldg128(K + row * a.ldk + h * DH + d) : zero128();
      uv.u = (kvalid[j] && d < DH) ? ldg128(V + row * a.ldv + h * DH + d) : zero128();
      fk[j][ks] = uk.h;
      fv[j][ks] = uv.h;
    }
  }

  f32x4 dkt[NKT][DT], dvt[NKT][DT];   // dK^T, dV^T [d = dt * 16 + g * 4 + r][key = l15]
#pragma unroll
  for (int j = 0; j < NKT; ++j)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dkt[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dvt[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const float c = a.scale * LOG2E;
  __syncthreads();

  // Stages = (query item that lists this kv item) x (32-query tile).  The global loads of stage s + 1 are issued into registers BEFORE the
  // MFMAs of stage s (the first version loaded, waited, staged and computed in sequence: every stage paid a full memory latency and the
  // kernel ran 18x off its MFMA time).
  constexpr int NCHK = (QB * CH + 255) / 256;
  const int p0 = a.inv_ptr[kit], p1 = a.inv_ptr[kit + 1];
  const int nqt = (a.nq + QB - 1) / QB;
  const int nstage = (p1 - p0) * nqt;
  U128 rq[NCHK], rd[NCHK];
  float rl = 0.f, rdl = 0.f;
  auto fetch = [&](int sidx) {
    const int qi = a.inv_item[p0 + sidx / nqt];
    const int q0 = (sidx % nqt) * QB;
#pragma unroll
    for (int n = 0; n < NCHK; ++n) {
      const int ci = tid + 256 * n;
      const int q = ci / CH, cc = ci - q * CH;
      const bool ok = ci < QB * CH && q0 + q < a.nq;
      const long row = (long)qi * a.nq + q0 + q;
      rq[n].u = ok ? ldg128(Q + row * a.ldq + h * DH + cc * 8) : zero128();
      rd[n].u = ok ? f32x8_to_f16(dO + row * a.lddo + h * DH + cc * 8) : zero128();
    }
    if (tid < QB) {
      const bool ok = q0 + tid < a.nq;
      const long row = (long)qi * a.nq + q0 + tid;
      rl = ok ? lse[row * a.heads + h] : 1.0e30f;   // rows past nq: P = 2^(-huge) = 0
      rdl = ok ? delta[row * a.heads + h] : 0.f;
    }
  };
  if (nstage > 0) fetch(0);
  for (int sidx = 0; sidx < nstage; ++sidx) {
    {
      // ---- stage 32 queries: Q and dO row-major and transposed (query slot order), lse, delta ----
#pragma unroll
      for (int n = 0; n < NCHK; ++n) {
        const int ci = tid + 256 * n;
        if (ci < QB * CH) {
          const int q = ci / CH, cc = ci - q * CH;
          *reinterpret_cast<uint4*>(sQ + q * RLD + cc * 8) = rq[n].u;
          *reinterpret_cast<uint4*>(sdO + q * RLD + cc * 8) = rd[n].u;
          const int pos = ((q >> 2) & 3) * 8 + (q >> 4) * 4 + (q & 3);   // query i * 16 + g * 4 + r sits at slot g * 8 + i * 4 + r
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sQt[(cc * 8 + e) * TLD + pos] = rq[n].e[e];
            sdOt[(cc * 8 + e) * TLD + pos] = rd[n].e[e];
          }
        }
      }
      if (tid < QB) {
        slse[tid] = rl;
        sdel[tid] = rdl;
      }
      __syncthreads();
      if (sidx + 1 < nstage) fetch(sidx + 1);

      // ---- S[q, key] = Q K^T, dP[q, key] = dO V^T: lane (key = l15, g), reg r <-> query i * 16 + g * 4 + r ----
      f32x4 s[NKT][2], dp[NKT][2];
#pragma unroll
      for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          s[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int ks = 0; ks < D32; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f16x8 aq = *reinterpret_cast<const f16x8*>(sQ + (i * 16 + l15) * RLD + ks * 32 + g * 8);
          const f16x8 ad = *reinterpret_cast<const f16x8*>(sdO + (i * 16 + l15) * RLD + ks * 32 + g * 8);
#pragma unroll
          for (int j = 0; j < NKT; ++j) {
            s[j][i] = mfma16(aq, fk[j][ks], s[j][i]);
            dp[j][i] = mfma16(ad, fv[j][ks], dp[j][i]);
          }
        }
      }
      // ---- P, dS -> fp16 operand B fragments (k-slot (g, i * 4 + r) = query i * 16 + g * 4 + r) ----
      f16x8 pf[NKT], dsf[NKT];
      {
        float L[2][4], Dl[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 l4 = *reinterpret_cast<const float4*>(slse + i * 16 + g * 4), d4 = *reinterpret_cast<const float4*>(sdel + i * 16 + g * 4);
          L[i][0] = l4.x; L[i][1] = l4.y; L[i][2] = l4.z; L[i][3] = l4.w;
          Dl[i][0] = d4.x; Dl[i][1] = d4.y; Dl[i][2] = d4.z; Dl[i][3] = d4.w;
        }
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
          U128 up, ud;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = kvalid[j] ? __builtin_amdgcn_exp2f(s[j][i][r] * c - L[i][r]) : 0.f;
              up.e[i * 4 + r] = (f16)pv;
              ud.e[i * 4 + r] = (f16)(pv * (dp[j][i][r] - Dl[i][r]));
            }
          pf[j] = up.h;
          dsf[j] = ud.h;
        }
      }
      // ---- dV^T[d, key] += dO^T[d, q] P[q, key],  dK^T[d, key] += Q^T[d, q] dS[q, key] ----
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const f16x8 at = *reinterpret_cast<const f16x8*>(sdOt + (dt * 16 + l15) * TLD + g * 8);
        const f16x8 aqt = *reinterpret_cast<const f16x8*>(sQt + (dt * 16 + l15) * TLD + g * 8);
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
          dvt[j][dt] = mfma16(at, pf[j], dvt[j][dt]);
          dkt[j][dt] = mfma16(aqt, dsf[j], dkt[j][dt]);
        }
      }
      __syncthreads();   // every wave is done with this stage's tiles
    }
  }

  // ---- accumulate into dK, dV: lane (key = l15, g) holds d = dt * 16 + g * 4 .. + 4 ----
  float* dK = reinterpret_cast<float*>(a.dK);
  float* dV = reinterpret_cast<float*>(a.dV);
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    if (!kvalid[j]) continue;
    const long row = (long)kit * a.nk + key0 + j * 16 + l15;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d >= DH) continue;
      float4* pk = reinterpret_cast<float4*>(dK + row * a.lddk + h * DH + d);
      float4* pv = reinterpret_cast<float4*>(dV + row * a.lddv + h * DH + d);
      float4 k4 = *pk, v4 = *pv;
      k4.x += dkt[j][dt][0] * a.scale; k4.y += dkt[j][dt][1] * a.scale; k4.z += dkt[j][dt][2] * a.scale; k4.w += dkt[j][dt][3] * a.scale;
      v4.x += dvt[j][dt][0]; v4.y += dvt[j][dt][1]; v4.z += dvt[j][dt][2]; v4.w += dvt[j][dt][3];
      *pk = k4;
      *pv = v4;
    }
  }
}

// ---- dQ: block = (query item, head, 64 queries); wave w owns 16 queries; keys walk through LDS 64 at a time ----
template <int DH, int KT>   // KT keys per stage (64; 32 at dh 160 so that the three tiles stay inside 64 KB of LDS)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const me_attn_bwd_args a) {
  constexpr int D32 = (DH + 31) / 32, DT = (DH + 15) / 16, CH = DH / 8;
  constexpr int NT = KT / 16, NKK = KT / 32;
  constexpr int RLD = D32 * 32 + 8;       // row-major K / V tiles [key][d]
  constexpr int TLD = KT + 8;             // transposed K tile [d][key slot]
  __shared__ __attribute__((aligned(16))) f16 sK[KT * RLD];
  __shared__ __attribute__((aligned(16))) f16 sV[KT * RLD];
  __shared__ __attribute__((aligned(16))) f16 sKt[DT * 16 * TLD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  const int nqb = (a.nq + 63) / 64;
  int w = blockIdx.x;
  const int qb = w % nqb;
  w /= nqb;
  const int h = w % a.heads;
  const int item = w / a.heads;

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  const float* __restrict__ dO = reinterpret_cast<const float*>(a.dO);

  for (int i = tid; i < KT * RLD; i += 256) { sK[i] = (f16)0.f; sV[i] = (f16)0.f; }
  for (int i = tid; i < DT * 16 * TLD; i += 256) sKt[i] = (f16)0.f;

  // this lane's query: Q and dO fragments as MFMA operand B, lse and delta as scalars
  const int q = qb * 64 + wave * 16 + l15;
  const bool qok = q < a.nq;
  const long qrow = (long)item * a.nq + (qok ? q : 0);
  f16x8 fq[D32], fdo[D32];
#pragma unroll
  for (int ks = 0; ks < D32; ++ks) {
    const int d = ks * 32 + g * 8;
    U128 uq, ud;
    uq.u = (qok && d < DH) ? ldg128(Q + qrow * a.ldq + h * DH + d) : zero128();
    ud.u = (qok && d < DH) ? f32x8_to_f16(dO + qrow * a.lddo + h * DH + d) : zero128();
    fq[ks] = uq.h;
    fdo[ks] = ud.h;
  }
  const float lse_q = qok ? reinterpret_cast<const float*>(a.lse)[qrow * a.heads + h] : 1.0e30f;
  const float del_q = qok ? reinterpret_cast<const float*>(a.delta)[qrow * a.heads + h] : 0.f;
  const float c = a.scale * LOG2E;

  f32x4 dqt[DT];   // dQ^T [d = dt * 16 + g * 4 + r][q = l15]
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // Stages = (segment) x (KT-key tile); the loads of stage s + 1 are in flight during the MFMAs of stage s (see attn_bwd_dkv_kernel).
  constexpr int NCHK = (KT * CH + 255) / 256;
  int nvalid = 0;
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    if (a.seg_item[item * a.nseg + sgi] < 0) break;
    ++nvalid;
  }
  const int nkt = (a.nk + KT - 1) / KT;
  const int nstage = nvalid * nkt;
  U128 rk[NCHK], rv[NCHK];
  auto fetch = [&](int sidx) {
    const int kit = a.seg_item[item * a.nseg + sidx / nkt];
    const int kt0 = (sidx % nkt) * KT;
#pragma unroll
    for (int n = 0; n < NCHK; ++n) {
      const int ci = tid + 256 * n;
      const int key = ci / CH, cc = ci - key * CH;
      const bool ok = ci < KT * CH && kt0 + key < a.nk;
      const long row = (long)kit * a.nk + kt0 + key;
      rk[n].u = ok ? ldg128(K + row * a.ldk + h * DH + cc * 8) : zero128();
      rv[n].u = ok ? ldg128(V + row * a.ldv + h * DH + cc * 8) : zero128();
    }
  };
  if (nstage > 0) fetch(0);
  for (int sidx = 0; sidx < nstage; ++sidx) {
    {
      const int kt0 = (sidx % nkt) * KT;
      // ---- stage KT keys: K row-major and transposed (key slot order), V row-major ----
#pragma unroll
      for (int n = 0; n < NCHK; ++n) {
        const int ci = tid + 256 * n;
        if (ci < KT * CH) {
          const int key = ci / CH, cc = ci - key * CH;
          *reinterpret_cast<uint4*>(sK + key * RLD + cc * 8) = rk[n].u;
          *reinterpret_cast<uint4*>(sV + key * RLD + cc * 8) = rv[n].u;
          // key kk * 32 + i * 16 + g * 4 + r sits at slot kk * 32 + g * 8 + i * 4 + r
          const int pos = (key & 32) | ((key & 12) << 1) | ((key & 16) >> 2) | (key & 3);
#pragma unroll
          for (int e = 0; e < 8; ++e) sKt[(cc * 8 + e) * TLD + pos] = rk[n].e[e];
        }
      }
      __syncthreads();
      if (sidx + 1 < nstage) fetch(sidx + 1);

      // ---- S^T[key, q] = K Q^T, dP^T[key, q] = V dO^T: lane (q = l15, g), reg r <-> key t * 16 + g * 4 + r ----
      f32x4 st[NT], dpt[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dpt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < D32; ++ks) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f16x8 ak = *reinterpret_cast<const f16x8*>(sK + (t * 16 + l15) * RLD + ks * 32 + g * 8);
          const f16x8 av = *reinterpret_cast<const f16x8*>(sV + (t * 16 + l15) * RLD + ks * 32 + g * 8);
          st[t] = mfma16(ak, fq[ks], st[t]);
          dpt[t] = mfma16(av, fdo[ks], dpt[t]);
        }
      }
      // ---- dS^T -> operand B of dQ^T[d, q] += K^T[d, key] dS^T[key, q] (k-slot (g, i * 4 + r) = key kk * 32 + i * 16 + g * 4 + r) ----
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        U128 ud;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = 2 * kk + i;
            const bool ok = kt0 + t * 16 + g * 4 + r < a.nk;
            const float pv = ok ? __builtin_amdgcn_exp2f(st[t][r] * c - lse_q) : 0.f;
            ud.e[i * 4 + r] = (f16)(pv * (dpt[t][r] - del_q));
          }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const f16x8 akt = *reinterpret_cast<const f16x8*>(sKt + (dt * 16 + l15) * TLD + kk * 32 + g * 8);
          dqt[dt] = mfma16(akt, ud.h, dqt[dt]);
        }
      }
      __syncthreads();
    }
  }

  if (!qok) return;
  float* dQ = reinterpret_cast<float*>(a.dQ);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int d = dt * 16 + g * 4;
    if (d >= DH) continue;
    float4* pq = reinterpret_cast<float4*>(dQ + qrow * a.lddq + h * DH + d);
    float4 q4 = *pq;
    q4.x += dqt[dt][0] * a.scale; q4.y += dqt[dt][1] * a.scale; q4.z += dqt[dt][2] * a.scale; q4.w += dqt[dt][3] * a.scale;
    *pq = q4;
  }
}

template <int DH, int NKT>
int launch_bwd(const me_attn_bwd_args* a, hipStream_t st) {
  const long rows = (long)a->n_items * a->nq;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows * a->heads + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const f16*>(a->O), a->ldo,
                     reinterpret_cast<const float*>(a->dO), a->lddo, reinterpret_cast<float*>(a->delta), rows, a->heads, a->dh);
  const int nkb = (a->nk + 64 * NKT - 1) / (64 * NKT);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, NKT>), dim3((unsigned)((long)a->n_kv_items * a->heads * nkb)), dim3(256), 0, st, *a);
  const int nqb = (a->nq + 63) / 64;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, (DH > 80 ? 32 : 64)>), dim3((unsigned)((long)a->n_items * a->heads * nqb)), dim3(256), 0, st, *a);
  me_set_kernel("attn_bwd_dkv_kernel+attn_bwd_dq_kernel");
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { me_set_hip_error("me_attn_bwd", (int)e); return ME_EHIP; }
  return ME_OK;
}

}  // namespace

extern "C" int me_attn_bwd(const me_attn_bwd_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->V || !a->O || !a->dO || !a->lse || !a->dQ || !a->dK || !a->dV || !a->delta || !a->seg_item || !a->inv_ptr || !a->inv_item) {
    me_set_error("me_attn_bwd: null pointer");
    return ME_EINVAL;
  }
  if (a->n_items <= 0 || a->n_kv_items <= 0 || a->nq <= 0 || a->nk <= 0 || a->heads <= 0 || a->nseg < 1 || a->nseg > 3) { me_set_error("me_attn_bwd: bad sizes"); return ME_EINVAL; }
  if (a->ldq % 8 || a->ldk % 8 || a->ldv % 8 || a->ldo % 8 || a->lddo % 4 || a->lddq % 4 || a->lddk % 4 || a->lddv % 4) {
    me_set_error("me_attn_bwd: row strides must be multiples of 8 (fp16 tensors) / 4 (fp32 gradients)");
    return ME_EINVAL;
  }
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O | (uintptr_t)a->dO | (uintptr_t)a->dQ | (uintptr_t)a->dK | (uintptr_t)a->dV) & 15) {
    me_set_error("me_attn_bwd: misaligned pointer");
    return ME_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  switch (a->dh) {
    case 40: {
      // keys per block = 64 * NKT.  NKT = 4 amortises a staged query tile over the most MFMAs but needs 376 VGPRs (one wave per SIMD, one block per
      // CU); NKT = 2 (196 VGPRs, two blocks per CU) measured 9.35 vs 11.4 ms for the whole L0 [prev | cur] backward, NKT = 1 11.1.  ME_ATTN_BWD_NKT for A/B.
      static const int nkt = getenv("ME_ATTN_BWD_NKT") ? atoi(getenv("ME_ATTN_BWD_NKT")) : 2;
      return nkt == 1 ? launch_bwd<40, 1>(a, st) : nkt == 2 ? launch_bwd<40, 2>(a, st) : launch_bwd<40, 4>(a, st);
    }
    case 80: {
      static const int nkt = getenv("ME_ATTN_BWD_NKT") ? atoi(getenv("ME_ATTN_BWD_NKT")) : 2;
      return nkt == 1 ? launch_bwd<80, 1>(a, st) : launch_bwd<80, 2>(a, st);
    }
    case 160: return launch_bwd<160, 1>(a, st);
    default: me_set_error("me_attn_bwd: head dim must be 40, 80 or 160"); return ME_EINVAL;
  }
}
