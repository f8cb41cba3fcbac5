// Fused multi-segment attention on MFMA (gfx950): O = softmax(Q K^T * scale) V with online softmax,
// scores and gathered keys never materialised.
//
// A query item (one frame of one batch row) attends `nseg` key segments, each a whole kv item
// (nk keys) named by seg_item[item][seg]:
//   attn1 un-edited  : [prev frame | cur frame]                         (attention_2d.py:732-740)
//   attn1 edited     : [src prev (dual-mask) | src cur (dual-mask) | edit cur]  (fully_control.py:372-447)
//   attn2 / ControlNet / adapter attn_pose : one segment
//   adapter sparse-causal : [first frame of chunk | prev frame]         (controlnet_adapter.py:352-361)
// DUAL segments implement the reference's "K*M | K*(1-M)" key duplication without duplicating
// anything: a masked key has logit m*s in the foreground copy and (1-m)*s in the background copy
// and BOTH copies carry the same (unmasked) V, so the key's total weight is exp(m s) + exp((1-m) s).
//
// Two kernels: attn2_kernel (below the first one) serves plain and BINARY dual segments -- every launch of the model with the
// binary masks the data ships; attn_kernel<GD = true> keeps the general, mask-reading dual modes.  Common structure
// (per block: one (item, head), waves x QT x 16 queries):
//   S^T = K Q^T   : MFMA A = K tile rows from LDS, B = Q fragments held in registers
//                   -> lane (q = lane&15, g = lane>>4) holds keys t*16 + g*4 + r: row max / sum need
//                      only 2 xor-shuffles across g; alpha and 1/l are lane-local for O^T.
//   O^T += V^T P^T: MFMA A = V^T fragments (attn_kernel: V transposed into LDS while staging; attn2: ds_read_b64_tr_b16),
//                   B = P^T straight from the S^T accumulator registers (fp16), no LDS round trip:
//                   MFMA k-slot (g, j) carries key kk*32 + (j>>2)*16 + g*4 + (j&3) for BOTH operands.
#include "me_common.h"
#include "../../include/motioned.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

// Block order: a block is one (item, head, query block), query blocks fastest; xcd_remap gives each XCD a contiguous run, so the query blocks of ONE head
// sit next to each other and its K | V tiles stay in the XCD's L2.  (Heads fastest for the 77-key text cross-attention -- the eight heads of the same
// query rows side by side, whole 128-byte lines of Q and O per XCD -- measured +-0: profiles/r04_attn_cross_order_ab.txt.)
extern "C" void me_set_kernel(const char* name);

namespace {

__device__ unsigned long long g_fallback_blocks;   // blocks of the fixed-offset kernels that re-ran with the running maximum (phase B)


constexpr float NEG_BIG = -1.0e30f;
constexpr int ATTN_HEAD_SLOWEST = 1 << 4;   // private bit of me_attn_args.general_dual inside attn2 launches (me_attn sets it; the field is 0 / 1 at the ABI)
// Ablation builds (tools/build_abl.sh, never the shipped library): bit 0 = no K/V DMA inside the sweep (the stage buffers keep the first stage),
// bit 1 = no barrier / vmcnt wait inside the sweep, bit 2 = no exp (P = cvt(S)), bit 3 = no tile arithmetic (DMA + barriers only), bit 4 = no re-basing test
#ifndef ME_ATTN_ABL
#define ME_ATTN_ABL 0
#endif
// dh = 40, phase A: QK^T as v_mfma_f32_32x32x16_f16 over K = 48 (3 steps: 40 dims + the two fold slots) instead of 16x16x32 over K = 64.
// ME_ATTN_W32=0 builds the 16x16x32 form (A/B, tools/build_abl.sh).
#ifndef ME_ATTN_W32
#define ME_ATTN_W32 1
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KT = 64;        // keys per tile
constexpr int VLD = KT + 8;   // V^T row stride in halves (144 B: 16 rows x 16 B hit 64 distinct banks)

template <int DH, int QT, bool GD, int MINW, int NBUF>
__global__ __launch_bounds__(256, MINW) void attn_kernel(const me_attn_args a) {
  constexpr int CH = DH / 8;
  constexpr int D32 = (DH + 31) / 32;
  constexpr int DT = (DH + 15) / 16;
  constexpr int KLD = D32 * 32 + 8;
  constexpr int BQ = 64 * QT;
  constexpr int NCH = KT * CH;          // 16-byte chunks per K (or V) tile
  constexpr int NFULL = NCH / 256;      // staging rounds in which every wave moves one chunk column (64 keys x 16 B)
  constexpr int RW = (NCH % 256) / 64;  // leftover chunk columns
  constexpr int NLD = NFULL + (RW ? 1 : 0);
  // V^T has padding rows when dh is not a multiple of 16 (40 -> 48, 80 -> 96): make the first one all ones, so the
  // PV MFMA accumulates the softmax denominator in accumulator row DH for free (no VALU row sums, and the sum is
  // taken over exactly the fp16-rounded P that multiplies V)
  constexpr bool ONES = DT * 16 > DH;

  // two LDS stages: tile t+1 is stored while tile t is consumed -> one barrier per tile
  __shared__ __attribute__((aligned(16))) f16 sKb[NBUF][KT * KLD];
  __shared__ __attribute__((aligned(16))) f16 sVtb[NBUF][DT * 16 * VLD];
  __shared__ __attribute__((aligned(16))) f16 sMb[NBUF][KT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int l15 = lane & 15;

  const int nqb = (a.nq + BQ - 1) / BQ;
  const int w = xcd_remap(blockIdx.x, a.n_items * a.heads * nqb);
  const int qb = w % nqb;
  const int rest = w / nqb;
  const int h = rest % a.heads;
  const int item = rest / a.heads;

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  const f16* __restrict__ Mk = reinterpret_cast<const f16*>(a.mask);
  f16* __restrict__ O = reinterpret_cast<f16*>(a.O);

  // zero the LDS padding that staging never writes (K columns >= DH, V^T rows >= DH)
  for (int i = tid; i < NBUF * KT * KLD; i += 256) (&sKb[0][0])[i] = (f16)0.f;
  for (int i = tid; i < NBUF * DT * 16 * VLD; i += 256) (&sVtb[0][0])[i] = (ONES && (i % (DT * 16 * VLD)) / VLD == DH) ? (f16)1.f : (f16)0.f;

  // Q fragments (MFMA operand B): lane (q = l15, g) holds Q[q][ks*32 + g*8 .. +8]
  f16x8 fq[QT][D32];
  int qrow[QT];
  const long hq = a.hsq > 0 ? a.hsq : DH;   // head-major Q panels (ABI 8) as in attn2_kernel, or heads as column slices of the rows
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qb * BQ + (wave * QT + qt) * 16 + l15;
    qrow[qt] = q < a.nq ? item * a.nq + q : -1;
#pragma unroll
    for (int ks = 0; ks < D32; ++ks) {
      const int d = ks * 32 + g * 8;
      U128 u;
      u.u = (qrow[qt] >= 0 && d < DH) ? ldg128(Q + (long)qrow[qt] * a.ldq + h * hq + d) : zero128();
      fq[qt][ks] = u.h;
    }
  }

  const int ntk = (a.nk + KT - 1) / KT;
  int nvalid = 0;  // segments before the first negative entry
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    if (a.seg_item[item * a.nseg + sgi] < 0) break;
    ++nvalid;
  }
  const int T = nvalid * ntk;
  const float c = a.scale * 1.4426950408889634f;  // fold log2(e): softmax via exp2

  // Staging map: lane = key inside the tile, wave = 16-byte chunk column (cc = wave + 4 * round), so a wave's
  // transposed b16 stores hit 32 distinct LDS banks and the global address is "uniform tile base + per-lane row
  // offset + immediate": no per-tile index arithmetic.  NCH % 256 leftover columns (dh 40: one, dh 80: two) go to
  // RW waves rotated over the tiles so that no wave is always the slow one in front of the barrier.
  uint4 rk[NLD], rv[NLD];
  // V^T rows keep their 64 keys in MFMA k-slot order: key kk*32 + j*16 + g*4 + r sits at kk*32 + g*8 + j*4 + r, so
  // the eight keys of a lane's PV operand-A fragment are one aligned 16-byte LDS read
  const int vpos = (lane & 32) | ((lane & 12) << 1) | ((lane & 16) >> 2) | (lane & 3);
  const long hk = a.hsk > 0 ? a.hsk : DH, hv = a.hsv > 0 ? a.hsv : DH;   // head-major K / V panels (ABI 6), or heads as column slices
  const int koff0 = lane * a.ldk;
  const int voff0 = lane * a.ldv;
  // load cursor = the tile being fetched (two ahead of the one consumed): uniform tile base pointers, bumped by a
  // constant per tile and re-derived from seg_item only when a segment ends
  int seg_l = 0, kt_l = 0;
  const f16 *kptr_l = K, *vptr_l = V;
  auto seg_base = [&]() {
    const int kit = a.seg_item[item * a.nseg + seg_l];
    kptr_l = K + (long)kit * a.nk * a.ldk + h * hk;
    vptr_l = V + (long)kit * a.nk * a.ldv + h * hv;
  };
  auto advance_l = [&]() {
    if (++kt_l == ntk) {
      kt_l = 0;
      if (++seg_l < nvalid) seg_base();
    } else {
      kptr_l += KT * a.ldk;
      vptr_l += KT * a.ldv;
    }
  };
  auto gload = [&](int ti) {
    int ko = koff0, vo = voff0;
    if ((kt_l + 1) * KT > a.nk) {   // tail tile: keys past nk re-read the last key (finite; their logits are masked)
      const int kl = min(lane, a.nk - 1 - kt_l * KT);
      ko = kl * a.ldk;
      vo = kl * a.ldv;
    }
#pragma unroll
    for (int i = 0; i < NFULL; ++i) {
      rk[i] = ldg128(kptr_l + ko + (wave + 4 * i) * 8);
      rv[i] = ldg128(vptr_l + vo + (wave + 4 * i) * 8);
    }
    if constexpr (RW > 0) {
      const int r = (wave - ti) & 3;
      if (r < RW) {
        rk[NFULL] = ldg128(kptr_l + ko + (4 * NFULL + r) * 8);
        rv[NFULL] = ldg128(vptr_l + vo + (4 * NFULL + r) * 8);
      }
    }
    advance_l();
  };
  int seg_s = 0, kt_s = 0;   // store cursor (general dual-mask instantiation only: which mask plane / keys)
  auto sstore = [&](int ti) {
    f16* sK = sKb[ti & (NBUF - 1)];
    f16* sVt = sVtb[ti & (NBUF - 1)];
    auto put = [&](int cc, const uint4& k, const uint4& v) {
      *reinterpret_cast<uint4*>(sK + lane * KLD + cc * 8) = k;
      U128 u;
      u.u = v;
#pragma unroll
      for (int e = 0; e < 8; ++e) sVt[(cc * 8 + e) * VLD + vpos] = u.e[e];
    };
#pragma unroll
    for (int i = 0; i < NFULL; ++i) put(wave + 4 * i, rk[i], rv[i]);
    if constexpr (RW > 0) {
      const int r = (wave - ti) & 3;
      if (r < RW) put(4 * NFULL + r, rk[NFULL], rv[NFULL]);
    }
    if constexpr (GD) {   // mask values of the tile's keys
      if (tid < KT) {
        const int mode = a.seg_mode[item * a.nseg + seg_s];
        f16 mv = (f16)0.f;
        const int key = kt_s * KT + tid;
        if (mode != ME_SEG_PLAIN && key < a.nk) {
          const int plane = mode == ME_SEG_DUAL_CUR ? h : (h > 0 ? h - 1 : 0);
          mv = Mk[(long)plane * a.nk + key];
        }
        sMb[ti & (NBUF - 1)][tid] = mv;
      }
      if (++kt_s == ntk) {
        kt_s = 0;
        ++seg_s;
      }
    }
  };

  using IC0 = std::integral_constant<int, ME_SEG_PLAIN>;
  using IC3 = std::integral_constant<int, ME_SEG_DUAL_BIN>;
  using ICG = std::integral_constant<int, ME_SEG_DUAL_CUR>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;

  f32x4 o[QT][DT];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = NEG_BIG;
    lrun[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  __syncthreads();  // LDS zero-fill visible before the first stage is written
  const int nfull = a.nk / KT;
  if (T > 0) {
    seg_base();
    gload(0);
    sstore(0);
    if (T > 1) gload(1);
  }
  __syncthreads();
  // One KV tile, specialised at compile time on the segment mode and on "tile fully inside nk": the mode / tail
  // tests are wave-uniform, and leaving them as run-time selects made hipcc if-convert BOTH softmax variants and
  // 32 tail compares into every tile (11 VALU per MFMA).  A scalar (readfirstlane) dispatch picks the body.
  auto tile = [&](auto mode_c, auto full_c, int ti, int kt) {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    const f16* sK = sKb[ti & (NBUF - 1)];
    const f16* sVt = sVtb[ti & (NBUF - 1)];
    const f16* sM = sMb[ti & (NBUF - 1)];
    const int kbase = kt * KT + g * 4;  // + t*16 + r

    // ---- S^T = K Q^T ----
    f32x4 s[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < 4; ++t) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D32; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 fk = *reinterpret_cast<const f16x8*>(sK + (t * 16 + l15) * KLD + ks * 32 + g * 8);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma16(fk, fq[qt][ks], s[qt][t]);
      }
    }

    // ---- online softmax (per query = per lane column), P^T packed to fp16 MFMA B fragments ----
    f16x8 pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if constexpr (!FULL) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + t * 16 + r >= a.nk) s[qt][t][r] = NEG_BIG;   // raw logit; c < 1 keeps NEG_BIG * c finite
      }
      float p[4][4];
      float psum = 0.f, alpha;
      if constexpr (MODE == ME_SEG_PLAIN || MODE == ME_SEG_DUAL_BIN) {
        float mr = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) mr = fmaxf(mr, s[qt][t][r]);
        mr = xor32_max(xor16_max(mr));
        // DUAL_BIN (binary mask): one of the (fg, bg) copies keeps the key, the other is a zero vector with logit 0,
        // both with the same V -> weight exp(s) + exp(0) whatever the mask bit says; the running max includes 0
        const float mnew = MODE == ME_SEG_PLAIN ? fmaxf(mrun[qt], mr * c) : fmaxf(mrun[qt], fmaxf(mr * c, 0.f));
        alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
        const float e0 = MODE == ME_SEG_DUAL_BIN ? __builtin_amdgcn_exp2f(-mnew) : 0.f;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): the VALU pipe is the bound of this kernel
        const f32x2 c2 = {c, c}, nm2 = {-mnew, -mnew}, e2 = {e0, e0};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2 sv = {s[qt][t][2 * h2], s[qt][t][2 * h2 + 1]};
            const f32x2 x = __builtin_elementwise_fma(sv, c2, nm2);
            f32x2 pv = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            if constexpr (MODE == ME_SEG_DUAL_BIN) pv += e2;
            p[t][2 * h2] = pv[0];
            p[t][2 * h2 + 1] = pv[1];
            if constexpr (!FULL && MODE == ME_SEG_DUAL_BIN) {
              if (kbase + t * 16 + 2 * h2 >= a.nk) p[t][2 * h2] = 0.f;
              if (kbase + t * 16 + 2 * h2 + 1 >= a.nk) p[t][2 * h2 + 1] = 0.f;
            }
            if constexpr (!ONES) psum += p[t][2 * h2] + p[t][2 * h2 + 1];
          }
      } else {
        // general (non-binary) masks: both copies' logits are needed.  Compiled only into the GD instantiation
        float x1[4][4], x2[4][4];
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          U64 mk;
          mk.u = *reinterpret_cast<const uint2*>(sM + t * 16 + g * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = s[qt][t][r] * c;
            const bool ok = sc > 0.5f * NEG_BIG * c;      // masked tail keys keep BOTH copies at -big
            const float fgv = sc * (float)mk.e[r];
            x1[t][r] = ok ? fgv : sc;
            x2[t][r] = ok ? sc - fgv : sc;
            mx = fmaxf(mx, fmaxf(x1[t][r], x2[t][r]));
          }
        }
        mx = xor32_max(xor16_max(mx));
        const float mnew = fmaxf(mrun[qt], mx);
        alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[t][r] = __builtin_amdgcn_exp2f(x1[t][r] - mnew) + __builtin_amdgcn_exp2f(x2[t][r] - mnew);
            if constexpr (!ONES) psum += p[t][r];
          }
      }
      if constexpr (!ONES) lrun[qt] = lrun[qt] * alpha + psum;
      // rescale only when some query of the wave saw its running max move (exact: alpha == 1 otherwise)
      if (__builtin_amdgcn_readfirstlane(__any(alpha != 1.0f))) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {   // four v_cvt_pk_f16_f32 per fragment, no per-element inserts
        union { f16x2 h[4]; f16x8 v; } f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          f.h[h2] = __builtin_convertvector((f32x2){p[2 * kk][2 * h2], p[2 * kk][2 * h2 + 1]}, f16x2);
          f.h[2 + h2] = __builtin_convertvector((f32x2){p[2 * kk + 1][2 * h2], p[2 * kk + 1][2 * h2 + 1]}, f16x2);
        }
        pf[qt][kk] = f.v;
      }
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const f16x8 fv = *reinterpret_cast<const f16x8*>(sVt + (dt * 16 + l15) * VLD + kk * 32 + g * 8);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma16(fv, pf[qt][kk], o[qt][dt]);
      }
    }
  };

  // One loop per (segment mode, full / tail) so that the accumulators stay in place between tiles: a single
  // loop dispatching on the mode made the compiler copy all of o[] on every tile to merge the variants.
  int ti = 0;
  auto run = [&](auto mode_c, auto full_c, int kt0, int count) {
    for (int n = 0; n < count; ++n, ++ti) {
      tile(mode_c, full_c, ti, kt0 + n);
      // stage tile ti+1 into the other buffer (every wave finished reading it one barrier ago), prefetch ti+2
      if (ti + 1 < T) {
        if (NBUF == 1) __syncthreads();  // single stage: everyone must be done reading it first
        sstore(ti + 1);
        if (ti + 2 < T) gload(ti + 2);
      }
      __syncthreads();
    }
  };
  for (int seg = 0; seg < nvalid; ++seg) {
    const int mode = __builtin_amdgcn_readfirstlane(a.seg_mode[item * a.nseg + seg]);
    if (mode == ME_SEG_PLAIN) {
      run(IC0{}, BT{}, 0, nfull);
      run(IC0{}, BF{}, nfull, ntk - nfull);
    } else if (mode == ME_SEG_DUAL_BIN) {
      run(IC3{}, BT{}, 0, nfull);
      run(IC3{}, BF{}, nfull, ntk - nfull);
    } else {
      if constexpr (GD) run(ICG{}, BF{}, 0, ntk);
      else ti += ntk;   // unreachable: me_attn routes general masks to the GD instantiation
    }
  }

  // ---- finalize: O^T[d = dt*16 + g*4 + r][q = l15] / l ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l;
    if constexpr (ONES) {   // denominator sits in accumulator row DH: tile DH/16, lanes g == (DH%16)/4, reg (DH%16)%4
      l = __shfl(o[qt][DH / 16][(DH % 16) % 4], ((DH % 16) / 4) * 16 + l15, 64);
    } else {
      l = xor32_sum(xor16_sum(lrun[qt]));
    }
    const float inv = 1.0f / l;
    if (qrow[qt] < 0) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d >= DH) continue;
      U64 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(o[qt][dt][r] * inv);
      *reinterpret_cast<uint2*>(O + (long)qrow[qt] * a.ldo + h * DH + d) = ov.u;
    }
  }
}


// =====================================================================================================================
// attn2: the production kernel for plain and BINARY-dual segments (everything except the general, non-binary masks).
//
// Differences to the kernel above (measured on the L0 [prev | cur] launch, 7.2 ms: MFMA 2.95 ms + staging 2.2 ms +
// softmax 1.6 ms, NOT overlapped -- profiles/r02_attn_ablation.txt):
//   * K and V tiles are staged by LDS-DMA (global_load_lds_dwordx4) in their ROW-MAJOR global layout: no VGPR round trip,
//     no ds_write, no per-element transposed b16 stores.  A lane's chunk never changes between tiles, so the per-lane
//     source offset is one register per DMA slot and a tile costs <= 3 DMA instructions per wave (dh 40).
//     Row pitches are an ODD number of 16-byte chunks (K: dh/8 [+1], V: dh/8 + ones / pad chunk), the pad / ones chunks
//     are written once at kernel start and skipped by the DMA (inactive lanes do not write LDS).
//   * V^T operand-A fragments come from ds_read_b64_tr_b16 (hardware 4x4 transpose): lane (c, g) addresses row
//     key0 + c/4, byte (c%4)*8 of a 16-column block and receives V[key0 .. key0+3][col0 + c].
//   * DUAL_BIN segments (binary fg/bg mask: weight exp(s) + 1 per source key) run the PLAIN tile code; the "+1" part is
//     query-independent -- sum_j V_j over the segment's keys -- and enters once, in the epilogue, from the column
//     sums `vsum` that me_attn computes beforehand:  O = (A f1 + f2 B) / (l f1 + f2 n),  A, l relative to the running
//     max m, m' = max(m, 0), f1 = 2^(m - m'), f2 = 2^(-m'), B = sum over the item's dual segments of vsum, n = their keys.
// =====================================================================================================================
// ds_read_b64_tr_b16 through inline asm: with the builtin, hipcc treats the pending LDS-DMA of the NEXT stage as a
// may-alias write and drains vmcnt(0) in front of the first transposed read of every tile (the DMA latency then sits
// on the critical path of each wave).  The asm form is invisible to that pass; completion is awaited by lds_wait<N>,
// which names the destination registers so that no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ uint2 lds_tr16(unsigned addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(uint2& r0, uint2& r1, uint2& r2, uint2& r3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "i"(N));
}
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

// One LDS-DMA instruction with a lane mask: lanes whose bit is clear neither fetch nor write LDS (pad / ones chunks keep
// their start-up contents).  Hand-written so that (a) the exec juggling is two scalar moves instead of the compare /
// saveexec / branch sequence hipcc builds around the builtin and (b) the transfer is invisible to hipcc's waitcnt pass --
// completion is awaited by the explicit vmcnt(0) in front of the stage barrier, nowhere else.
__device__ __forceinline__ void dma16_masked(unsigned lds_dst, const char* base, int voff, unsigned long long mask) {
  unsigned keep_m0;
  unsigned long long keep_exec;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_mov_b64 exec, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %4\n\t"
      "s_mov_b64 exec, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep_m0), "=&s"(keep_exec)
      : "s"(lds_dst), "v"(voff), "s"(base), "s"(mask)
      : "memory");
}

template <int DH, int QT, int NW, int MINW, int NBUF, int NSUB, bool FOLD, bool KVRES = false>
__global__ __launch_bounds__(64 * NW, MINW) void attn2_kernel(const me_attn_args a) {
  static_assert(!FOLD || DH % 32 != 0, "FOLD needs a spare k-slot in the last QK^T step");
  static_assert(!KVRES || !FOLD, "KVRES (keys resident across query blocks) is a form of the classic sweep");
  constexpr int NTHR = 64 * NW;
  constexpr int D32 = (DH + 31) / 32;
  constexpr int DT = (DH + 15) / 16;
  constexpr int CKR = DH / 8;                         // real 16-byte chunks per K / V row
  // K row pitch in chunks (odd: 16 rows x one chunk column hit distinct 16-byte slots); FOLD adds the ones chunk CKR
  constexpr int CK = FOLD ? ((CKR + 1) | 1) : ((CKR & 1) ? CKR : CKR + 1);
  constexpr int GS = CKR - (D32 - 1) * 4;             // FOLD: lane group whose last-step Q fragment starts at k-slot DH
  constexpr bool ONES = DT * 16 > DH;                 // spare V^T rows: row DH = all ones -> the PV MFMA also produces the softmax denominator
  constexpr int CV = DH == 40 ? 6 : (DH == 80 ? 10 : 22);   // V row pitch in chunks (8 consecutive rows x 32 B on distinct banks)
  constexpr int KPB = CK * 16, VPB = CV * 16;         // pitches in bytes
  constexpr int KBYTES = KT * KPB, VBYTES = KT * VPB, SUB = KBYTES + VBYTES;   // one 64-key sub-tile: K rows, then V rows
  constexpr int STAGE = NSUB * SUB;                   // keys per barrier = NSUB * 64
  constexpr int NINS = CK + CV;                       // DMA instructions per sub-tile (each 64 lanes x 16 B = 1 KB of LDS image)
  constexpr int NSLOT = (NINS + NW - 1) / NW;         // per wave
  constexpr int BQ = 16 * QT * NW;
  static_assert(KBYTES % 1024 == 0 && VBYTES % 1024 == 0, "stage layout");

  // KVRES: + a wave-private scratch of QT * 16 rows x dh halves, where the output tile turns from the MFMA layout into whole 16-byte row pieces (finalize_rows)
  constexpr int OROWS = QT * 16, OSCR = KVRES ? OROWS * DH * 2 : 0;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * STAGE + 64 + NW * OSCR];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int l15 = lane & 15;

  const int nqb = (a.nq + BQ - 1) / BQ;
  // KVRES (round 6; the 77-key text cross-attention, attention_2d.py:343): every key of the item fits ONE stage, so a block stages K | V once and
  // walks `qpb` consecutive query blocks of its (item, head) over it -- LDS image, ones / pad chunks, DMA and barriers paid once per qpb * BQ queries;
  // the walk itself has no barrier (the waves drift apart and cover each other's Q loads).  qpb travels in the private high bits of general_dual.
  const int qpb = KVRES ? max(a.general_dual >> 8, 1) : 1;
  const int nqg = (nqb + qpb - 1) / qpb;
  const int w = xcd_remap(blockIdx.x, a.n_items * a.heads * nqg);
  int qb = (w % nqg) * qpb;
  const int rest = w / nqg;
  // HEAD_SLOWEST (round 5, multi-segment launches): blocks ordered (head, item, query block) -- with 8 heads an XCD's contiguous run is ONE head over all
  // items, so the frame that item f reads as `cur` is still in that XCD's L2 when item f + 1 reads it as `prev` (4 items of a head run at a time on the
  // XCD's 32 CUs: 5 frames x 655 KB of K | V at dh = 40, N = 4096 -- inside the 4 MB).  In the (item, head) order the XCD ran 4 (item, head) pairs of
  // 2 x 2 frames = 5.2 MB and met every K | V frame twice, a dozen block rounds apart: the 1.55 - 1.67 x HBM traffic of the round-3 / round-4 PMC passes.
  const bool head_slowest = (a.general_dual & ATTN_HEAD_SLOWEST) != 0;
  const int h = head_slowest ? rest / a.n_items : rest % a.heads;
  int item = head_slowest ? rest % a.n_items : rest / a.heads;
  if (head_slowest && a.item_order) item = __builtin_amdgcn_readfirstlane(a.item_order[item]);   // (ABI 8) the caller's order of the items inside a head's run

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const char* __restrict__ K = reinterpret_cast<const char*>(a.K);
  const char* __restrict__ V = reinterpret_cast<const char*>(a.V);
  f16* __restrict__ O = reinterpret_cast<f16*>(a.O);

  // LDS image: zero everywhere (pad chunks, the tail of the last row a fragment read may run into), ones chunks of V
  for (int i = tid; i < (NBUF * STAGE + 64) / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = zero128();
  __syncthreads();
  if constexpr (ONES) {
    for (int i = tid; i < NBUF * NSUB * KT; i += NTHR)
      *reinterpret_cast<f16*>(smem + (i / KT) * SUB + KBYTES + (i % KT) * VPB + DH * 2) = (f16)1.f;
  }
  if constexpr (FOLD) {   // K column DH = 1: the Q fragment's k-slot DH then carries -(reference offset) into every logit
    for (int i = tid; i < NBUF * NSUB * KT; i += NTHR)
      *reinterpret_cast<f16*>(smem + (i / KT) * SUB + (i % KT) * KPB + DH * 2) = (f16)1.f;
  }
  const float c = a.scale * 1.4426950408889634f;  // fold log2(e): softmax via exp2

  const long hq = a.hsq > 0 ? a.hsq : DH;   // head-major Q panels (ABI 8), or heads as column slices of the rows
  // Q fragments (MFMA operand B): lane (q = l15, g) holds Q[q][ks*32 + g*8 .. +8], zero beyond dh.
  // scaled: pre-multiplied by c, so the MFMA result is the logit in log2 units and needs no VALU pass before exp2.
  f16x8 fq[QT][D32];
  int qrow[QT];
  auto load_q = [&](auto scaled_c) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qb * BQ + (wave * QT + qt) * 16 + l15;
      qrow[qt] = q < a.nq ? item * a.nq + q : -1;
      const long qsrc = (long)(a.q_items > 0 ? item % a.q_items : item) * a.nq + q;   // queries shared by several batch entries
#pragma unroll
      for (int ks = 0; ks < D32; ++ks) {
        const int d = ks * 32 + g * 8;
        U128 u;
        u.u = (qrow[qt] >= 0 && d < DH) ? ldg128(Q + qsrc * a.ldq + h * hq + d) : zero128();
        if constexpr (decltype(scaled_c)::value) {
#pragma unroll
          for (int e = 0; e < 8; ++e) u.e[e] = (f16)((float)u.e[e] * c);
        }
        fq[qt][ks] = u.h;
      }
      if constexpr (FOLD) {
        if (g == GS) fq[qt][D32 - 1][1] = (f16)-1.f;   // against the pad marks of K column DH+1
      }
    }
  };
  // W32 (dh = 40, phase A): the wave's 32 queries as ONE 32-wide block.  S^T = K Q^T by v_mfma_f32_32x32x16_f16: operand A = K rows
  // (lane (r = lane & 31, hi = lane >> 5) holds K[key(r)][ks*16 + hi*8 .. +8]), operand B = Q (lane (q = lane & 31, hi) holds
  // Q[q][ks*16 + hi*8 .. +8]), three k-steps cover d = 0..39 and the fold slots 40 (offset) / 41 (pad mark): 6 MFMAs of 32 cycles per
  // 64 keys instead of 16 of 16.  D: lane (q, hi), register i <-> MFMA row 8*(i>>2) + 4*hi + (i&3).  After exp2 / cvt_pkrtz, four
  // v_permlane16_swap per 32 keys move the packed P of queries 16..31 out of the lanes of queries 0..15 and leave the two 16-query B operands
  // of the (unchanged) 16x16x32 PV MFMAs; the MFMA row -> key map (bits 2 and 3 swapped) is chosen so that their k-slot order equals the
  // 16x16 path's (lane group g <-> keys 4g .. 4g+3 of each 16): the V^T fragment reads, their bank pattern and everything behind PV stay as they are.
  constexpr bool W32 = FOLD && DH == 40 && QT == 2 && (ME_ATTN_W32 != 0);
  const int hi32 = lane >> 5;
  const int krow32 = (lane & 19) | ((lane & 4) << 1) | ((lane & 8) >> 1);   // K row (inside a 32-key block) this lane feeds to MFMA row lane & 31
  f16x8 fqw[3];
  float mq = 0.f;   // W32: the fixed offset of query lane & 31 (both hi lanes agree)
  auto load_qw = [&]() {
    const int q = qb * BQ + wave * 32 + (lane & 31);
    const long qsrc = (long)(a.q_items > 0 ? item % a.q_items : item) * a.nq + q;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int d = ks * 16 + hi32 * 8;
      U128 u;
      u.u = (q < a.nq && d < DH) ? ldg128(Q + qsrc * a.ldq + h * hq + d) : zero128();
#pragma unroll
      for (int e = 0; e < 8; ++e) u.e[e] = (f16)((float)u.e[e] * c);
      fqw[ks] = u.h;
    }
    if (hi32 == 1) fqw[2][1] = (f16)-1.f;   // k-slot 41, against the pad marks of K column DH + 1
  };
  if constexpr (W32) {
    load_qw();
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qb * BQ + (wave * QT + qt) * 16 + l15;
      qrow[qt] = q < a.nq ? item * a.nq + q : -1;
    }
  } else {
    load_q(std::integral_constant<bool, FOLD>{});
  }

  const int ntk = (a.nk + KT - 1) / KT;          // 64-key sub-tiles per segment
  const int nst = (ntk + NSUB - 1) / NSUB;        // stages per segment
  // the item's segment list, read once (uniform): kv item per segment, number of valid / binary-dual segments
  int kit3[3] = {-1, -1, -1}, dual3[3] = {0, 0, 0};
  int nvalid = 0, ndual = 0;
#pragma unroll
  for (int sgi = 0; sgi < 3; ++sgi) {
    if (sgi < a.nseg && nvalid == sgi) {
      const int v = __builtin_amdgcn_readfirstlane(a.seg_item[item * a.nseg + sgi]);
      if (v >= 0) {
        kit3[sgi] = v;
        ++nvalid;
        dual3[sgi] = __builtin_amdgcn_readfirstlane(a.seg_mode[item * a.nseg + sgi]) == ME_SEG_DUAL_BIN;
        ndual += dual3[sgi];
      }
    }
  }
  const int T = nvalid * nst;                     // stages in total

  // DMA slots: instruction q = wave + NW i covers image bytes [q * 1024, +1024) of a sub-tile (K rows first, then V rows);
  // lane -> chunk q * 64 + lane of that operand.  off[i] = byte offset of the lane's source chunk from the sub-tile's base
  // pointer; msk[i] = the lanes that carry a real chunk (pad / ones chunks and unused slots are masked off).
  auto slot_off = [&](int i, int maxkey) -> int {
    const int q = wave + NW * i;
    if (q >= NINS) return -1;
    const bool isk = q < CK;
    const int cl = (isk ? q : q - CK) * 64 + lane;
    const int cpr = isk ? CK : CV;
    const int key = cl / cpr, ch = cl - key * cpr;
    if (ch >= CKR) return -1;
    const int kc = min(key, maxkey);   // tail sub-tile: rows past nk re-read the last key (finite; their logits are masked)
    return (kc * (isk ? a.ldk : a.ldv) + ch * 8) * 2;   // (the head's offset travels in the base pointers)
  };
  int off[NSLOT];
  unsigned long long msk[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    off[i] = slot_off(i, KT - 1);
    msk[i] = __ballot(off[i] >= 0);
    off[i] = max(off[i], 0);
  }
  const unsigned smem_base = (unsigned)(size_t)smem;

  int seg_l = 0, st_l = 0;    // load cursor: segment, stage inside the segment
  const long hk = a.hsk > 0 ? a.hsk : DH, hv = a.hsv > 0 ? a.hsv : DH;   // head-major K / V panels (ABI 6), or heads as column slices of the rows
  const char *kptr_l = K, *vptr_l = V;
  auto seg_base = [&]() {
    const int kit = seg_l == 0 ? kit3[0] : (seg_l == 1 ? kit3[1] : kit3[2]);
    kptr_l = K + ((long)kit * a.nk * a.ldk + h * hk) * 2;
    vptr_l = V + ((long)kit * a.nk * a.ldv + h * hv) * 2;
  };
  auto dma_stage = [&](int si) {   // fetch the load cursor's stage into buffer si % NBUF, advance the cursor
    const unsigned st = smem_base + (si & (NBUF - 1)) * STAGE;
    if constexpr (FOLD) {
      // pad marks: K column DH+1 (next to the ones column, never touched by the DMA) is 60000 for the rows of this stage that lie
      // past nk and 0 elsewhere; the Q fragments carry -1 in that k-slot, so those logits come out of the MFMA at -60000 and
      // their P is exactly 0 -- the tail tile needs no masking code (and the kernel no second tile variant).
      if (a.nk % KT != 0 && wave == NW - 1) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          const int kt = st_l * NSUB + j;
          *reinterpret_cast<f16*>(smem + (si & (NBUF - 1)) * STAGE + j * SUB + lane * KPB + DH * 2 + 2) = (kt * KT + lane >= a.nk) ? (f16)60000.f : (f16)0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      const int kt = st_l * NSUB + j;
      if (kt < ntk) {
        const bool tail = (kt + 1) * KT > a.nk;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          const int q = wave + NW * i;
          if (q < NINS) {
            const char* base = (q < CK ? kptr_l : vptr_l) + (long)j * KT * (q < CK ? a.ldk : a.ldv) * 2;
            const int o = tail ? max(slot_off(i, a.nk - 1 - kt * KT), 0) : off[i];
            dma16_masked(st + j * SUB + q * 1024, base, o, msk[i]);
          }
        }
      }
    }
    if (++st_l == nst) {
      st_l = 0;
      if (++seg_l < nvalid) seg_base();
    } else {
      kptr_l += (long)NSUB * KT * a.ldk * 2;
      vptr_l += (long)NSUB * KT * a.ldv * 2;
    }
  };

  f32x4 o[QT][DT];
  float mrun[QT], lrun[QT];
  auto reset_acc = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      mrun[qt] = NEG_BIG;
      lrun[qt] = 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto prime = [&]() {   // load cursor to the first stage, fetch it, wait for it
    seg_l = st_l = 0;
    if (T > 0) {
      seg_base();
      dma_stage(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  reset_acc();
  __syncthreads();   // LDS image initialised before the first DMA lands
  // FOLD: the probe -- the K rows at the block's OWN query positions in the item's last segment (its own frame in every table
  // of the model) go into the K regions of the stage buffers first.  A query's heaviest keys are itself and its neighbours
  // in the row order; the fixed offset of phase A is taken from this 64-key neighbourhood (below), not from whatever the
  // first keys of the first segment happen to be, so that trained, peaky attention maps do not trip the overflow check.
  constexpr int NPROBE = FOLD ? ((BQ / KT) < NBUF * NSUB ? (BQ / KT > 0 ? BQ / KT : 1) : NBUF * NSUB) : 0;
  if constexpr (FOLD) {
    if (T > 0) {
      const int kitp = nvalid == 1 ? kit3[0] : (nvalid == 2 ? kit3[1] : kit3[2]);   // (no dynamic index: the array would move to scratch)
      const char* kseg = K + ((long)kitp * a.nk * a.ldk + h * hk) * 2;
#pragma unroll
      for (int t = 0; t < NPROBE; ++t) {
        const int ktp = min(qb * (BQ / KT) + t, ntk - 1);
        const unsigned st = smem_base + (t / NSUB) * STAGE + (t % NSUB) * SUB;
        const bool tail = (ktp + 1) * KT > a.nk;
        if (a.nk % KT != 0 && wave == NW - 1)
          *reinterpret_cast<f16*>(smem + (t / NSUB) * STAGE + (t % NSUB) * SUB + lane * KPB + DH * 2 + 2) = (ktp * KT + lane >= a.nk) ? (f16)60000.f : (f16)0.f;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          const int q = wave + NW * i;
          if (q < CK) dma16_masked(st + q * 1024, kseg + (long)ktp * KT * a.ldk * 2, tail ? max(slot_off(i, a.nk - 1 - ktp * KT), 0) : off[i], msk[i]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // FOLD (phase A, the speculative fast path): the reference offset m~ of a query (mrun) rides in k-slot DH of its Q fragment,
  // against the ones column of K: the QK^T MFMA then delivers  s * c - m~  directly and P = 2^(that), no VALU pass between.
  // m~ = ceil(row maximum over the query's PROBE tile: the 64 keys around its own position in its own frame), an integer
  // (exact in fp16), and stays fixed: a softmax is exact for any offset as long as P stays inside fp16's range.  Keys
  // lighter than the probe's maximum underflow exactly as they do under a running maximum; a key more than 2^16 x HEAVIER would overflow fp16, which is checked ONCE, after the
  // sweep, on the denominators (below): if any query of the block may have seen a saturated P, the block discards phase A
  // and recomputes with the classic online softmax (phase B).  Conversions round towards zero, so such a P saturates at 65504 -- finite -- and phase A can
  // neither produce inf / NaN nor fault while it runs to its end.
  auto set_ref = [&](int qt, float m) {
    const f16 hm = (f16)(-fminf(fmaxf(m, -2000.f), 2000.f));
    if (g == GS) fq[qt][D32 - 1][0] = hm;
    mrun[qt] = -(float)hm;
  };
  auto set_refw = [&](float m) {
    const f16 hm = (f16)(-fminf(fmaxf(m, -2000.f), 2000.f));
    if (hi32 == 1) fqw[2][0] = hm;
    mq = -(float)hm;
  };
  auto mfma32 = [](f16x8 x, f16x8 y, f32x16 z) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, z, 0, 0, 0); };
  // S^T of one 32-key block (keys b*32 .. +31 of the sub-tile at sK) against the wave's 32 queries
  auto qk32 = [&](const char* sK, int b) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const char* row = sK + (b * 32 + krow32) * KPB + hi32 * 16;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) acc = mfma32(*reinterpret_cast<const f16x8*>(row + ks * 32), fqw[ks], acc);
    return acc;
  };
  if constexpr (W32) {
    if (T > 0) {
      const int tw = min((wave * QT * 16) / KT, NPROBE - 1);
      const char* sK = smem + (tw / NSUB) * STAGE + (tw % NSUB) * SUB;
      float mr = NEG_BIG;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const f32x16 s0 = qk32(sK, b);
#pragma unroll
        for (int i = 0; i < 16; ++i) mr = fmaxf(mr, s0[i]);   // keys past nk: pad marks put them at -60000
      }
      set_refw(ceilf(xor32_max(mr)));
    }
    __syncthreads();   // every wave has read its probe tile: the buffers may be overwritten
  } else if constexpr (FOLD) {
    if (T > 0) {   // initial offset = ceil(row maximum over the probe tile that holds this wave's queries): one extra QK^T, no P, no PV
      const int tw = min((wave * QT * 16) / KT, NPROBE - 1);
      const char* sK = smem + (tw / NSUB) * STAGE + (tw % NSUB) * SUB;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 s0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D32; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            s0[t] = mfma16(*reinterpret_cast<const f16x8*>(sK + (t * 16 + l15) * KPB + (ks * 4 + g) * 16), fq[qt][ks], s0[t]);
        float mr = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) mr = fmaxf(mr, s0[t][r]);   // keys past nk: pad marks put them at -60000
        mr = xor32_max(xor16_max(mr));
        set_ref(qt, ceilf(mr));
      }
    }
    __syncthreads();   // every wave has read its probe tile: the buffers may be overwritten
  }
  prime();

  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;
  auto tile = [&](auto full_c, auto fold_c, const char* st, int kt) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr bool FOLDT = decltype(fold_c)::value;
    const char* sK = st;
    const char* sV = st + KBYTES;
    const int kbase = kt * KT + g * 4;  // + t*16 + r

    // ---- S^T = K Q^T ----
    f32x4 s[QT][4];
    f16x8 pf[QT][2];
    auto qk = [&]() {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < D32; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f16x8 fk = *reinterpret_cast<const f16x8*>(sK + (t * 16 + l15) * KPB + (ks * 4 + g) * 16);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma16(fk, fq[qt][ks], s[qt][t]);
        }
      }
      if constexpr (!FULL) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kbase + t * 16 + r >= a.nk) s[qt][t][r] = NEG_BIG;   // raw logit; c < 1 keeps NEG_BIG * c finite
      }
    };
    auto pv = [&]() {
      // ---- O^T += V^T P^T: operand A k-slot (g, j) = key kk*32 + (j>>2)*16 + g*4 + (j&3), as P^T above ----
      const unsigned vlane = (unsigned)(size_t)(sV + (g * 4 + (l15 >> 2)) * VPB + (l15 & 3) * 8);
      {
        // transposed reads one d-block ahead of the MFMAs that consume them
        uint2 rv[2][4];   // [parity of dt][kk*2 + j]
        auto issue = [&](auto dt_c) {
          constexpr int dt = decltype(dt_c)::value;
          rv[dt & 1][0] = lds_tr16<(0) * VPB + dt * 32>(vlane);
          rv[dt & 1][1] = lds_tr16<(16) * VPB + dt * 32>(vlane);
          rv[dt & 1][2] = lds_tr16<(32) * VPB + dt * 32>(vlane);
          rv[dt & 1][3] = lds_tr16<(48) * VPB + dt * 32>(vlane);
        };
        issue(std::integral_constant<int, 0>{});
        static_for(std::make_integer_sequence<int, DT>{}, [&](auto dt_c) {
          constexpr int dt = decltype(dt_c)::value;
          if constexpr (dt + 1 < DT) {
            issue(std::integral_constant<int, dt + 1>{});
            lds_wait<4>(rv[dt & 1][0], rv[dt & 1][1], rv[dt & 1][2], rv[dt & 1][3]);
          } else {
            lds_wait<0>(rv[dt & 1][0], rv[dt & 1][1], rv[dt & 1][2], rv[dt & 1][3]);
          }
  #pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            union { uint2 u[2]; f16x8 v; } fv;
            fv.u[0] = rv[dt & 1][kk * 2];
            fv.u[1] = rv[dt & 1][kk * 2 + 1];
  #pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma16(fv.v, pf[qt][kk], o[qt][dt]);
          }
        });
      }
    };

    if constexpr (FOLDT) {
      // P = 2^(MFMA result), rounded TOWARDS ZERO to fp16 (the denominator is the sum of the same rounded values, so the
      // bias cancels); P >= 0, so the unsigned order of the bit patterns is the value order.
      qk();
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float psum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          union { f16x2 h[4]; f16x8 v; } f;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
#if ME_ATTN_ABL & 4
            const float p0 = s[qt][2 * kk][2 * h2], p1 = s[qt][2 * kk][2 * h2 + 1], p2 = s[qt][2 * kk + 1][2 * h2], p3 = s[qt][2 * kk + 1][2 * h2 + 1];
#else
            const float p0 = __builtin_amdgcn_exp2f(s[qt][2 * kk][2 * h2]), p1 = __builtin_amdgcn_exp2f(s[qt][2 * kk][2 * h2 + 1]);
            const float p2 = __builtin_amdgcn_exp2f(s[qt][2 * kk + 1][2 * h2]), p3 = __builtin_amdgcn_exp2f(s[qt][2 * kk + 1][2 * h2 + 1]);
#endif
            f.h[h2] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            f.h[2 + h2] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(p2, p3));
            if constexpr (!ONES) psum += (p0 + p1) + (p2 + p3);
          }
          pf[qt][kk] = f.v;
        }
        if constexpr (!ONES) lrun[qt] += psum;
      }
      pv();
    } else {
      qk();
      // ---- online softmax (per query = per lane column), P^T packed to fp16 MFMA B fragments ----
  #pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mr = fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), s[qt][0][2]);
        mr = fmaxf(fmaxf(mr, s[qt][0][3]), s[qt][1][0]);
        mr = fmaxf(fmaxf(mr, s[qt][1][1]), s[qt][1][2]);
        mr = fmaxf(fmaxf(mr, s[qt][1][3]), s[qt][2][0]);
        mr = fmaxf(fmaxf(mr, s[qt][2][1]), s[qt][2][2]);
        mr = fmaxf(fmaxf(mr, s[qt][2][3]), s[qt][3][0]);
        mr = fmaxf(fmaxf(mr, s[qt][3][1]), s[qt][3][2]);
        mr = fmaxf(mr, s[qt][3][3]);
        mr = xor32_max(xor16_max(mr));
        const float mnew = fmaxf(mrun[qt], mr * c);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        mrun[qt] = mnew;
        float p[4][4];
        float psum = 0.f;
        const f32x2 c2 = {c, c}, nm2 = {-mnew, -mnew};
  #pragma unroll
        for (int t = 0; t < 4; ++t)
  #pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2 sv = {s[qt][t][2 * h2], s[qt][t][2 * h2 + 1]};
            const f32x2 x = __builtin_elementwise_fma(sv, c2, nm2);
            p[t][2 * h2] = __builtin_amdgcn_exp2f(x[0]);
            p[t][2 * h2 + 1] = __builtin_amdgcn_exp2f(x[1]);
            if constexpr (!ONES) psum += p[t][2 * h2] + p[t][2 * h2 + 1];
          }
        if constexpr (!ONES) lrun[qt] = lrun[qt] * alpha + psum;
        // rescale only when some query of the wave saw its running max move (exact: alpha == 1 otherwise)
        if (__builtin_amdgcn_readfirstlane(__any(alpha != 1.0f))) {
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
        }
  #pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          union { f16x2 h[4]; f16x8 v; } f;
  #pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            f.h[h2] = __builtin_convertvector((f32x2){p[2 * kk][2 * h2], p[2 * kk][2 * h2 + 1]}, f16x2);
            f.h[2 + h2] = __builtin_convertvector((f32x2){p[2 * kk + 1][2 * h2], p[2 * kk + 1][2 * h2 + 1]}, f16x2);
          }
          pf[qt][kk] = f.v;
        }
      }

      pv();
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the next tile's K fragment reads below this tile's PV: 8 x 4 registers the kernel does not have
  };

#ifndef ME_ATTN_REBASE_AT
#define ME_ATTN_REBASE_AT 512.f
#endif
  constexpr float REBASE_AT = ME_ATTN_REBASE_AT, REBASE_TO_LOG2 = 5.f;   // (a lane of the kernels without the ones row holds a quarter of the denominator: it re-bases a little later)
  // W32 re-basing (see "FOLD hedge" below), taken once per 128 keys in the MIDDLE of a tile -- after the tile's first QK^T MFMAs are issued, so that
  // the test reads PV accumulators that were finished a tile ago instead of stalling on the ones just issued (at the stage end it cost 2.4 % of
  // the kernel).  Offsets live per query lane (mq, k-slot 40 of fqw[2]); the denominators sit in the PV accumulators' lane order (query (qt, l15)):
  // lane L's query is ((L >> 4) & 1, L & 15), so the shift of ITS query is dsh[(L >> 4) & 1] as computed in this very lane.  The logits `sw` that
  // are already in flight carry the old offset: they move down by the same (integer, exact) amount.
  auto rebase_w32 = [&](f32x16& sw) {
    if constexpr (!W32) return;
    constexpr int LT = W32 ? DH / 16 : 0;   // (the accumulator tile that holds the ones row; only the W32 instantiation has it)
    float dsh[QT];
    bool any = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      dsh[qt] = 0.f;
      const float lq = (g == (DH % 16) / 4) ? o[qt][LT][(DH % 16) % 4] : 0.f;
      if (__builtin_amdgcn_readfirstlane(__any(lq > REBASE_AT))) {
        any = true;
        const float l = __shfl(o[qt][LT][(DH % 16) % 4], ((DH % 16) / 4) * 16 + l15, 64);
        dsh[qt] = (l > REBASE_AT && l < 65504.f) ? ceilf(__log2f(l)) - REBASE_TO_LOG2 : 0.f;
        const float sc = __builtin_amdgcn_exp2f(-dsh[qt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= sc;
      }
    }
    if (any) {
      const float dq = (lane & 16) ? dsh[QT - 1] : dsh[0];
      set_refw(mq + dq);
#pragma unroll
      for (int i = 0; i < 16; ++i) sw[i] -= dq;
    }
  };
  // W32 tile (phase A only): 6 MFMAs (32x32x16) -> 32 exp2 + 16 cvt_pkrtz + 8 permlane16_swap -> the 12 PV MFMAs of the 16x16 path
  auto tile32 = [&](const char* st, auto check_c) {
    const char* sK = st;
    const char* sV = st + KBYTES;
    f16x8 pf[QT][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x16 sw = qk32(sK, b);
      if constexpr (decltype(check_c)::value) {
        if (b == 0) rebase_w32(sw);
      }
      unsigned cw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#if ME_ATTN_ABL & 4
        cw[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(sw[2 * j], sw[2 * j + 1]));
#else
        cw[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(sw[2 * j]), __builtin_amdgcn_exp2f(sw[2 * j + 1])));
#endif
      }
      // (c0, c2), (c1, c3), (c4, c6), (c5, c7): the first of each pair ends up with queries 0..15 in all four lane groups, the second with 16..31
      const auto r0 = __builtin_amdgcn_permlane16_swap(cw[0], cw[2], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(cw[1], cw[3], false, false);
      const auto r2 = __builtin_amdgcn_permlane16_swap(cw[4], cw[6], false, false);
      const auto r3 = __builtin_amdgcn_permlane16_swap(cw[5], cw[7], false, false);
      union { unsigned u[4]; f16x8 v; } f0, f1;
      f0.u[0] = r0[0]; f0.u[1] = r1[0]; f0.u[2] = r2[0]; f0.u[3] = r3[0];
      f1.u[0] = r0[1]; f1.u[1] = r1[1]; f1.u[2] = r2[1]; f1.u[3] = r3[1];
      pf[0][b] = f0.v;
      pf[QT - 1][b] = f1.v;   // (QT == 2 whenever this lambda is called)
    }
    // ---- O^T += V^T P^T, as in tile() ----
    const unsigned vlane = (unsigned)(size_t)(sV + (g * 4 + (l15 >> 2)) * VPB + (l15 & 3) * 8);
    uint2 rv[2][4];
    auto issue = [&](auto dt_c) {
      constexpr int dt = decltype(dt_c)::value;
      rv[dt & 1][0] = lds_tr16<(0) * VPB + dt * 32>(vlane);
      rv[dt & 1][1] = lds_tr16<(16) * VPB + dt * 32>(vlane);
      rv[dt & 1][2] = lds_tr16<(32) * VPB + dt * 32>(vlane);
      rv[dt & 1][3] = lds_tr16<(48) * VPB + dt * 32>(vlane);
    };
    issue(std::integral_constant<int, 0>{});
    static_for(std::make_integer_sequence<int, DT>{}, [&](auto dt_c) {
      constexpr int dt = decltype(dt_c)::value;
      if constexpr (dt + 1 < DT) {
        issue(std::integral_constant<int, dt + 1>{});
        lds_wait<4>(rv[dt & 1][0], rv[dt & 1][1], rv[dt & 1][2], rv[dt & 1][3]);
      } else {
        lds_wait<0>(rv[dt & 1][0], rv[dt & 1][1], rv[dt & 1][2], rv[dt & 1][3]);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        union { uint2 u[2]; f16x8 v; } fv;
        fv.u[0] = rv[dt & 1][kk * 2];
        fv.u[1] = rv[dt & 1][kk * 2 + 1];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma16(fv.v, pf[qt][kk], o[qt][dt]);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };

  // FOLD hedge: once per stage, a query whose denominator has grown past 2^9 -- its keys are getting heavier than its probe tile
  // promised (or there are simply many of them) -- moves its fixed offset up by d = ceil(log2(denominator)) - 5 and scales its
  // accumulators by 2^-d (a power of two; numerator and denominator alike), which puts the denominator back into (16, 32].  Gradual
  // growth away from the probe thus never reaches fp16's range -- the softmax is exact for any offset -- while keys down to 2^-19 of the
  // denominator keep full fp16 precision (re-basing all the way down to 1, as a running maximum does, pushed the light keys of a peaky row
  // into fp16's subnormals: 2-3 ulp on such rows instead of 1).  One compare and a wave-uniform branch per query tile and stage; only a
  // jump of more than ~2^7 in the denominator inside ONE stage still ends in phase B.
  // (a lane without the ones row tests its QUARTER of the denominator; where a stage holds 128 keys -- dh = 80 at 32 queries per wave -- the test comes half
  // as often per key, so its threshold is a quarter: the growth one test interval may bring, 65504 / (4 x threshold), stays 2^7)
  constexpr float REBASE_LANE = (!ONES && NSUB > 1) ? REBASE_AT * 0.25f : REBASE_AT;
  auto rebase = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float lq;
      if constexpr (ONES) lq = (g == (DH % 16) / 4) ? o[qt][DH / 16][(DH % 16) % 4] : 0.f;   // the ones row of the PV accumulators
      else lq = lrun[qt];                                                                     // this lane's share of the denominator
      if (__builtin_amdgcn_readfirstlane(__any(lq > REBASE_LANE))) {
        float l;
        if constexpr (ONES) l = __shfl(o[qt][DH / 16][(DH % 16) % 4], ((DH % 16) / 4) * 16 + l15, 64);
        else l = xor32_sum(xor16_sum(lrun[qt]));
        // per query (lane column); the four lane groups of a query agree.  A denominator at or beyond 65504 may already contain a saturated P:
        // that query is left alone -- its denominator can only grow, and the check after the sweep sends the block to phase B.
        const float d = (l > REBASE_LANE && l < 65504.f) ? ceilf(__log2f(l)) - REBASE_TO_LOG2 : 0.f;
        const float sc = __builtin_amdgcn_exp2f(-d);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= sc;
        if constexpr (!ONES) lrun[qt] *= sc;
        set_ref(qt, mrun[qt] + d);
      }
    }
  };
  const int nfull = a.nk / KT;
  auto sweep = [&](auto fold_c) {
    int st_c = 0;   // compute cursor: stage inside the segment
    for (int si = 0; si < T; ++si) {
      if (NBUF > 1 && si + 1 < T && !(ME_ATTN_ABL & 1)) dma_stage(si + 1);
      const char* st = smem + (si & (NBUF - 1)) * STAGE;
      if constexpr (W32 && decltype(fold_c)::value) {
        static_for(std::make_integer_sequence<int, NSUB>{}, [&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          if (st_c * NSUB + j < ntk && !(ME_ATTN_ABL & 8)) tile32(st + j * SUB, std::integral_constant<bool, (NSUB == 1 || (j & 1) == 1) && !(ME_ATTN_ABL & 16)>{});
        });
      }
#pragma unroll
      for (int j = 0; j < NSUB; ++j) {
        const int kt = st_c * NSUB + j;
        if constexpr (W32 && decltype(fold_c)::value) {
        } else if constexpr (FOLD) {   // keys past nk are masked through the K image (pad marks, see dma_stage): one tile variant only
          if (kt < ntk && !(ME_ATTN_ABL & 8)) tile(BT{}, fold_c, st + j * SUB, kt);
        } else {
          if (kt < nfull) tile(BT{}, fold_c, st + j * SUB, kt);
          else if (kt < ntk) tile(BF{}, fold_c, st + j * SUB, kt);
        }
      }
      if constexpr (FOLD && !W32 && decltype(fold_c)::value && !(ME_ATTN_ABL & 16)) rebase();
      if (++st_c == nst) st_c = 0;
      if (NBUF == 1 && si + 1 < T) {
        __syncthreads();   // single buffer: every wave is done reading it
        dma_stage(si + 1);
      }
      if (!(ME_ATTN_ABL & 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next stage have landed ...
        __syncthreads();                                    // ... and so have everyone else's; this stage is free
      }
    }
  };
  // ---- finalize: O^T[d = dt*16 + g*4 + r][q = l15] ----
  const float* __restrict__ vsum = reinterpret_cast<const float*>(a.vsum);
  auto finalize = [&]() {
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l;
    if constexpr (ONES) {   // denominator sits in accumulator row DH: tile DH/16, lanes g == (DH%16)/4, reg (DH%16)%4
      l = __shfl(o[qt][DH / 16][(DH % 16) % 4], ((DH % 16) / 4) * 16 + l15, 64);
    } else {
      l = xor32_sum(xor16_sum(lrun[qt]));
    }
    // log2 of the softmax denominator in exp2 units (offset + log2 of the sum): what me_attn_bwd needs to rebuild P per tile
    if (a.lse && ndual == 0 && g == 0 && qrow[qt] >= 0) reinterpret_cast<float*>(a.lse)[(long)qrow[qt] * a.heads + h] = mrun[qt] + __log2f(l);
    float f1 = 1.0f, f2 = 0.f;
    if (ndual > 0) {   // wave-uniform
      const float mp = fmaxf(mrun[qt], 0.f);
      f1 = __builtin_amdgcn_exp2f(mrun[qt] - mp);
      f2 = __builtin_amdgcn_exp2f(-mp);
      l = l * f1 + f2 * (float)(ndual * a.nk);
    }
    const float inv = 1.0f / l;
    if (qrow[qt] < 0) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d >= DH) continue;
      f32x4 acc = o[qt][dt];
      if (ndual > 0) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sgi = 0; sgi < 3; ++sgi) {
          if (dual3[sgi]) b += *reinterpret_cast<const f32x4*>(vsum + ((long)kit3[sgi] * a.heads + h) * DH + d);
        }
        acc = acc * f1 + b * f2;
      }
      U64 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(acc[r] * inv);
      *reinterpret_cast<uint2*>(O + (long)qrow[qt] * a.ldo + h * DH + d) = ov.u;
    }
  }
  };
  // KVRES with 65 .. 80 keys -- the 77-key text cross-attention itself: the item's five 16-key tiles in ONE pass.  Every logit of a query is in registers at
  // once (QT x 20), so the softmax is the plain one: one maximum, one exp2 sweep, no running rescale, no second 64-key tile that is 80 % padding
  // (20 + 18 MFMAs per 32 queries instead of 32 + 24, 40 exp2 instead of 64, a quarter of the classic tile's bookkeeping).
  auto tile80 = [&](const char* st) {
    constexpr int NT = 5, NKK = 3;
    static_assert(NSUB >= 2 || !KVRES, "tile80 reads the first tile of the second sub-tile");
    f32x4 s[QT][NT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < NT; ++t) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D32; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 fk = *reinterpret_cast<const f16x8*>(st + (t >> 2) * SUB + ((t & 3) * 16 + l15) * KPB + (ks * 4 + g) * 16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma16(fk, fq[qt][ks], s[qt][t]);
      }
    }
    f16x8 pf[QT][NKK];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((NT - 1) * 16 + g * 4 + r >= a.nk) s[qt][NT - 1][r] = NEG_BIG;   // (64 < nk <= 80: only the last tile has keys past nk)
      float mr = s[qt][0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mr = fmaxf(mr, s[qt][t][r]);
      mr = xor32_max(xor16_max(mr));
      const float mnew = mr * c;
      mrun[qt] = mnew;
      float psum = 0.f;
      f16x2 ph[2 * NKK][2];
#pragma unroll
      for (int t = 0; t < 2 * NKK; ++t)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          if (t < NT) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][t][2 * h2], c, -mnew)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][t][2 * h2 + 1], c, -mnew));
            if constexpr (!ONES) psum += p0 + p1;
            ph[t][h2] = __builtin_convertvector((f32x2){p0, p1}, f16x2);
          } else {
            ph[t][h2] = f16x2{(f16)0.f, (f16)0.f};
          }
        }
      if constexpr (!ONES) lrun[qt] = psum;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        union { f16x2 h[4]; f16x8 v; } f;
        f.h[0] = ph[2 * kk][0]; f.h[1] = ph[2 * kk][1]; f.h[2] = ph[2 * kk + 1][0]; f.h[3] = ph[2 * kk + 1][1];
        pf[qt][kk] = f.v;
      }
    }
    // O^T = V^T P^T: k-slot (g, j) of step kk = key kk*32 + (j>>2)*16 + g*4 + (j&3), i.e. 16-key tiles 2 kk and 2 kk + 1 (tile 5 does not exist: zeros)
    const unsigned vlane = (unsigned)(size_t)(st + KBYTES + (g * 4 + (l15 >> 2)) * VPB + (l15 & 3) * 8);
    static_for(std::make_integer_sequence<int, DT>{}, [&](auto dt_c) {
      constexpr int dt = decltype(dt_c)::value;
      uint2 r0 = lds_tr16<0 * 16 * VPB + dt * 32>(vlane), r1 = lds_tr16<1 * 16 * VPB + dt * 32>(vlane);
      uint2 r2 = lds_tr16<2 * 16 * VPB + dt * 32>(vlane), r3 = lds_tr16<3 * 16 * VPB + dt * 32>(vlane);
      uint2 r4 = lds_tr16<SUB + dt * 32>(vlane);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4));
      union { uint2 u[2]; f16x8 v; } f0, f1, f2;
      f0.u[0] = r0; f0.u[1] = r1;
      f1.u[0] = r2; f1.u[1] = r3;
      f2.u[0] = r4; f2.u[1] = uint2{0u, 0u};
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        o[qt][dt] = mfma16(f0.v, pf[qt][0], o[qt][dt]);
        o[qt][dt] = mfma16(f1.v, pf[qt][1], o[qt][dt]);
        o[qt][dt] = mfma16(f2.v, pf[qt][2], o[qt][dt]);
      }
    });
  };
  // KVRES epilogue: the launch is paced by the CU's in-order memory pipeline (profiles/r06_attn_kvres.txt: loads + arithmetic 0.09 ms, with the direct
  // epilogue's stores 0.196 -- 8-byte stores that touch 16 rows each), so the wave parks its fp16 tile in its private LDS scratch (MFMA layout: lane (q, g)
  // holds 4 consecutive d of query q) and stores it as 16-byte pieces of whole dh-wide row slices, 12.8 / 6.4 rows per instruction instead of 16 rows per
  // HALF-size instruction.  Wave-private, LDS operations of a wave execute in order: no barrier.
  auto finalize_rows = [&]() {
    constexpr int PITCH = DH * 2, CPR = DH / 8, NPIECE = OROWS * CPR, NIT = (NPIECE + 63) / 64;
    char* sc = smem + NBUF * STAGE + 64 + wave * OSCR;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float l;
      if constexpr (ONES) l = __shfl(o[qt][DH / 16][(DH % 16) % 4], ((DH % 16) / 4) * 16 + l15, 64);
      else l = xor32_sum(xor16_sum(lrun[qt]));
      if (a.lse && g == 0 && qrow[qt] >= 0) reinterpret_cast<float*>(a.lse)[(long)qrow[qt] * a.heads + h] = mrun[qt] + __log2f(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + g * 4;
        if (d >= DH) continue;
        U64 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(o[qt][dt][r] * inv);
        *reinterpret_cast<uint2*>(sc + (qt * 16 + l15) * PITCH + d * 2) = ov.u;
      }
    }
    const int q0 = qb * BQ + wave * OROWS;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int p = i * 64 + lane;
      const int row = p / CPR, ch = p - row * CPR;
      if (p < NPIECE && q0 + row < a.nq)
        *reinterpret_cast<uint4*>(O + ((long)item * a.nq + q0 + row) * a.ldo + h * DH + ch * 8) = *reinterpret_cast<const uint4*>(sc + row * PITCH + ch * 16);
    }
  };
  if constexpr (KVRES) {   // (T == 1: one stage holds every key; prime() above has fetched it)
    const char* st = smem;
    for (int it = 0; it < qpb && qb < nqb; ++it, ++qb) {
      if (it > 0) {
        load_q(BF{});
        reset_acc();
      }
#ifndef ME_KVRES_ABL   // ablation builds (tools/build_abl.sh): 1 = no tile arithmetic, 2 = no O stores (an impossible condition keeps the code alive)
#define ME_KVRES_ABL 0
#endif
      if (T > 0 && !(ME_KVRES_ABL & 1)) tile80(st);   // (me_attn launches this form for 64 < nk <= 80 only)
      if (!(ME_KVRES_ABL & 2) || a.nk < 0) {
        if (ME_KVRES_ABL & 4) finalize();   // (ablation: the direct epilogue)
        else finalize_rows();
      }
    }
    return;
  }
  if constexpr (FOLD) {
    sweep(BT{});
    // one check per block: could any P have saturated?  P >= 0, so a query's denominator (the sum of its P, accumulated in
    // fp32 by the ones row of the PV MFMA, or in lrun) bounds every one of them: denominator < 65504 => every P was below
    // fp16's largest value, i.e. exact.  A saturated P (stored as 65504) always trips it; the test is conservative (flat
    // attention over more than 65504 keys would trip it too) and costs nothing per tile.
    // Then phase A's result is discarded: phase B = the classic running-maximum sweep over the same keys with the unscaled Q
    // (K's ones column then meets a zero k-slot).
    __shared__ int trip_flag;
    if (tid == 0) trip_flag = 0;
    __syncthreads();
    bool trip = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // (rows past nq carry a zero query: P = 1 for every key, a denominator of n keys -- they must not trip the block)
      if constexpr (ONES) trip |= qrow[qt] >= 0 && (g == (DH % 16) / 4) && !(o[qt][DH / 16][(DH % 16) % 4] < 65504.f);
      else trip |= qrow[qt] >= 0 && !(lrun[qt] < 65504.f);
    }
    if (trip) trip_flag = 1;
    __syncthreads();
    if constexpr (W32) {   // the epilogue wants the offsets in the accumulators' lane order
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) mrun[qt] = __shfl(mq, qt * 16 + l15, 64);
    }
    if (trip_flag) {
      if (tid == 0) atomicAdd(&g_fallback_blocks, 1ull);   // diagnostic: me_attn_fallback_blocks()
      load_q(BF{});
      reset_acc();
      prime();
      sweep(BF{});
    }
  } else {
    sweep(BF{});
  }

  finalize();
}

// column sums of V per kv item: vsum[kit][c] = sum_key V[kit*nk + key][c], fp32 (the query-independent "+1" part of the
// binary dual segments).  Two deterministic passes: block (kit, row split rs) sums its rows with 16-byte loads -- a thread owns
// one 8-column vector and walks down the rows, row lanes are folded through LDS in a fixed order -- into part[rs][kit][C];
// the second pass adds the RS partials in order.  (The first version gave every kv item to ONE block per 128 columns: 288
// blocks of 4-byte loads, 0.9 TB/s.)
constexpr int COLSUM_RS = 16;
__global__ __launch_bounds__(256) void colsum_part_kernel(const f16* __restrict__ V, int ldv, int nk, int C, float* __restrict__ part, int n_items, int dh, long hsv) {
  __shared__ float red[256][8];
  const int kit = blockIdx.x, rs = blockIdx.y;
  const int tpr = C / 8, RL = 256 / tpr;          // vectors per row, row lanes (C <= 2048)
  const int rl = threadIdx.x / tpr, vc = threadIdx.x - rl * tpr;
  const int rows = (nk + COLSUM_RS - 1) / COLSUM_RS, r0 = rs * rows, r1 = min(r0 + rows, nk);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    const int col = vc * 8, hd = col / dh;   // head-major V: the vector's head panel (8 | dh: a vector never straddles heads)
    const f16* p = V + (long)kit * nk * ldv + (hsv > 0 ? hd * hsv + (col - hd * dh) : col);
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
      U128 u;
      u.u = ldg128(p + (long)r * ldv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)u.e[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float sacc = 0.f;
    for (int l = 0; l < RL; ++l) sacc += red[l * tpr + c / 8][c % 8];
    part[((long)rs * n_items + kit) * C + c] = sacc;
  }
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float sacc = 0.f;
#pragma unroll
  for (int rs = 0; rs < COLSUM_RS; ++rs) sacc += part[rs * n + i];
  out[i] = sacc;
}

template <int DH, int QT, int NW, int MINW, int NBUF, int NSUB, bool FOLD = (DH % 32 != 0), bool KVRES = false>
int launch_attn2(const me_attn_args* a, hipStream_t st) {
  constexpr int BQ = 16 * QT * NW;
  const int nqb = (a->nq + BQ - 1) / BQ;
  long total = (long)a->n_items * a->heads * nqb;
  me_attn_args b = *a;
  {   // block order (see attn2_kernel): heads slowest for launches whose items share K | V frames with their neighbours (>= 2 segments); ME_ATTN_ORDER=0: A/B
    const char* e = getenv("ME_ATTN_ORDER");
    if (a->nseg >= 2 && !(e && e[0] == '0')) b.general_dual |= ATTN_HEAD_SLOWEST;
  }
  if constexpr (KVRES) {
    // query blocks per block: the largest divisor of nqb that still leaves >= 6 blocks per CU (three rounds of the two resident ones) -- each block stages
    // its K | V once; the walk length hardly matters beyond 4 (tools/exp_kvres_qpanels.py, profiles/r06_attn_kvres.txt)
    int qpb = 1;
    for (int d = 1; d <= nqb; ++d)
      if (nqb % d == 0 && (long)a->n_items * a->heads * (nqb / d) >= 1536) qpb = d;
    if (const char* e = getenv("ME_ATTN_KVRES")) {   // tests: ME_ATTN_KVRES=n (n >= 2) forces n query blocks per block, divisor of nqb or not
      const int n = atoi(e);
      if (n >= 2) qpb = n < nqb ? n : nqb;
    }
    b.general_dual |= qpb << 8;
    total = (long)a->n_items * a->heads * ((nqb + qpb - 1) / qpb);
  }
  hipLaunchKernelGGL((attn2_kernel<DH, QT, NW, MINW, NBUF, NSUB, FOLD, KVRES>), dim3((unsigned)total), dim3(64 * NW), 0, st, b);
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "attn2_kernel<%d,%d,%d,%s>", DH, QT, NW, KVRES ? "kvres" : (FOLD ? "fold" : "classic"));   // (the variants of a shape are different kernels: bench.py / pmc_summary.py key on this)
    me_set_kernel(nm);
  }
  return hipGetLastError() == hipSuccess ? ME_OK : ME_EHIP;
}

template <int DH, int QT, bool GD, int MINW, int NBUF>
int launch_attn(const me_attn_args* a, hipStream_t st) {
  constexpr int BQ = 64 * QT;
  const int nqb = (a->nq + BQ - 1) / BQ;
  const long total = (long)a->n_items * a->heads * nqb;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL((attn_kernel<DH, QT, GD, MINW, NBUF>), dim3((unsigned)total), dim3(256), 0, st, *a);
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "attn_kernel<%d,%d,general-dual>", DH, QT);
    me_set_kernel(nm);
  }
  return hipGetLastError() == hipSuccess ? ME_OK : ME_EHIP;
}

}  // namespace

extern "C" void me_set_error(const char* msg);

extern "C" int64_t me_attn_fallback_blocks(int32_t reset) {
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fallback_blocks), sizeof(v)) != hipSuccess) return -1;   // synchronises with the device
  if (reset) {
    const unsigned long long z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fallback_blocks), &z, sizeof(z)) != hipSuccess) return -1;
  }
  return (int64_t)v;
}

extern "C" int64_t me_attn_vsum_bytes(int32_t n_kv_items, int32_t channels) {
  return n_kv_items > 0 && channels > 0 ? (int64_t)(1 + COLSUM_RS) * n_kv_items * channels * (int64_t)sizeof(float) : 0;
}

extern "C" int me_attn(const me_attn_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->V || !a->O || !a->seg_item || !a->seg_mode) { me_set_error("me_attn: null pointer"); return ME_EINVAL; }
  if (a->n_items <= 0 || a->nq <= 0 || a->nk <= 0 || a->heads <= 0 || a->nseg < 1 || a->nseg > 3) { me_set_error("me_attn: bad sizes"); return ME_EINVAL; }
  if (a->ldq % 8 || a->ldk % 8 || a->ldv % 8 || a->ldo % 4) { me_set_error("me_attn: row strides must be multiples of 8 (Q,K,V) / 4 (O)"); return ME_EINVAL; }
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V) & 15 || ((uintptr_t)a->O & 7)) { me_set_error("me_attn: misaligned pointer"); return ME_EINVAL; }
  if (a->q_items < 0 || (a->general_dual && (a->q_items > 0 || a->lse))) { me_set_error("me_attn: q_items / lse are not served by the general-dual kernel"); return ME_EINVAL; }
  if (a->lse && a->vsum) { me_set_error("me_attn: lse is written for plain segments only"); return ME_EINVAL; }
  if (a->hsk < 0 || a->hsv < 0 || a->hsq < 0 || a->hsk % 8 || a->hsv % 8 || a->hsq % 8) { me_set_error("me_attn: head strides must be non-negative multiples of 8"); return ME_EINVAL; }
  if (a->item_order && ((uintptr_t)a->item_order & 3)) { me_set_error("me_attn: misaligned item_order"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (a->general_dual) {   // non-binary masks: the mask-reading kernel
    switch (a->dh) {
      case 40: rc = launch_attn<40, 2, true, 2, 1>(a, st); break;
      case 80: rc = launch_attn<80, 2, true, 2, 1>(a, st); break;
      case 160: rc = launch_attn<160, 1, true, 2, 1>(a, st); break;
      default: me_set_error("me_attn: head dim must be 40, 80 or 160"); return ME_EINVAL;
    }
  } else {
    if (a->vsum) {   // binary dual segments: per-kv-item column sums of V first (same stream)
      if (a->n_kv_items <= 0) { me_set_error("me_attn: vsum needs n_kv_items"); return ME_EINVAL; }
      const int C = a->heads * a->dh;
      if (C % 8 || C > 2048 || ((uintptr_t)a->vsum & 15)) { me_set_error("me_attn: vsum needs heads * dh to be a multiple of 8 and <= 2048, 16-byte aligned scratch"); return ME_EINVAL; }
      // vsum = [n_kv_items][C] sums, followed by COLSUM_RS x [n_kv_items][C] partials (scratch of the two-pass reduction)
      float* vs = reinterpret_cast<float*>(a->vsum);
      const long n = (long)a->n_kv_items * C;
      hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)a->n_kv_items, COLSUM_RS), dim3(256), 0, st, reinterpret_cast<const f16*>(a->V), a->ldv, a->nk, C, vs + n,
                         a->n_kv_items, a->dh, (long)a->hsv);
      hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vs + n, vs, n);
    }
    // 8 waves x 32 queries when the launch has whole 256-query blocks (halves the K/V fill per query), else 4 waves
    static const bool fold_on = !(getenv("ME_ATTN_FOLD") && atoi(getenv("ME_ATTN_FOLD")) == 0);
    const bool fold = fold_on && a->nk >= 256;
    // keys resident across query blocks (round 6): one segment of 65 .. 80 keys -- the 77-key text cross-attention (attention_2d.py:343) -- with whole
    // multi-block query items.  ME_ATTN_KVRES=0: A/B switch (read per call: the tests flip it in-process).
    const char* kvr = getenv("ME_ATTN_KVRES");
    const bool kvres = a->nseg == 1 && a->nk > 64 && a->nk <= 80 && !a->vsum && a->ldo % 8 == 0 && !((uintptr_t)a->O & 15) && !(kvr && kvr[0] == '0');
    switch (a->dh) {
      // fold (speculative fixed-offset softmax with the classic sweep as in-kernel fallback) pays from a few tiles per query on:
      // its start-up is one extra QK^T tile.  The 77-key text cross-attention stays classic.  ME_ATTN_FOLD=0: A/B switch.
      case 40:
        // 16 waves x 32 queries and 256 keys per barrier, one block per CU, when the launch has whole 512-query blocks: half the K/V fill per query
        // of the 8-wave form (measured -2.2 %; ME_ATTN_NW16=0 builds without it)
#if !defined(ME_ATTN_NW16) || ME_ATTN_NW16
        if (fold && a->nq >= 512) { rc = launch_attn2<40, 2, 16, 4, 2, 4, true>(a, st); break; }
#endif
#ifndef ME_KV40_QT   // (experiment builds: query tiles per wave / waves per SIMD of the resident-key kernel)
#define ME_KV40_QT 2
#define ME_KV40_MINW 4
#endif
        if (kvres && a->nq >= 512) { rc = launch_attn2<40, ME_KV40_QT, 8, ME_KV40_MINW, 1, 2, false, true>(a, st); break; }
        if (fold) rc = a->nq >= 256 ? launch_attn2<40, 2, 8, 4, 2, 2, true>(a, st) : launch_attn2<40, 2, 4, 3, 2, 1, true>(a, st);
        else rc = a->nq >= 256 ? launch_attn2<40, 2, 8, 4, 2, 2, false>(a, st) : launch_attn2<40, 2, 4, 3, 2, 1, false>(a, st);
        break;
      case 80:
        if (kvres && a->nq >= 256) { rc = launch_attn2<80, 1, 8, 3, 1, 2, false, true>(a, st); break; }
        // 8 waves x 32 queries and 128 keys per barrier when the launch has whole 256-query blocks (round 6): at 16 queries per wave every K / V fragment read
        // from LDS feeds ONE MFMA (22 KB of fragment reads per 22 MFMAs: the LDS pipe as busy as the matrix pipe); 32 queries halve that.  Level-1 [prev | cur]
        // 0.702 -> 0.648 ms, edited 0.888 -> 0.819, bitwise the same output (profiles/r06_attn80_qt2.txt; with 64-key stages: +-0).  ME_ATTN_80_QT2=0: A/B.
        if (fold && a->nq >= 256) {
          const char* e = getenv("ME_ATTN_80_QT2");
          if (!(e && e[0] == '0')) { rc = launch_attn2<80, 2, 8, 2, 2, 2, true>(a, st); break; }
        }
        if (fold) rc = a->nq >= 128 ? launch_attn2<80, 1, 8, 3, 2, 1, true>(a, st) : launch_attn2<80, 2, 4, 2, 2, 1, true>(a, st);
        else rc = a->nq >= 128 ? launch_attn2<80, 1, 8, 3, 2, 1, false>(a, st) : launch_attn2<80, 2, 4, 2, 2, 1, false>(a, st);
        break;
      case 160:
        // (level 2: 256 queries per item -- two 128-query blocks walk the staged keys; 0.080 -> 0.062 ms per launch, profiles/r06_attn_kvres.txt)
        if (kvres && a->nq >= 256) { rc = launch_attn2<160, 1, 8, 2, 1, 2, false, true>(a, st); break; }
        rc = launch_attn2<160, 1, 4, 2, 1, 1>(a, st);
        break;
      default: me_set_error("me_attn: head dim must be 40, 80 or 160"); return ME_EINVAL;
    }
  }
  if (rc != ME_OK) me_set_error("me_attn: kernel launch failed");
  return rc;
}
