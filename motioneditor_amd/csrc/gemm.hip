// gather-GEMM / implicit convolution on MFMA (gfx950).
//
//   C[m, n] = epilogue( sum_{tap} sum_{c < K} X[src(m, tap), c] * W[n, tap, c] )
//
// One kernel serves nn.Linear, 1x1 conv, 3x3 conv (stride 1/2, optional nearest-2x upsample of the
// input folded into the gather) and the k=3 temporal conv: only the row-gather src(m, tap) differs.
// Activations are channels-last so every tap is a contiguous K-run of one source row.
//
// Tiling: BM x BN x 64 block tile, waves in a (BM/64) x 2 grid, each wave 64 x (BN/2) built from
// v_mfma_f32_16x16x32_f16 with the WEIGHT fragment as MFMA operand A and the ACTIVATION fragment as operand B, so a
// lane ends up holding 4 consecutive output channels of one output row (8-byte stores, epilogue without cross-lane
// traffic).  256 x 320 (8 waves, one block per CU) when the grid still fills the chip, 128 x {160, 128, 64} otherwise.
// The bias is the accumulators' initial value; the epilogue packs the tile to fp16 and adds the row-vector /
// residual terms with packed fp16 adds (see epilogue_rows).
//
// Staging (STAGE_GLDS, default): global_load_lds_dwordx4 -- each wave instruction DMAs 8 rows x 128 B
// straight into LDS (no VGPR round trip, no ds_write).  The LDS image is lane-linear, so the
// bank-conflict fix is an XOR swizzle applied on the per-lane SOURCE chunk and again on the
// ds_read_b128 address (chunk ^= (row >> 1) & 7): every 16-lane group of a fragment read then hits
// 16 distinct 16-byte slots.  Padding taps / tails read a 16-byte zero buffer.  Two LDS buffers:
// the next K-slab's DMA is in flight while the current slab feeds the MFMAs; one barrier per slab.
// STAGE_REG keeps the first implementation (global -> VGPR -> padded LDS) for A/B runs.
//
// Kernels in this file and who gets what (me_gemm, bottom):
//   gemm8p_kernel<256 | 192, 320 | 256, GATHER>  the 8-phase ping-pong schedule (two wave groups half a phase apart, DMA parts in flight across
//                                                 barriers): grids of >= 512 tiles of 256 rows, or >= 448 tiles of 192 rows when 256-row tiles would leave
//                                                 the last block round half empty; GEGLU takes the 256-wide tile
//   conv3_halo_kernel                             3x3 stride-1 convolutions on grids of >= 512 (16 x 16 pixel, 320 channel) patches: the input patch
//                                                 with its halo is staged once per 64 channels, only the weight slab changes per tap
//   gemm_kernel<128, 160 | 128 | 64>              everything smaller (one barrier per slab); grids of < 400 blocks with long K loops are split along K
//                                                 (fp32 partials in me_gemm_args.work + gemm_splitk_reduce_kernel), N >= 1280 only
//   gemm_kernel<256, 320>                         the one-barrier form of the big tile, kept for A/B (ME_GEMM_8P=0) and K % 64 != 0
// The gemm / gemm8p kernels accumulate an output element over (tap, 64-channel slab) in the same order, so their results are bitwise equal
// whatever the tile (tested); the halo kernel walks (slab, tap) and a K split adds partial sums: those two change the fp32 summation order,
// which is why their selection can be pinned to a larger launch's (me_gemm_args.sel_rows) and the split is confined to N >= 1280.
#include "me_common.h"
#include "../../include/motioned.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int BK = 64;
#ifndef WIDE_TERMS
#define WIDE_TERMS 0
#endif
constexpr int STAGE_REG = 0, STAGE_GLDS = 1, STAGE_BUF = 2;
constexpr int ROWEPI_FLAG = 1 << 30, TILEORDER_FLAG = 1 << 29;   // me_gemm_args.splits_ (private to me_gemm): the 8-phase kernel takes the row-contiguous epilogue

// epilogue stores.  NT = non-temporal: used by the GEGLU epilogue only (a [rows, 4C] tensor that the next GEMM streams once: L0 0.832 -> 0.805 ms,
// L2 0.576 -> 0.560); on epilogues that read a residual -- usually the very lines they then write -- non-temporal stores cost 20-50 %.
template <bool NT = false>
__device__ __forceinline__ void st16(f16* p, uint4 v) {
  if constexpr (NT) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}
template <bool NT = false>
__device__ __forceinline__ void st8(f16* p, uint2 v) {
  if constexpr (NT) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((u32x2){v.x, v.y}, reinterpret_cast<u32x2*>(p));
  } else {
    *reinterpret_cast<uint2*>(p) = v;
  }
}

__device__ uint4 g_zero16;  // zero-initialised: source of every padded 16-byte chunk in the GLDS path

struct RowInfo {
  // CONV3: base = img * Hin * Win, y0 = oy*stride - 1, x0 = ox*stride - 1
  // TCONV: base = m, y0 = frame index inside its chunk
  // DENSE: base = m
  int base, y0, x0;
  bool valid;
};

__device__ __forceinline__ RowInfo make_row(const me_gemm_args& a, int m) {
  RowInfo r;
  r.valid = m < a.M;
  r.base = m;
  r.y0 = 0;
  r.x0 = 0;
  if (!r.valid) return r;
  if (a.gather == ME_GATHER_CONV3) {
    const int hw = a.Hout * a.Wout;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int oy = rem / a.Wout;
    const int ox = rem - oy * a.Wout;
    r.base = img * a.Hin * a.Win;
    r.y0 = oy * a.stride - (a.pad0 ? 0 : 1);
    r.x0 = ox * a.stride - (a.pad0 ? 0 : 1);
  } else if (a.gather == ME_GATHER_TCONV) {
    const int bf = m / a.npix;
    r.y0 = bf % a.frames;                                  // local frame
    r.x0 = (bf / a.frames) * a.npix + (m - bf * a.npix);   // b * npix + p: offset inside a halo block
  }
  return r;
}

// source row of (row, tap) or -1 when the tap falls into zero padding
__device__ __forceinline__ int src_row(const me_gemm_args& a, const RowInfo& r, int tap) {
  if (!r.valid) return -1;
  if (a.gather == ME_GATHER_DENSE) return r.base;
  if (a.gather == ME_GATHER_CONV3) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = r.y0 + ky, ix = r.x0 + kx;
    const int sh = a.ups ? 1 : 0;   // ups 1: nearest 2x, ups 2: zero-stuffed 2x (odd virtual pixels are zeros)
    const int Hv = a.Hin << sh, Wv = a.Win << sh;
    if (iy < 0 || iy >= Hv || ix < 0 || ix >= Wv) return -1;
    if (a.ups == 2 && ((iy | ix) & 1)) return -1;
    return r.base + (iy >> sh) * a.Win + (ix >> sh);
  }
  const int dt = tap - 1;  // TCONV over global frames: same chunk, inside the clip
  const int ftot = a.frames_total > 0 ? a.frames_total : a.frames;
  const int gf = a.frame0 + r.y0, gs = gf + dt;
  if (gs < 0 || gs >= ftot || gs / a.chunk != gf / a.chunk) return -1;
  const int ls = r.y0 + dt;
  if (ls >= 0 && ls < a.frames) return r.base + dt * a.npix;
  const int hb = ls < 0 ? a.halo_prev : a.halo_next;    // neighbour rank's boundary frame
  return hb < 0 ? -1 : hb + r.x0;
}

// The bias enters as the accumulators' initial value (bias / alpha, so that alpha * acc adds exactly the bias): its
// loads happen before the K loop, when registers are free, and the epilogue never sees it.
template <int NT, int MT, int WN>
__device__ __forceinline__ void init_acc(const me_gemm_args& a, f32x4 (&acc)[NT][MT], int n0, int wn, int lane, bool use_bias = true) {
  const f16* __restrict__ bias = use_bias ? reinterpret_cast<const f16*>(a.bias) : nullptr;
  const int nb = n0 + wn * WN + (lane >> 4) * 4;
  const float inv_alpha = bias ? 1.0f / a.alpha : 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    U64 b;
    b.u = (bias && nb + 16 * j < a.N) ? *reinterpret_cast<const uint2*>(bias + nb + 16 * j) : make_uint2(0u, 0u);
    const f32x4 v = {(float)b.e[0] * inv_alpha, (float)b.e[1] * inv_alpha, (float)b.e[2] * inv_alpha, (float)b.e[3] * inv_alpha};
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = v;
  }
}

// ABI 9 -- LayerNorm folded into the projection (me_gemm_args.ln_stats): the accumulators hold x W'^T of the UN-normalised rows; map them through
//     rstd[m] * (acc - mean[m] * colsum[n]) + cvec[n]
// before the usual epilogue.  (mean, rstd) come from the partial row sums (sum x, sum x^2) the producing projection's epilogue (or me_ln_stats) left per
// 320-column part; the fused multiply-adds are spelled out so that every kernel of this file rounds the same way (the tile shape must not show).
//   mbase / nbase: first row / column of this wave's tile; lane holds rows mbase + 16 i + (lane & 15), columns nbase + 16 j + 4 (lane >> 4) + r.
template <int NT, int MT>
__device__ __forceinline__ void ln_fold_acc(const me_gemm_args& a, f32x4 (&acc)[NT][MT], int mbase, int nbase, int lane) {
  const float* __restrict__ st = reinterpret_cast<const float*>(a.ln_stats);
  const float* __restrict__ cs = reinterpret_cast<const float*>(a.ln_colsum);
  const float* __restrict__ cv = reinterpret_cast<const float*>(a.ln_cvec);
  const float invK = 1.0f / (float)a.K;
  float nmean[MT], rstd[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = min(mbase + i * 16 + (lane & 15), a.M - 1);   // rows past M: any row that exists (never stored)
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < a.ln_parts; ++p) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(st + (long)p * a.ln_stride + 2 * (long)m);
      s1 += v[0];
      s2 += v[1];
    }
    const float mu = s1 * invK;
    const float var = __builtin_fmaf(-mu, mu, s2 * invK);
    nmean[i] = -mu;
    rstd[i] = rsqrtf(fmaxf(var, 0.f) + a.ln_eps);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = nbase + j * 16 + (lane >> 4) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {
      s = *reinterpret_cast<const f32x4*>(cs + n);
      c = *reinterpret_cast<const f32x4*>(cv + n);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][i][r] = __builtin_fmaf(rstd[i], __builtin_fmaf(nmean[i], s[r], acc[j][i][r]), c[r]);
  }
}

// The same map with its operands read from LDS: the 8-phase kernels (one block per CU -- nothing hides a global load's latency in their epilogue: the
// global form cost 1.4 us per 256-row tile, +2.5 ms per step on each of the two LayerNorm-folded kernel families) fetch the tile's partial row sums, column sums
// and constants by LDS-DMA together with the first K tile (gemm8p_kernel, LN_LDS).  Image: [parts <= 4][BM rows][2 floats] at 2 KB per part | colsum [BN] at
// + 8 KB | cvec [BN] at + 10 KB.  Same values, same operation order as ln_fold_acc: bitwise the same result.
// Measured and not kept (profiles/r06_lnfold_ab.txt): the accumulators STARTING at cvec sd - mean colsum behind the prologue's barrier and the epilogue scaling
// row m by rstd where it scaled by alpha anyway -- the same 2 operations per accumulator, half of them moved to the tile's head: GEGLU kernel 16.2 vs 15.6 ms
// per step (14.2 without the fold), the 320-wide kernel 24.5 vs 24.6 (23.2): at one block per CU VALU work at EITHER end of a tile is exposed.
constexpr int LN_LDS_BYTES = 12288, LN_LDS_COLSUM = 8192, LN_LDS_CVEC = 10240;
template <int NT, int MT>
__device__ __forceinline__ void ln_fold_acc_lds(const me_gemm_args& a, f32x4 (&acc)[NT][MT], const char* img, int rloc, int cloc, int lane) {
  const float invK = 1.0f / (float)a.K;
  // ALL four part slots of the image are read, branch-free (the DMA zero-fills the slots past ln_parts: + 0.0f is exact, the sum keeps the order of
  // ln_fold_acc).  The first version looped over ln_parts: hipcc built a vectorised + remainder loop pair per row, every trip behind its own s_waitcnt --
  // ~300 instructions of control and waits for 8 x ln_parts reads, 2 - 3 us per tile at one block per CU (tools/kbench.py lnfold: L0 q | k | v 0.452 vs 0.399 ms).
  uint2 v[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = rloc + i * 16 + (lane & 15);
#pragma unroll
    for (int p = 0; p < 4; ++p) v[i][p] = *reinterpret_cast<const uint2*>(img + p * 2048 + row * 8);
  }
  float nmean[MT], rstd[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      s1 += __uint_as_float(v[i][p].x);
      s2 += __uint_as_float(v[i][p].y);
    }
    const float mu = s1 * invK;
    const float var = __builtin_fmaf(-mu, mu, s2 * invK);
    nmean[i] = -mu;
    rstd[i] = rsqrtf(fmaxf(var, 0.f) + a.ln_eps);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = cloc + j * 16 + (lane >> 4) * 4;
    const uint4 su = *reinterpret_cast<const uint4*>(img + LN_LDS_COLSUM + col * 4), cu = *reinterpret_cast<const uint4*>(img + LN_LDS_CVEC + col * 4);
    const f32x4 s = {__uint_as_float(su.x), __uint_as_float(su.y), __uint_as_float(su.z), __uint_as_float(su.w)};
    const f32x4 c = {__uint_as_float(cu.x), __uint_as_float(cu.y), __uint_as_float(cu.z), __uint_as_float(cu.w)};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][i][r] = __builtin_fmaf(rstd[i], __builtin_fmaf(nmean[i], s[r], acc[j][i][r]), c[r]);
  }
}

// lane holds D[n = (lane>>4)*4 + r][m = lane & 15] of each 16x16 tile acc[j][i].
// sC != nullptr: instead of 8-byte global stores (32-byte segments), park the finished fp16 values in an LDS
// tile [128][CLD] so that the block can write whole 16-byte x row-contiguous vectors afterwards.
// rowfn(i) = global output row of this lane's column in m tile i, or -1 (tail); m0 only addresses the sC tile.
//
// F fixes the set of optional terms at compile time (bit 1 row vector, 2 residual, 3 second residual; no
// activation) for the combinations the model's large GEMMs use; everything else takes epilogue_generic.
//   * With the wave-uniform tests evaluated inside the 40 (i, j) iterations, each iteration was a chain of scalar
//     branches with a load -> wait pair behind every one: the epilogue of a 256 x 320 tile cost as much as four
//     K slabs (~10 us of a 27 us tile at K = 320).
//   * vmcnt retires in order, so a load issued after a store cannot be waited for without draining that store:
//     every load of the epilogue is issued before its first store.
//   * 160 accumulator registers leave no room for 80 registers of residual, so the tile is first packed to fp16
//     (4 -> 2 registers per (i, j), the accumulators die), then each term is fetched whole and added with packed
//     fp16 adds -- the rounding the reference's own half-precision `conv(x) + temb`, `attn(x) + x` perform.
template <int F, int NT, int MT, int WN, class RowFn>
__device__ __forceinline__ void epilogue_rows(const me_gemm_args& a, f32x4 (&acc)[NT][MT], RowFn rowfn, int m0, int n0, int wn, int lane, f16* sC, int CLD) {
  static_assert(F >= 0 && (F & 1) == 0, "specialised epilogue");
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(a.rowvec);
  const f16* res = reinterpret_cast<const f16*>(a.res);  // may alias C (in-place residual)
  const f16* res2 = reinterpret_cast<const f16*>(a.res2);
  f16* C = reinterpret_cast<f16*>(a.C);
  constexpr bool has_rv = (F & 2) != 0, has_res = (F & 4) != 0, has_res2 = (F & 8) != 0;
  const int nb = n0 + wn * WN + (lane >> 4) * 4;   // first column of n tile j is nb + 16 j
  union P4 { f16x2 h[2]; uint2 u; };
  P4 o[MT][NT];
  int mrow[MT];
  auto rr = [&](int m) { return a.res_rows > 0 ? m % a.res_rows : m; };   // residual shared by several batch entries (wave-uniform test)
  auto rr2 = [&](int m) { return a.res2_rows > 0 ? m % a.res2_rows : m; };
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mrow[i] = rowfn(i);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      o[i][j].h[0] = __builtin_convertvector((f32x2){acc[j][i][0] * a.alpha, acc[j][i][1] * a.alpha}, f16x2);
      o[i][j].h[1] = __builtin_convertvector((f32x2){acc[j][i][2] * a.alpha, acc[j][i][3] * a.alpha}, f16x2);
    }
  }
  // Wide path (sC == nullptr): pairs of n tiles (j, j+1) trade halves between lanes l and l ^ 16 with v_permlane16_swap, after
  // which a lane holds 8 CONSECUTIVE output columns -- even quarter-rows g: tile j, columns 4g .. 4g+7; odd g: tile j+1, columns
  // 4(g-1) .. 4(g-1)+7 -- so every term load and the store move 16 bytes per lane (64 contiguous bytes per output row and
  // instruction instead of 32): the direct epilogue of the 256 x 320 tile is issue-bound on its 40 stores per lane.
  const bool wide = (F == 0 || WIDE_TERMS) && sC == nullptr && a.N % 8 == 0 && a.ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                    (!has_rv || (a.ldrv % 8 == 0 && (reinterpret_cast<uintptr_t>(a.rowvec) & 15) == 0)) &&
                    (!has_res || (a.ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)) &&
                    (!has_res2 || (a.ldr2 % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res2) & 15) == 0));
  constexpr int NP = NT / 2;   // tile pairs; an odd last tile keeps the 8-byte path
  const int gq = lane >> 4;
  if (wide) {
    union P8 { f16x2 h[4]; uint4 u; };
    const int nw = n0 + wn * WN + ((gq & 1) ? 16 + 4 * (gq - 1) : 4 * gq);   // first of this lane's 8 columns inside pair 0; pair jp adds 32 jp
    P8 w[MT][NP > 0 ? NP : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const auto rx = __builtin_amdgcn_permlane16_swap(o[i][2 * jp].u.x, o[i][2 * jp + 1].u.x, false, false);
        const auto ry = __builtin_amdgcn_permlane16_swap(o[i][2 * jp].u.y, o[i][2 * jp + 1].u.y, false, false);
        w[i][jp].u = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      }
    auto add_wide = [&](const f16* src, auto rowoff) {
      P8 t[MT][NP > 0 ? NP : 1];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const f16* p = src + rowoff(mrow[i] < 0 ? 0 : mrow[i]) + nw;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) t[i][jp].u = nw + 32 * jp + 7 < a.N ? *reinterpret_cast<const uint4*>(p + 32 * jp) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
#pragma unroll
          for (int q = 0; q < 4; ++q) w[i][jp].h[q] += t[i][jp].h[q];
    };
    auto add_narrow_last = [&](const f16* src, auto rowoff) {   // the unpaired last tile (NT odd)
      if constexpr (NT % 2 == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          P4 t;
          t.u = nb + 16 * (NT - 1) < a.N ? *reinterpret_cast<const uint2*>(src + rowoff(mrow[i] < 0 ? 0 : mrow[i]) + nb + 16 * (NT - 1)) : make_uint2(0u, 0u);
          o[i][NT - 1].h[0] += t.h[0];
          o[i][NT - 1].h[1] += t.h[1];
        }
      }
    };
    if constexpr (has_rv) { add_wide(rowvec, [&](int m) { return (long)(m / a.rows_per_vec) * a.ldrv; }); add_narrow_last(rowvec, [&](int m) { return (long)(m / a.rows_per_vec) * a.ldrv; }); }
    if constexpr (has_res) { add_wide(res, [&](int m) { return (long)rr(m) * a.ldr; }); add_narrow_last(res, [&](int m) { return (long)rr(m) * a.ldr; }); }
    if constexpr (has_res2) { add_wide(res2, [&](int m) { return (long)rr2(m) * a.ldr2; }); add_narrow_last(res2, [&](int m) { return (long)rr2(m) * a.ldr2; }); }
    if constexpr (F == 0) {
      if (a.C2) {   // head-major second output (ABI 6): a lane's 8 (4) consecutive columns never straddle a head (c2_dh % 8 == 0)
        f16* C2 = reinterpret_cast<f16*>(a.C2);
        long hoff[NP + 1];   // element offset of the lane's columns inside C2 (row 0), or -1: the columns stay in C
#pragma unroll
        for (int jp = 0; jp <= NP; ++jp) {
          const int col = jp < NP ? nw + 32 * jp : nb + 16 * (NT - 1);
          const int n2 = col - a.c2_col0, hh = n2 / a.c2_dh;
          hoff[jp] = n2 >= 0 ? (long)hh * a.c2_hs + (n2 - hh * a.c2_dh) : -1;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (mrow[i] < 0) continue;
          f16* crow = C + (long)mrow[i] * a.ldc;
          f16* hrow = C2 + (long)mrow[i] * a.c2_dh;
#pragma unroll
          for (int jp = 0; jp < NP; ++jp)
            if (nw + 32 * jp + 7 < a.N) st16(hoff[jp] >= 0 ? hrow + hoff[jp] : crow + nw + 32 * jp, w[i][jp].u);
          if constexpr (NT % 2 == 1) {
            if (nb + 16 * (NT - 1) < a.N) st8(hoff[NP] >= 0 ? hrow + hoff[NP] : crow + nb + 16 * (NT - 1), o[i][NT - 1].u);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (mrow[i] < 0) continue;
      f16* crow = C + (long)mrow[i] * a.ldc;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
        if (nw + 32 * jp + 7 < a.N) st16(crow + nw + 32 * jp, w[i][jp].u);
      if constexpr (NT % 2 == 1) {
        if (nb + 16 * (NT - 1) < a.N) st8(crow + nb + 16 * (NT - 1), o[i][NT - 1].u);
      }
    }
    return;
  }
  // o += src[row(i) * ld + column]: all 40 loads, then 80 packed adds
  auto add_term = [&](const f16* src, auto rowoff) {
    P4 t[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const f16* p = src + rowoff(mrow[i] < 0 ? 0 : mrow[i]) + nb;
#pragma unroll
      for (int j = 0; j < NT; ++j) t[i][j].u = nb + 16 * j < a.N ? *reinterpret_cast<const uint2*>(p + 16 * j) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        o[i][j].h[0] += t[i][j].h[0];
        o[i][j].h[1] += t[i][j].h[1];
      }
  };
  if constexpr (has_rv) add_term(rowvec, [&](int m) { return (long)(m / a.rows_per_vec) * a.ldrv; });
  if constexpr (has_res) add_term(res, [&](int m) { return (long)rr(m) * a.ldr; });
  if constexpr (has_res2) add_term(res2, [&](int m) { return (long)rr2(m) * a.ldr2; });
  if constexpr (F == 0) {
    if (a.C2 && sC == nullptr) {   // head-major second output through the 8-byte path (me_gemm never stages such a launch through LDS)
      f16* C2 = reinterpret_cast<f16*>(a.C2);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = nb + 16 * j;
        if (col >= a.N) continue;
        const int n2 = col - a.c2_col0, hh = n2 / a.c2_dh;
        const long hoff = n2 >= 0 ? (long)hh * a.c2_hs + (n2 - hh * a.c2_dh) : -1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (mrow[i] < 0) continue;
          st8(hoff >= 0 ? C2 + (long)mrow[i] * a.c2_dh + hoff : C + (long)mrow[i] * a.ldc + col, o[i][j].u);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (mrow[i] < 0) continue;
    f16* crow = C + (long)mrow[i] * a.ldc + nb;                // + 16 j
    f16* lrow = sC + (mrow[i] - m0) * CLD + (nb - n0);         // never step below the LDS tile: the address is 32-bit
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (nb + 16 * j >= a.N) continue;
      if (sC) *reinterpret_cast<uint2*>(lrow + 16 * j) = o[i][j].u;
      else st8(crow + 16 * j, o[i][j].u);
    }
  }
}

// Row-contiguous epilogue of the 8-phase kernels (round 5; `ME_GEMM_ROWEPI=0` restores epilogue_rows for the A/B).
//
// Measured (tools/ubench_epi.hip, tools/ubench_fill.hip, profiles/r05_epi_fill.txt): the output stores of a 256 x 320 tile cost as much as its whole
// main loop at K = 320 (24.4 us per tile with, 13.4 without), they do not overlap with the next tile's fills whatever the ring depth or the
// counted-wait credit (the CU's memory pipeline takes fills and stores in order), staggering the CUs changes nothing, and even stores into an
// L2-resident window keep 8 of the 11 us: the cost is the store INSTRUCTIONS -- about 53 cycles per 1 KB instruction plus 1.8 per distinct
// output row it touches (tools/ubench_store.hip).  The direct epilogue writes 16 rows x 64 B per instruction (16 rows x 32 B with terms).
// Here a wave parks its fp16 tile, 32 rows at a time, in a PRIVATE LDS scratch (the staging buffers are dead by then; no block barrier) and
// reads it back row-major, so that every 16-byte store / term load of a lane continues its neighbour's: 160 contiguous bytes per output row
// and wave, 6.4 rows per instruction.  The ring-GEMM model went 24.0 -> 19.7 us per tile with it.
//
// Arithmetic is exactly epilogue_rows': the tile is packed to fp16 (acc * alpha, bias already inside), then rowvec / res / res2 are added
// with packed fp16 adds in that order -- the results are bitwise those of the direct epilogue (test_gemm_8phase_kernel compares the two).
// vmcnt retires in order: the term loads of chunk c + 1 are issued BEFORE the stores of chunk c.
//   mw0 / nw0: first output row / column of this wave's MT*16 x WN tile;  scr: 32 * WN * 2 bytes of LDS owned by this wave.
//   ln_out (ABI 9): the partial row sums (sum y, sum y^2) of the FINAL fp16 rows, for the LayerNorm-folded projection that reads this output next: every 16-byte
//   piece is summed with v_dot2_f32_f16 (against ones / itself), the wave's CH pieces of a row meet in its scratch, the four waves of a row in `red`
//   (float2 [2 wave rows][4 wave columns][GR rows], behind a block barrier), and 2 * GR threads store one (sum, sum of squares) pair per row of the
//   tile -- a fixed order throughout.  lnx = {first row of the tile, wave row, wave column, thread id}.
struct LnOutCtx { int m0, wr, wc, tid; char* red; };
template <int F, int NT, int MT, int WN>
__device__ __forceinline__ void epilogue_rowpass(const me_gemm_args& a, f32x4 (&acc)[NT][MT], int mw0, int nw0, int lane, char* scr, const LnOutCtx& lnx) {
  static_assert(F >= 0 && (F & 1) == 0 && MT % 2 == 0 && WN % 16 == 0, "row-pass epilogue");
  constexpr bool has_rv = (F & 2) != 0, has_res = (F & 4) != 0, has_res2 = (F & 8) != 0;
    // a chunk = two 16-row MFMA row blocks = 32 rows x WN columns = 32 * WN / 8 16-byte pieces, IT per lane (5 at WN = 80: no masked piece)
  // scratch row pitch = the 160-byte segment + 16: at 40 dwords the 16 rows of an MFMA-layout ds_write_b64 (bank = dword address mod 32, lanes 0-15 / 16-31 / ...
  // together) fall on 4 bank groups -- a 4-way conflict, +10 LDS cycles on each of the 40 writes of a wave tile (SQ_LDS_BANK_CONFLICT: a fifth of the LDS cycles);
  // at 44 dwords rows m and m + 8 still meet (2-way: +2), which is the best a 16-byte-aligned pitch can do, and the ds_read_b128 readback stays aligned
  constexpr int NCHK = MT / 2, ROWS = 32, CH = WN / 8, PIECES = ROWS * CH, IT = PIECES / 64, PITCH = WN * 2 + 16;
  static_assert(PIECES % 64 == 0, "whole instructions per chunk");
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(a.rowvec);
  const f16* res = reinterpret_cast<const f16*>(a.res);    // may alias C (in-place residual): a piece is read and written by the same lane
  const f16* res2 = reinterpret_cast<const f16*>(a.res2);
  f16* C = reinterpret_cast<f16*>(a.C);
  union P4 { f16x2 h[2]; uint2 u; };
  union P8 { f16x2 h[4]; uint4 u; };
  P4 o[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      o[i][j].h[0] = __builtin_convertvector((f32x2){acc[j][i][0] * a.alpha, acc[j][i][1] * a.alpha}, f16x2);
      o[i][j].h[1] = __builtin_convertvector((f32x2){acc[j][i][2] * a.alpha, acc[j][i][3] * a.alpha}, f16x2);
    }
  // this lane's pieces: piece t of a chunk = 16-byte piece p = 64 t + lane = (row p / CH, columns 8 (p % CH) ...) -- the same in every chunk
  int prow[IT], pcol[IT], poff[IT];
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    const int p = t * 64 + lane;
    prow[t] = p / CH;
    pcol[t] = nw0 + (p - prow[t] * CH) * 8;
    poff[t] = prow[t] * PITCH + (p - prow[t] * CH) * 16;      // the piece inside the scratch
  }
  // row maps of the terms without a division per piece: a wave tile spans at most two row vectors / one wrap of a shared residual whenever those
  // periods are at least the tile height (every launch of the model); the general forms stay behind a wave-uniform test
  // (a ragged last tile: the rows of a wave -- the whole second wave group, even -- may lie past M.  Term loads of such rows read row M - 1 instead, a row
  //  that exists; nothing is stored for them.  mwc = the wave's first row, clamped the same way: the base of the row maps.)
  constexpr int TH = MT * 16;
  const int mwc = min(mw0, a.M - 1);
  int v0 = 0, vb = 0x7fffffff, r0 = 0, q0 = 0;
  if constexpr (has_rv) { v0 = mwc / a.rows_per_vec; vb = (v0 + 1) * a.rows_per_vec; }
  if constexpr (has_res) r0 = a.res_rows > 0 ? mwc % a.res_rows : 0;
  if constexpr (has_res2) q0 = a.res2_rows > 0 ? mwc % a.res2_rows : 0;
  auto vrow = [&](int m) { return a.rows_per_vec >= TH ? v0 + (m >= vb ? 1 : 0) : m / a.rows_per_vec; };   // m in [mwc, M)
  auto wrap = [&](int m, int period, int first) {
    if (period <= 0) return m;
    if (period < TH) return m % period;
    const int x = first + (m - mwc);
    return x >= period ? x - period : x;
  };
  // The row vector (the time embedding of temp_conv1 / conv1: one row per batch entry) depends on the column alone while the wave tile lies inside one
  // vector row -- every tile but the few that straddle two batch entries: its IT pieces are then fetched ONCE per tile, not once per chunk (wave-uniform
  // test; the straddling tiles fetch per chunk like a residual).  The residuals are the pipelined terms: NRES of them per chunk.
  constexpr int NRES = (has_res ? 1 : 0) + (has_res2 ? 1 : 0), NRV = NRES > 0 ? NRES : 1;
  bool rv_once = false;
  P8 rvp[has_rv ? IT : 1];
  if constexpr (has_rv) {
    const int mlast = max(min(mw0 + TH, a.M) - 1, mwc);
    rv_once = vrow(mwc) == vrow(mlast);
    if (rv_once) {
#pragma unroll
      for (int t = 0; t < IT; ++t) rvp[t].u = *reinterpret_cast<const uint4*>(rowvec + (long)vrow(mwc) * a.ldrv + pcol[t]);
    }
  }
  auto load_rv = [&](int c) {     // a tile that straddles two vector rows: per chunk
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int m = min(mw0 + c * ROWS + prow[t], a.M - 1);
      rvp[t].u = *reinterpret_cast<const uint4*>(rowvec + (long)vrow(m) * a.ldrv + pcol[t]);
    }
  };
  auto load_terms = [&](int c, P8 (&tv)[NRV][IT]) {
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int m = min(mw0 + c * ROWS + prow[t], a.M - 1);     // a row that exists: the value of a row past M is never stored
      int k = 0;
      if constexpr (has_res) tv[k++][t].u = *reinterpret_cast<const uint4*>(res + (long)wrap(m, a.res_rows, r0) * a.ldr + pcol[t]);
      if constexpr (has_res2) tv[k++][t].u = *reinterpret_cast<const uint4*>(res2 + (long)wrap(m, a.res2_rows, q0) * a.ldr2 + pcol[t]);
    }
  };
  const int wrow = lane & 15, wq = lane >> 4;
  auto park = [&](int c, P8 (&d)[IT]) {     // MFMA layout -> this wave's LDS scratch -> row-major 16-byte pieces
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NT; ++j) *reinterpret_cast<uint2*>(scr + (ii * 16 + wrow) * PITCH + (j * 16 + wq * 4) * 2) = o[c * 2 + ii][j].u;
#pragma unroll
    for (int t = 0; t < IT; ++t) d[t].u = *reinterpret_cast<const uint4*>(scr + poff[t]);
  };
  auto add_rv = [&](P8 (&d)[IT]) {          // order of the packed fp16 adds as in epilogue_rows: rowvec, res, res2
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[t].h[q] += rvp[t].h[q];
  };
  auto add_terms = [&](P8 (&d)[IT], P8 (&tv)[NRV][IT]) {
#pragma unroll
    for (int k = 0; k < NRES; ++k)
#pragma unroll
      for (int t = 0; t < IT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[t].h[q] += tv[k][t].h[q];
  };
  // head-major second output (F == 0 only): element offset of a piece's 8 columns inside C2's row block, or -1 (the columns stay in C)
  long hoff[IT];
  const bool c2 = F == 0 && a.C2 != nullptr;
  if (c2) {
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int n2 = pcol[t] - a.c2_col0, hh = n2 / a.c2_dh;
      hoff[t] = n2 >= 0 ? (long)hh * a.c2_hs + (n2 - hh * a.c2_dh) : -1;
    }
  }
  auto store = [&](int c, P8 (&d)[IT]) {
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int m = mw0 + c * ROWS + prow[t];
      if (m >= a.M) continue;
      f16* dst = C + (long)m * a.ldc + pcol[t];
      if (c2 && hoff[t] >= 0) dst = reinterpret_cast<f16*>(a.C2) + (long)m * a.c2_dh + hoff[t];
      st16(dst, d[t].u);
    }
  };
  const bool lnout = a.ln_out != nullptr;    // (wave-uniform)
  static_assert(CH % 2 == 0, "row sums: half the pieces of a row per half wave");
  f32x2 rsum[NCHK];                           // lanes < 32: (sum, sum of squares) of row 32 c + lane over this wave's WN columns
  auto row_sums = [&](int c, P8 (&d)[IT]) {
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1 = __builtin_amdgcn_fdot2(d[t].h[q], one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(d[t].h[q], d[t].h[q], s2, false);
      }
      const int p = t * 64 + lane;            // piece p = (row p / CH, piece p % CH): the partials of a row are CH consecutive float2
      *reinterpret_cast<uint2*>(scr + p * 8) = make_uint2(__float_as_uint(s1), __float_as_uint(s2));   // (uint2 like park's stores: one access type per scratch)
    }
    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CH / 2; ++k) {
      const uint2 v = *reinterpret_cast<const uint2*>(scr + ((lane & 31) * CH + (lane >> 5) * (CH / 2) + k) * 8);
      acc2 += f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
    }
    rsum[c] = f32x2{xor32_sum(acc2[0]), xor32_sum(acc2[1])};
  };
  // software pipeline over the chunks: the residual pieces of chunk c + 1 are requested before chunk c is stored
  P8 tv[2][NRV][IT];
  if constexpr (NRES > 0) load_terms(0, tv[0]);
#pragma unroll
  for (int c = 0; c < NCHK; ++c) {
    P8 d[IT];
    if constexpr (has_rv) {
      if (!rv_once) load_rv(c);       // (wave-uniform, rare)
    }
    park(c, d);
    if constexpr (has_rv) add_rv(d);
    if constexpr (NRES > 0) {
      add_terms(d, tv[c & 1]);
      if (c + 1 < NCHK) load_terms(c + 1, tv[(c + 1) & 1]);
    }
    store(c, d);
    if (lnout) row_sums(c, d);
  }
  if (lnout) {
    constexpr int GR = MT * 16;
    uint2* red = reinterpret_cast<uint2*>(lnx.red);
    if (lane < 32) {
#pragma unroll
      for (int c = 0; c < NCHK; ++c) red[(lnx.wr * 4 + lnx.wc) * GR + c * 32 + lane] = make_uint2(__float_as_uint(rsum[c][0]), __float_as_uint(rsum[c][1]));
    }
    __syncthreads();
    if (lnx.tid < 2 * GR) {
      const int g = lnx.tid / GR, r = lnx.tid - g * GR, m = lnx.m0 + lnx.tid;
      f32x2 sum = {0.f, 0.f};
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const uint2 v = red[(g * 4 + w4) * GR + r];
        sum += f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
      }
      if (m < a.M) *reinterpret_cast<f32x2*>(reinterpret_cast<float*>(a.ln_out) + (long)((nw0 - lnx.wc * WN) / 320) * a.ln_out_stride + 2 * (long)m) = sum;
    }
  }
}

// GEGLU form of the row-contiguous epilogue (round 5): the 256 x 256 tile of the 8-phase GEGLU kernel yields 256 rows x 128 output columns, and a wave's
// share is 128 rows x 32 columns -- 64-byte row segments however a wave cuts them (16 rows per 1 KB store instruction in the direct epilogue).  Here all
// eight waves park a * gelu(g) in ONE LDS tile [256 rows][16 chunks of 16 B] behind a block barrier (the staging buffers are dead), and every wave then stores
// 32 whole 256-byte rows: 4 rows per instruction.  The chunk position inside a row is XOR-swizzled with the row index (c ^ (row & 15)): the MFMA-layout
// writes (16 rows x one column chunk per ds_write_b64) and the row-major reads both spread over all banks; a lane stores the chunk its position holds, so
// the 16 lanes of a row still cover the row's 256 bytes.  Same values as epilogue_geglu (same products, same rounding): bitwise.
template <int NT, int MT, int WN>
__device__ __forceinline__ void epilogue_geglu_rowpass(const me_gemm_args& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wr, int wc, int lane, char* tile) {
  static_assert(NT == 4 && WN == 64, "256-wide GEGLU tile: 4 waves x 32 output columns");
  constexpr int NO = NT / 2, GR = MT * 16;   // output column tiles per wave; rows per wave group
  f16* C = reinterpret_cast<f16*>(a.C);
  const int wrow = lane & 15, wq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int jj = 0; jj < NO; ++jj) {
      float v[4];
      {   // alpha == 1 (me_gemm checks)
        const f32x2 v01 = geglu2(f32x2{acc[2 * jj][i][0], acc[2 * jj][i][1]}, f32x2{acc[2 * jj + 1][i][0], acc[2 * jj + 1][i][1]});
        const f32x2 v23 = geglu2(f32x2{acc[2 * jj][i][2], acc[2 * jj][i][3]}, f32x2{acc[2 * jj + 1][i][2], acc[2 * jj + 1][i][3]});
        v[0] = v01[0]; v[1] = v01[1]; v[2] = v23[0]; v[3] = v23[1];
      }
      union { f16x2 h[2]; uint2 u; } o;
      o.h[0] = __builtin_convertvector((f32x2){v[0], v[1]}, f16x2);
      o.h[1] = __builtin_convertvector((f32x2){v[2], v[3]}, f16x2);
      const int row = wr * GR + i * 16 + wrow;                       // row of the 2 * GR-row tile
      const int col = wc * 32 + jj * 16 + wq * 4;                    // output column inside the tile's 128
      const int pos = (col >> 3) ^ (row & 15);                       // swizzled chunk position
      *reinterpret_cast<uint2*>(tile + row * 256 + pos * 16 + (col & 4) * 2) = o.u;
    }
  __syncthreads();
  // wave w stores rows [32 w, 32 w + 32) of the tile: piece p = 64 t + lane -> row 32 w + p / 16, position p % 16
  const int wave = wr * 4 + wc;
  constexpr int RW = 2 * GR / 8;              // rows per wave (32, or 24 at 192-row tiles -- the GEGLU kernel only has the 256-row form)
  const int Nout0 = n0 / 2;
#pragma unroll
  for (int t = 0; t < RW / 4; ++t) {
    const int p = t * 64 + lane, row = wave * RW + (p >> 4), pos = p & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(tile + row * 256 + pos * 16);
    const int m = m0 + row, n = Nout0 + ((pos ^ (row & 15)) << 3);
    if (m < a.M) st16<true>(C + (long)m * a.ldc + n, v);
  }
}


// any combination of terms, tested element by element (activations, rare combinations)
template <int NT, int MT, int WN, class RowFn>
__device__ __forceinline__ void epilogue_generic(const me_gemm_args& a, f32x4 (&acc)[NT][MT], RowFn rowfn, int m0, int n0, int wn, int lane, f16* sC, int CLD) {
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(a.rowvec);
  const f16* res = reinterpret_cast<const f16*>(a.res);
  const f16* res2 = reinterpret_cast<const f16*>(a.res2);
  f16* C = reinterpret_cast<f16*>(a.C);
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = rowfn(i);
    if (m < 0) continue;
    const f16* rv = rowvec ? rowvec + (long)(m / a.rows_per_vec) * a.ldrv : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WN + j * 16 + nq;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] * a.alpha;
      if (rv) {
        U64 b;
        b.u = *reinterpret_cast<const uint2*>(rv + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b.e[r];
      }
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (a.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      const int mr = a.res_rows > 0 ? m % a.res_rows : m;
      if (res) {
        U64 b;
        b.u = *reinterpret_cast<const uint2*>(res + (long)mr * a.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b.e[r];
      }
      if (res2) {
        U64 b;
        b.u = *reinterpret_cast<const uint2*>(res2 + (long)(a.res2_rows > 0 ? m % a.res2_rows : m) * a.ldr2 + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b.e[r];
      }
      U64 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o.e[r] = (f16)v[r];
      if (sC) *reinterpret_cast<uint2*>(sC + (m - m0) * CLD + (n - n0)) = o.u;
      else *reinterpret_cast<uint2*>(C + (long)m * a.ldc + n) = o.u;
    }
  }
}

// GEGLU: tile 2jj = value rows, tile 2jj+1 = gate rows of the same 16 output columns (weights.py packs them so)
template <int NT, int MT, int WN, class RowFn>
__device__ __forceinline__ void epilogue_geglu(const me_gemm_args& a, f32x4 (&acc)[NT][MT], RowFn rowfn, int m0, int n0, int wn, int lane, f16* sC, int CLD) {
  f16* C = reinterpret_cast<f16*>(a.C);
  constexpr int NO = NT / 2;                     // 16-column output tiles per wave
  const int nb = n0 + wn * WN + (lane >> 4) * 4;
  const int nob = (n0 + wn * WN) / 2 + (lane >> 4) * 4;
  const int Nout = a.N / 2;
  union P4 { f16x2 h[2]; uint2 u; };
  P4 o[MT][NO];
  int mrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mrow[i] = rowfn(i);
#pragma unroll
    for (int jj = 0; jj < NO; ++jj) {
      float v[4];
      {   // alpha == 1 (me_gemm checks)
        const f32x2 v01 = geglu2(f32x2{acc[2 * jj][i][0], acc[2 * jj][i][1]}, f32x2{acc[2 * jj + 1][i][0], acc[2 * jj + 1][i][1]});
        const f32x2 v23 = geglu2(f32x2{acc[2 * jj][i][2], acc[2 * jj][i][3]}, f32x2{acc[2 * jj + 1][i][2], acc[2 * jj + 1][i][3]});
        v[0] = v01[0]; v[1] = v01[1]; v[2] = v23[0]; v[3] = v23[1];
      }
      o[i][jj].h[0] = __builtin_convertvector((f32x2){v[0], v[1]}, f16x2);
      o[i][jj].h[1] = __builtin_convertvector((f32x2){v[2], v[3]}, f16x2);
    }
  }
  // wide path as in epilogue_rows: pairs of output tiles trade halves across lanes l ^ 16 -> 16-byte stores
  const bool wide = sC == nullptr && Nout % 8 == 0 && a.ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
  constexpr int NP = NO / 2;
  const int gq = lane >> 4;
  const int nw = (n0 + wn * WN) / 2 + ((gq & 1) ? 16 + 4 * (gq - 1) : 4 * gq);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mrow[i];
    if (wide) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {   // the swaps run in every lane (uniform control flow), only the stores are predicated
        const auto rx = __builtin_amdgcn_permlane16_swap(o[i][2 * jp].u.x, o[i][2 * jp + 1].u.x, false, false);
        const auto ry = __builtin_amdgcn_permlane16_swap(o[i][2 * jp].u.y, o[i][2 * jp + 1].u.y, false, false);
        if (m >= 0 && nw + 32 * jp + 7 < Nout) st16<true>(C + (long)m * a.ldc + nw + 32 * jp, make_uint4(rx[0], ry[0], rx[1], ry[1]));
      }
      if constexpr (NO % 2 == 1) {
        if (m >= 0 && nob + 16 * (NO - 1) < Nout) st8<true>(C + (long)m * a.ldc + nob + 16 * (NO - 1), o[i][NO - 1].u);
      }
      continue;
    }
    if (m < 0) continue;
#pragma unroll
    for (int jj = 0; jj < NO; ++jj) {
      if (nb + 32 * jj >= a.N) continue;
      const int no = nob + 16 * jj;  // output column
      if (sC) *reinterpret_cast<uint2*>(sC + (m - m0) * CLD + (no - n0 / 2)) = o[i][jj].u;
      else st8<true>(C + (long)m * a.ldc + no, o[i][jj].u);
    }
  }
}

template <int NT, int MT, int WN, class RowFn>
__device__ __forceinline__ void epilogue(const me_gemm_args& a, f32x4 (&acc)[NT][MT], RowFn rowfn, int m0, int n0, int wn, int lane, f16* sC, int CLD) {
  if (a.geglu) return epilogue_geglu<NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);
  const int f = a.act != 0 ? -1 : ((a.rowvec ? 2 : 0) | (a.res ? 4 : 0) | (a.res2 ? 8 : 0));
  switch (f) {   // wave-uniform
    case 0: return epilogue_rows<0, NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);   // projections, (biased) linears / convs
    case 2: return epilogue_rows<2, NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);   // resnet conv1 + time embedding
    case 4: return epilogue_rows<4, NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);   // out projection / ff2 / conv2 / tconv + residual
    case 6: return epilogue_rows<6, NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);    // temp_conv1: + time embedding + residual
    case 12: return epilogue_rows<12, NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);  // temp_conv2: + residual + shortcut
    default: return epilogue_generic<NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);
  }
}

// BM = 128 (4 waves, 2 blocks/CU) or 256 (8 waves, 1 block/CU).  The 256 x 320 tile halves the L2 -> LDS fill per
// FLOP (142 vs 71 flop/byte of staged operands): the 128-row tiles measured fill-bound at ~8 TB/s for K <= 640.
template <int BM, int BN, int STAGE, int WM = 64>
__global__ __launch_bounds__(BM / WM * 128, WM == 128 ? 1 : 2) void gemm_kernel(const me_gemm_args a) {
  constexpr int NTHR = BM / WM * 128;  // WM rows x 2 wave columns per 64 threads
  constexpr int RSTR = NTHR / 8;  // row stride between a thread's staged rows
  constexpr int WN = BN / 2;      // per-wave N extent
  constexpr int NT = WN / 16;     // 16-wide n tiles per wave
  constexpr int MT = WM / 16;     // 16-wide m tiles per wave
  constexpr int XROWS = BM / RSTR;  // X rows staged per thread
  constexpr int WROWS = BN / RSTR;  // W rows staged per thread
  constexpr int LD = STAGE != STAGE_REG ? BK : BK + 8;  // LDS row stride in halves

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sX = reinterpret_cast<f16*>(smem);  // [2][BM][LD]
  f16* sW = sX + 2 * BM * LD;              // [2][BN][LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nbn = (a.N + BN - 1) / BN;
  const int nbm = (a.M - a.m_off + BM - 1) / BM;   // tiles cover the rows [m_off, M)
  const int w = xcd_remap(blockIdx.x, nbm * nbn);
  const int tile_n = w % nbn, tile_m = w / nbn;
  const int m0 = a.m_off + tile_m * BM, n0 = tile_n * BN;

  const f16* __restrict__ X = reinterpret_cast<const f16*>(a.X);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(a.W);

  const int taps = a.gather == ME_GATHER_CONV3 ? 9 : (a.gather == ME_GATHER_TCONV ? 3 : 1);
  const int nkc = (a.K + BK - 1) / BK;
  // 3x3 convs with fewer than 64 input channels (the ControlNet conditioning embedding: 16, 32) pack (tap, channel)
  // into ONE K axis of 9 K -- the weights [N][9][K] are contiguous in exactly that order -- instead of nine 64-wide
  // slabs that are 75 % / 50 % zero padding: 3 / 5 slabs instead of 9.  A lane's 16-byte chunk then belongs to
  // the tap (slab * 64 + chunk offset) / K, different lanes gather different taps of their rows.
  const bool packk = STAGE == STAGE_GLDS && a.gather == ME_GATHER_CONV3 && a.K < BK && BK % a.K == 0;   // (STAGE_BUF is only launched with K % 64 == 0)
  const int nit = packk ? (9 * a.K + BK - 1) / BK : taps * nkc;

  // staging assignment
  //   REG : thread -> rows tid/8 + 32*i, 16-byte chunk tid%8
  //   GLDS: wave instruction q = wave + 4*i covers rows 8q..8q+7; lane -> row 8q + lane/8, LDS chunk slot lane%8,
  //         which holds source chunk (lane%8) ^ swz, swz = (row >> 1) & 7 = (4*(wave&1) + (lane>>4)) & 7 for every i
  int srow, scol;
  if (STAGE != STAGE_REG) {
    srow = wave * 8 + (lane >> 3);
    scol = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  } else {
    srow = tid >> 3;
    scol = (tid & 7) * 8;
  }

  RowInfo rinfo[XROWS];
#pragma unroll
  for (int i = 0; i < XROWS; ++i) rinfo[i] = make_row(a, m0 + srow + RSTR * i);

  long xoff[XROWS];  // element offset of the source row for the current tap, or -1
  int cur_tap = -1;
  auto set_tap = [&](int tap) {
    if (tap != cur_tap) {
      cur_tap = tap;
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const int s = src_row(a, rinfo[i], tap);
        xoff[i] = s < 0 ? -1L : (long)s * a.ldx;
      }
    }
  };

  // split-K (me_gemm sets splits_ for small grids, STAGE_BUF only): blockIdx.y owns K tiles [it0, it1) and stores raw fp32 partial sums
  const int S = STAGE == STAGE_BUF ? a.splits_ : 0;
  const int it0 = S > 1 ? (int)((long)nit * blockIdx.y / S) : 0;
  const int it1 = S > 1 ? (int)((long)nit * (blockIdx.y + 1) / S) : nit;
  f32x4 acc[NT][MT];
  init_acc<NT, MT, WN>(a, acc, n0, wn, lane, S <= 1);

  const int frow = lane & 15;
  const int fg = lane >> 4;

  auto compute = [&](int buf) {
    const f16* bx = sX + buf * BM * LD;
    const f16* bw = sW + buf * BN * LD;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      f16x8 fx[MT], fw[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * WM + i * 16 + frow;
        const int ch = STAGE != STAGE_REG ? ((ks * 4 + fg) ^ ((row >> 1) & 7)) : (ks * 4 + fg);
        fx[i] = *reinterpret_cast<const f16x8*>(bx + row * LD + ch * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + frow;
        const int ch = STAGE != STAGE_REG ? ((ks * 4 + fg) ^ ((row >> 1) & 7)) : (ks * 4 + fg);
        fw[j] = *reinterpret_cast<const f16x8*>(bw + row * LD + ch * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(fw[j], fx[i], acc[j][i]);
    }
  };

  if constexpr (STAGE == STAGE_BUF) {
    // Staging through buffer_load_dwordx4 ... lds (K % 64 == 0): the operands are addressed as "buffer resource + per-lane
    // 32-bit byte offset + scalar slab offset".  The lane offsets change only with the tap, the slab advance is ONE scalar add,
    // and padding taps / tail rows are offsets past the resource's range, which the hardware turns into zeros in LDS -- the
    // global_load_lds path above spends ~250 VALU instructions per 64-wide slab (64-bit per-lane pointers, selects against a zero
    // buffer, an integer division) next to its 80 MFMAs, and on this chip VALU time adds to MFMA time.
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr unsigned OOB = 0x80000000u;
    // X window: rows are addressed relative to the block's lowest source row, so the 32-bit offsets stay below 2 GB on any tensor
    long brow = m0;
    if (a.gather == ME_GATHER_CONV3) brow = (long)(m0 / (a.Hout * a.Wout)) * a.Hin * a.Win;
    else if (a.gather == ME_GATHER_TCONV) brow = m0 > a.npix ? m0 - a.npix : 0;
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(X + brow * a.ldx), 0, OOB, 0x00020000);
    const int wrows = min(a.N - n0, BN);
    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W + (long)n0 * taps * a.K), 0, (unsigned)((long)wrows * taps * a.K * 2), 0x00020000);
    unsigned xo[XROWS], wo[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) wo[i] = (unsigned)(((long)(srow + RSTR * i) * taps * a.K + scol) * 2);
    auto set_tap_off = [&](int tap) {
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const int sr = src_row(a, rinfo[i], tap);
        xo[i] = sr < 0 ? OOB : (unsigned)(((long)(sr - brow) * a.ldx + scol) * 2);
      }
    };
    int tap_l = it0 / nkc, kc_l = it0 - (it0 / nkc) * nkc;   // load cursor
    set_tap_off(tap_l);
    auto gload = [&](int buf) {
      char* dx = reinterpret_cast<char*>(sX + buf * BM * LD) + wave * 1024;
      char* dw = reinterpret_cast<char*>(sW + buf * BN * LD) + wave * 1024;
      const int sx = kc_l * (BK * 2), sw = (tap_l * a.K + kc_l * BK) * 2;
#pragma unroll
      for (int i = 0; i < XROWS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(dx + i * (RSTR * 128)), 16, (int)xo[i], sx, 0, 0);
#pragma unroll
      for (int i = 0; i < WROWS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(dw + i * (RSTR * 128)), 16, (int)wo[i], sw, 0, 0);
      if (++kc_l == nkc) {
        kc_l = 0;
        if (++tap_l < taps) set_tap_off(tap_l);
      }
    };
    gload(0);
    __syncthreads();
    for (int it = it0; it < it1; ++it) {
      const int buf = (it - it0) & 1;
      if (it + 1 < it1) gload(buf ^ 1);
      compute(buf);
      __syncthreads();  // LDS-DMA pending -> the compiler drains vmcnt(0) here: next slab landed, this slab free
    }
    if (S > 1) {   // raw partial sums: work[split][m][n], the lane's four consecutive columns as one 16-byte store
      float* wk = reinterpret_cast<float*>(a.work) + (long)blockIdx.y * a.M * a.N;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * WM + i * 16 + (lane & 15);
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
          if (n < a.N) *reinterpret_cast<f32x4*>(wk + (long)m * a.N + n) = acc[j][i];
        }
      }
      return;
    }
  } else if constexpr (STAGE == STAGE_GLDS) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const f16* zsrc = reinterpret_cast<const f16*>(&g_zero16);
    auto gload = [&](int it, int buf) {
      int tap = it / nkc;
      int c = (it - tap * nkc) * BK + scol;
      bool kok = c < a.K;
      long wk = (long)tap * a.K + c;       // offset inside a weight row
      if (packk) {
        wk = it * BK + scol;
        tap = (int)wk / a.K;
        c = (int)wk - tap * a.K;
        kok = tap < 9;
        if (!kok) tap = 8;
      }
      set_tap(tap);
      char* dx = reinterpret_cast<char*>(sX + buf * BM * LD) + wave * 1024;
      char* dw = reinterpret_cast<char*>(sW + buf * BN * LD) + wave * 1024;
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const f16* src = (kok && xoff[i] >= 0) ? X + xoff[i] + c : zsrc;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dx + i * (RSTR * 128)), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WROWS; ++i) {
        const int n = n0 + srow + RSTR * i;
        const f16* src = (kok && n < a.N) ? W + (long)n * taps * a.K + wk : zsrc;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dw + i * (RSTR * 128)), 16, 0, 0);
      }
    };
    gload(0, 0);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const int buf = it & 1;
      if (it + 1 < nit) gload(it + 1, buf ^ 1);
      compute(buf);
      __syncthreads();  // LDS-DMA pending -> the compiler drains vmcnt(0) here: next slab landed, this slab free
    }
  } else {
    uint4 rx[XROWS], rw[WROWS];
    auto gload = [&](int it) {
      const int tap = it / nkc;
      const int c = (it - tap * nkc) * BK + scol;
      set_tap(tap);
      const bool kok = c < a.K;
#pragma unroll
      for (int i = 0; i < XROWS; ++i) rx[i] = (kok && xoff[i] >= 0) ? ldg128(X + xoff[i] + c) : zero128();
#pragma unroll
      for (int i = 0; i < WROWS; ++i) {
        const int n = n0 + srow + RSTR * i;
        rw[i] = (kok && n < a.N) ? ldg128(W + ((long)n * taps + tap) * a.K + c) : zero128();
      }
    };
    auto sstore = [&](int buf) {
      f16* dx = sX + buf * BM * LD;
      f16* dw = sW + buf * BN * LD;
#pragma unroll
      for (int i = 0; i < XROWS; ++i) *reinterpret_cast<uint4*>(dx + (srow + RSTR * i) * LD + scol) = rx[i];
#pragma unroll
      for (int i = 0; i < WROWS; ++i) *reinterpret_cast<uint4*>(dw + (srow + RSTR * i) * LD + scol) = rw[i];
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const int buf = it & 1;
      if (it + 1 < nit) gload(it + 1);
      compute(buf);
      if (it + 1 < nit) sstore(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  if (a.ln_stats) ln_fold_acc<NT, MT>(a, acc, m0 + wm * WM, n0 + wn * WN, lane);   // (wave-uniform) LayerNorm folded into this projection
  const int ncols = a.geglu ? BN / 2 : BN;            // output columns of this block
  const int Nout = a.geglu ? a.N / 2 : a.N;
  constexpr bool CFITS = (size_t)BM * (BN + 8) <= (size_t)2 * (BM + BN) * LD;   // the C tile fits the staging buffers
  const bool wide_store = CFITS && (Nout % 8 == 0) && (a.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0) && !a.C2;
  auto rowfn = [&](int i) {
    const int m = m0 + wm * WM + i * 16 + (lane & 15);
    return m < a.M ? m : -1;
  };
  if (!wide_store) {
    epilogue<NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, nullptr, 0);
    return;
  }
  // the K loop ended on a barrier: the staging buffers are free and become the C tile
  constexpr int CLD = BN + 8;
  f16* sC = reinterpret_cast<f16*>(smem);
  epilogue<NT, MT, WN>(a, acc, rowfn, m0, n0, wn, lane, sC, CLD);
  __syncthreads();
  f16* C = reinterpret_cast<f16*>(a.C);
  const int vpr = ncols / 8;                           // 16-byte vectors per row
  const int nb0 = a.geglu ? n0 / 2 : n0;
  for (int idx = tid; idx < BM * vpr; idx += NTHR) {
    const int row = idx / vpr, c8 = (idx - row * vpr) * 8;
    const int m = m0 + row, n = nb0 + c8;
    if (m < a.M && n < Nout) *reinterpret_cast<uint4*>(C + (long)m * a.ldc + n) = *reinterpret_cast<const uint4*>(sC + row * CLD + c8);
  }
}

// ---- 3x3 / stride-1 convolution with an LDS halo tile ----
// The gather kernel above re-stages every activation row once per tap (9 x 32 KB of X per 64-channel slab beside
// 9 x 40 KB of W) and is bound by the L2 -> LDS fill (~14 B/clk/CU), not by MFMA.  Here one block owns a 16 x 16
// output patch of one image x 320 output channels: the 18 x 18 input patch (halo included) of a 64-channel slab
// is DMA'd ONCE (41 KB) and all nine taps read their operand-B fragments from it at shifted pixel offsets, so
// only the weight slab changes per tap -> 45 KB instead of 72 KB staged per 256 x 320 x 64 MACs.
// Each MFMA m tile is one patch row (16 consecutive pixels = 16 consecutive halo rows at a fixed tap), so the
// usual (row >> 1) & 7 chunk swizzle keeps the fragment reads conflict-free.
constexpr int HALO_W = 18, HALO_ROWS = HALO_W * HALO_W, HALO_INSTR = (HALO_ROWS + 7) / 8;
constexpr int HALO_BYTES = HALO_INSTR * 1024, CONVW_BYTES = 320 * 128;

__global__ __launch_bounds__(512, 2) void conv3_halo_kernel(const me_gemm_args a) {
  constexpr int BN = 320, WN = 160, NT = 10, MT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;                 // [HALO_INSTR * 8 halo pixels][64 ch], single buffer
  char* sWt = smem + HALO_BYTES;   // [2][320][64 ch]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nbn = a.N / BN;
  const int tx_n = a.Win >> 4, ppi = tx_n * (a.Hin >> 4);
  const int hw = a.Hin * a.Win;
  const int w = xcd_remap(blockIdx.x, (a.M / hw) * ppi * nbn);
  const int tile_n = w % nbn, patch = w / nbn;
  const int img = patch / ppi, pr = patch - img * ppi;
  const int ty = pr / tx_n, tx = pr - ty * tx_n;
  const int y0 = ty * 16, x0 = tx * 16, n0 = tile_n * BN;

  const f16* __restrict__ X = reinterpret_cast<const f16*>(a.X);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(a.W);

  // DMA lane mapping as in gemm_kernel: instruction q = wave + 8*i covers LDS rows 8q..8q+7
  const int srow = wave * 8 + (lane >> 3);
  const int scol = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  // buffer-resource addressing as in gemm_kernel<STAGE_BUF>: per-lane byte offsets relative to the image, scalar slab / tap
  // offsets, zero padding = offsets past the resource range
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr unsigned OOB = 0x80000000u;
  const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(X + (long)img * hw * a.ldx), 0, OOB, 0x00020000);
  const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W + (long)n0 * 9 * a.K), 0, (unsigned)((long)BN * 9 * a.K * 2), 0x00020000);
  unsigned hoff[6];  // byte offset of halo pixel srow + 64*i inside the image, OOB = zero padding
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int hr = srow + 64 * i;
    unsigned o = OOB;
    if (hr < HALO_ROWS) {
      const int hy = hr / HALO_W, hx = hr - hy * HALO_W;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) o = (unsigned)(((long)(iy * a.Win + ix) * a.ldx + scol) * 2);
    }
    hoff[i] = o;
  }
  auto load_halo = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (wave + 8 * i < HALO_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(sH + (wave + 8 * i) * 1024), 16, (int)hoff[i], kc * (BK * 2), 0, 0);
    }
  };
  unsigned woff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) woff[i] = (unsigned)(((long)(srow + 64 * i) * 9 * a.K + scol) * 2);
  auto load_w = [&](int tap, int kc, int buf) {
    char* dw = sWt + buf * CONVW_BYTES + wave * 1024;
    const int so = (tap * a.K + kc * BK) * 2;
#pragma unroll
    for (int i = 0; i < 5; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(dw + i * 8192), 16, (int)woff[i], so, 0, 0);
  };

  f32x4 acc[NT][MT];
  init_acc<NT, MT, WN>(a, acc, n0, wn, lane);

  const int frow = lane & 15, fg = lane >> 4;
  auto compute = [&](int tap, int buf) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* bw = sWt + buf * CONVW_BYTES;
    const int hbase = (wm * 4 + ky) * HALO_W + kx + frow;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      f16x8 fx[MT], fw[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = hbase + i * HALO_W;
        fx[i] = *reinterpret_cast<const f16x8*>(sH + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + frow;
        fw[j] = *reinterpret_cast<const f16x8*>(bw + row * 128 + (((ks * 4 + fg) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(fw[j], fx[i], acc[j][i]);
    }
  };

  const int nkc = a.K / BK;
  load_halo(0);
  load_w(0, 0, 0);
  __syncthreads();
  int it = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    for (int tap = 0; tap < 9; ++tap, ++it) {
      const int buf = it & 1;
      const bool last_tap = tap == 8;
      if (!last_tap) load_w(tap + 1, kc, buf ^ 1);
      else if (kc + 1 < nkc) load_w(0, kc + 1, buf ^ 1);
      compute(tap, buf);
      if (last_tap && kc + 1 < nkc) {
        __syncthreads();       // every wave is done with this slab's halo tile
        load_halo(kc + 1);
      }
      __syncthreads();         // drains the DMAs: next weight slab (and halo tile) landed, this weight buffer free
    }
  }

  auto rowfn = [&](int i) { return img * hw + (y0 + wm * 4 + i) * a.Win + x0 + (lane & 15); };
  epilogue<NT, MT, WN>(a, acc, rowfn, 0, n0, wn, lane, nullptr, 0);
}

// ---- 8-phase ping-pong gather-GEMM: 256 x BN x 64 tiles, waves 2 (M) x 4 (N), two wave groups half a phase apart ----
// The one-barrier-per-slab loop of gemm_kernel keeps the two waves of a SIMD in lockstep: both read fragments, both run their
// MFMAs, both wait for the slab's DMA at the barrier (vmcnt(0)).  Here the K tile is cut into four phases -- the quadrants
// (A0,B0) (A0,B1) (A1,B1) (A1,B0) of the wave's 128 x WN output, A0/A1 = its two 64-row halves, B0/B1 = its first NT0 / last NT1
// column tiles -- and every phase is  { read the quadrant's new fragments; issue ONE part of a later K tile's DMA }  barrier
// { MFMAs }  barrier.  Waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so on each SIMD one wave's
// MFMA cluster runs beside the other's LDS reads and DMA issue (s_setprio favours the cluster).  DMAs are never drained inside
// the loop: ONE counted s_waitcnt vmcnt(2 + NT1) per K tile (in phase 3) leaves the two youngest parts in flight across barriers.
//
// Parts of K tile t live in LDS buffer t & 1 and are issued, in program order,
//     phase 0 of tile t: A1(t+1)    phase 1: B0(t+1)    phase 2: A0(t+2)    phase 3: B1(t+2), then vmcnt(2 + NT1)
// Read-after-DMA: a part is read at least one full phase (two barriers) after the counted wait that covers it -- the wait of
// phase 3 of tile t retires everything up to B0(t+1); A0(t+1) / B0(t+1) are first read in phase 0 of t+1, B1(t+1) in phase 1,
// A1(t+1) in phase 2 -- so the other wave group (half a phase behind with its own waits) has waited too.
// DMA-after-read: a part's slot is re-issued two phases after its last read (A0: read phase 0, issued phase 2; B1: 1 -> 3;
// A1: 2 -> 0; B0: 3 -> 1).  Tiles past the end are issued as out-of-range offsets (zero fill, no memory traffic) so that the
// counts stay uniform; everything is drained before the epilogue.
// EPI (round 6): the epilogue compiled into this instantiation -- the row-contiguous epilogue with exactly one term set (0 / 2 / 4 / 6 / 12: none, rowvec,
// res, rowvec + res, res + res2; BN = 256: 0 = the GEGLU row pass) or, EPI = -1, every direct epilogue behind run-time tests.  One kernel with all of them
// behind a switch carried 530 spilled VGPRs (hipcc hoists the lane-constant address arithmetic of EVERY variant above the K loop, where 160 accumulators
// and 56 fragment registers leave room for none of it); one variant per instantiation: 0 (22 for res + res2).  Same arithmetic, bitwise the same output.
template <int BM, int BN, bool GATHER, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const me_gemm_args a) {
  // BM = 256, or 192 for grids whose 256-row tiles would leave the last block round half empty (M = 24576 x N = 1280: 384 tiles = 1.5 rounds of
  // the 256 CUs, 512 tiles of 192 rows = 2): wave tile GR x WN with GR = BM / 2 rows per wave group, A halves of HALF = GR / 2 rows.
  constexpr int WN = BN / 4, NT = WN / 16, NT0 = (NT + 1) / 2, NT1 = NT / 2;
  constexpr int GR = BM / 2, HALF = GR / 2, MH = HALF / 16, MT = 2 * MH;
  constexpr int NPA = 2 * HALF / 8;   // 8-row DMA pieces per A part (both wave groups): 16, or 12 -- then waves 4-7 issue one piece, waves 0-3 two
  constexpr int ABYTES = BM * 128, BUFBYTES = (BM + BN) * 128;
  static_assert(BN % 64 == 0 && NT1 >= 1 && HALF % 16 == 0 && NPA >= 8 && NPA <= 16, "wave tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A BM rows | B BN rows][128 B], then (GATHER) the source-row table [taps][256]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nbn = (a.N + BN - 1) / BN;
  const int nbm = (a.M - a.m_off + BM - 1) / BM;   // tiles cover the rows [m_off, M)
  const int w = xcd_remap(blockIdx.x, nbm * nbn);
  // tile order inside an XCD's contiguous run: column tiles fastest (default: the nbn tiles of a row block run side by side and share its A rows in L2) or, with
  // TILEORDER_FLAG (experiment, ME_GEMM_TILE_ORDER=1), row blocks fastest: 32 row blocks of ONE column tile at a time share its weight slabs
  int tile_n, tile_m;
  if (a.splits_ & TILEORDER_FLAG) { tile_m = w % nbm; tile_n = w / nbm; }
  else { tile_n = w % nbn; tile_m = w / nbn; }
  const int m0 = a.m_off + tile_m * BM, n0 = tile_n * BN;

  const f16* __restrict__ X = reinterpret_cast<const f16*>(a.X);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(a.W);
  const int taps = !GATHER ? 1 : (a.gather == ME_GATHER_CONV3 ? 9 : (a.gather == ME_GATHER_TCONV ? 3 : 1));
  const int nkc = a.K / BK;
  const int nit = taps * nkc;

  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr unsigned OOB = 0x80000000u;
  long brow = m0;
  if (GATHER) {
    if (a.gather == ME_GATHER_CONV3) brow = (long)(m0 / (a.Hout * a.Wout)) * a.Hin * a.Win;
    else if (a.gather == ME_GATHER_TCONV) brow = m0 > a.npix ? m0 - a.npix : 0;
  }
  const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(X + brow * a.ldx), 0, OOB, 0x00020000);
  const int wrows = min(a.N - n0, BN);
  const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W + (long)n0 * taps * a.K), 0, (unsigned)((long)wrows * taps * a.K * 2), 0x00020000);

  // DMA lane mapping as in gemm_kernel: one wave instruction = 8 LDS rows x 128 B, lane -> row + lane / 8, 16-byte slot lane % 8
  // holding source chunk (lane % 8) ^ ((row >> 1) & 7); every piece of this wave starts on a row whose bit 3 is wave & 1
  const int prow = lane >> 3;
  const int scol = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  // first LDS row of this wave's piece i of a part (wave-uniform; the DMA's LDS address travels in M0)
  int rA[2][2], rB0[NT0], rB1[NT1];
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // A0 / A1: piece p = wave + 8 i (< NPA) -> wave group p / (HALF / 8), rows 8 (p % (HALF / 8)) of that group's half
    const int p = wave + 8 * i;
    rA[0][i] = __builtin_amdgcn_readfirstlane((p / (HALF / 8)) * GR + 8 * (p % (HALF / 8)));
    rA[1][i] = __builtin_amdgcn_readfirstlane((p / (HALF / 8)) * GR + HALF + 8 * (p % (HALF / 8)));
  }
  const bool two_a = __builtin_amdgcn_readfirstlane(wave + 8 < NPA ? 1 : 0) != 0;   // this wave issues two pieces per A part (always at BM = 256)
#pragma unroll
  for (int i = 0; i < NT0; ++i) {   // B0: 8 NT0 pieces
    const int p = wave + 8 * i;
    rB0[i] = __builtin_amdgcn_readfirstlane((p / (2 * NT0)) * WN + 8 * (p % (2 * NT0)));
  }
#pragma unroll
  for (int i = 0; i < NT1; ++i) {   // B1: 8 NT1 pieces
    const int p = wave + 8 * i;
    rB1[i] = __builtin_amdgcn_readfirstlane((p / (2 * NT1)) * WN + 16 * NT0 + 8 * (p % (2 * NT1)));
  }

  // A-operand source offsets.  Dense: one byte offset per staged row, fixed.  Gathers: the offsets change with the tap, and a branch
  // that recomputes them inside the K loop makes hipcc drain the DMA queue (vmcnt(0)) where the paths join -- so every (tap, row)
  // offset of the tile is computed ONCE into an LDS table (9 KB for a 3x3 convolution) and each issue reads its two entries.
  unsigned* tab = reinterpret_cast<unsigned*>(smem + 2 * BUFBYTES);   // [taps][256]: byte offset of the source row (chunk 0), or OOB
  unsigned xo[2][2];
  if constexpr (GATHER) {
    const int r = tid & 255;
    const RowInfo ri = make_row(a, r < BM ? m0 + r : a.M);
    for (int tap = tid >> 8; tap < taps; tap += 2) {
      const int sr = src_row(a, ri, tap);
      tab[tap * 256 + r] = sr < 0 ? OOB : (unsigned)((long)(sr - brow) * a.ldx * 2);
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + rA[h][i] + prow;
        xo[h][i] = (m < a.M && (i == 0 || two_a)) ? (unsigned)(((long)(m - brow) * a.ldx + scol) * 2) : OOB;
      }
  }
  unsigned wo0[NT0], wo1[NT1];
#pragma unroll
  for (int i = 0; i < NT0; ++i) wo0[i] = (unsigned)(((long)(rB0[i] + prow) * taps * a.K + scol) * 2);
#pragma unroll
  for (int i = 0; i < NT1; ++i) wo1[i] = (unsigned)(((long)(rB1[i] + prow) * taps * a.K + scol) * 2);

  // issue cursors, one per part: tl = K tile the next issue of the part belongs to.  The weight rows are [taps][K] contiguous and K is a
  // multiple of 64, so a B part's scalar offset is simply tl * 128 bytes; the A parts of a gather also track (tap, k chunk).
  struct Cur { int tap, kc, tl; };
  Cur cA0 = {0, 0, 0}, cA1 = {0, 0, 0};
  int tB0 = 0, tB1 = 0;
  auto issueA = [&](Cur& c, int half) {
    char* base = smem + (c.tl & 1) * BUFBYTES;
    const int sx = __builtin_amdgcn_readfirstlane((GATHER ? c.kc : c.tl) * (BK * 2));
    const bool live = c.tl < nit;
    unsigned v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (GATHER) v[i] = tab[__builtin_amdgcn_readfirstlane(min(c.tap, taps - 1)) * 256 + min(rA[half][i] + prow, 255)] + (unsigned)(scol * 2);
      else v[i] = xo[half][i];
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(base + rA[half][0] * 128), 16, (int)(live ? v[0] : OOB), sx, 0, 0);
    if (BM == 256 || two_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(base + rA[half][1] * 128), 16, (int)(live ? v[1] : OOB), sx, 0, 0);
    ++c.tl;
    if constexpr (GATHER) {
      if (++c.kc == nkc) { c.kc = 0; ++c.tap; }
    }
  };
  auto issueB0 = [&]() {
    char* base = smem + (tB0 & 1) * BUFBYTES + ABYTES;
    const int sw = __builtin_amdgcn_readfirstlane(tB0 * (BK * 2));
    const bool live = tB0 < nit;
#pragma unroll
    for (int i = 0; i < NT0; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(base + rB0[i] * 128), 16, (int)(live ? wo0[i] : OOB), sw, 0, 0);
    ++tB0;
  };
  auto issueB1 = [&]() {
    char* base = smem + (tB1 & 1) * BUFBYTES + ABYTES;
    const int sw = __builtin_amdgcn_readfirstlane(tB1 * (BK * 2));
    const bool live = tB1 < nit;
#pragma unroll
    for (int i = 0; i < NT1; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(base + rB1[i] * 128), 16, (int)(live ? wo1[i] : OOB), sw, 0, 0);
    ++tB1;
  };

  f32x4 acc[NT][MT];
  init_acc<NT, MT, WN>(a, acc, n0, wc, lane);

  // fragment reads: lane -> row (tile base + lane & 15), 16-byte chunk ((ks * 4 + lane >> 4) ^ ((row >> 1) & 7)); every tile base is a
  // multiple of 16 rows, so the swizzle term depends on the lane alone and the two k-steps differ by an XOR with 64 bytes
  const int frow = lane & 15, fg = lane >> 4;
  const int c0 = (fg ^ ((frow >> 1) & 7)) << 4;
  const int la0 = (wr * GR + frow) * 128 + c0, la1 = la0 ^ 64;
  const int lb0 = ABYTES + (wc * WN + frow) * 128 + c0, lb1 = lb0 ^ 64;
  f16x8 fa[2][MH], fb[2][NT0];
  auto readA = [&](int buf, int half) {
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      fa[0][i] = *reinterpret_cast<const f16x8*>(smem + buf * BUFBYTES + la0 + (half * MH + i) * 2048);
      fa[1][i] = *reinterpret_cast<const f16x8*>(smem + buf * BUFBYTES + la1 + (half * MH + i) * 2048);
    }
  };
  auto readB = [&](int buf, int j0, int nj) {
#pragma unroll
    for (int j = 0; j < NT0; ++j) {
      if (j < nj) {
        fb[0][j] = *reinterpret_cast<const f16x8*>(smem + buf * BUFBYTES + lb0 + (j0 + j) * 2048);
        fb[1][j] = *reinterpret_cast<const f16x8*>(smem + buf * BUFBYTES + lb1 + (j0 + j) * 2048);
      }
    }
  };
  auto cluster = [&](int half, int j0, int nj) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NT0; ++j)
        if (j < nj) {
#pragma unroll
          for (int i = 0; i < MH; ++i) acc[j0 + j][half * MH + i] = mfma16(fb[ks][j], fa[ks][i], acc[j0 + j][half * MH + i]);
        }
    __builtin_amdgcn_s_setprio(0);
  };
#define ME_BAR()                               \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_s_barrier();              \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)
#define ME_LGKM0()                                       \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

  // the counted wait: everything but the two youngest parts (one A part: 2 or 1 pieces of this wave, B1: NT1 pieces) has landed
  auto wait_young = [&]() {
    static_assert(NT1 == 2, "vmcnt literals below");
    if (BM == 256 || two_a) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  };
  // LayerNorm-folded launch (term-free epilogue, dense): the fold's operands travel with the first K tile -- the oldest requests of every wave, so the
  // prologue's counted wait covers them; wave w moves pieces w and w + 8 of { 2 per part: the tile's BM x (sum, sum of squares) | 2: colsum | 2: cvec }
  constexpr bool LN_LDS = EPI == 0 && !GATHER;
  if constexpr (LN_LDS) {
    if (a.ln_stats) {   // (me_gemm: ln_parts <= 4; the part slots past ln_parts get a zero-length range: the DMA fills them with zeros)
      constexpr int np = 8;
      for (int q = wave; q < np + 4; q += 8) {
        const char* src;
        unsigned range;
        int dst;
        if (q < np) {
          const bool live = (q >> 1) < a.ln_parts;
          src = reinterpret_cast<const char*>(a.ln_stats) + (live ? ((long)(q >> 1) * a.ln_stride + 2 * (long)m0) * 4 : 0L);
          range = live ? (unsigned)min(a.M - m0, BM) * 8u : 0u;
          dst = (q >> 1) * 2048 + (q & 1) * 1024;
        } else {
          const int k = q - np;
          src = reinterpret_cast<const char*>((k >> 1) ? a.ln_cvec : a.ln_colsum) + (long)n0 * 4;
          range = (unsigned)min(a.N - n0, BN) * 4u;
          dst = ((k >> 1) ? LN_LDS_CVEC : LN_LDS_COLSUM) + (k & 1) * 1024;
        }
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, range, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(smem + 2 * BUFBYTES + dst), 16, (q & 1) * 1024 + lane * 16, 0, 0, 0);
      }
    }
  }
  // prologue: tile 0 whole, the first two parts of tile 1
  issueA(cA0, 0);
  issueB0();
  issueB1();
  issueA(cA1, 1);
  issueA(cA0, 0);
  issueB1();
  wait_young();
  ME_BAR();
  if (wr == 1) ME_BAR();   // the second wave of every SIMD runs one barrier behind the first

  for (int t = 0; t < nit; ++t) {
    const int buf = t & 1;
    // phase 0: quadrant (A0, B0)
    readB(buf, 0, NT0);
    readA(buf, 0);
    issueA(cA1, 1);
    ME_BAR();
    ME_LGKM0();
    cluster(0, 0, NT0);
    ME_BAR();
    // phase 1: (A0, B1)
    readB(buf, NT0, NT1);
    issueB0();
    ME_BAR();
    ME_LGKM0();
    cluster(0, NT0, NT1);
    ME_BAR();
    // phase 2: (A1, B1)
    readA(buf, 1);
    issueA(cA0, 0);
    ME_BAR();
    ME_LGKM0();
    cluster(1, NT0, NT1);
    ME_BAR();
    // phase 3: (A1, B0)
    readB(buf, 0, NT0);
    issueB1();
    wait_young();
    ME_BAR();
    ME_LGKM0();
    cluster(1, 0, NT0);
    ME_BAR();
  }
  if (wr == 0) ME_BAR();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail DMAs still write (zeros) into this block's LDS
#undef ME_BAR
#undef ME_LGKM0

  if constexpr (EPI <= 0) {   // (the LayerNorm-folded projections -- q | k | v, to_q, GEGLU -- have no row-vector / residual terms: me_gemm checks)
    if (a.ln_stats) {         // (wave-uniform; ONE form per instantiation: two alternatives behind a run-time test cost 168 spilled registers at the join)
      if constexpr (LN_LDS) ln_fold_acc_lds<NT, MT>(a, acc, smem + 2 * BUFBYTES, wr * GR, wc * WN, lane);   // (the image landed under the prologue's counted wait + barrier)
      else ln_fold_acc<NT, MT>(a, acc, m0 + wr * GR, n0 + wc * WN, lane);
    }
  }

  if constexpr (EPI >= 0 && BN == 256) {   // GEGLU: one 64 KB LDS tile for the whole block (me_gemm has checked geglu, N % 256 == 0, ldc % 8 == 0, 16-byte aligned C)
    __builtin_amdgcn_s_barrier();
    return epilogue_geglu_rowpass<NT, MT, WN>(a, acc, m0, n0, wr, wc, lane, smem);
  } else if constexpr (EPI >= 0) {         // row-contiguous epilogue; me_gemm has checked alignments, N % 320 == 0, no activation / GEGLU, and the term set is EPI
    // every wave's DMAs have landed (its own vmcnt(0) above + this barrier): the staging buffers are dead, each wave takes 32 x 160 B of them
    __builtin_amdgcn_s_barrier();
    char* scr = smem + wave * (32 * (WN * 2 + 16));
    const int mw0 = m0 + wr * GR, nw0 = n0 + wc * WN;
    const LnOutCtx lnx = {m0, wr, wc, tid, smem + 8 * (32 * (WN * 2 + 16))};    // `red` (8 KB) behind the eight scratches
    return epilogue_rowpass<EPI, NT, MT, WN>(a, acc, mw0, nw0, lane, scr, lnx);
  } else {
    auto rowfn = [&](int i) {
      const int m = m0 + wr * GR + i * 16 + (lane & 15);
      return m < a.M ? m : -1;
    };
    epilogue<NT, MT, WN>(a, acc, rowfn, m0, n0, wc, lane, nullptr, 0);
  }
}

// split-K second pass: C = epilogue(alpha * sum_s work[s] + bias ...) with the epilogue semantics of the one-pass kernels (no activation: round to
// fp16, then add rowvec / res / res2 in fp16; with an activation: everything in fp32, one rounding)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const me_gemm_args a) {
  const int vpr = a.N / 4;
  const long total = (long)a.M * vpr;
  const float* __restrict__ wk = reinterpret_cast<const float*>(a.work);
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(a.bias);
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(a.rowvec);
  const f16* res = reinterpret_cast<const f16*>(a.res);
  const f16* res2 = reinterpret_cast<const f16*>(a.res2);
  f16* C = reinterpret_cast<f16*>(a.C);
  const long plane = (long)a.M * a.N;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / vpr);
    const int n = (int)(idx - (long)m * vpr) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(wk + (long)m * a.N + n);
    for (int s_ = 1; s_ < a.splits_; ++s_) v += *reinterpret_cast<const f32x4*>(wk + s_ * plane + (long)m * a.N + n);
    v *= a.alpha;
    if (bias) {
      U64 b;
      b.u = *reinterpret_cast<const uint2*>(bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)b.e[r];
    }
    U64 rv, r1, r2, o;
    if (rowvec) rv.u = *reinterpret_cast<const uint2*>(rowvec + (long)(m / a.rows_per_vec) * a.ldrv + n);
    if (res) r1.u = *reinterpret_cast<const uint2*>(res + (long)(a.res_rows > 0 ? m % a.res_rows : m) * a.ldr + n);
    if (res2) r2.u = *reinterpret_cast<const uint2*>(res2 + (long)(a.res2_rows > 0 ? m % a.res2_rows : m) * a.ldr2 + n);
    if (a.act == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f16 h = (f16)v[r];
        if (rowvec) h = h + rv.e[r];
        if (res) h = h + r1.e[r];
        if (res2) h = h + r2.e[r];
        o.e[r] = h;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float f = v[r];
        if (rowvec) f += (float)rv.e[r];
        f = a.act == 1 ? fmaxf(f, 0.f) : silu_f(f);
        if (res) f += (float)r1.e[r];
        if (res2) f += (float)r2.e[r];
        o.e[r] = (f16)f;
      }
    }
    *reinterpret_cast<uint2*>(C + (long)m * a.ldc + n) = o.u;
  }
}

// split-K decision (shared by me_gemm and me_gemm_work_bytes): small grids of the 128-row kernels with a long K loop.  `blocks` = tiles of the kernel
// that would run; returns the number of K splits (1 = none).
long split_below() {   // ME_GEMM_SPLITK: grids of fewer blocks than this are split along K (0 = never)
  const char* e = getenv("ME_GEMM_SPLITK");
  return e ? atol(e) : 400;
}
int choose_split(const me_gemm_args* a, long blocks, int nit) {
  // N >= 1280 only (the 16 x 16- and 8 x 8-latent levels): a split changes the fp32 summation order, and the level-0 / level-1 launches must give
  // the same rows whatever the batch size -- the UNet graph runs its first blocks on half the batch (classifier-free-guidance prefix) and the
  // step has to stay bitwise the same.  (A 20-tile K loop measured slower split than whole: 32 tiles at least.)
  if (a->geglu || a->K % 64 || a->N < 1280 || blocks >= split_below() || nit < 32 || a->C2 || a->m_off || a->ln_stats) return 1;
  int S = (int)((640 + blocks - 1) / blocks);
  if (S > 4) S = 4;
  if (S > nit / 4) S = nit / 4;
  return S < 2 ? 1 : S;
}

int stage_impl() {
  static int impl = -1;
  if (impl < 0) {
    const char* e = getenv("ME_GEMM_STAGE");
    impl = (e && e[0] == 'r') ? STAGE_REG : STAGE_GLDS;  // ME_GEMM_STAGE=reg selects the register-staged kernel
  }
  return impl;
}

long big_min_blocks() {   // ME_GEMM_BIG_MIN: smallest grid (in 256x320 blocks) that gets the big tile; 0 disables nothing, huge disables it
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("ME_GEMM_BIG_MIN");
    v = e ? atol(e) : 512;
  }
  return v;
}

long halo_min_blocks() {   // ME_CONV_HALO_MIN: smallest grid (16x16-pixel patches x 320-channel tiles) that takes the halo kernel
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("ME_CONV_HALO_MIN");
    v = e ? atol(e) : 512;
  }
  return v;
}

bool conv_halo() {   // ME_CONV_HALO=0 sends 3x3 convolutions back to the gather kernel
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("ME_CONV_HALO");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

bool buf_stage() {   // ME_GEMM_BUF=0: keep the global_load_lds staging for every shape (A/B)
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("ME_GEMM_BUF");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

int use_8p() {   // ME_GEMM_8P=0: every big-tile GEMM stays on the one-barrier-per-slab kernel (A/B); default: K tiles >= ME_GEMM_8P (2).
  const char* e = getenv("ME_GEMM_8P");   // read at every call so that tests and tools/kbench.py can flip it inside one process
  return e ? atoi(e) : 2;
}

bool row_epilogue() {   // ME_GEMM_ROWEPI=0: the 8-phase kernels keep the direct epilogue (A/B; read per call so that tests can flip it in-process)
  const char* e = getenv("ME_GEMM_ROWEPI");
  return !(e && e[0] == '0');
}

bool tile_order_exp() {   // ME_GEMM_TILE_ORDER=1: row blocks fastest inside an XCD's run (an experiment of round 5; read once)
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("ME_GEMM_TILE_ORDER");
    on = (e && e[0] == '1') ? 1 : 0;
  }
  return on == 1;
}

long min_tiles_192() {   // ME_GEMM_8P_192: smallest grid (in 192 x 320 tiles) that takes the 192-row 8-phase kernel (0 = never)
  const char* e = getenv("ME_GEMM_8P_192");
  const long v = e ? atol(e) : 192;
  return v > 0 ? v : (1L << 60);
}

long min_tiles_128() {   // ME_GEMM_8P_128: smallest grid (in 128 x 320 tiles) that takes the 128-row 8-phase kernel (0 = never; read per call)
  const char* e = getenv("ME_GEMM_8P_128");
  const long v = e ? atol(e) : 192;
  return v > 0 ? v : (1L << 60);
}

long geglu_min_tiles() {   // ME_GEMM_GEGLU_MIN: smallest grid (in 256 x 256 tiles) that takes the 8-phase GEGLU kernel (read per call: tools/ flip it)
  const char* e = getenv("ME_GEMM_GEGLU_MIN");
  return e ? atol(e) : 240;
}

int min_ktiles_192() {   // ME_GEMM_192_MINK: fewest K tiles (of 64) for the 192-row 8-phase kernel (read per call: tools/kbench.py flips it)
  const char* e = getenv("ME_GEMM_192_MINK");
  return e ? atoi(e) : 4;
}

long n64_below() {   // ME_GEMM_N64_BELOW: grids of fewer 128 x 128 tiles than this take 128 x 64 tiles (M = 1536 convolutions: 0.136 -> 0.117 ms)
  const char* e = getenv("ME_GEMM_N64_BELOW");
  return e ? atol(e) : 200;
}

bool tile160() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("ME_GEMM_TILE160");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

}  // namespace

extern "C" void me_set_error(const char* msg);
extern "C" void me_set_kernel(const char* name);
extern "C" const char* me_last_kernel(void);

static thread_local bool g_ln_fused = false;   // the launch me_gemm just made writes me_gemm_args.ln_out from its own epilogue (else me_gemm appends me_ln_stats)

template <int BM, int BN, int STAGE, int WM = 64>
static int launch_gemm(const me_gemm_args* a, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * (STAGE != STAGE_REG ? BK : BK + 8) * sizeof(f16);
  static bool attr_set_dev[64] = {};   // the attribute is per device: a process that drives several GPUs sets it on each
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, STAGE, WM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      me_set_error("me_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return ME_EHIP;
    }
    attr_set = true;
  }
  const int nbm = (a->M - a->m_off + BM - 1) / BM, nbn = (a->N + BN - 1) / BN;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  me_gemm_args b = *a;
  b.splits_ = 0;
  int S = 1;
  if (STAGE == STAGE_BUF && a->work) {
    const int nit = (a->K / 64) * (a->gather == ME_GATHER_CONV3 ? 9 : (a->gather == ME_GATHER_TCONV ? 3 : 1));
    // (the split decision looks at the grid of the launch the caller SELECTS by -- sel_rows: a row-range piece or a sub-batch must sum in the order of
    // the full launch)
    const int Msel = a->sel_rows > a->M ? a->sel_rows : a->M;
    S = choose_split(a, (long)((Msel + BM - 1) / BM) * nbn, nit);
    if ((int64_t)S * a->M * a->N * 4 > a->work_bytes || a->N % 4 || ((uintptr_t)a->work & 15)) S = 1;
  }
  if (S > 1) {
    b.splits_ = S;
    hipLaunchKernelGGL((gemm_kernel<BM, BN, STAGE, WM>), dim3(nbm * nbn, S), dim3(BM / WM * 128), lds, st, b);
    const long vec = (long)a->M * (a->N / 4);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((vec + 255) / 256 < 4096 ? (vec + 255) / 256 : 4096)), dim3(256), 0, st, b);
  } else {
    hipLaunchKernelGGL((gemm_kernel<BM, BN, STAGE, WM>), dim3(nbm * nbn), dim3(BM / WM * 128), lds, st, b);
  }
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "gemm_kernel<%d,%d%s>%s", BM, BN, STAGE == STAGE_REG ? ",reg" : "", S > 1 ? "+splitk" : "");
    me_set_kernel(nm);
  }
  if (hipGetLastError() != hipSuccess) {
    me_set_error("me_gemm: kernel launch failed");
    return ME_EHIP;
  }
  return ME_OK;
}

template <int BM, int BN, bool GATHER, int EPI>
static int launch_gemm8p_epi(const me_gemm_args& b, hipStream_t st) {
  const int lds = 2 * (BM + BN) * 128 + (GATHER ? 9 * 256 * 4 : (EPI == 0 ? LN_LDS_BYTES : 0));   // (dense, term-free: room for the LayerNorm fold's operands)
  static bool attr_set_dev[64] = {};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<BM, BN, GATHER, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      me_set_error("me_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return ME_EHIP;
    }
    attr_set = true;
  }
  const int nbm = (b.M - b.m_off + BM - 1) / BM, nbn = (b.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm8p_kernel<BM, BN, GATHER, EPI>), dim3(nbm * nbn), dim3(512), lds, st, b);
  return ME_OK;
}

template <int BM, int BN, bool GATHER>
static int launch_gemm8p(const me_gemm_args* a, hipStream_t st) {
  (void)hipGetLastError();
  me_gemm_args b = *a;
  b.splits_ = 0;
  int epi = -1;   // the direct epilogues
  {   // row-contiguous epilogue (epilogue_rowpass): 16-byte pieces of every tensor it touches, one of the specialised term sets
    auto al = [](const void* p, int ld) { return p == nullptr || (ld % 8 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0); };
    const int f = (a->rowvec ? 2 : 0) | (a->res ? 4 : 0) | (a->res2 ? 8 : 0);
    if (BN == 320 && row_epilogue() && !a->geglu && a->act == 0 && a->N % 320 == 0 && (f == 0 || f == 2 || f == 4 || f == 6 || f == 12) && al(a->C, a->ldc) &&
        al(a->rowvec, a->ldrv) && al(a->res, a->ldr) && al(a->res2, a->ldr2) && (!a->C2 || f == 0))
      epi = f;
    if (tile_order_exp()) b.splits_ |= TILEORDER_FLAG;
    if (BN == 256 && BM == 256 && row_epilogue() && a->geglu && a->N % 256 == 0 && al(a->C, a->ldc) && !a->C2) epi = 0;
    if (epi >= 0) b.splits_ |= ROWEPI_FLAG;
    g_ln_fused = BN == 320 && epi >= 0 && a->ln_out != nullptr;
    if (!g_ln_fused) b.ln_out = nullptr;
  }
  int rc;
  if constexpr (BN == 256) {
    rc = epi == 0 ? launch_gemm8p_epi<BM, BN, GATHER, 0>(b, st) : launch_gemm8p_epi<BM, BN, GATHER, -1>(b, st);
  } else {
    switch (epi) {
      case 0: rc = launch_gemm8p_epi<BM, BN, GATHER, 0>(b, st); break;
      case 2: rc = launch_gemm8p_epi<BM, BN, GATHER, 2>(b, st); break;
      case 4: rc = launch_gemm8p_epi<BM, BN, GATHER, 4>(b, st); break;
      case 6: rc = launch_gemm8p_epi<BM, BN, GATHER, 6>(b, st); break;
      case 12: rc = launch_gemm8p_epi<BM, BN, GATHER, 12>(b, st); break;
      default: rc = launch_gemm8p_epi<BM, BN, GATHER, -1>(b, st); break;
    }
  }
  if (rc != ME_OK) return rc;
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "gemm8p_kernel<%d,%d,%s>", BM, BN, GATHER ? "true" : "false");
    me_set_kernel(nm);
  }
  if (hipGetLastError() != hipSuccess) {
    me_set_error("me_gemm: kernel launch failed");
    return ME_EHIP;
  }
  return ME_OK;
}

static int launch_conv_halo(const me_gemm_args* a, hipStream_t st) {
  const int lds = HALO_BYTES + 2 * CONVW_BYTES;
  static bool attr_set_dev[64] = {};   // the attribute is per device: a process that drives several GPUs sets it on each
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      me_set_error("me_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return ME_EHIP;
    }
    attr_set = true;
  }
  const long blocks = (long)(a->M / 256) * (a->N / 320);
  (void)hipGetLastError();
  hipLaunchKernelGGL(conv3_halo_kernel, dim3((unsigned)blocks), dim3(512), lds, st, *a);
  me_set_kernel("conv3_halo_kernel");
  if (hipGetLastError() != hipSuccess) {
    me_set_error("me_gemm: kernel launch failed");
    return ME_EHIP;
  }
  return ME_OK;
}

extern "C" int64_t me_gemm_work_bytes(const me_gemm_args* a) {
  // upper bound over the tile choices of the 128-row kernels (the smallest tile, 128 x 64, gives the most blocks and so the fewest splits: use the
  // largest tile, 128 x 160, for the bound): 4 splits at most
  if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->geglu || a->K % 64 || a->N % 4) return 0;
  const int nit = (a->K / 64) * (a->gather == ME_GATHER_CONV3 ? 9 : (a->gather == ME_GATHER_TCONV ? 3 : 1));
  const long blocks = (long)((a->M + 127) / 128) * ((a->N + 159) / 160);
  const int S = choose_split(a, blocks, nit);
  return S > 1 ? (int64_t)4 * a->M * a->N * 4 : 0;
}

static int gemm_dispatch(const me_gemm_args* a, void* stream);

extern "C" int me_gemm(const me_gemm_args* a, void* stream) {
  if (!a || !a->X || !a->W || !a->C) { me_set_error("me_gemm: null pointer"); return ME_EINVAL; }
  if (a->ln_stats) {
    if (a->gather != ME_GATHER_DENSE || a->bias || a->alpha != 1.0f || a->rowvec || a->res || a->res2 || a->act || !a->ln_colsum || !a->ln_cvec || a->ln_parts < 1 || a->ln_parts > 4 || a->ln_stride < 2 * (int64_t)a->M ||
        (a->ln_stride & 1) || ((uintptr_t)a->ln_stats & 7) || (((uintptr_t)a->ln_colsum | (uintptr_t)a->ln_cvec) & 15) || !(a->ln_eps > 0.f)) {
      me_set_error("me_gemm: a LayerNorm-folded launch is dense, has no bias (it is inside ln_cvec), no rowvec / residual / activation and alpha == 1; ln_colsum / ln_cvec fp32 [N] 16-byte aligned, "
                   "ln_stats 8-byte aligned with 1 <= ln_parts <= 4 parts ln_stride >= 2 M floats apart, ln_eps > 0");
      return ME_EINVAL;
    }
  }
  if (a->ln_out && (a->geglu || a->C2 || a->N % 8 || a->N > 1536 || a->ldc % 8 || ((uintptr_t)a->C & 15) || ((uintptr_t)a->ln_out & 7) || a->ln_out_stride < 2 * (int64_t)a->M || (a->ln_out_stride & 1))) {
    me_set_error("me_gemm: ln_out needs a plain fp16 output of N % 8 == 0, N <= 1536 columns (no GEGLU, no head-major panels), ldc % 8 == 0, and parts >= 2 M floats apart");
    return ME_EINVAL;
  }
  g_ln_fused = false;
  const int rc = gemm_dispatch(a, stream);
  if (rc != ME_OK || !a->ln_out || g_ln_fused) return rc;
  // the kernel that took the launch has no row sums in its epilogue: a read-only pass over the rows it wrote
  char nm[96];
  snprintf(nm, sizeof(nm), "%s", me_last_kernel());
  const int rc2 = me_ln_stats(reinterpret_cast<const f16*>(a->C) + (long)a->m_off * a->ldc, a->ldc, (int64_t)a->M - a->m_off, a->N,
                              reinterpret_cast<float*>(a->ln_out) + 2 * (long)a->m_off, a->ln_out_stride, stream);
  me_set_kernel(nm);
  return rc2;
}

static int gemm_dispatch(const me_gemm_args* a, void* stream) {
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) { me_set_error("me_gemm: non-positive dimension"); return ME_EINVAL; }
  if (a->K % 8 || a->ldx % 8 || a->ldc % 4 || a->N % 4) { me_set_error("me_gemm: K, ldx must be multiples of 8 and N, ldc of 4"); return ME_EINVAL; }
  if (((uintptr_t)a->X | (uintptr_t)a->W) & 15 || ((uintptr_t)a->C & 7)) { me_set_error("me_gemm: misaligned pointer"); return ME_EINVAL; }
  if (a->gather < 0 || a->gather > 2) { me_set_error("me_gemm: bad gather mode"); return ME_EINVAL; }
  if (a->gather == ME_GATHER_CONV3) {
    if (a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0 || (a->stride != 1 && a->stride != 2) || a->ups < 0 || a->ups > 2 ||
        (a->pad0 != 0 && a->pad0 != 1) || a->M % (a->Hout * a->Wout)) { me_set_error("me_gemm: bad conv geometry"); return ME_EINVAL; }
  }
  if (a->gather == ME_GATHER_TCONV) {
    const int ftot = a->frames_total > 0 ? a->frames_total : a->frames;
    // (a row-range launch, m_off > 0 or M cut short, ends where its caller says: rows must then be whole pixels' rows of the full problem, which the caller owns)
    if (a->frames <= 0 || a->npix <= 0 || a->chunk <= 0 || ftot % a->chunk || (a->m_off == 0 && a->sel_rows <= a->M && a->M % (a->frames * a->npix)) || a->frame0 < 0 ||
        a->frame0 + a->frames > ftot) { me_set_error("me_gemm: bad tconv geometry"); return ME_EINVAL; }
  }
  if (a->rowvec && (a->rows_per_vec <= 0 || a->ldrv % 4 || ((uintptr_t)a->rowvec & 7))) { me_set_error("me_gemm: bad rowvec"); return ME_EINVAL; }
  if (a->res && (a->ldr % 4 || ((uintptr_t)a->res & 7))) { me_set_error("me_gemm: bad residual"); return ME_EINVAL; }
  if (a->res2 && (a->ldr2 % 4 || ((uintptr_t)a->res2 & 7))) { me_set_error("me_gemm: bad second residual"); return ME_EINVAL; }
  if (a->res_rows < 0 || a->res2_rows < 0) { me_set_error("me_gemm: negative res_rows"); return ME_EINVAL; }
  if (a->act < 0 || a->act > 2) { me_set_error("me_gemm: bad activation"); return ME_EINVAL; }
  if (a->bias && ((uintptr_t)a->bias & 7)) { me_set_error("me_gemm: misaligned bias"); return ME_EINVAL; }
  if (a->geglu && (a->N % 32 || a->rowvec || a->res || a->res2 || a->act || a->alpha != 1.0f)) { me_set_error("me_gemm: geglu needs N % 32 == 0, alpha == 1 and no rowvec/res/act"); return ME_EINVAL; }
  if (a->m_off < 0 || a->m_off >= a->M || (a->m_off > 0 && a->gather == ME_GATHER_CONV3)) { me_set_error("me_gemm: m_off must lie in [0, M) and is for dense / TemporalConv launches"); return ME_EINVAL; }
  if (a->C2 && (a->geglu || a->act || a->rowvec || a->res || a->res2 || a->c2_dh <= 0 || a->c2_dh % 8 || a->c2_col0 < 0 || a->c2_col0 % 16 || a->c2_col0 >= a->N ||
                (a->N - a->c2_col0) % a->c2_dh || a->c2_hs % 8 || a->c2_hs < (int64_t)a->M * a->c2_dh || ((uintptr_t)a->C2 & 15) || a->gather == ME_GATHER_CONV3)) {
    me_set_error("me_gemm: head-major output (C2) needs a term-free epilogue, c2_dh % 8 == 0, c2_col0 % 16 == 0, whole heads, c2_hs % 8 == 0 and >= M * c2_dh, 16-byte alignment");
    return ME_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // N tile: every channel count of the model (320 ... 10240) is a multiple of 160 -> exact 128x160 tiles;
  // GEGLU needs whole (value, gate) 32-row pairs per wave -> 128; leftovers (4, 16, 32, 96, 256) -> 128 / 64 with a tail
  const bool wide = a->N % 128 == 0 || a->N % 64 != 0;
  if (stage_impl() == STAGE_GLDS) {
    // big tile when the grid still fills the chip: every model width is a multiple of 320
    // EVERY tile-count heuristic below uses Msel (round-4 advisor finding: only the big-tile and halo choices did, so the interior and boundary
    // pieces of a row-range TemporalConv could land on different kernel families than the unsplit launch).
    // kernel selection looks at the grid the launch WOULD have with sel_rows rows (> 0: the caller computes a sub-batch of a larger launch once -- the
    // classifier-free-guidance prefix of the UNet graph -- and wants the rows it gets to be bitwise those of the full launch: the halo kernel and the
    // gather kernels add the (tap, channel slab) products in different orders)
    const int Msel = a->sel_rows > a->M ? a->sel_rows : a->M;
    const long big_blocks = (long)((Msel + 255) / 256) * (a->N / 320);
    if (a->gather == ME_GATHER_CONV3 && a->stride == 1 && a->ups == 0 && !a->pad0 && a->N % 320 == 0 && a->K % 64 == 0 && a->Hin % 16 == 0 &&
        a->Win % 16 == 0 && !a->geglu && big_blocks >= halo_min_blocks() && conv_halo())
      return launch_conv_halo(a, st);
    const bool buf = a->K % 64 == 0 && buf_stage();   // scalar-offset buffer staging (no K tail, no packed-tap mode)
    const int nit8 = (a->K / 64) * (a->gather == ME_GATHER_CONV3 ? 9 : (a->gather == ME_GATHER_TCONV ? 3 : 1));
    const bool dense = a->gather == ME_GATHER_DENSE;
    // GEGLU on the 256-wide 8-phase kernel from geglu_min_tiles() tiles of 256 x 256 on (round 6: its own threshold -- it used to share the 256 x 320 kernels')
    if (a->geglu && dense && buf && use_8p() > 0 && nit8 >= use_8p() && a->N % 256 == 0 && (long)((Msel + 255) / 256) * (a->N / 256) >= geglu_min_tiles())
      return launch_gemm8p<256, 256, false>(a, st);
    if (a->N % 320 == 0 && big_blocks >= big_min_blocks()) {
      // the 8-phase kernel: K tiles of 64, at least use_8p() of them; GEGLU pairs (value, gate) column tiles inside a wave -> 256-wide
      // tiles with 4 column tiles per wave (every GEGLU width of the model, 2560 ... 10240, is a multiple of 256)
      if (buf && use_8p() > 0 && nit8 >= use_8p()) {
        if (!a->geglu) return dense ? launch_gemm8p<256, 320, false>(a, st) : launch_gemm8p<256, 320, true>(a, st);
        if (dense && a->N % 256 == 0 && (long)((Msel + 255) / 256) * (a->N / 256) >= big_min_blocks()) return launch_gemm8p<256, 256, false>(a, st);
      }
      return buf ? launch_gemm<256, 320, STAGE_BUF>(a, st) : launch_gemm<256, 320, STAGE_GLDS>(a, st);
    }
    // grids of 256 < tiles < 512 (the M = 24576 level at N = 1280: 384 tiles = 1.5 rounds of the 256 CUs): 192-row tiles of the 8-phase kernel
    // make it 512 = 2 full rounds
    // (K >= 512 only: the K = 320 projections of this size are bound by their residual / output traffic and measured 7 % slower)
    if (a->N % 320 == 0 && !a->geglu && buf && use_8p() > 0 && nit8 >= min_ktiles_192() && nit8 >= use_8p() && (long)((Msel + 191) / 192) * (a->N / 320) >= min_tiles_192())
      return dense ? launch_gemm8p<192, 320, false>(a, st) : launch_gemm8p<192, 320, true>(a, st);
    // ... and 128-row tiles of it (round 6) for DENSE grids from min_tiles_128() tiles of 128 x 320 on (the M = 6144 level at N = 1280: 192 tiles): +4 ... 30 % with
    // terms / row sums, +-0 without; the gather form measured 10 % SLOWER than 128 x 128 tiles there and stays out (profiles/r06_gemm_dispatch.txt)
    if (dense && a->N % 320 == 0 && !a->geglu && buf && use_8p() > 0 && nit8 >= min_ktiles_192() && nit8 >= use_8p() && (long)((Msel + 127) / 128) * (a->N / 320) >= min_tiles_128())
      return launch_gemm8p<128, 320, false>(a, st);
    // small grids (level 3 / mid block / ControlNet): 128-wide N tiles give 25 % more blocks until the 256 CUs have
    // two each (+8 % on those shapes; 64-row tiles measured worse)
    const long blocks160 = (long)((Msel + 127) / 128) * ((a->N + 159) / 160);
    {   // grids that leave more than half of the 256 CUs without a 128 x 128 tile (the M = 1536 level: 120 tiles): 64-wide tiles double the block count
      const long blocks128 = (long)((Msel + 127) / 128) * ((a->N + 127) / 128);
      if (a->N % 64 == 0 && blocks128 < n64_below()) return buf ? launch_gemm<128, 64, STAGE_BUF>(a, st) : launch_gemm<128, 64, STAGE_GLDS>(a, st);
    }
    if (a->N % 128 == 0 && blocks160 < 512) return buf ? launch_gemm<128, 128, STAGE_BUF>(a, st) : launch_gemm<128, 128, STAGE_GLDS>(a, st);
    if (!a->geglu && a->N % 160 == 0 && tile160()) return buf ? launch_gemm<128, 160, STAGE_BUF>(a, st) : launch_gemm<128, 160, STAGE_GLDS>(a, st);
    if (wide) return buf ? launch_gemm<128, 128, STAGE_BUF>(a, st) : launch_gemm<128, 128, STAGE_GLDS>(a, st);
    return launch_gemm<128, 64, STAGE_GLDS>(a, st);
  }
  return wide ? launch_gemm<128, 128, STAGE_REG>(a, st) : launch_gemm<128, 64, STAGE_REG>(a, st);
}
