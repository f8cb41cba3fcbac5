// Temporal (per-pixel, over frames) causal self-attention.  Rows are (b*F + frame)*npix + pixel, so
// the reference's "(b f) d c -> (b d) f c" rearrange (attention_2d.py:535,545) is pure indexing here.
// HBM-bound: algorithmic bytes = 4 tensors x rows x C x 2.
//
// One block = one (batch, pixel, 320-column slice): the slice's K and V rows of all F frames are staged ONCE
// into LDS with coalesced 16-byte loads (the first version let each of the F query threads re-read them
// through L1 and was L1-bandwidth-bound).  One thread = one (head, query frame): the F x F score row lives
// in registers, K/V come from LDS (the F lanes of a head read the same address -> broadcast).
#include "me_common.h"
#include "../../include/motioned.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int SLICE = 320;      // columns per block: 8 heads x 40, 4 x 80 or 2 x 160
constexpr int SLD = SLICE + 8;  // LDS row stride in halves

template <int F>
__global__ __launch_bounds__(512) void tattn_kernel(const me_tattn_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* sK = reinterpret_cast<f16*>(smem);  // [F][SLD]
  f16* sV = sK + F * SLD;

  const int nslice = (a.heads * a.dh) / SLICE;
  int bid = blockIdx.x;
  const int sl = bid % nslice;
  bid /= nslice;
  const int p = bid % a.npix;
  const int b = bid / a.npix;
  const int kb = a.kv_map[b];
  const int hps = SLICE / a.dh;  // heads per slice
  const int QF = a.q_frames > 0 ? a.q_frames : F;          // local query frames (frame sharding) ...
  const int q0 = a.q_frames > 0 ? a.q_frame0 : 0;          // ... starting at this global frame
  const int parts = a.kv_parts > 1 ? a.kv_parts : 1;
  const int fpp = F / parts;                               // frames per all-gathered K/V part
  const int nthr = hps * QF;
  const int tid = threadIdx.x;

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  f16* __restrict__ O = reinterpret_cast<f16*>(a.O);
  const int col0 = sl * SLICE;

  // stage K, V: F rows x 40 chunks of 16 bytes each
  for (int idx = tid; idx < F * (SLICE / 8); idx += blockDim.x) {
    const int j = idx / (SLICE / 8), c = idx - j * (SLICE / 8);
    const long row = ((long)(j / fpp) * a.batch + kb) * fpp * a.npix + (long)(j % fpp) * a.npix + p;
    *reinterpret_cast<uint4*>(sK + j * SLD + c * 8) = ldg128(K + row * a.ldk + col0 + c * 8);
    *reinterpret_cast<uint4*>(sV + j * SLD + c * 8) = ldg128(V + row * a.ldv + col0 + c * 8);
  }
  __syncthreads();
  if (tid >= nthr) return;

  const int hl = tid / QF, il = tid - hl * QF;
  const int i = q0 + il;                                   // global frame of this query
  const int lcol = hl * a.dh;
  // query/output row: (b, local frame, pixel), or part-major like K/V after the frame<->pixel all-to-all (q_parts > 1)
  const long qrow = a.q_parts > 1 ? ((long)(i / fpp) * a.batch + b) * fpp * a.npix + (long)(i % fpp) * a.npix + p : ((long)b * QF + il) * a.npix + p;
  const int nch = a.dh / 8;

  float s[F];
#pragma unroll
  for (int j = 0; j < F; ++j) s[j] = 0.f;
  for (int cc = 0; cc < nch; ++cc) {
    U128 q;
    q.u = ldg128(Q + qrow * a.ldq + col0 + lcol + cc * 8);
    const f16x2* q2 = reinterpret_cast<const f16x2*>(&q);
#pragma unroll
    for (int j = 0; j < F; ++j) {
      U128 k;
      k.u = *reinterpret_cast<const uint4*>(sK + j * SLD + lcol + cc * 8);
      const f16x2* k2 = reinterpret_cast<const f16x2*>(&k);
      float acc = s[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2(q2[e], k2[e], acc, false);
      s[j] = acc;
    }
  }

  // causal softmax over j <= i (the reference adds -10000 above the diagonal: exp underflows to 0 in fp32)
  const float c = a.scale * 1.4426950408889634f;
  float mx = -1.0e30f;
#pragma unroll
  for (int j = 0; j < F; ++j) {
    s[j] = j <= i ? s[j] * c : -1.0e30f;
    mx = fmaxf(mx, s[j]);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < F; ++j) {
    s[j] = __builtin_amdgcn_exp2f(s[j] - mx);
    l += s[j];
  }
  const float inv = 1.0f / l;

  for (int cc = 0; cc < nch; ++cc) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < F; ++j) {
      U128 v;
      v.u = *reinterpret_cast<const uint4*>(sV + j * SLD + lcol + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += s[j] * (float)v.e[e];
    }
    U128 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov.e[e] = (f16)(o[e] * inv);
    *reinterpret_cast<uint4*>(O + qrow * a.ldo + col0 + lcol + cc * 8) = ov.u;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The MFMA form (dh 40 / 80 / 160, up to 64 frames).  The per-thread kernel above re-reads every K and V row of a head from
// LDS once per QUERY THREAD (24 threads x 48 ds_read_b128 per 8-column chunk): the LDS pipe, not HBM, paces it (2 TB/s).
// Here one WAVE owns a (pixel, head): S^T = K Q^T and O^T = V^T P^T are 16x16x32 MFMAs in the layouts of attn.hip -- K rows
// as operand A straight from the staged row-major tile, Q fragments straight from global memory, P^T from the S^T
// accumulators as operand B, V^T through ds_read_b64_tr_b16 -- so a head reads its K and V tile from LDS once.
// Frames are padded to KP = 32 or 64 keys / queries with zero rows; the causal mask also removes the pad keys.
template <int OFF>
__device__ __forceinline__ uint2 t_lds_tr16(unsigned addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int... I, class Fn>
__device__ __forceinline__ void t_static_for(std::integer_sequence<int, I...>, Fn&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

template <int DH, int KP>
__global__ __launch_bounds__(64 * (SLICE / DH)) void tattn_mfma_kernel(const me_tattn_args a) {
  constexpr int HPS = SLICE / DH, NTHR = 64 * HPS;
  constexpr int D32 = (DH + 31) / 32, DT = (DH + 15) / 16, NT = KP / 16, KK = KP / 32;
  constexpr int SLDB = SLD * 2;                                   // row pitch in bytes (656: an odd number of 16-byte chunks)
  __shared__ __attribute__((aligned(16))) char smem[2 * KP * SLDB + 128];
  char* sK = smem;
  char* sV = smem + KP * SLDB;

  const int F = a.frames;
  const int nslice = (a.heads * a.dh) / SLICE;
  int bid = blockIdx.x;
  const int sl = bid % nslice;
  bid /= nslice;
  const int p = bid % a.npix;
  const int b = bid / a.npix;
  const int kb = a.kv_map[b];
  const int QF = a.q_frames > 0 ? a.q_frames : F;
  const int q0 = a.q_frames > 0 ? a.q_frame0 : 0;
  const int parts = a.kv_parts > 1 ? a.kv_parts : 1;
  const int fpp = F / parts;
  const int tid = threadIdx.x, lane = tid & 63, hl = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  f16* __restrict__ O = reinterpret_cast<f16*>(a.O);
  const int col0 = sl * SLICE, lcol = hl * DH;

  // stage K and V: KP rows x 41 chunks (40 real + the pad chunk, zeroed); rows >= F are zero
  for (int idx = tid; idx < KP * (SLD / 8); idx += NTHR) {
    const int j = idx / (SLD / 8), c = idx - j * (SLD / 8);
    uint4 kq = make_uint4(0u, 0u, 0u, 0u), vq = kq;
    if (j < F && c < SLICE / 8) {
      const long row = ((long)(j / fpp) * a.batch + kb) * fpp * a.npix + (long)(j % fpp) * a.npix + p;
      kq = ldg128(K + row * a.ldk + col0 + c * 8);
      vq = ldg128(V + row * a.ldv + col0 + c * 8);
    }
    *reinterpret_cast<uint4*>(sK + j * SLDB + c * 16) = kq;
    *reinterpret_cast<uint4*>(sV + j * SLDB + c * 16) = vq;
  }
  if (tid < 8) reinterpret_cast<uint4*>(smem + 2 * KP * SLDB)[tid] = make_uint4(0u, 0u, 0u, 0u);   // slack behind the last row

  // Q fragments (operand B): lane (q = l15, g) holds Q[frame qt*16 + l15][ks*32 + g*8 .. +8], zero beyond dh / QF
  f16x8 fq[NT][D32];
  long qrow[NT];
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int il = qt * 16 + l15, i = q0 + il;
    qrow[qt] = il < QF ? (a.q_parts > 1 ? ((long)(i / fpp) * a.batch + b) * fpp * a.npix + (long)(i % fpp) * a.npix + p : ((long)b * QF + il) * a.npix + p) : -1;
#pragma unroll
    for (int ks = 0; ks < D32; ++ks) {
      const int d = ks * 32 + g * 8;
      U128 u;
      u.u = (qrow[qt] >= 0 && d < DH) ? ldg128(Q + qrow[qt] * a.ldq + col0 + lcol + d) : make_uint4(0u, 0u, 0u, 0u);
      fq[qt][ks] = u.h;
    }
  }
  __syncthreads();

  // S^T[key tile kt][query tile qt]
  f32x4 s[NT][NT];
#pragma unroll
  for (int qt = 0; qt < NT; ++qt)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < D32; ++ks)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f16x8 fk = *reinterpret_cast<const f16x8*>(sK + (kt * 16 + l15) * SLDB + (lcol + ks * 32 + g * 8) * 2);
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) s[qt][kt] = mfma16(fk, fq[qt][ks], s[qt][kt]);
    }

  // causal softmax per query (= per lane column; the 4 lane groups g hold different keys of it)
  const float c = a.scale * 1.4426950408889634f;
  f16x8 pf[NT][KK];
  float linv[NT];
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int i = q0 + qt * 16 + l15;
    float mx = -1.0e30f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[qt][kt][r] = (kt * 16 + g * 4 + r <= i) ? s[qt][kt][r] * c : -1.0e30f;   // the reference adds -10000 above the diagonal
        mx = fmaxf(mx, s[qt][kt][r]);
      }
    mx = xor32_max(xor16_max(mx));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r] - mx);
        l += s[qt][kt][r];
      }
    linv[qt] = 1.0f / xor32_sum(xor16_sum(l));
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {   // operand B k-slot (g, j) = key kk*32 + (j>>2)*16 + g*4 + (j&3)
      union { f16x2 h[4]; f16x8 v; } f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        f.h[h2] = __builtin_convertvector((f32x2){s[qt][2 * kk][2 * h2], s[qt][2 * kk][2 * h2 + 1]}, f16x2);
        f.h[2 + h2] = __builtin_convertvector((f32x2){s[qt][2 * kk + 1][2 * h2], s[qt][2 * kk + 1][2 * h2 + 1]}, f16x2);
      }
      pf[qt][kk] = f.v;
    }
  }

  // O^T[d tile dt][query tile qt] = V^T P^T; lane (c = l15, g) of a transposed read addresses row key0 + c/4, byte (c%4)*8 of a
  // 16-column block and receives V[key0 .. key0+3][col + c]
  const unsigned vlane = (unsigned)(size_t)(sV + (g * 4 + (l15 >> 2)) * SLDB + (l15 & 3) * 8 + lcol * 2);
  t_static_for(std::make_integer_sequence<int, DT>{}, [&](auto dt_c) {
    constexpr int dt = decltype(dt_c)::value;
    f32x4 o[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) o[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    t_static_for(std::make_integer_sequence<int, KK>{}, [&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      union { uint2 u[2]; f16x8 v; } fv;
      fv.u[0] = t_lds_tr16<(kk * 32) * SLDB + dt * 32>(vlane);
      fv.u[1] = t_lds_tr16<(kk * 32 + 16) * SLDB + dt * 32>(vlane);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv.u[0]), "+v"(fv.u[1]));
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) o[qt] = mfma16(fv.v, pf[qt][kk], o[qt]);
    });
    const int d = dt * 16 + g * 4;
    if (d < DH) {
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        if (qrow[qt] < 0) continue;
        U64 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(o[qt][r] * linv[qt]);
        *reinterpret_cast<uint2*>(O + qrow[qt] * a.ldo + col0 + lcol + d) = ov.u;
      }
    }
  });
}

template <int DH, int KP>
int launch_tattn_mfma(const me_tattn_args* a, hipStream_t st) {
  const int nslice = (a->heads * a->dh) / SLICE;
  const long blocks = (long)a->batch * a->npix * nslice;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL((tattn_mfma_kernel<DH, KP>), dim3((unsigned)blocks), dim3(64 * (SLICE / DH)), 0, st, *a);
  return hipGetLastError() == hipSuccess ? ME_OK : ME_EHIP;
}

template <int F>
int launch_tattn(const me_tattn_args* a, hipStream_t st) {
  const int nslice = (a->heads * a->dh) / SLICE;
  const int hps = SLICE / a->dh;
  const int threads = ((hps * (a->q_frames > 0 ? a->q_frames : F) + 63) / 64) * 64;
  const long blocks = (long)a->batch * a->npix * nslice;
  const size_t lds = (size_t)2 * F * SLD * sizeof(f16);
  static bool attr_set_dev[64] = {};   // per device: a process that drives several GPUs sets the attribute on each
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set && lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ME_EHIP;
    attr_set = true;
  }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(tattn_kernel<F>, dim3((unsigned)blocks), dim3(threads), lds, st, *a);
  return hipGetLastError() == hipSuccess ? ME_OK : ME_EHIP;
}

}  // namespace

extern "C" void me_set_error(const char* msg);

extern "C" int me_tattn(const me_tattn_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->V || !a->O) { me_set_error("me_tattn: null pointer"); return ME_EINVAL; }
  if (a->batch <= 0 || a->batch > 8 || a->npix <= 0 || a->heads <= 0 || a->dh <= 0 || a->dh % 8) { me_set_error("me_tattn: bad sizes"); return ME_EINVAL; }
  if (SLICE % a->dh || (a->heads * a->dh) % SLICE) { me_set_error("me_tattn: head dim must divide 320 and heads*dh be a multiple of 320"); return ME_EINVAL; }
  if (a->ldq % 8 || a->ldk % 8 || a->ldv % 8 || a->ldo % 8) { me_set_error("me_tattn: row strides must be multiples of 8"); return ME_EINVAL; }
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) & 15) { me_set_error("me_tattn: misaligned pointer"); return ME_EINVAL; }
  if (a->q_frames < 0 || a->q_frame0 < 0 || (a->q_frames > 0 && a->q_frame0 + a->q_frames > a->frames) ||
      (a->kv_parts > 1 && a->frames % a->kv_parts) || (a->q_parts > 1 && (a->q_parts != a->kv_parts || a->q_frames > 0))) { me_set_error("me_tattn: bad frame-shard geometry"); return ME_EINVAL; }
  for (int b = 0; b < a->batch; ++b)
    if (a->kv_map[b] < 0 || a->kv_map[b] >= a->batch) { me_set_error("me_tattn: kv_map out of range"); return ME_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  static const bool mfma_on = !(getenv("ME_TATTN_MFMA") && atoi(getenv("ME_TATTN_MFMA")) == 0);   // A/B: the per-thread kernel
  if (mfma_on && a->frames <= 64 && a->ldo % 4 == 0 && (a->dh == 40 || a->dh == 80 || a->dh == 160)) {
    const bool big = a->frames > 32;
    switch (a->dh) {
      case 40: rc = big ? launch_tattn_mfma<40, 64>(a, st) : launch_tattn_mfma<40, 32>(a, st); break;
      case 80: rc = big ? launch_tattn_mfma<80, 64>(a, st) : launch_tattn_mfma<80, 32>(a, st); break;
      default: rc = big ? launch_tattn_mfma<160, 64>(a, st) : launch_tattn_mfma<160, 32>(a, st); break;
    }
    if (rc != ME_OK) me_set_error("me_tattn: kernel launch failed");
    return rc;
  }
  switch (a->frames) {
    case 8: rc = launch_tattn<8>(a, st); break;
    case 16: rc = launch_tattn<16>(a, st); break;
    case 24: rc = launch_tattn<24>(a, st); break;
    case 32: rc = launch_tattn<32>(a, st); break;
    case 40: rc = launch_tattn<40>(a, st); break;
    case 48: rc = launch_tattn<48>(a, st); break;
    default: me_set_error("me_tattn: frames must be one of 8,16,24,32,40,48"); return ME_EINVAL;
  }
  if (rc != ME_OK) me_set_error("me_tattn: kernel launch failed");
  return rc;
}
