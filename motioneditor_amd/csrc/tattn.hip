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
