// Temporal (per-pixel, over frames) causal self-attention.  Rows are (b*F + frame)*npix + pixel, so
// the reference's "(b f) d c -> (b d) f c" rearrange (attention_2d.py:535,545) is pure indexing here.
// The F x F score matrix of one (pixel, head) is tiny (F <= 48): one thread owns one query frame,
// scores live in registers, K/V rows are shared by the F threads of a (pixel, head) through L1.
// HBM-bound: bytes = 4 tensors x rows x C x 2.
#include "me_common.h"
#include "../../include/motioned.h"

namespace {

template <int F>
__global__ __launch_bounds__(256) void tattn_kernel(const me_tattn_args a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.batch * a.npix * a.heads * F;
  if (idx >= total) return;
  const int i = (int)(idx % F);
  long r = idx / F;
  const int hd = (int)(r % a.heads);
  r /= a.heads;
  const int p = (int)(r % a.npix);
  const int b = (int)(r / a.npix);
  const int kb = a.kv_map[b];

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(a.Q);
  const f16* __restrict__ K = reinterpret_cast<const f16*>(a.K);
  const f16* __restrict__ V = reinterpret_cast<const f16*>(a.V);
  f16* __restrict__ O = reinterpret_cast<f16*>(a.O);

  const int col = hd * a.dh;
  const long qrow = ((long)b * F + i) * a.npix + p;
  const long krow0 = (long)kb * F * a.npix + p;  // + j * npix
  const int nch = a.dh / 8;

  float s[F];
#pragma unroll
  for (int j = 0; j < F; ++j) s[j] = 0.f;

  for (int cc = 0; cc < nch; ++cc) {
    U128 q;
    q.u = ldg128(Q + qrow * a.ldq + col + cc * 8);
    const f16x2* q2 = reinterpret_cast<const f16x2*>(&q);
#pragma unroll
    for (int j = 0; j < F; ++j) {
      U128 k;
      k.u = ldg128(K + (krow0 + (long)j * a.npix) * a.ldk + col + cc * 8);
      const f16x2* k2 = reinterpret_cast<const f16x2*>(&k);
      float acc = s[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2(q2[e], k2[e], acc, false);
      s[j] = acc;
    }
  }

  // causal softmax over j <= i (exp2 with folded log2 e)
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
      v.u = ldg128(V + (krow0 + (long)j * a.npix) * a.ldv + col + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += s[j] * (float)v.e[e];
    }
    U128 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov.e[e] = (f16)(o[e] * inv);
    *reinterpret_cast<uint4*>(O + qrow * a.ldo + col + cc * 8) = ov.u;
  }
}

template <int F>
int launch_tattn(const me_tattn_args* a, hipStream_t st) {
  const long total = (long)a->batch * a->npix * a->heads * F;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(tattn_kernel<F>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a);
  return hipGetLastError() == hipSuccess ? ME_OK : ME_EHIP;
}

}  // namespace

extern "C" void me_set_error(const char* msg);

extern "C" int me_tattn(const me_tattn_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->V || !a->O) { me_set_error("me_tattn: null pointer"); return ME_EINVAL; }
  if (a->batch <= 0 || a->batch > 8 || a->npix <= 0 || a->heads <= 0 || a->dh <= 0 || a->dh % 8) { me_set_error("me_tattn: bad sizes"); return ME_EINVAL; }
  if (a->ldq % 8 || a->ldk % 8 || a->ldv % 8 || a->ldo % 8) { me_set_error("me_tattn: row strides must be multiples of 8"); return ME_EINVAL; }
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) & 15) { me_set_error("me_tattn: misaligned pointer"); return ME_EINVAL; }
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
