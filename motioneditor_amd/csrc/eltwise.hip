// Small element-wise / layout kernels of libmotioned (gfx950): direct 3x3 conv for tiny C_in,
// strided adds / copies, SiLU / ReLU, timestep embedding, fused CFG + DDIM update, layout converts.
#include "me_common.h"
#include "../../include/motioned.h"

namespace {

// Direct 3x3 conv for C_in <= 8.  One thread = one output pixel; blockIdx.y picks a slice of output
// channels.  The 9*C_in inputs sit in registers; weights/bias are fp32 and indexed only by wave-uniform
// values, so they arrive through the scalar cache (s_load) and feed v_fma as SGPR operands.
template <int CIN>
__global__ __launch_bounds__(256) void conv_small_kernel(const me_conv_small_args a, int co_per_block) {
  const long npix = (long)a.n_img * a.H * a.Wd;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int x = (int)(pix % a.Wd);
  const int y = (int)((pix / a.Wd) % a.H);
  const int img = (int)(pix / ((long)a.Wd * a.H));
  long base;
  if (a.frames > 0) base = (long)(img / a.frames) * a.img_stride + (long)(img % a.frames) * a.frame_stride;
  else base = (long)img * a.img_stride;

  float in[9][CIN];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const long off = base + (long)ci * a.ch_stride + (long)iy * a.Wd + ix;
      float v = 0.f;
      if (ok) v = a.in_is_f16 ? (float)reinterpret_cast<const f16*>(a.in)[off] : reinterpret_cast<const float*>(a.in)[off];
      in[tap][ci] = v;
    }
  }
  const float* __restrict__ W = reinterpret_cast<const float*>(a.W);
  const float* __restrict__ bias = reinterpret_cast<const float*>(a.bias);
  f16* out = reinterpret_cast<f16*>(a.out) + pix * a.Cout;
  const int c0 = blockIdx.y * co_per_block;
  for (int cb = c0; cb < c0 + co_per_block && cb < a.Cout; cb += 8) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[cb + e] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* w = W + (long)(cb + e) * 9 * CIN;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[e] = __builtin_fmaf(in[tap][ci], w[tap * CIN + ci], acc[e]);
    }
    U128 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = (f16)(a.silu ? silu_f(acc[e]) : acc[e]);
    *reinterpret_cast<uint4*>(out + cb) = o.u;
  }
}

// The same convolution for wide outputs (conv_in 4 -> 320, the VAE encoder's 3 -> 128): the kernel above has every lane store 16
// bytes into a different 640-byte row (0.43 ms for a 251 MB output).  Here a block owns 64 consecutive pixels x ALL output
// channels: wave w computes channel quarter w for the block's 64 pixels (weights still wave-uniform -> scalar operands), parks
// its results in an LDS tile [64 pixels][C_out] and the block then writes the tile -- one contiguous run of 64 * C_out halves in
// global memory -- with 16-byte row-contiguous stores.
template <int CIN>
__global__ __launch_bounds__(256) void conv_small_tile_kernel(const me_conv_small_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_cs[];
  f16* sOut = reinterpret_cast<f16*>(smem_cs);
  const int OLD = a.Cout + 4;   // halves per pixel row of the tile (+8 bytes: 16-byte stores of consecutive pixels land on distinct banks)
  const long npix = (long)a.n_img * a.H * a.Wd;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pix0 = (long)blockIdx.x * 64;
  const long pix = pix0 + lane;
  if (pix < npix) {
    const int x = (int)(pix % a.Wd);
    const int y = (int)((pix / a.Wd) % a.H);
    const int img = (int)(pix / ((long)a.Wd * a.H));
    long base;
    if (a.frames > 0) base = (long)(img / a.frames) * a.img_stride + (long)(img % a.frames) * a.frame_stride;
    else base = (long)img * a.img_stride;
    float in[9][CIN];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const long off = base + (long)ci * a.ch_stride + (long)iy * a.Wd + ix;
        float v = 0.f;
        if (ok) v = a.in_is_f16 ? (float)reinterpret_cast<const f16*>(a.in)[off] : reinterpret_cast<const float*>(a.in)[off];
        in[tap][ci] = v;
      }
    }
    const float* __restrict__ W = reinterpret_cast<const float*>(a.W);
    const float* __restrict__ bias = reinterpret_cast<const float*>(a.bias);
    const int cq = a.Cout / 4, c0 = wave * cq;
    for (int cb = c0; cb < c0 + cq; cb += 8) {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[cb + e] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* w = W + (long)(cb + e) * 9 * CIN;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) acc[e] = __builtin_fmaf(in[tap][ci], w[tap * CIN + ci], acc[e]);
      }
      U128 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (f16)(a.silu ? silu_f(acc[e]) : acc[e]);
      *reinterpret_cast<uint2*>(sOut + lane * OLD + cb) = make_uint2(o.u.x, o.u.y);
      *reinterpret_cast<uint2*>(sOut + lane * OLD + cb + 4) = make_uint2(o.u.z, o.u.w);
    }
  }
  __syncthreads();
  f16* out = reinterpret_cast<f16*>(a.out) + pix0 * a.Cout;
  const int vpr = a.Cout / 8;
  const int rows = (int)(npix - pix0 < 64 ? npix - pix0 : 64);
  for (int idx = threadIdx.x; idx < rows * vpr; idx += 256) {
    const int r = idx / vpr, c8 = (idx - r * vpr) * 8;
    const uint2 lo = *reinterpret_cast<const uint2*>(sOut + r * OLD + c8), hi = *reinterpret_cast<const uint2*>(sOut + r * OLD + c8 + 4);
    *reinterpret_cast<uint4*>(out + (long)r * a.Cout + c8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

// Y = X + alpha * A over a [rows, cols] view, 4 halves per thread
__global__ __launch_bounds__(256) void axpy_rows_kernel(f16* Y, int ldy, const f16* X, int ldx, const f16* A, int lda, long rows, int cols, float alpha) {
  const int vpr = cols / 4;
  const long n = rows * vpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const long r = idx / vpr;
    const int c = (int)(idx - r * vpr) * 4;
    U64 x, av, o;
    x.u = *reinterpret_cast<const uint2*>(X + r * ldx + c);
    av.u = *reinterpret_cast<const uint2*>(A + r * lda + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) o.e[e] = (f16)((float)x.e[e] + alpha * (float)av.e[e]);
    *reinterpret_cast<uint2*>(Y + r * ldy + c) = o.u;
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(f16* Y, int ldy, const f16* X, int ldx, long rows, int cols) {
  const int vpr = cols / 8;
  const long n = rows * vpr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const long r = idx / vpr;
    const int c = (int)(idx - r * vpr) * 8;
    *reinterpret_cast<uint4*>(Y + r * ldy + c) = ldg128(X + r * ldx + c);
  }
}

__global__ __launch_bounds__(256) void copy_blocks_kernel(f16* Y, int ldy, const f16* X, int ldx, int n1, long rows, int cols, long total, long ys0, long ys1,
                                                          long xs0, long xs1) {
  const int vpr = cols / 8;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    long r = idx / vpr;
    const int c = (int)(idx - r * vpr) * 8;
    const long blk = r / rows;
    r -= blk * rows;
    const long i = blk / n1, j = blk - i * n1;
    *reinterpret_cast<uint4*>(Y + (i * ys0 + j * ys1 + r) * ldy + c) = ldg128(X + (i * xs0 + j * xs1 + r) * ldx + c);
  }
}

template <int OP>  // 0 silu, 1 relu
__global__ __launch_bounds__(256) void unary_kernel(f16* Y, const f16* X, long n) {
  for (long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 8; idx < n; idx += (long)gridDim.x * 256 * 8) {
    if (idx + 8 <= n) {
      U128 u, o;
      u.u = ldg128(X + idx);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (float)u.e[e];
        o.e[e] = (f16)(OP == 0 ? silu_f(v) : fmaxf(v, 0.f));
      }
      *reinterpret_cast<uint4*>(Y + idx) = o.u;
    } else {
      for (long k = idx; k < n; ++k) {
        const float v = (float)X[k];
        Y[k] = (f16)(OP == 0 ? silu_f(v) : fmaxf(v, 0.f));
      }
    }
  }
}

// tp != nullptr: the timestep is read from device memory (hipGraph replay: the captured launch stays valid across steps)
__global__ void timestep_embed_kernel(f16* out, int rows, int dim, float t, const float* __restrict__ tp) {
  if (tp) t = tp[0];
  const int half = dim / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * dim) return;
  const int d = idx % dim;
  const int k = d < half ? d : d - half;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
  const float ang = t * freq;
  out[idx] = (f16)(d < half ? cosf(ang) : sinf(ang));
}

__global__ __launch_bounds__(256) void cfg_ddim_kernel(float* lat_out, const float* lat_in, const f16* eps, int lde, int nb, int C, int frames,
                                                       int npix, float guidance, float ca, float cb, const float* __restrict__ pp) {
  if (pp) {   // device-resident step parameters [t, guidance, ca, cb]
    guidance = pp[1];
    ca = pp[2];
    cb = pp[3];
  }
  const long total = (long)nb * C * frames * npix;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % npix);
  long r = idx / npix;
  const int f = (int)(r % frames);
  r /= frames;
  const int c = (int)(r % C);
  const int b = (int)(r / C);
  const long row_u = ((long)b * frames + f) * npix + p;
  const long row_c = ((long)(b + nb) * frames + f) * npix + p;
  const float eu = (float)eps[row_u * lde + c];
  const float ec = (float)eps[row_c * lde + c];
  const float e = eu + guidance * (ec - eu);
  lat_out[idx] = ca * lat_in[idx] + cb * e;
}

__global__ __launch_bounds__(256) void gaussian_sample_kernel(float* out, const f16* __restrict__ mom, int ldm, const float* __restrict__ noise, int n_img, int npix, float scale) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // (img, c, p)
  if (idx >= (long)n_img * 4 * npix) return;
  const int p = (int)(idx % npix);
  const int c = (int)((idx / npix) % 4);
  const long row = (idx / (4L * npix)) * npix + p;
  const float mean = (float)mom[row * ldm + c];
  const float logvar = fminf(fmaxf((float)mom[row * ldm + 4 + c], -30.f), 20.f);
  out[idx] = (mean + __expf(0.5f * logvar) * noise[idx]) * scale;
}

__global__ __launch_bounds__(256) void nchw_to_rows_kernel(f16* Y, int ldy, const float* X, long img_stride, long ch_stride, int n_img, int C, int npix) {
  const long total = (long)n_img * npix * C;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long row = idx / C;
  const int p = (int)(row % npix);
  const int img = (int)(row / npix);
  Y[row * ldy + c] = (f16)X[(long)img * img_stride + (long)c * ch_stride + p];
}

__global__ __launch_bounds__(256) void rows_to_nchw_kernel(float* Y, long img_stride, long ch_stride, const f16* X, int ldx, int n_img, int C, int npix) {
  const long total = (long)n_img * npix * C;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % npix);
  long r = idx / npix;
  const int c = (int)(r % C);
  const int img = (int)(r / C);
  Y[(long)img * img_stride + (long)c * ch_stride + p] = (float)X[((long)img * npix + p) * ldx + c];
}

inline unsigned grid_for(long n, long cap = 16384) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace

extern "C" void me_set_error(const char* msg);

extern "C" void me_set_hip_error(const char* what, int err);

#define ME_CHECK_LAUNCH(name)                                         \
  {                                                                   \
    const hipError_t e_ = hipGetLastError();                          \
    if (e_ != hipSuccess) {                                           \
      me_set_hip_error(name ": kernel launch failed", (int)e_);       \
      return ME_EHIP;                                                 \
    }                                                                 \
  }                                                                   \
  return ME_OK;

extern "C" int me_conv_small(const me_conv_small_args* a, void* stream) {
  if (!a || !a->in || !a->W || !a->out) { me_set_error("me_conv_small: null pointer"); return ME_EINVAL; }
  if ((a->Cin != 3 && a->Cin != 4) || a->Cout % 8 || a->n_img <= 0 || a->H <= 0 || a->Wd <= 0) { me_set_error("me_conv_small: C_in must be 3 or 4, C_out a multiple of 8"); return ME_EINVAL; }
  const long npix = (long)a->n_img * a->H * a->Wd;
  if (a->Cout >= 64 && a->Cout % 32 == 0 && a->Cout <= 480) {   // wide outputs: 64 pixels x all channels per block, row-contiguous stores through LDS
    const size_t lds = (size_t)64 * (a->Cout + 4) * sizeof(f16);
    (void)hipGetLastError();
    if (a->Cin == 3) hipLaunchKernelGGL(conv_small_tile_kernel<3>, dim3((unsigned)((npix + 63) / 64)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), *a);
    else hipLaunchKernelGGL(conv_small_tile_kernel<4>, dim3((unsigned)((npix + 63) / 64)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), *a);
    ME_CHECK_LAUNCH("me_conv_small")
  }
  const int co_per_block = a->Cout > 80 ? 80 : a->Cout;   // wide outputs: slices of 80 channels per block row
  const dim3 grid((unsigned)((npix + 255) / 256), (unsigned)((a->Cout + co_per_block - 1) / co_per_block));
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  if (a->Cin == 3) hipLaunchKernelGGL(conv_small_kernel<3>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a, co_per_block);
  else hipLaunchKernelGGL(conv_small_kernel<4>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a, co_per_block);
  ME_CHECK_LAUNCH("me_conv_small")
}

extern "C" int me_axpy_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, const void* A, int32_t lda, int64_t rows, int32_t cols, float alpha,
                            void* stream) {
  if (!Y || !X || !A || rows <= 0 || cols <= 0 || cols % 4 || ldy % 4 || ldx % 4 || lda % 4) { me_set_error("me_axpy_rows: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(axpy_rows_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y), ldy,
                     reinterpret_cast<const f16*>(X), ldx, reinterpret_cast<const f16*>(A), lda, (long)rows, cols, alpha);
  ME_CHECK_LAUNCH("me_axpy_rows")
}

extern "C" int me_copy_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, int64_t rows, int32_t cols, void* stream) {
  if (!Y || !X || rows <= 0 || cols <= 0 || cols % 8 || ldy % 8 || ldx % 8) { me_set_error("me_copy_rows: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y), ldy,
                     reinterpret_cast<const f16*>(X), ldx, (long)rows, cols);
  ME_CHECK_LAUNCH("me_copy_rows")
}

extern "C" int me_copy_blocks(void* Y, int32_t ldy, const void* X, int32_t ldx, int32_t n0, int32_t n1, int64_t rows, int32_t cols, int64_t ys0, int64_t ys1,
                              int64_t xs0, int64_t xs1, void* stream) {
  if (!Y || !X || n0 <= 0 || n1 <= 0 || rows <= 0 || cols <= 0 || cols % 8 || ldy % 8 || ldx % 8 || ys0 < 0 || ys1 < 0 || xs0 < 0 || xs1 < 0 ||
      (((uintptr_t)Y | (uintptr_t)X) & 15)) { me_set_error("me_copy_blocks: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  const long total = (long)n0 * n1 * rows * (cols / 8);
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y), ldy,
                     reinterpret_cast<const f16*>(X), ldx, n1, (long)rows, cols, total, (long)ys0, (long)ys1, (long)xs0, (long)xs1);
  ME_CHECK_LAUNCH("me_copy_blocks")
}

extern "C" int me_silu(void* Y, const void* X, int64_t n, void* stream) {
  if (!Y || !X || n <= 0 || (((uintptr_t)Y | (uintptr_t)X) & 15)) { me_set_error("me_silu: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(unary_kernel<0>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y),
                     reinterpret_cast<const f16*>(X), (long)n);
  ME_CHECK_LAUNCH("me_silu")
}

extern "C" int me_relu(void* Y, const void* X, int64_t n, void* stream) {
  if (!Y || !X || n <= 0 || (((uintptr_t)Y | (uintptr_t)X) & 15)) { me_set_error("me_relu: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(unary_kernel<1>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y),
                     reinterpret_cast<const f16*>(X), (long)n);
  ME_CHECK_LAUNCH("me_relu")
}

extern "C" int me_timestep_embed(void* out, int32_t rows, int32_t dim, float t, void* stream) {
  if (!out || rows <= 0 || dim <= 0 || dim % 2) { me_set_error("me_timestep_embed: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((rows * dim + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(out), rows,
                     dim, t, (const float*)nullptr);
  ME_CHECK_LAUNCH("me_timestep_embed")
}

extern "C" int me_timestep_embed_dev(void* out, int32_t rows, int32_t dim, const float* step_params, void* stream) {
  if (!out || !step_params || rows <= 0 || dim <= 0 || dim % 2) { me_set_error("me_timestep_embed_dev: bad arguments"); return ME_EINVAL; }
  (void)hipGetLastError();
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((rows * dim + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(out), rows,
                     dim, 0.f, step_params);
  ME_CHECK_LAUNCH("me_timestep_embed_dev")
}

extern "C" int me_cfg_ddim(float* lat_out, const float* lat_in, const void* eps, int32_t lde, int32_t nb, int32_t C, int32_t frames, int32_t npix,
                           float guidance, float ca, float cb, void* stream) {
  if (!lat_out || !lat_in || !eps || nb <= 0 || C <= 0 || frames <= 0 || npix <= 0 || lde < C) { me_set_error("me_cfg_ddim: bad arguments"); return ME_EINVAL; }
  const long total = (long)nb * C * frames * npix;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), lat_out, lat_in,
                     reinterpret_cast<const f16*>(eps), lde, nb, C, frames, npix, guidance, ca, cb, (const float*)nullptr);
  ME_CHECK_LAUNCH("me_cfg_ddim")
}

extern "C" int me_cfg_ddim_dev(float* lat_out, const float* lat_in, const void* eps, int32_t lde, int32_t nb, int32_t C, int32_t frames, int32_t npix,
                               const float* step_params, void* stream) {
  if (!lat_out || !lat_in || !eps || !step_params || nb <= 0 || C <= 0 || frames <= 0 || npix <= 0 || lde < C) { me_set_error("me_cfg_ddim_dev: bad arguments"); return ME_EINVAL; }
  const long total = (long)nb * C * frames * npix;
  (void)hipGetLastError();
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), lat_out, lat_in,
                     reinterpret_cast<const f16*>(eps), lde, nb, C, frames, npix, 0.f, 0.f, 0.f, step_params);
  ME_CHECK_LAUNCH("me_cfg_ddim_dev")
}

extern "C" int me_gaussian_sample(float* out, const void* moments, int32_t ldm, const float* noise, int32_t n_img, int32_t npix, float scale, void* stream) {
  if (!out || !moments || !noise || n_img <= 0 || npix <= 0 || ldm < 8) { me_set_error("me_gaussian_sample: bad arguments"); return ME_EINVAL; }
  const long total = (long)n_img * 4 * npix;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out,
                     reinterpret_cast<const f16*>(moments), ldm, noise, n_img, npix, scale);
  ME_CHECK_LAUNCH("me_gaussian_sample")
}

extern "C" int me_nchw_to_rows(void* Y, int32_t ldy, const float* X, int64_t img_stride, int64_t ch_stride, int32_t n_img, int32_t C, int32_t npix,
                               void* stream) {
  if (!Y || !X || n_img <= 0 || C <= 0 || npix <= 0 || ldy < C) { me_set_error("me_nchw_to_rows: bad arguments"); return ME_EINVAL; }
  const long total = (long)n_img * npix * C;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(nchw_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<f16*>(Y), ldy, X,
                     (long)img_stride, (long)ch_stride, n_img, C, npix);
  ME_CHECK_LAUNCH("me_nchw_to_rows")
}

extern "C" int me_rows_to_nchw(float* Y, int64_t img_stride, int64_t ch_stride, const void* X, int32_t ldx, int32_t n_img, int32_t C, int32_t npix,
                               void* stream) {
  if (!Y || !X || n_img <= 0 || C <= 0 || npix <= 0 || ldx < C) { me_set_error("me_rows_to_nchw: bad arguments"); return ME_EINVAL; }
  const long total = (long)n_img * npix * C;
  (void)hipGetLastError();  // drop stale errors left by other HIP users in this thread
  hipLaunchKernelGGL(rows_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), Y, (long)img_stride,
                     (long)ch_stride, reinterpret_cast<const f16*>(X), ldx, n_img, C, npix);
  ME_CHECK_LAUNCH("me_rows_to_nchw")
}
