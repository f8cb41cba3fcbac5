/*
 * motioned.h -- C ABI of libmotioned.so, the MI355X (gfx950) kernel library behind the
 * MotionEditor two-branch DDIM denoising step.
 *
 * The reference (Francis-Rings/MotionEditor) has NO FFI: its seams are Python call signatures and
 * every arithmetic op is a torch / xformers / cuDNN call (SURVEY.md layer L1).  Each entry point
 * below replaces one family of those L1 calls; the reference call sites it stands in for are cited
 * per function (paths relative to the reference repo root).  The Python mirror of the reference
 * interface (motioneditor_amd.pipelines / .attn_control / .models) binds these through ctypes --
 * see INTEGRATION.md for the stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C types only; all tensors are raw DEVICE pointers owned by the caller
 *   - activations are fp16 ("u16" storage), token-major / channels-last: row = (batch*frame, pixel),
 *     columns = channels, with an explicit row stride (ld*) in ELEMENTS; 16-byte aligned rows
 *   - weights are fp16, [N_out][taps][C_in] (C_in contiguous) -- torch Linear layout, conv kernels
 *     repacked tap-major by the host packer
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no hidden sync
 *   - return 0 on success, negative ME_E* code on failure; me_last_error() gives the message
 *   - not thread-safe per stream; one process per GPU
 */
#ifndef MOTIONED_H
#define MOTIONED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_OK 0
#define ME_EINVAL (-1) /* bad argument (shape / alignment / unsupported size)          */
#define ME_EHIP (-2)   /* HIP runtime error at launch                                   */

#define ME_ABI_VERSION 9

/* ---- library ------------------------------------------------------------------------------ */
int me_abi_version(void);
const char* me_last_error(void);
/* name of the device kernel the last compute call of this thread launched, e.g. "gemm_kernel<256,320>",
 * "attn2_kernel<40,2,8>" -- the names rocprofv3 reports; used by bench.py's per-kernel roofline */
const char* me_last_kernel(void);
/* number of CUs / XCD-count etc. of the current device; returns ME_EHIP if no gfx950 device */
int me_device_info(int* cus, int* lds_bytes, char* arch, int arch_len);

/* ---- gather-GEMM (implicit convolution) ---------------------------------------------------- *
 * C[m, n] = epilogue( sum_{tap, c} X[src(m, tap), c] * W[n, tap, c] )
 * Replaces: nn.Linear (attention_2d.py:83-92, temporal_attn.py:63-72, controlnet_adapter.py:55-64),
 * InflatedConv3d 3x3 / 1x1 / stride-2 / after nearest-2x (resnet_2d.py:28-36, 39-125),
 * TemporalConv (resnet_2d.py:10-26, controlnet_adapter.py:411-434), GEGLU FeedForward
 * (diffusers FeedForward, used attention_2d.py:427,531), Transformer2DModel proj_in/out
 * (attention_2d.py:307,336), time_emb_proj (resnet_2d.py:172,211).
 */
#define ME_GATHER_DENSE 0  /* taps = 1, src(m) = m                                                  */
#define ME_GATHER_CONV3 1  /* taps = 9, 3x3 pad 1 over [img][Hin][Win] pixels; stride 1|2; ups 0|1|2 */
#define ME_GATHER_TCONV 2  /* taps = 3, k=3 pad 1 over frames inside chunks of `chunk` frames       */

typedef struct me_gemm_args {
  const void* X;      /* fp16 [rows_in, ldx]                                   */
  const void* W;      /* fp16 [N, taps, K]                                     */
  void* C;            /* fp16 [M, ldc]                                         */
  int32_t M, N, K;    /* K = input channels per tap (multiple of 8)            */
  int32_t ldx, ldc;   /* row strides in elements (multiples of 8 / 4)          */
  int32_t gather;     /* ME_GATHER_*                                           */
  /* CONV3: M = n_img * Hout * Wout.  ups = 1: the input is read through a nearest-neighbour 2x upsample (Upsample2D folded into
   * the gather); ups = 2: through a ZERO-STUFFED 2x upsample (virtual pixel (2y, 2x) = input pixel (y, x), every other virtual
   * pixel is zero) -- the input gradient of a stride-2 convolution is the stride-1 correlation of the zero-stuffed output gradient */
  int32_t Hin, Win, Hout, Wout, stride, ups;
  int32_t pad0;       /* CONV3: 0 = padding 1 on every side; 1 = no padding at the top / left, one row / column at the bottom /
                         right (diffusers Downsample2D(padding=0) of the VAE encoder: F.pad(x, (0,1,0,1)) + stride-2 conv) */
  /* TCONV: row m = ((b * frames + fr) * npix + p).  Frame-sharded operation: this rank holds
   * `frames` consecutive frames starting at global frame `frame0` of `frames_total`; taps that leave the local
   * range read the one-frame halos the host appended to X at rows halo_prev / halo_next (+ b*npix + p).
   * frames_total == 0 means unsharded (frame0 = 0, frames_total = frames, no halos). */
  int32_t frames, npix, chunk;
  int32_t frame0, frames_total, halo_prev, halo_next;
  /* epilogue (all optional, applied in this order).  Arithmetic: fp32 accumulation of alpha * (X W^T) + bias (the
   * bias enters as the accumulators' initial value); without an activation the tile is then rounded to fp16 and
   * rowvec / res / res2 are added in fp16 -- the roundings the reference's half-precision `conv(x) + temb`,
   * `attn(x) + x` perform; with act != 0 every term is added in fp32 and rounded once.  res (or res2) may alias C. */
  const void* bias;   /* fp16 [N]                                              */
  const void* rowvec; /* fp16: += rowvec[(m / rows_per_vec) * ldrv + n]        */
  int32_t ldrv, rows_per_vec;
  const void* res;    /* fp16 [M, ldr]: += res[m, n]                           */
  int32_t ldr;
  const void* res2;   /* second residual, fp16 [M, ldr2]                       */
  int32_t ldr2;
  int32_t geglu;      /* 1: W rows interleaved (16 a-rows, 16 gate-rows); C gets N/2 columns a*gelu(g) */
  int32_t act;        /* applied after bias+rowvec, before residuals: 0 none, 1 ReLU, 2 SiLU */
  float alpha;        /* scale applied to the accumulator before the epilogue adds */
  int32_t res_rows;   /* > 0: res holds res_rows rows and output row m reads row m % res_rows (a residual shared by several batch entries,
                         e.g. the adapter's ControlNet-only half broadcast over the edit rows); 0: row m reads row m */
  int32_t res2_rows;  /* the same for res2 */
  /* Optional split-K scratch (ABI 5).  Grids too small to fill the chip (the 8 x 8-latent level, batch-1 backward passes) are split along K:
   * every split stores fp32 partial sums into `work`, a second kernel adds them in a fixed order and applies the epilogue.  work: device
   * memory of at least me_gemm_work_bytes(a) bytes, 16-byte aligned, or NULL (never split).  splits_ is set by me_gemm itself. */
  void* work;
  int64_t work_bytes;
  int32_t splits_;
  int32_t sel_rows;   /* > M: choose between the LDS-halo convolution kernel and the gather kernels as a launch of sel_rows rows would (they add the
                         (tap, channel slab) products in different orders): a caller that computes a sub-batch of a launch once gets bitwise the rows
                         of the full launch.  0: decide on M */
  /* ABI 6: HEAD-MAJOR second output (C2 != NULL).  Output columns n >= c2_col0 are not written to C but to C2 as [head][M][c2_dh]:
   *     C2[((n - c2_col0) / c2_dh) * c2_hs + m * c2_dh + (n - c2_col0) % c2_dh]
   * -- the fused q|k|v projection of a self-attention (attention_2d.py:705-768) hands K and V to me_attn as contiguous [keys][dh] panels per
   * head (me_attn_args.hsk / hsv) instead of dh-wide column slices of 3C-wide rows: the attention's K/V tile fill touches 2.5 x fewer cache
   * lines, and the projection's own stores land 80 ... 320 bytes apart instead of 1920 ... 7680.  Requires: no geglu / act / rowvec / res /
   * res2, c2_dh % 8 == 0, c2_col0 % 16 == 0, (N - c2_col0) % c2_dh == 0, c2_hs % 8 == 0, C2 16-byte aligned.  c2_col0 = 0 (round 5): EVERY column leaves
   * as panels -- q | k | v all head-major (me_attn_args.hsq) -- and C is never written (it must still be a valid, aligned address). */
  void* C2;
  int32_t c2_col0, c2_dh;
  int64_t c2_hs;      /* elements between the panels of consecutive heads (>= M * c2_dh) */
  /* ABI 6: row range.  m_off > 0: the launch computes output rows [m_off, M) only (row indices, gather geometry, residual and rowvec rows all stay
   * absolute).  A frame-sharded TemporalConv (resnet_2d.py:18-26 over a rank's frames) is issued as an interior launch -- frames that need no remote
   * data -- behind the posted halo exchange, and boundary launches after it.  DENSE / TCONV only; such a launch is never split along K. */
  int32_t m_off;
  /* ABI 9: LayerNorm folded into the projection that consumes it (BasicTransformerBlock, attention_2d.py:493-547: norm1 -> attn1.to_q|k|v, norm2 -> attn2.to_q,
   * norm3 -> ff.net.0.proj, norm_temp -> attn_temp.to_q|k|v; the adapter's norm_temp / ff_norm / norm_self_temp, controlnet_adapter.py:497-534).
   *   LN(x) W^T + b  =  rstd * (x W'^T - mean * colsum(W')) + (W beta + b),   W' = W diag(gamma)
   * so the normalised tensor never exists: X holds the UN-normalised rows, W = W' (packed once, fp16), and with ln_stats != NULL every kernel of me_gemm
   * maps its fp32 accumulators through  rstd[m] * (acc - mean[m] * ln_colsum[n]) + ln_cvec[n]  before its usual epilogue (GEGLU included).
   * ln_stats: fp32 partial row sums [ln_parts][...][2] = (sum x, sum x^2) of row m over the columns [320 p, 320 p + 320) of X -- the format the epilogue of
   * the PRODUCING projection writes (ln_out below) or me_ln_stats computes; part p of row m at ln_stats[p * ln_stride + 2 m].  mean = S1 / K,
   * rstd = rsqrt(max(S2 / K - mean^2, 0) + ln_eps).  ln_colsum / ln_cvec: fp32 [N].  Requires DENSE, bias == NULL (it is inside ln_cvec), no rowvec / res / res2 / act, alpha == 1;
   * such a launch is never split along K. */
  const void* ln_stats;
  const void* ln_colsum;
  const void* ln_cvec;
  int64_t ln_stride;  /* floats between consecutive parts of ln_stats (>= 2 * rows of X) */
  int32_t ln_parts;   /* K / 320 for the model's widths (1, 2, 4); 1 when K is not a multiple of 320 */
  float ln_eps;
  /* ... and the producer side: ln_out != NULL asks for the partial row sums (sum y, sum y^2) of THIS launch's fp16 output rows, part p = columns
   * [320 p, 320 p + 320) at ln_out[p * ln_out_stride + 2 m] -- from the row-contiguous epilogue of the 8-phase kernels where the launch takes them
   * (fixed order: 8-column pieces, then the wave's 10 pieces, then the 4 waves of a row), from a read-only pass over C behind the launch otherwise.
   * Requires N % 8 == 0, no GEGLU / C2. */
  void* ln_out;
  int64_t ln_out_stride;
} me_gemm_args;

int me_gemm(const me_gemm_args* a, void* stream);
/* ABI 9: partial row sums of X [rows, C] (fp16, row stride ldx) in the ln_stats format above: stats[p * stride + 2 m] = (sum, sum of squares) of row m over
 * columns [320 p, 320 p + 320), p < C / 320 (one part over the whole row when C is not a multiple of 320).  C % 8 == 0, C <= 1536. */
int me_ln_stats(const void* X, int32_t ldx, int64_t rows, int32_t C, void* stats, int64_t stride, void* stream);
/* bytes of `work` that me_gemm may use for these arguments (0: the launch is never split) */
int64_t me_gemm_work_bytes(const me_gemm_args* a);

/* ---- direct convolution for tiny channel counts (C_in < 8) --------------------------------- *
 * conv_in of the UNet / ControlNet on fp32 latents in the REFERENCE layout, and the first conv of
 * controlnet_cond_embedding.  Replaces InflatedConv3d conv_in (unet_2d_condition.py:160,451).
 * in : fp32, element (img, c, y, x) at  in[img*img_stride + c*ch_stride + y*Win + x]
 * out: fp16 channels-last [n_img*H*W, Cout]; W, bias fp32, W [Cout][9][Cin] (read through the scalar
 * cache); C_in in {3, 4}; 3x3 pad 1 stride 1.
 */
typedef struct me_conv_small_args {
  const void* in;
  const void* W;
  const void* bias;
  void* out;
  int32_t n_img, Cin, Cout, H, Wd;
  int64_t img_stride, ch_stride;
  int32_t in_is_f16; /* 0: fp32 input, 1: fp16 input */
  int32_t silu;      /* apply SiLU to the output */
  int32_t frames;    /* >0: img = b*frames + f and in offset = b*img_stride + f*frame_stride (5-D latents) */
  int64_t frame_stride;
} me_conv_small_args;

int me_conv_small(const me_conv_small_args* a, void* stream);

/* ---- fused attention ------------------------------------------------------------------------ *
 * O[item, q, h, :] = softmax_over_all_segments( Q.K^T * scale ) . V, never materialising scores
 * or gathered keys.  Replaces xformers.ops.memory_efficient_attention and baddbmm/softmax/bmm
 * (attention_2d.py:172-201,246-253, fully_control_utils.py:48-66,191-203, controlnet_adapter.py:144-225)
 * AND the spatial editor's masked 5N-key attention (fully_control.py:372-460) via seg modes.
 */
#define ME_SEG_PLAIN 0     /* weight exp(s)                                                            */
#define ME_SEG_DUAL_CUR 1  /* weight exp(m*s) + exp((1-m)*s), m = mask[head][key]                      */
#define ME_SEG_DUAL_PREV 2 /* same with m = mask[max(head-1,0)][key]                                   */
#define ME_SEG_DUAL_BIN 3  /* DUAL with a BINARY mask: weight exp(s) + 1 for either mask value (no mask read) */

typedef struct me_attn_args {
  const void* Q;  /* fp16 rows (item*nq + q), cols head*dh + d */
  const void* K;  /* fp16 rows (kv_item*nk + key)               */
  const void* V;
  void* O;
  int32_t ldq, ldk, ldv, ldo;
  int32_t heads, dh; /* dh in {40, 80, 160} */
  int32_t n_items, nq, nk, nseg; /* nseg in 1..3 */
  const int32_t* seg_item; /* device int32 [n_items][nseg]: kv item index of each segment; a negative
                              entry ends the item's segment list (skipped segments must come last) */
  const int32_t* seg_mode; /* device int32 [n_items][nseg]: ME_SEG_*                       */
  const void* mask;        /* fp16 [8][nk] mask planes (only for DUAL_CUR / DUAL_PREV)     */
  float scale;
  int32_t general_dual;    /* 1 when any seg_mode is DUAL_CUR / DUAL_PREV (selects the kernel built with that path) */
  /* Required when any seg_mode is ME_SEG_DUAL_BIN: device scratch of me_attn_vsum_bytes(n_kv_items, heads*dh) bytes, 16-byte
   * aligned.  me_attn fills its head, fp32 [n_kv_items][heads*dh], with the per-kv-item column sums of V (the
   * query-independent "+1" part of every binary dual key, fully_control.py:381-413) on `stream` before the attention kernel,
   * which adds it in its epilogue; the rest holds the partials of the fixed-order reduction.  NULL when no segment is DUAL_BIN. */
  void* vsum;
  int32_t n_kv_items;      /* kv items in K / V (rows / nk); only read with vsum */
  int32_t q_items;         /* > 0: Q holds q_items query items and item i reads item i % q_items (queries shared by several batch entries:
                              the adapter's pose queries, controlnet_adapter.py:519-523, broadcast over the edit rows); 0: item i reads item i */
  void* lse;               /* optional fp32 [n_items * nq][heads]: log2 of the softmax denominator in exp2 units, log2(sum_j 2^(s_j scale log2 e)),
                              stashed for me_attn_bwd (plain segments only; NULL = not written) */
  /* ABI 6: head-major K / V (me_gemm_args.C2).  hsk / hsv > 0: element (row, head, d) of K / V lies at row * ldk + head * hsk + d (ldk = dh for
   * contiguous per-head panels); 0: at row * ldk + head * dh + d (heads are column slices of the rows, as Q and O always are). */
  int64_t hsk, hsv;
  /* ABI 8: head-major Q.  hsq > 0: element (row, head, d) of Q lies at row * ldq + head * hsq + d (the fused q|k|v projection writes all three as
   * per-head [rows, dh] panels, me_gemm_args.c2_col0 = 0); 0: at row * ldq + head * dh + d.  With the heads-slowest block order of the multi-segment
   * launches (each XCD owns one head) a head's 80-byte slices of 640-byte Q rows cost 2.4 x their bytes in cache lines per XCD; panels cost 1 x.
   * O stays a row tensor (the out-projection reads it as its A operand).  Served by every kernel of me_attn (round 6: the general-dual one too). */
  int64_t hsq;
  /* ABI 8: optional processing order of the query items inside a head's run of the heads-slowest block order: device int32 [n_items], a permutation; the
   * k-th item a head's blocks work on is item_order[k].  NULL: ascending.  Scheduling only -- the output is bitwise the same.  The edited launches pass
   * (recon frame g, edit frame g, recon frame g + 1, ...): an edit item reads its source's K | V right after the reconstruction item did, while they
   * are still in the XCD's L2 (ascending order puts two dozen items between the two). */
  const int32_t* item_order;
} me_attn_args;

int me_attn(const me_attn_args* a, void* stream);
int64_t me_attn_vsum_bytes(int32_t n_kv_items, int32_t channels);
/* Diagnostic of the fixed-offset softmax (the dh = 40 / 80 kernels for segments of >= 256 keys): the number of thread blocks, on the current
 * device since the last reset, whose speculative pass met a probability beyond fp16's range and that therefore re-ran with the classic
 * running maximum (correct either way; each such block costs about twice).  Synchronises with the device.  reset != 0 zeroes the counter.
 * Returns -1 on a HIP error.  With trained checkpoints this says whether the per-stage re-basing keeps the fast path. */
int64_t me_attn_fallback_blocks(int32_t reset);

/* ---- temporal (per-pixel, over frames) causal attention ------------------------------------- *
 * rows (b*frames + fr)*npix + p.  For batch b, K/V are read from batch kv_map[b] (the temporal
 * editor's recon->edit replacement, temporal_control.py:70-88).  Causal: key frame <= query frame
 * (the reference adds -10000 above the diagonal, attention_2d.py:542-543; identical in fp32).
 * Replaces TemporalSelfAttention._attention (temporal_attn.py:152-181) and the patched closure
 * (temporal_control_utils.py:81-123).
 */
typedef struct me_tattn_args {
  const void* Q;
  const void* K;
  const void* V;
  void* O;
  int32_t ldq, ldk, ldv, ldo;
  int32_t heads, dh;
  int32_t batch, frames, npix; /* frames (of K/V) in {8,16,24,32,40,48} */
  int32_t kv_map[8];           /* batch <= 8 */
  float scale;
  /* frame sharding: Q/O hold q_frames local frames starting at global frame q_frame0 (0, 0 = all frames);
   * K/V is the all-gather of kv_parts equal frame shards, part-major: row of (b, global frame j, p) =
   * ((j / fpp) * batch + b) * fpp * npix + (j % fpp) * npix + p with fpp = frames / kv_parts (0 or 1 = one part) */
  int32_t q_frames, q_frame0, kv_parts;
  /* pixel sharding (the frame<->pixel all-to-all of a frame-sharded run): q_parts > 1 = Q and O hold ALL frames in the
   * same part-major row order as K/V (q_parts == kv_parts, q_frames == 0); npix is then this rank's pixel slice */
  int32_t q_parts;
} me_tattn_args;

int me_tattn(const me_tattn_args* a, void* stream);

/* ---- GroupNorm (channels-last) -------------------------------------------------------------- *
 * Statistics over rows_per_group rows x (C/32) channels.  rows_per_group = frames*npix reproduces
 * the reference's 5-D GroupNorm whose statistics span ALL frames (resnet_2d.py:202,230);
 * rows_per_group = npix is the per-frame GroupNorm of Transformer2DModel (attention_2d.py:303,348).
 */
typedef struct me_groupnorm_args {
  const void* X;   /* fp16 [rows, ldx]                                        */
  void* Y;         /* fp16 [rows, ldy]                                        */
  const void* gamma;
  const void* beta;
  void* stats;     /* device scratch of me_groupnorm_scratch_bytes(rows, rows_per_group, groups) bytes, 16-byte aligned.  It starts
                      with the statistics proper, fp64 [rows / rows_per_group][groups][2] = (sum, sum of squares) -- the part a
                      frame-sharded caller all-reduces between me_groupnorm_stats and me_groupnorm_apply -- followed by the
                      per-chunk partial sums of the deterministic (fixed-order, atomic-free) reduction */
  int32_t rows, rows_per_group;
  int32_t C, ldx, ldy;
  int32_t groups;  /* 32 */
  float eps;
  int32_t silu;
} me_groupnorm_args;

int me_groupnorm(const me_groupnorm_args* a, void* stream);
int64_t me_groupnorm_scratch_bytes(int32_t rows, int32_t rows_per_group, int32_t groups);
/* The two halves of me_groupnorm, for frame-sharded runs: stats zeroes a->stats and accumulates this rank's
 * (sum, sum of squares) in fp64; the host all-reduces the first rows / rows_per_group * groups * 2 doubles of a->stats over the
 * ranks; apply normalises with the GLOBAL element
 * count rows_per_group_total * (C / groups). */
int me_groupnorm_stats(const me_groupnorm_args* a, void* stream);
int me_groupnorm_apply(const me_groupnorm_args* a, int64_t rows_per_group_total, void* stream);

/* ---- LayerNorm over the channel axis (nn.LayerNorm, eps 1e-5; attention_2d.py:443-463) ------ */
typedef struct me_layernorm_args {
  const void* X;
  void* Y;
  const void* gamma;
  const void* beta;
  int32_t rows, C, ldx, ldy;
  float eps;
} me_layernorm_args;

int me_layernorm(const me_layernorm_args* a, void* stream);

/* ---- small element-wise helpers -------------------------------------------------------------- */
/* Y[r, c] = X[r, c] + alpha * A[r, c]  on fp16 views (ControlNet/adapter residual adds,
 * unet_2d_condition.py:487-494,508-509) */
int me_axpy_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, const void* A, int32_t lda,
                 int64_t rows, int32_t cols, float alpha, void* stream);
/* copy a [rows, cols] fp16 view (skip concat, unet_2d_blocks.py "torch.cat([hidden, res], dim=1)") */
int me_copy_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, int64_t rows, int32_t cols, void* stream);
/* n0 x n1 blocks of [rows, cols] fp16: block (i, j) goes from row i*xs0 + j*xs1 of X to row i*ys0 + j*ys1 of Y (strides in
 * rows).  The (batch*frame, pixel slice) <-> (pixel slice, batch*frame) reorder either side of the frame<->pixel all-to-all
 * of a frame-sharded run; the reference has no counterpart (its rearranges are views of one device's tensor). */
int me_copy_blocks(void* Y, int32_t ldy, const void* X, int32_t ldx, int32_t n0, int32_t n1, int64_t rows, int32_t cols,
                   int64_t ys0, int64_t ys1, int64_t xs0, int64_t xs1, void* stream);
/* y = silu(x), n fp16 elements */
int me_silu(void* Y, const void* X, int64_t n, void* stream);
/* y = relu(x), n fp16 elements (adapter ResnetBlock.act, controlnet_adapter.py:452,504) */
int me_relu(void* Y, const void* X, int64_t n, void* stream);
/* sinusoidal timestep embedding, diffusers get_timestep_embedding(flip_sin_to_cos=True, shift=0):
 * out fp16 [rows, dim] all rows equal (unet_2d_condition.py:430-432) */
int me_timestep_embed(void* out, int32_t rows, int32_t dim, float t, void* stream);
/* Classifier-free guidance + DDIM update (pipeline_motion_editor.py:643-648; util.py:77-87):
 * eps channels-last fp16 rows ((b*frames+f)*npix+p) with b in [0, 2*nb): [uncond x nb, cond x nb];
 * latents fp32 in the reference layout [nb, C, frames, npix];  out = ca*x + cb*(eu + g*(ec-eu)). */
int me_cfg_ddim(float* lat_out, const float* lat_in, const void* eps, int32_t lde, int32_t nb, int32_t C,
                int32_t frames, int32_t npix, float guidance, float ca, float cb, void* stream);
/* The same two calls with their per-step scalars in DEVICE memory -- step_params = fp32 [4] {t, guidance, ca, cb} -- so that a
 * whole denoising step captured into a hipGraph (MotionEditorPipeline.denoise_step_graphed) replays for every timestep:
 * the host writes the four floats, then launches the graph. */
int me_timestep_embed_dev(void* out, int32_t rows, int32_t dim, const float* step_params, void* stream);
int me_cfg_ddim_dev(float* lat_out, const float* lat_in, const void* eps, int32_t lde, int32_t nb, int32_t C,
                    int32_t frames, int32_t npix, const float* step_params, void* stream);
/* DiagonalGaussianDistribution.sample of the VAE encoder (inference.py:262: vae.encode(x).latent_dist.sample() * 0.18215):
 * moments fp16 channels-last rows (img*npix + p) x [mean 0..3 | logvar 4..7]; noise, out fp32 [n_img, 4, npix];
 * out = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale */
int me_gaussian_sample(float* out, const void* moments, int32_t ldm, const float* noise, int32_t n_img, int32_t npix, float scale, void* stream);
/* fp32 [n_img, C, H*W] (img/channel strides in elements) -> fp16 channels-last [n_img*H*W, ldy] */
int me_nchw_to_rows(void* Y, int32_t ldy, const float* X, int64_t img_stride, int64_t ch_stride,
                    int32_t n_img, int32_t C, int32_t npix, void* stream);
/* fp16 channels-last [n_img*npix, ldx] -> fp32 [n_img, C, npix] */
int me_rows_to_nchw(float* Y, int64_t img_stride, int64_t ch_stride, const void* X, int32_t ldx,
                    int32_t n_img, int32_t C, int32_t npix, void* stream);

/* Y[r, :] = softmax(X[r, :]) over `cols` fp16 logits per row, fp32 arithmetic (already scaled: the producing
 * me_gemm applies 1/sqrt(d) through alpha).  Used by the VAE decoder's single-head 512-wide attention, whose head
 * dimension is outside me_attn's {40, 80, 160}: diffusers AttentionBlock, softmax(Q K^T / sqrt(512)).
 * cols % 8 == 0, cols <= 8192; X and Y may alias. */
int me_softmax_rows(void* Y, int32_t ldy, const void* X, int32_t ldx, int64_t rows, int32_t cols, void* stream);

/* ---- backward (input-gradient) primitives ---------------------------------------------------- *
 * What motioneditor_amd/autodiff.py calls for the null-text optimisation (p2p/null_text_optimization.py:133-166 differentiates
 * the guided prev_step loss through these layers with torch autograd).  Gradients are fp32 [rows, ld] views, activations fp16.
 * The input gradient of me_gemm needs no entry of its own: it is me_gemm on transposed, tap-reversed weights.
 */
/* GEGLU (attention_2d.py FeedForward/GEGLU): pre = biased pre-activation [M, N] fp16 in the packed (16 value | 16 gate) column
 * order, dy fp32 [M, N/2] -> dpre fp16 [M, N] */
int me_geglu_bwd(void* dpre, int32_t ldd, const void* pre, int32_t ldp, const void* dy, int32_t lddy, int64_t M, int32_t N, void* stream);
/* nn.LayerNorm: dx fp32 [rows, C] from x fp16, gamma fp16, dy fp32 */
int me_layernorm_bwd(void* dx, int32_t lddx, const void* x, int32_t ldx, const void* gamma, const void* dy, int32_t lddy, int64_t rows, int32_t C, float eps,
                     void* stream);
/* GroupNorm (+ SiLU when silu != 0; statistics over rows_per_group rows x C/groups channels, as me_groupnorm): dx fp32.  Row-parallel passes
 * (statistics, per-chunk gradient sums in a fixed order, apply); scratch: me_groupnorm_bwd_scratch_bytes(rows, rows_per_group, groups) bytes, 16-byte aligned */
int me_groupnorm_bwd(void* dx, int32_t lddx, const void* x, int32_t ldx, const void* gamma, const void* beta, const void* dy, int32_t lddy, int64_t rows,
                     int32_t rows_per_group, int32_t C, int32_t groups, float eps, int32_t silu, void* scratch, void* stream);
int64_t me_groupnorm_bwd_scratch_bytes(int32_t rows, int32_t rows_per_group, int32_t groups);

/* Temporal causal attention (me_tattn, plain row order, identity kv_map): dq, dk, dv fp32 [batch*frames*npix, ld] from q, k, v fp16
 * and dout fp32; frames <= 64 */
int me_tattn_bwd(void* dq, int32_t lddq, void* dk, int32_t lddk, void* dv, int32_t lddv, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv,
                 const void* dout, int32_t lddo, int32_t batch, int32_t frames, int32_t npix, int32_t heads, int32_t dh, float scale, void* stream);

/* Row softmax backward, fp16: dS = P * (dP - sum_j P_j dP_j) * scale (the first, matrix-materialising form of the spatial
 * attention backward composes it with me_gemm and me_softmax_rows) */
int me_softmax_bwd_rows(void* dS, int32_t ldds, const void* P, int32_t ldp, const void* dP, int32_t lddp, int64_t rows, int32_t cols, float scale, void* stream);

/* ReLU epilogue backward (adapter TemporalConv -> ReLU, controlnet_adapter.py:452,504): dx = dy where the forward output > 0 */
int me_relu_bwd(void* dx, int32_t lddx, const void* dy, int32_t lddy, const void* out, int32_t ldo, int64_t rows, int32_t cols, void* stream);

/* ---- fused attention backward (plain segments) ------------------------------------------------ *
 * (dQ, dK, dV) += the input gradients of me_attn for PLAIN key segments -- [prev | cur] (attention_2d.py:705-768), per-frame
 * self-attention, the text cross-attention (:115-201), the adapter's [first | prev] (controlnet_adapter.py:332-407) -- that torch
 * autograd computes for the reference (p2p/null_text_optimization.py:149-156: loss.backward() through the UNet;
 * train_adaptor.py:372: accelerator.backward(loss)).  Flash-style: P is rebuilt per tile from the log-sum-exp me_attn stashed
 * (me_attn_args.lse); no score matrix is materialised; two deterministic kernels (key-centric for dK / dV, query-centric for dQ),
 * every output element has one owner and is accumulated into.  dO and the outputs are fp32 views (the tape's gradient buffers;
 * dO is loss-scaled by the caller: it travels through the MFMAs as fp16).
 */
typedef struct me_attn_bwd_args {
  const void* Q;  /* fp16, as me_attn */
  const void* K;
  const void* V;
  const void* O;      /* fp16 [n_items * nq, ldo]: the forward output */
  const void* dO;     /* fp32 [n_items * nq, lddo] */
  const void* lse;    /* fp32 [n_items * nq][heads] written by me_attn */
  void* dQ;           /* fp32 [n_items * nq, lddq]      += */
  void* dK;           /* fp32 [n_kv_items * nk, lddk]   += */
  void* dV;           /* fp32 [n_kv_items * nk, lddv]   += */
  void* delta;        /* fp32 scratch [n_items * nq][heads]: sum_d dO O */
  int32_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int32_t heads, dh;  /* dh in {40, 80, 160} */
  int32_t n_items, nq, nk, nseg, n_kv_items;
  const int32_t* seg_item; /* device int32 [n_items][nseg], as me_attn (every listed segment is PLAIN) */
  const int32_t* inv_ptr;  /* device int32 [n_kv_items + 1]: CSR of the inverse table ... */
  const int32_t* inv_item; /* ... inv_item[inv_ptr[k] .. inv_ptr[k + 1]) = the query items that list kv item k (once per listing) */
  float scale;
} me_attn_bwd_args;

int me_attn_bwd(const me_attn_bwd_args* a, void* stream);

/* ---- parameter gradients, gradient bookkeeping, optimiser ------------------------------------- *
 * The adapter training step (train_adaptor.py:364-385: loss.backward() into controlnet_adapter.*, DDP gradient average,
 * clip_grad_norm_(1.0), AdamW(lr 3e-5, betas (0.9, 0.999), weight_decay 1e-2, eps 1e-8)) and the null-text optimisation's loss
 * and Adam (p2p/null_text_optimization.py:140-160) on the device.  Reductions over the token axis use fixed-order partials.
 */
typedef struct me_gemm_dw_args {
  const void* dY;     /* [M, lddy] gradient of me_gemm's (pre-epilogue) output: fp32, or fp16 when dy_is_f16 */
  const void* X;      /* fp16 [rows, ldx]: the input me_gemm read */
  void* dW;           /* fp32 [N][taps][K]   += alpha * sum_m dY[m, n] X[src(m, tap), k]   (this call: one tap) */
  void* work;         /* scratch of me_gemm_dw_work_bytes(M, N, K) bytes, 16-byte aligned */
  int32_t M, N, K, lddy, ldx, dy_is_f16;
  int32_t taps, tap;
  int32_t gather;     /* ME_GATHER_DENSE (taps 1) or ME_GATHER_TCONV (taps 3, unsharded) */
  int32_t frames, npix, chunk;
  float alpha;
} me_gemm_dw_args;

int me_gemm_dw(const me_gemm_dw_args* a, void* stream);
int64_t me_gemm_dw_work_bytes(int32_t M, int32_t N, int32_t K);
/* out[n] += alpha * sum_m dY[m, n]  (bias gradients); work: me_colsum_work_bytes(N) bytes */
int me_colsum(float* out, const void* dY, int32_t lddy, int32_t dy_is_f16, int64_t M, int32_t N, float alpha, float* work, void* stream);
int64_t me_colsum_work_bytes(int32_t N);
/* nn.LayerNorm parameter gradients: dgamma[c] += alpha * sum_m dy[m, c] xhat[m, c], dbeta[c] += alpha * sum_m dy[m, c] (either may be NULL);
 * x fp16, dy fp32; work: me_layernorm_bwd_params_work_bytes(rows, C) bytes */
int me_layernorm_bwd_params(float* dgamma, float* dbeta, const void* x, int32_t ldx, const void* dy, int32_t lddy, int64_t rows, int32_t C, float eps, float alpha,
                            float* work, void* stream);
int64_t me_layernorm_bwd_params_work_bytes(int64_t rows, int32_t C);
/* dst[r, c] += alpha * src[r, c] on an fp32 [rows, cols] view (src fp32 or fp16): the gradient accumulation of the reverse-mode tape.
 * pool_h, pool_w > 0: dst pixel (img, y, x) of a pool_h x pool_w grid collects the 2 x 2 block (2y + a, 2x + b) of the 2 pool_h x 2 pool_w
 * source grid -- the input gradient of the nearest-2x upsample in front of a convolution (resnet_2d.py:77).
 * src_is_f16: bit 0 = src is fp16 (else fp32); bit 1 = STORE, dst = alpha * src without reading dst -- the first contribution to a gradient
 * buffer that was allocated but never zeroed (saves the fill and the read) */
int me_grad_acc(void* dst, int32_t lddst, const void* src, int32_t ldsrc, int32_t src_is_f16, int64_t rows, int32_t cols, float alpha, int32_t pool_h, int32_t pool_w,
                void* stream);
/* out = {sum x^2, max |x|} of an fp32 vector (gradient-norm clipping, loss, loss-scale selection); work: me_sumsq_work_bytes() bytes */
int me_sumsq_absmax(float* out, const float* x, int64_t n, float* work, void* stream);
int64_t me_sumsq_work_bytes(void);
/* One AdamW step on fp32 master parameters (torch.optim.AdamW; weight_decay 0 = torch.optim.Adam).  bias_c1 = 1 - beta1^t, bias_c2 = 1 - beta2^t.
 * The gradient used is g * grad_scale * clip, clip = min(1, max_grad_norm / (sqrt(gnorm_sq[0]) * grad_scale + 1e-6)) when gnorm_sq (a DEVICE
 * scalar, me_sumsq_absmax's out[0] over the whole gradient bucket) is given -- torch.nn.utils.clip_grad_norm_ without a host round trip */
int me_adamw(float* p, float* m, float* v, const float* g, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_c1, float bias_c2,
             const float* gnorm_sq, float max_grad_norm, float grad_scale, void* stream);
/* dst fp16 [n] = src fp32 [n]: refresh of the packed fp16 weights from their fp32 masters */
int me_cast_f16(void* dst, const float* src, int64_t n, void* stream);
/* dst fp16 [rows, lddst] = src fp32 [rows, ldsrc] over columns [0, cols), zeros in [cols, pad_cols): a gradient view as the fp16 operand of the
 * input-gradient / weight-gradient MFMA kernels (pad_cols: the next multiple of 8 the transposed weights are padded to) */
int me_cast_rows_f16(void* dst, int32_t lddst, const float* src, int32_t ldsrc, int64_t rows, int32_t cols, int32_t pad_cols, void* stream);
/* Loss seed of both optimisations: rec = ca * x + cb * (eps_u + guidance * (eps_c - eps_u)) (eps_c NULL: rec = ca * x + cb * eps_u; x NULL: no x term),
 * diff = rec - target (fp32 [nb, C, frames, npix], reference layout), d_eps[row, c] = coef * diff on channels-last rows (fp32, ld ldd).
 * prev_step + mse of p2p/null_text_optimization.py:26-36,150-151; the mse of train_adaptor.py:368 with ca = 0, cb = 1 */
int me_mse_seed(float* diff, float* d_eps, int32_t ldd, const void* eps_u, int32_t ldu, const void* eps_c, int32_t ldc, const float* x, const float* target, int32_t nb,
                int32_t C, int32_t frames, int32_t npix, float guidance, float ca, float cb, float coef, void* stream);

/* ---- the denoising step as ONE call: launch list recorded once, re-issued from C (csrc/plan.hip) ------------------------------- *
 * Replaces: one iteration of the reference's denoising loop body, pipeline_motion_editor.py:603-648 (ControlNet forward :613-625,
 * UNet3D forward with the adapter :632-640, classifier-free guidance :643-645, DDIMScheduler.step :648) -- ~1100 kernel launches
 * on two HIP streams.  SURVEY.md section 8(b) sketched `me_plan` / `me_denoise_step`; the launch graph itself (which kernel on which
 * rows: models/graph.py) stays host logic, its execution moves behind this boundary:
 *
 *   me_plan_begin(&plan, stream)        this thread starts recording: every kernel a me_* entry point launches is executed as usual
 *                                       AND appended to the plan {kernel, grid, block, LDS bytes, stream, argument bytes}
 *   ... one eager step through the per-family entry points above, its per-step scalars read from DEVICE memory
 *       (me_timestep_embed_dev / me_cfg_ddim_dev), cross-stream dependencies stated with me_plan_event_record / _wait ...
 *   me_plan_end(plan)
 *   me_plan_bind(plan, ...)             names the static buffers the recorded step reads (latents, text embeddings, step scalars)
 *                                       and writes (updated latents)
 *   me_denoise_step(plan, ...)          per timestep: copies the inputs in (when they are not the bound buffers themselves), writes
 *                                       {t, guidance, ca, cb}, re-issues every launch / event in recorded order on the live streams
 *
 * The caller guarantees what a captured hipGraph also needs: every buffer whose address a recorded launch holds stays allocated
 * and is not handed to anyone else while the plan lives (the Python mirror records inside a private allocator pool and keeps it),
 * and the step's shapes, key-segment tables and editor gating are those of the recorded step.  A plan is bound to the device,
 * the thread-independent streams and the buffers it was recorded on; stream index 0 (the stream given to me_plan_begin) is
 * replaced by me_denoise_step's `stream` argument, side streams are used as recorded.  Not thread-safe per plan. */
typedef struct me_plan me_plan;

typedef struct me_plan_stats {
  int64_t launches;      /* kernel launches in the plan                       */
  int64_t event_records; /* cross-stream dependencies: events recorded ...    */
  int64_t event_waits;   /* ... and waited on                                 */
  int64_t arg_bytes;     /* bytes of recorded kernel arguments (16-byte aligned each) */
  int64_t replays;       /* successful me_denoise_step calls so far           */
  int32_t streams;       /* distinct streams the step used (1 or 2)           */
} me_plan_stats;

#define ME_PLAN_LAUNCH 0
#define ME_PLAN_RECORD 1
#define ME_PLAN_WAIT 2
typedef struct me_plan_node_info {
  int32_t kind;          /* ME_PLAN_LAUNCH / ME_PLAN_RECORD / ME_PLAN_WAIT    */
  int32_t stream;        /* stream index (0 = main)                           */
  int32_t event;         /* event id of a RECORD / WAIT node, -1 for launches */
  uint32_t grid[3], block[3];
  uint32_t lds_bytes;
  int32_t n_args;
  int64_t arg_bytes;     /* total bytes of the launch's arguments, densely packed */
} me_plan_node_info;

int me_plan_begin(me_plan** out, void* main_stream);
/* states, while recording, what `hipEventRecord(ev, stream)` / `hipStreamWaitEvent(stream, ev)` state to the runtime (the caller still
 * issues those itself for the recording pass; the plan owns the events it replays with) */
int me_plan_event_record(void* stream, int32_t* event_id);
int me_plan_event_wait(void* stream, int32_t event_id);
int me_plan_end(me_plan* plan);
/* 1 while this thread records a plan */
int me_plan_recording(void);
/* latents_in / latents_out: fp32 [nb, 4, f, h, w] of latents_bytes each; text_emb: the step's text-embedding buffer (text_bytes, may be
 * 0 / NULL when the step reads none); step_params: fp32 [4] {t, guidance, ca, cb} in device memory, the buffer the recorded
 * me_timestep_embed_dev / me_cfg_ddim_dev launches read */
int me_plan_bind(me_plan* plan, void* latents_in, int64_t latents_bytes, void* text_emb, int64_t text_bytes, float* step_params, void* latents_out);
/* One denoising step.  latents_in / text_emb: device buffers copied into the bound ones first (NULL, or the bound address itself: nothing is
 * copied); latents_out likewise receives a copy of the bound output.  ca, cb: the DDIM update's coefficients for timestep t
 * (prev = ca * x + cb * eps, schedulers.DDIMScheduler.coeffs).  Only enqueues; returns ME_EHIP naming the failing node otherwise. */
int me_denoise_step(me_plan* plan, const void* latents_in, void* latents_out, const void* text_emb, float t, float guidance, float ca, float cb, void* stream);
int me_plan_info(const me_plan* plan, me_plan_stats* out);
/* node `index` of the plan (0 <= index < launches + event_records + event_waits, recorded order); arg_bytes_out (may be NULL) receives up to
 * arg_bytes_cap bytes of the launch's arguments, densely packed in declaration order */
int me_plan_node(const me_plan* plan, int64_t index, me_plan_node_info* out, void* arg_bytes_out, int64_t arg_bytes_cap);
void me_plan_destroy(me_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* MOTIONED_H */
