"""Thin tensor-level wrappers over the C ABI.  PyTorch-ROCm is used ONLY for device memory and the
current HIP stream; every arithmetic op below is a ``libmotioned.so`` kernel.

Activation convention: 2-D fp16 tensors ``[rows, C]`` (possibly strided views, unit column stride),
rows ordered ``(batch, frame, pixel)`` -- channels-last / token-major everywhere, so the reference's
``rearrange`` calls between "(b f) c h w", "(b f) (h w) c" and "(b d) f c" are free.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import os

import torch

from . import capi
from .capi import AttnArgs, AttnBwdArgs, ConvSmallArgs, GemmArgs, GemmDwArgs, GroupNormArgs, LayerNormArgs, TAttnArgs

F16 = torch.float16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of torch's current stream.  The raw accessors cost ~0.3 us; torch.cuda.current_stream() builds a Stream
    object per call (~8 us, a third of the host's enqueue time of a step at ~1100 launches)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk2d(t: torch.Tensor, name: str) -> None:
    if t.dtype != F16 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA fp16 2-D tensor with unit column stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")


# ---- optional per-launch profiling (bench.py): HIP events on the launch stream around each kernel call ----
SELECT_ROWS_SCALE = 1   # > 1 while the UNet graph computes a shared sub-batch (classifier-free-guidance prefix): me_gemm then selects its kernel as for the full batch
PROFILE = None  # None = off; else a list receiving (family, algorithmic_flops, algorithmic_bytes, start_event, end_event)


def _pb():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _pe(e0, family: str, flops: float, nbytes: float, detail: str = "", kernel: str = "", exec_flops=None) -> None:
    """flops = reference-semantics work of the call (what the reference's own ops would execute), exec_flops = what the
    kernel actually multiplies (a binary dual segment is computed once, not as 2 nk materialised keys)."""
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((family, flops, nbytes, e0, e1, detail, kernel or family, flops if exec_flops is None else exec_flops))


def _last_kernel() -> str:
    return capi.lib().me_last_kernel().decode()


def empty(rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((rows, cols), dtype=F16, device=like.device)


ATTN_ITEM_ORDER = os.environ.get("ME_ATTN_ITEM_ORDER", "1") != "0"   # the edited launches walk their items (recon g, edit g, recon g + 1, ...): me_attn_args.item_order
ROW_RANGE = True       # this backend implements gemm(row_range=...) (me_gemm_args.m_off)
LN_FOLD = True         # ... and gemm(ln=..., ln_out=...) / ln_stats (ABI 9: LayerNorm folded into the projection that consumes it)
HEAD_MAJOR_KV = True   # this backend implements gemm(head_major=...) / 3-D k, v in attention (the CPU emulation and the autodiff recorder do not)


def gemm(x: torch.Tensor, w: torch.Tensor, *, M: Optional[int] = None, out: Optional[torch.Tensor] = None,
         bias: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None, rows_per_vec: int = 0,
         res: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None, geglu: bool = False, act: int = 0, alpha: float = 1.0,
         conv: Optional[Tuple[int, int, int, int, int, int]] = None,
         tconv: Optional[Tuple[int, ...]] = None, res_rows: int = 0, res2_rows: int = 0, head_major: Optional[Tuple[int, int]] = None,
         row_range: Optional[Tuple[int, int]] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, float]] = None, ln_out: bool = False):
    """out[m, n] = epilogue(sum_{tap,c} x[src(m,tap), c] * w[n, tap, c]).

    w: fp16 [N, taps, K] (taps = 1 dense, 9 for ``conv=(Hin, Win, Hout, Wout, stride, ups)``,
    3 for ``tconv=(frames, npix, chunk)``).  res_rows / res2_rows > 0: res / res2 holds that many rows, output row m reads row m % rows.
    row_range = (lo, hi): only the output rows [lo, hi) of the M-row problem are computed (into `out`, which must be given and hold all M rows): the
    interior / boundary launches of a frame-sharded TemporalConv (me_gemm_args.m_off).
    head_major = (col0, dh): the output columns from col0 on leave as a second tensor [(N - col0) / dh, M, dh] -- one contiguous [rows, dh]
    panel per head (me_gemm_args.C2) -- and the call returns (out[:, :col0], panels); col0 = 0: every column does, the call returns (None, panels).
    ln = (stats, colsum, cvec, eps): x holds UN-normalised rows and w = W diag(gamma) (weights.Packed.ln_fold): the LayerNorm in front of this projection
    is applied to the accumulators, rstd (acc - mean colsum) + cvec, from the partial row sums `stats` fp32 [parts, rows, 2] (ln_stats(), or ln_out of the
    projection that produced x).  ln_out: the call also returns the partial row sums of ITS output rows, (out, stats)."""
    _chk2d(x, "gemm.x")
    if w.dtype != F16 or not w.is_contiguous() or w.dim() != 3:
        raise ValueError("gemm.w: expected contiguous fp16 [N, taps, K]")
    N, taps, K = w.shape
    a = GemmArgs()
    a.gather = capi.GATHER_DENSE
    if conv is not None:
        a.gather = capi.GATHER_CONV3
        a.Hin, a.Win, a.Hout, a.Wout, a.stride, a.ups = conv[:6]
        a.pad0 = conv[6] if len(conv) > 6 else 0     # 1: pad (0,1,0,1) instead of 1 all round (VAE encoder downsampling)
        if taps != 9:
            raise ValueError("gemm: conv needs 9 taps")
    elif tconv is not None:
        a.gather = capi.GATHER_TCONV
        a.frames, a.npix, a.chunk = tconv[:3]
        if len(tconv) > 3:   # frame-sharded: (frames, npix, chunk, frame0, frames_total, halo_prev_row, halo_next_row)
            a.frame0, a.frames_total, a.halo_prev, a.halo_next = tconv[3:]
        if taps != 3:
            raise ValueError("gemm: tconv needs 3 taps")
    elif taps != 1:
        raise ValueError("gemm: dense needs 1 tap")
    if M is None:
        M = x.shape[0]
    if x.shape[1] < K:
        raise ValueError(f"gemm: x has {x.shape[1]} columns < K={K}")
    n_out = N // 2 if geglu else N
    panels = None
    if head_major is not None:
        col0, hdh = head_major
        if out is not None or geglu or (N - col0) % hdh or col0 < 0:
            raise ValueError("gemm: head_major needs whole heads behind col0 >= 0 and allocates its own outputs")
        panels = torch.empty(((N - col0) // hdh, M, hdh), dtype=F16, device=x.device)
        a.C2, a.c2_col0, a.c2_dh, a.c2_hs = panels.data_ptr(), col0, hdh, M * hdh
        n_out = col0
        if col0 == 0:      # EVERY column leaves as panels (q | k | v all head-major): C is never written -- any valid, aligned address will do
            out = panels.view(-1)[:8 * M].view(M, 8)
    if out is None:
        if row_range is not None:
            raise ValueError("gemm: row_range writes into a caller-provided `out` of all M rows")
        out = empty(M, n_out, x)
    _chk2d(out, "gemm.out")
    if out.shape[0] < M or out.shape[1] < n_out:
        raise ValueError("gemm: out too small")
    a.X, a.W, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    a.M, a.N, a.K = M, N, K
    if row_range is not None:
        lo, hi = row_range
        if not (0 <= lo < hi <= M) or conv is not None:
            raise ValueError("gemm: row_range must lie inside [0, M) (dense / TemporalConv launches)")
        a.m_off, a.M = lo, hi
        a.sel_rows = max(a.sel_rows, M)      # pick kernels as the whole launch would
    a.ldx, a.ldc = x.stride(0), out.stride(0)
    a.bias = _p(bias)
    if rowvec is not None:
        _chk2d(rowvec, "gemm.rowvec")
        a.rowvec, a.ldrv, a.rows_per_vec = rowvec.data_ptr(), rowvec.stride(0), rows_per_vec
    if res is not None:
        _chk2d(res, "gemm.res")
        a.res, a.ldr = res.data_ptr(), res.stride(0)
    if res2 is not None:
        _chk2d(res2, "gemm.res2")
        a.res2, a.ldr2 = res2.data_ptr(), res2.stride(0)
    a.geglu = 1 if geglu else 0
    a.act = act
    a.alpha = alpha
    a.res_rows, a.res2_rows = res_rows, res2_rows
    if ln is not None:
        st_, cs_, cv_, eps_ = ln
        if st_.dtype != torch.float32 or st_.dim() != 3 or st_.shape[2] != 2 or st_.stride(2) != 1 or st_.stride(1) != 2 or st_.shape[1] < M \
                or cs_.dtype != torch.float32 or cv_.dtype != torch.float32 or cs_.numel() != N or cv_.numel() != N or not cs_.is_contiguous() or not cv_.is_contiguous():
            raise ValueError("gemm: ln = (fp32 stats [parts, rows, 2], fp32 colsum [N], fp32 cvec [N], eps)")
        a.ln_stats, a.ln_colsum, a.ln_cvec, a.ln_eps = st_.data_ptr(), cs_.data_ptr(), cv_.data_ptr(), eps_
        a.ln_parts, a.ln_stride = st_.shape[0], st_.stride(0)
    so = None
    if ln_out:
        if panels is not None or geglu:
            raise ValueError("gemm: ln_out is for plain row outputs")
        so = torch.empty((n_out // 320 if n_out % 320 == 0 else 1, M, 2), dtype=torch.float32, device=x.device)
        a.ln_out, a.ln_out_stride = so.data_ptr(), so.stride(0)
    if SELECT_ROWS_SCALE > 1:
        a.sel_rows = M * SELECT_ROWS_SCALE
    for r_, n_ in ((res, res_rows), (res2, res2_rows)):
        if r_ is not None and r_.shape[0] < (n_ or M):
            raise ValueError("gemm: residual has fewer rows than the output reads")
    if N >= 1280 and M <= 8192:   # (the library splits only small grids at N >= 1280: do not even ask for the ~500 large launches of a step)
        nb = capi.lib().me_gemm_work_bytes(C.byref(a))   # > 0: a grid too small to fill the chip -- me_gemm may split it along K through this scratch
        if nb > 0:
            a.work, a.work_bytes = _work(nb, x.device, "splitk").data_ptr(), nb
    e0 = _pb()
    capi.check(capi.lib().me_gemm(C.byref(a), _stream()), "me_gemm")
    if e0 is not None:
        # algorithmic HBM bytes: X once (a gather's taps re-read it from L2), the weights, the output, and every residual term the epilogue reads
        rows_in = (M // (conv[2] * conv[3])) * conv[0] * conv[1] if conv is not None else M
        n_terms = (res is not None) + (res2 is not None)
        n_hm = panels.numel() // M if panels is not None else 0      # output columns that leave as head-major panels (counted like any other output byte)
        _pe(e0, "gemm", 2.0 * M * N * K * taps, 2.0 * (rows_in * K + N * K * taps + M * n_out * (1 + n_terms) + M * n_hm), f"M{M} N{N} K{K} taps{taps}{' geglu' if geglu else ''}{' +b' if bias is not None else ''}{' +rv' if rowvec is not None else ''}"
            f"{' +res' if res is not None else ''}{' +res2' if res2 is not None else ''}{' act' + str(act) if act else ''}{' a' + str(alpha) if alpha != 1.0 else ''}{' ln' if ln is not None else ''}{' lnout' if ln_out else ''}", _last_kernel())
    if panels is not None and n_out == 0:
        return None, panels
    out = out[:M, :n_out] if (out.shape[0] != M or out.shape[1] != n_out) else out
    if so is not None:
        return out, so
    return out if panels is None else (out, panels)


def ln_stats(x: torch.Tensor) -> torch.Tensor:
    """Partial row sums (sum, sum of squares) of x over its 320-column parts, fp32 [parts, rows, 2]: what gemm(ln=...) derives a row's mean / rstd from
    when the producer of x could not leave them behind (me_ln_stats)."""
    _chk2d(x, "ln_stats.x")
    rows, Cc = x.shape
    st = torch.empty((Cc // 320 if Cc % 320 == 0 else 1, rows, 2), dtype=torch.float32, device=x.device)
    e0 = _pb()
    capi.check(capi.lib().me_ln_stats(x.data_ptr(), x.stride(0), rows, Cc, st.data_ptr(), st.stride(0), _stream()), "me_ln_stats")
    _pe(e0, "layernorm", 3.0 * rows * Cc, 2.0 * rows * Cc, "stats", "ln_stats")
    return st


def gemm_splits_k(M: int, N: int, K: int, taps: int = 1) -> bool:
    """Would me_gemm split this launch along K (small grids at N >= 1280, me_gemm_work_bytes > 0)?  A split changes the fp32 summation order, so a caller that
    wants row-range pieces (never split) to equal the one-launch form bit for bit asks first."""
    if not (N >= 1280 and M <= 8192):
        return False
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.gather = capi.GATHER_TCONV if taps == 3 else (capi.GATHER_CONV3 if taps == 9 else capi.GATHER_DENSE)
    return capi.lib().me_gemm_work_bytes(C.byref(a)) > 0


def conv_small(inp: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, n_img: int, Cin: int, H: int, Wd: int,
               img_stride: int, ch_stride: int, frames: int = 0, frame_stride: int = 0, silu: bool = False) -> torch.Tensor:
    if w.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        raise ValueError("conv_small: weights and bias must be fp32")
    Cout = w.shape[0]
    out = torch.empty((n_img * H * Wd, Cout), dtype=F16, device=inp.device)
    a = ConvSmallArgs()
    a.inp, a.W, a.bias, a.out = inp.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr()
    a.n_img, a.Cin, a.Cout, a.H, a.Wd = n_img, Cin, Cout, H, Wd
    a.img_stride, a.ch_stride = img_stride, ch_stride
    if inp.dtype == torch.float32:
        a.in_is_f16 = 0
    elif inp.dtype == F16:
        a.in_is_f16 = 1
    else:
        raise ValueError("conv_small: input must be fp32 or fp16")
    a.silu = 1 if silu else 0
    a.frames, a.frame_stride = frames, frame_stride
    capi.check(capi.lib().me_conv_small(C.byref(a), _stream()), "me_conv_small")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, heads: int, dh: int, n_items: int, nq: int, nk: int,
              seg_item: torch.Tensor, seg_mode: torch.Tensor, mask: Optional[torch.Tensor] = None,
              scale: Optional[float] = None, out: Optional[torch.Tensor] = None, q_items: int = 0, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q_items > 0: q holds q_items query items, item i reads item i % q_items.  lse: fp32 [n_items * nq, heads] that receives the
    log2-domain log-sum-exp of every (query, head) -- what attention_bwd rebuilds P from (plain segments only)."""
    hm = k.dim() == 3   # head-major K / V: [heads, rows, dh] panels (gemm(head_major=...)); strides (head, row, 1)
    if hm:
        if v.dim() != 3 or k.shape[0] != heads or v.shape[0] != heads or k.shape[2] != dh or v.shape[2] != dh or k.stride(2) != 1 or v.stride(2) != 1 \
                or k.dtype != F16 or v.dtype != F16:
            raise ValueError("attention: head-major k / v must be fp16 [heads, rows, dh] with unit column stride")
        if q.dim() == 3:   # head-major Q as well (ABI 8, me_attn_args.hsq): [heads, rows, dh] panels
            if q.shape[0] != heads or q.shape[2] != dh or q.stride(2) != 1 or q.dtype != F16 or q.stride(1) % 8 or q.stride(0) % 8 or q.data_ptr() % 16:
                raise ValueError("attention: head-major q must be fp16 [heads, rows, dh] with unit column stride")
        else:
            _chk2d(q, "attention.q")
    else:
        for t, n in ((q, "q"), (k, "k"), (v, "v")):
            _chk2d(t, "attention." + n)
    qhm = q.dim() == 3
    if q.shape[1 if qhm else 0] < (q_items or n_items) * nq:
        raise ValueError("attention: q has fewer rows than the items read")
    if seg_item.dtype != torch.int32 or seg_mode.dtype != torch.int32 or seg_item.shape != seg_mode.shape or seg_item.shape[0] != n_items:
        raise ValueError("attention: seg tables must be int32 [n_items, nseg]")
    if out is None:
        out = empty(n_items * nq, heads * dh, q)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1 if qhm else 0), k.stride(1 if hm else 0), v.stride(1 if hm else 0), out.stride(0)
    if hm:
        a.hsk, a.hsv = k.stride(0), v.stride(0)
    if qhm:
        a.hsq = q.stride(0)
    a.heads, a.dh = heads, dh
    a.n_items, a.nq, a.nk, a.nseg = n_items, nq, nk, seg_item.shape[1]
    a.seg_item, a.seg_mode, a.mask = seg_item.data_ptr(), seg_mode.data_ptr(), _p(mask)
    a.scale = dh ** -0.5 if scale is None else scale
    from . import segments
    gd = segments.GENERAL_DUAL.get(seg_mode.data_ptr())
    bd = segments.BINARY_DUAL.get(seg_mode.data_ptr())
    if gd is None or bd is None:   # table not built (and kept alive) by segments.py: inspect it, never cache a transient pointer
        gd = bool(((seg_mode == 1) | (seg_mode == 2)).any().item())
        bd = bool((seg_mode == 3).any().item())
    a.general_dual = 1 if gd else 0
    a.q_items = q_items
    order = segments.ITEM_ORDER.get(seg_item.data_ptr())
    if order is not None and order.device == q.device and order.numel() == n_items and ATTN_ITEM_ORDER:   # (A/B switch, read once at import)
        a.item_order = order.data_ptr()
    if lse is not None:
        if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != n_items * nq * heads:
            raise ValueError("attention: lse must be a contiguous fp32 [n_items * nq, heads] tensor")
        a.lse = lse.data_ptr()
    if bd and not gd:   # binary dual keys: me_attn needs fp32 scratch for the per-kv-item column sums of V
        n_kv = k.shape[1 if hm else 0] // nk
        vsum = torch.empty(capi.lib().me_attn_vsum_bytes(n_kv, heads * dh) // 4, dtype=torch.float32, device=q.device)
        a.vsum, a.n_kv_items = vsum.data_ptr(), n_kv
    e0 = _pb()
    capi.check(capi.lib().me_attn(C.byref(a), _stream()), "me_attn")
    if e0 is not None:
        # reference-semantics key count: a dual (fg|bg) segment is 2*nk materialised keys (fully_control.py:381-413)
        units = segments.KEY_UNITS.get(seg_item.data_ptr())
        if units is None:
            sm, si = seg_mode.tolist(), seg_item.tolist()
            units = sum((2 if m != 0 else 1) for ri, rm in zip(si, sm) for i_, m in zip(ri, rm) if i_ >= 0)
        keys = nk * units
        nseg_exec = segments.SEG_COUNT.get(seg_item.data_ptr())
        if nseg_exec is None:
            nseg_exec = int((seg_item >= 0).sum().item())
        _pe(e0, f"attn_dh{dh}", 4.0 * heads * nq * keys * dh, 2.0 * 4 * n_items * nq * heads * dh, "", _last_kernel(),
            4.0 * heads * nq * nk * nseg_exec * dh)
    return out


def temporal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, heads: int, dh: int, batch: int, frames: int, npix: int,
                       kv_map: Optional[Sequence[int]] = None, scale: Optional[float] = None, q_frames: int = 0, q_frame0: int = 0,
                       kv_parts: int = 1, q_parts: int = 1) -> torch.Tensor:
    """frames = K/V frames.  Frame-sharded: q holds q_frames local frames from global frame q_frame0; k, v are the
    all-gather (part-major) of kv_parts equal frame shards.  Pixel-sharded (after the frame<->pixel all-to-all): q, k, v and
    the output all hold every frame of this rank's npix pixels, part-major (q_parts == kv_parts)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk2d(t, "temporal_attention." + n)
    out = empty(batch * (q_frames or frames) * npix, heads * dh, q)
    a = TAttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.heads, a.dh, a.batch, a.frames, a.npix = heads, dh, batch, frames, npix
    km = list(kv_map) if kv_map is not None else list(range(batch))
    for i in range(8):
        a.kv_map[i] = km[i] if i < len(km) else 0
    a.scale = dh ** -0.5 if scale is None else scale
    a.q_frames, a.q_frame0, a.kv_parts, a.q_parts = q_frames, q_frame0, kv_parts, q_parts
    e0 = _pb()
    capi.check(capi.lib().me_tattn(C.byref(a), _stream()), "me_tattn")
    _pe(e0, "tattn", 4.0 * batch * npix * heads * frames * frames * dh, 2.0 * 4 * batch * frames * npix * heads * dh)
    return out


_gn_scratch = {}


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, rows_per_group: int, eps: float, silu: bool,
              groups: int = 32, out: Optional[torch.Tensor] = None, reduce=None, rows_per_group_total: Optional[int] = None) -> torch.Tensor:
    """reduce: callable(stats fp32 tensor) -> None that all-reduces the (sum, sumsq) statistics over the frame shards
    (the reference's 5-D GroupNorm spans all frames); rows_per_group_total = the global rows per group."""
    _chk2d(x, "groupnorm.x")
    rows, Cc = x.shape
    if out is None:
        out = empty(rows, Cc, x)
    nsg = rows // rows_per_group
    nbytes = capi.lib().me_groupnorm_scratch_bytes(rows, rows_per_group, groups)
    # per stream (two streams never share a statistics buffer) and per capture (a buffer born inside a hipGraph's pool stays there)
    key = (x.device, nbytes, _stream(), torch.cuda.is_current_stream_capturing())
    stats = _gn_scratch.get(key)
    if stats is None:
        stats = _gn_scratch[key] = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device)
    a = GroupNormArgs()
    a.X, a.Y, a.gamma, a.beta, a.stats = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr()
    a.rows, a.rows_per_group, a.C, a.ldx, a.ldy = rows, rows_per_group, Cc, x.stride(0), out.stride(0)
    a.groups, a.eps, a.silu = groups, eps, 1 if silu else 0
    e0 = _pb()
    if reduce is None:
        capi.check(capi.lib().me_groupnorm(C.byref(a), _stream()), "me_groupnorm")
    else:
        capi.check(capi.lib().me_groupnorm_stats(C.byref(a), _stream()), "me_groupnorm_stats")
        reduce(stats[:nsg * groups * 2])   # fp64 (sum, sum of squares) of every (sample-group, channel group)
        capi.check(capi.lib().me_groupnorm_apply(C.byref(a), rows_per_group_total or rows_per_group, _stream()), "me_groupnorm_apply")
    _pe(e0, "groupnorm", 8.0 * rows * Cc, 4.0 * rows * Cc)
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _chk2d(x, "layernorm.x")
    out = empty(x.shape[0], x.shape[1], x)
    a = LayerNormArgs()
    a.X, a.Y, a.gamma, a.beta = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.rows, a.C, a.ldx, a.ldy, a.eps = x.shape[0], x.shape[1], x.stride(0), out.stride(0), eps
    e0 = _pb()
    capi.check(capi.lib().me_layernorm(C.byref(a), _stream()), "me_layernorm")
    _pe(e0, "layernorm", 8.0 * x.shape[0] * x.shape[1], 4.0 * x.shape[0] * x.shape[1])
    return out


def softmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row softmax of fp16 logits [rows, cols] (fp32 arithmetic); out may be x."""
    if x.dtype != F16 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("softmax_rows: expected an fp16 [rows, cols] view with unit column stride")
    if out is None:
        out = torch.empty_like(x)
    e0 = _pb()
    capi.check(capi.lib().me_softmax_rows(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], _stream()), "me_softmax_rows")
    _pe(e0, "softmax", 0.0, 4.0 * x.shape[0] * x.shape[1])
    return out


def axpy_rows(y: torch.Tensor, x: torch.Tensor, a_: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """y = x + alpha * a_ on equal-shape [rows, cols] views (y may alias x)."""
    for t, n in ((y, "y"), (x, "x"), (a_, "a")):
        _chk2d(t, "axpy_rows." + n)
    capi.check(capi.lib().me_axpy_rows(y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0), a_.data_ptr(), a_.stride(0),
                                       y.shape[0], y.shape[1], alpha, _stream()), "me_axpy_rows")
    return y


def copy_rows(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _chk2d(y, "copy_rows.y")
    _chk2d(x, "copy_rows.x")
    capi.check(capi.lib().me_copy_rows(y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], _stream()), "me_copy_rows")
    return y


def copy_blocks(y: torch.Tensor, x: torch.Tensor, n0: int, n1: int, rows: int, *, ys0: int, ys1: int, xs0: int, xs1: int) -> torch.Tensor:
    """n0 x n1 blocks of [rows, cols]: block (i, j) from row i*xs0 + j*xs1 of x to row i*ys0 + j*ys1 of y."""
    _chk2d(y, "copy_blocks.y")
    _chk2d(x, "copy_blocks.x")
    if (n0 - 1) * xs0 + (n1 - 1) * xs1 + rows > x.shape[0] or (n0 - 1) * ys0 + (n1 - 1) * ys1 + rows > y.shape[0] or x.shape[1] > y.shape[1]:
        raise ValueError("copy_blocks: block grid exceeds a tensor")
    capi.check(capi.lib().me_copy_blocks(y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0), n0, n1, rows, x.shape[1], ys0, ys1, xs0, xs1, _stream()),
               "me_copy_blocks")
    return y


NATIVE = True   # this backend is libmotioned (tests/emu_ops.py, the torch emulation of the ABI, has no such flag): data movement below stays inside the library


def clone_rows(x: torch.Tensor) -> torch.Tensor:
    """x.clone() of an fp16 row tensor as a library copy (a torch kernel inside a step would be missing from a recorded plan's replays, plan.py)."""
    return copy_rows(torch.empty_like(x), x)


def _as_f16_row(t: torch.Tensor) -> torch.Tensor:
    """A contiguous tensor's bytes as one fp16 row [1, n] (n a multiple of 8): what copy_rows moves."""
    v = t.reshape(-1).view(F16)
    if v.numel() % 8 or v.data_ptr() % 16:
        raise ValueError("byte copy: size must be a multiple of 16 bytes and the address 16-byte aligned")
    return v.reshape(1, -1)


def repeat_batch(x: torch.Tensor, times: int) -> torch.Tensor:
    """torch.cat([x] * times) along dim 0 of a contiguous tensor (the classifier-free-guidance duplication of the latents,
    pipeline_motion_editor.py:605) as `times` library copies."""
    if not x.is_contiguous():
        raise ValueError("repeat_batch: contiguous input")
    out = torch.empty((times * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    for k in range(times):
        copy_rows(_as_f16_row(out[k * x.shape[0]:(k + 1) * x.shape[0]]), _as_f16_row(x))
    return out


def to_f16_rows(ehs: torch.Tensor) -> torch.Tensor:
    """[B, n, C] (fp32 or fp16; contiguous, or batch entries that are each contiguous -- a `[1::2]` slice) -> fp16 rows [B * n, C]:
    `ehs.to(float16).reshape(-1, C)` with the cast done by me_cast_f16 / the copy by me_copy_rows."""
    Cc = ehs.shape[-1]
    if ehs.dtype == F16 and ehs.is_contiguous():
        return ehs.reshape(-1, Cc)
    B = ehs.shape[0]
    n = ehs[0].numel() // Cc
    out = torch.empty((B * n, Cc), dtype=F16, device=ehs.device)
    parts = [(out, ehs)] if ehs.is_contiguous() else [(out[b * n:(b + 1) * n], ehs[b]) for b in range(B)]
    for dst, src in parts:
        if not src.is_contiguous():
            raise ValueError("to_f16_rows: batch entries must be contiguous")
        if src.dtype == torch.float32:
            cast_f16(dst, src)
        elif src.dtype == F16:
            copy_rows(dst, src.reshape(-1, Cc))
        else:
            raise ValueError(f"to_f16_rows: unsupported dtype {src.dtype}")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    capi.check(capi.lib().me_silu(out.data_ptr(), x.data_ptr(), x.numel(), _stream()), "me_silu")
    return out


def relu(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    capi.check(capi.lib().me_relu(out.data_ptr(), x.data_ptr(), x.numel(), _stream()), "me_relu")
    return out


# Device-resident step scalars {t, guidance, ca, cb} (fp32 [4]) while a denoising step is captured into / replayed from a
# hipGraph: the launches below then read them from memory instead of baking this step's values into the captured kernel
# arguments.  Set by pipelines.MotionEditorPipeline.denoise_step_graphed; None = plain scalar arguments.
STEP_PARAMS: Optional[torch.Tensor] = None


def timestep_embed(rows: int, dim: int, t: float, device) -> torch.Tensor:
    out = torch.empty((rows, dim), dtype=F16, device=device)
    if STEP_PARAMS is not None:
        capi.check(capi.lib().me_timestep_embed_dev(out.data_ptr(), rows, dim, STEP_PARAMS.data_ptr(), _stream()), "me_timestep_embed_dev")
        return out
    capi.check(capi.lib().me_timestep_embed(out.data_ptr(), rows, dim, float(t), _stream()), "me_timestep_embed")
    return out


def cfg_ddim(latents: torch.Tensor, eps_rows: torch.Tensor, *, guidance: float, ca: float, cb: float) -> torch.Tensor:
    """latents fp32 [nb, C, f, h, w] (reference layout); eps_rows fp16 [(2nb*f*h*w), >=C] = [uncond | cond]."""
    nb, Cc, f, h, w = latents.shape
    if latents.dtype != torch.float32 or not latents.is_contiguous():
        raise ValueError("cfg_ddim: latents must be contiguous fp32")
    out = torch.empty_like(latents)
    if STEP_PARAMS is not None:
        capi.check(capi.lib().me_cfg_ddim_dev(out.data_ptr(), latents.data_ptr(), eps_rows.data_ptr(), eps_rows.stride(0), nb, Cc, f, h * w,
                                              STEP_PARAMS.data_ptr(), _stream()), "me_cfg_ddim_dev")
        return out
    capi.check(capi.lib().me_cfg_ddim(out.data_ptr(), latents.data_ptr(), eps_rows.data_ptr(), eps_rows.stride(0), nb, Cc, f, h * w,
                                      guidance, ca, cb, _stream()), "me_cfg_ddim")
    return out


def gaussian_sample(moments: torch.Tensor, noise: torch.Tensor, n_img: int, npix: int, scale: float = 1.0) -> torch.Tensor:
    """moments fp16 rows [(n_img*npix), >= 8] = (mean | logvar); noise fp32 [n_img, 4, npix] -> fp32 [n_img, 4, npix]."""
    if noise.dtype != torch.float32 or not noise.is_contiguous():
        raise ValueError("gaussian_sample: noise must be contiguous fp32")
    out = torch.empty((n_img, 4, npix), dtype=torch.float32, device=moments.device)
    capi.check(capi.lib().me_gaussian_sample(out.data_ptr(), moments.data_ptr(), moments.stride(0), noise.data_ptr(), n_img, npix, float(scale), _stream()), "me_gaussian_sample")
    return out


def nchw_to_rows(x: torch.Tensor, n_img: int, Cc: int, npix: int, img_stride: int, ch_stride: int) -> torch.Tensor:
    out = torch.empty((n_img * npix, Cc), dtype=F16, device=x.device)
    capi.check(capi.lib().me_nchw_to_rows(out.data_ptr(), out.stride(0), x.data_ptr(), img_stride, ch_stride, n_img, Cc, npix, _stream()), "me_nchw_to_rows")
    return out


def rows_to_nchw(x: torch.Tensor, n_img: int, Cc: int, npix: int) -> torch.Tensor:
    """fp16 rows [(n_img*npix), >=Cc] -> fp32 [n_img, Cc, npix]."""
    out = torch.empty((n_img, Cc, npix), dtype=torch.float32, device=x.device)
    capi.check(capi.lib().me_rows_to_nchw(out.data_ptr(), Cc * npix, npix, x.data_ptr(), x.stride(0), n_img, Cc, npix, _stream()), "me_rows_to_nchw")
    return out


def nchw5_to_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32 [B, C, f, h, w] (reference layout) -> fp16 rows [(B f h w), C]."""
    B, Cc, f, h, w = x.shape
    x = x.contiguous().float()
    out = torch.empty((B * f * h * w, Cc), dtype=F16, device=x.device)
    for b in range(B):
        capi.check(capi.lib().me_nchw_to_rows(out[b * f * h * w:].data_ptr(), out.stride(0), x[b].data_ptr(), h * w, f * h * w, f, Cc, h * w, _stream()),
                   "me_nchw_to_rows")
    return out


def rows_to_nchw5(rows: torch.Tensor, B: int, Cc: int, f: int, h: int, w: int) -> torch.Tensor:
    """fp16 rows [(B f h w), >=C] -> fp32 [B, C, f, h, w]."""
    out = torch.empty((B, Cc, f, h, w), dtype=torch.float32, device=rows.device)
    for b in range(B):
        capi.check(capi.lib().me_rows_to_nchw(out[b].data_ptr(), h * w, f * h * w, rows[b * f * h * w:].data_ptr(), rows.stride(0), f, Cc, h * w, _stream()),
                   "me_rows_to_nchw")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Backward primitives (the kernel-level contract of motioneditor_amd/autodiff.py; SURVEY.md 8f rank 1 "null-text" and rank 4
# "adapter training").  Their CPU statements live in tests/emu_ops.py and are pinned, through util.null_optimization /
# util.adapter_training_grads, against the reference's own optimisation.  Every entry is a HIP kernel (csrc/bwd.hip, attn_bwd.hip,
# train.hip) or me_gemm itself on transposed weights; torch only allocates and moves data (transposes of constant weights, fp16
# casts of gradients) -- no torch arithmetic, no CPU fallback.  Contract: gradients are fp32 [rows, ld] views of the tape's
# buffers; entries that take `dst` ACCUMULATE into it (+=).  Not differentiated (they raise): edited / masked attention segments,
# shared query items, frame- / pixel-sharded row orders, the temporal editor's kv_map, 3x3-convolution weights, the pad-(0,1,0,1)
# convolution of the VAE encoder.
# ---------------------------------------------------------------------------------------------------------------------
_wT_cache = {}


def _w_transposed(w: torch.Tensor) -> torch.Tensor:
    """[N][taps][K] -> [K][taps reversed][N] (padded to N % 8 == 0): the weights of the input-gradient GEMM / correlation.  Built
    once per FROZEN weight tensor (a transpose of constants at first use) and kept: the backward doubles the weight memory.
    Entries of tensors that are rewritten (trained parameters) are dropped by invalidate_transposed()."""
    key = (w.data_ptr(), tuple(w.shape))
    hit = _wT_cache.get(key)
    if hit is None:
        N, taps, K = w.shape
        n8 = (N + 7) // 8 * 8
        t = torch.zeros((K, taps, n8), dtype=w.dtype, device=w.device)
        t[:, :, :N] = w.flip(1).permute(2, 1, 0)
        hit = _wT_cache[key] = (t.contiguous(), w)   # keep w alive: the key is its address
    return hit[0]


def invalidate_transposed(tensors=None) -> None:
    """Forget the cached transposes of `tensors` (all when None): call after a packed weight has been rewritten in place (the adapter
    training step) or replaced -- a stale transpose would feed old weights to gemm_dx, a dead one pins device memory."""
    if tensors is None:
        _wT_cache.clear()
        return
    for t in tensors:
        _wT_cache.pop((t.data_ptr(), tuple(t.shape)), None)


def _chk_grad(t: torch.Tensor, name: str) -> None:
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA fp32 2-D gradient view with unit column stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")


def grad_acc(dst: torch.Tensor, src: torch.Tensor, alpha: float = 1.0, pool: Optional[Tuple[int, int]] = None, store: bool = False) -> torch.Tensor:
    """dst (fp32 view) += alpha * src (fp32 or fp16, at least dst's rows x cols; pool=(H, W): src holds the 2H x 2W grid whose 2 x 2 blocks
    are summed into dst's H x W pixels).  1-D / n-D contiguous operands are treated as one row.  store=True: dst = alpha * src (dst is not
    read: the first contribution to a gradient buffer that was never zeroed)."""
    if dst.dim() != 2:
        if not (dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel() and dst.numel() % 4 == 0):
            raise ValueError("grad_acc: non-2-D operands must be contiguous, of equal size and a multiple of 4 elements")
        return grad_acc(dst.reshape(1, -1), src.reshape(1, -1), alpha, None, store).reshape(dst.shape)
    _chk_grad(dst, "grad_acc.dst")
    if src.dtype not in (torch.float32, F16) or src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("grad_acc: src must be an fp32 / fp16 2-D view with unit column stride")
    rows, cols = dst.shape
    if src.shape[1] < cols or src.shape[0] < (4 * rows if pool else rows):
        raise ValueError(f"grad_acc: src {tuple(src.shape)} does not cover dst {tuple(dst.shape)}")
    ph, pw = pool if pool else (0, 0)
    capi.check(capi.lib().me_grad_acc(dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), (1 if src.dtype == F16 else 0) | (2 if store else 0), rows, cols, float(alpha), ph, pw,
                                      _stream()), "me_grad_acc")
    return dst


def _f16(dy: torch.Tensor, cols: int) -> torch.Tensor:
    """fp32 gradient view -> contiguous fp16 [rows, cols >= dy.shape[1]] (zero-padded columns): the MFMA operand (me_cast_rows_f16)."""
    if dy.dtype == F16 and dy.shape[1] == cols and dy.stride(1) == 1:
        return dy
    _chk_grad(dy, "gradient")
    d16 = torch.empty((dy.shape[0], cols), dtype=F16, device=dy.device)
    capi.check(capi.lib().me_cast_rows_f16(d16.data_ptr(), d16.stride(0), dy.data_ptr(), dy.stride(0), dy.shape[0], dy.shape[1], cols, _stream()), "me_cast_rows_f16")
    return d16


def gemm_dx(dy, w, *, dst, M, alpha=1.0, conv=None, tconv=None, store=False):
    """dst (fp32 view of dX, [x_rows, K]) += the input gradient of me_gemm's y = alpha * gather(x) @ w^T.  me_gemm itself on the
    [K][taps reversed][N] weights: a dense GEMM, the stride-1 3x3 correlation / TemporalConv with reversed taps; the stride-2
    convolution's input gradient is that correlation over the ZERO-STUFFED dy (gather mode ups = 2), the nearest-upsampled
    convolution's the 2 x 2 block sum of it (me_grad_acc's pooling).  dy arrives loss-scaled and is cast to fp16 like every activation.
    store=True: dst was never written (nor zeroed) -- this product is its first contribution and is stored."""
    N, taps, K = w.shape
    wt = _w_transposed(w)
    d16 = _f16(dy[:M], wt.shape[2])
    pool = None
    if conv is not None:
        Hin, Win, Hout, Wout, stride, ups = conv[:6]
        if len(conv) > 6 and conv[6]:
            raise NotImplementedError("gemm_dx: the pad-(0,1,0,1) convolution (VAE encoder) is not differentiated")
        if stride == 2:      # dX = corr(zero-stuffed dy, reversed taps) at the input resolution
            du = gemm(d16, wt, M=(M // (Hout * Wout)) * Hin * Win, alpha=alpha, conv=(Hout, Wout, Hin, Win, 1, 2))
        else:
            du = gemm(d16, wt, alpha=alpha, conv=(Hout, Wout, Hout, Wout, 1, 0))
            if ups:          # y = conv(nearest2x(x)): every input pixel collects its 2 x 2 block
                pool = (Hin, Win)
    elif tconv is not None:
        if len(tconv) > 3:
            raise NotImplementedError("gemm_dx: the frame-sharded TemporalConv is not differentiated")
        du = gemm(d16, wt, alpha=alpha, tconv=tuple(tconv))
    else:
        du = gemm(d16, wt, alpha=alpha)
    rows = du.shape[0] // 4 if pool else du.shape[0]
    if store and (rows < dst.shape[0] or du.shape[1] < dst.shape[1]):   # the product does not cover the whole (never zeroed) buffer
        dst.zero_()
        store = False
    grad_acc(dst[:rows], du, 1.0, pool, store)
    return dst


def geglu_bwd(pre, dy):
    """d(pre-activation) fp16 [M, N] from dy fp32 [M, N/2], pre in the packed (16 value | 16 gate) column order."""
    _chk2d(pre, "geglu_bwd.pre")
    _chk_grad(dy, "geglu_bwd.dy")
    M, N = pre.shape
    out = empty(M, N, pre)
    capi.check(capi.lib().me_geglu_bwd(out.data_ptr(), out.stride(0), pre.data_ptr(), pre.stride(0), dy.data_ptr(), dy.stride(0), M, N, _stream()), "me_geglu_bwd")
    return out


def attention_fallback_blocks(reset: bool = False) -> int:
    """Blocks of the fixed-offset attention kernels that re-ran with the running maximum since the last reset (me_attn_fallback_blocks): a
    diagnostic for trained checkpoints -- 0 means the speculative fast path (with its per-stage re-basing) held everywhere.  Synchronises."""
    n = capi.lib().me_attn_fallback_blocks(1 if reset else 0)
    if n < 0:
        raise capi.MotionedError("me_attn_fallback_blocks: HIP error")
    return int(n)


def attention_bwd(q, k, v, out, dout, *, dq, dk, dv, lse, heads, dh, n_items, nq, nk, seg_item, seg_mode, mask=None, scale=None, q_items=0):
    """(dq, dk, dv) += the input gradients of me_attn for PLAIN segments ([prev | cur], self, text, [first | prev]): the fused flash-style
    backward (csrc/attn_bwd.hip): P rebuilt per tile from the forward's log-sum-exp `lse`, no score matrix, any key count.  dq / dk / dv are
    fp32 views of the tape's gradient buffers; dout arrives loss-scaled."""
    from . import segments
    if mask is not None or segments.has_dual(seg_mode):
        raise NotImplementedError("attention_bwd: only plain segments are differentiated (the edited / masked attention is not)")
    if q_items:
        raise NotImplementedError("attention_bwd: shared query items are not differentiated")
    if lse is None:
        raise ValueError("attention_bwd: the forward must have stashed its log-sum-exp (attention(..., lse=...))")
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        _chk2d(t, "attention_bwd." + n)
    for t, n in ((dout, "dout"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk_grad(t, "attention_bwd." + n)
    n_kv = k.shape[0] // nk
    inv_ptr, inv_item = segments.inverse(seg_item, n_kv)
    a = AttnBwdArgs()
    a.Q, a.K, a.V, a.O, a.dO, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr()
    a.dQ, a.dK, a.dV = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    delta = torch.empty((n_items * nq, heads), dtype=torch.float32, device=q.device)
    a.delta = delta.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo, a.lddo = q.stride(0), k.stride(0), v.stride(0), out.stride(0), dout.stride(0)
    a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.heads, a.dh, a.n_items, a.nq, a.nk, a.nseg, a.n_kv_items = heads, dh, n_items, nq, nk, seg_item.shape[1], n_kv
    a.seg_item, a.inv_ptr, a.inv_item = seg_item.data_ptr(), inv_ptr.data_ptr(), inv_item.data_ptr()
    a.scale = dh ** -0.5 if scale is None else scale
    capi.check(capi.lib().me_attn_bwd(C.byref(a), _stream()), "me_attn_bwd")


def temporal_attention_bwd(q, k, v, out, dout, *, heads, dh, batch, frames, npix, kv_map=None, scale=None, q_frames=0, q_frame0=0, kv_parts=1, q_parts=1):
    """(dq, dk, dv) fp32 of me_tattn in the plain row order (no editor kv_map, no frame / pixel sharding: the differentiated UNet
    runs un-edited on one GPU)."""
    if (kv_map is not None and list(kv_map) != list(range(batch))) or q_frames or kv_parts > 1 or q_parts > 1:
        raise NotImplementedError("temporal_attention_bwd: editor kv_map / sharded row orders are not differentiated")
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk2d(t, "temporal_attention_bwd." + n)
    _chk_grad(dout, "temporal_attention_bwd.dout")
    rows, C_ = batch * frames * npix, heads * dh
    dq, dk, dv = (torch.empty((rows, C_), dtype=torch.float32, device=q.device) for _ in range(3))
    capi.check(capi.lib().me_tattn_bwd(dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), q.data_ptr(), q.stride(0),
                                       k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), dout.data_ptr(), dout.stride(0), batch, frames, npix, heads, dh,
                                       dh ** -0.5 if scale is None else scale, _stream()), "me_tattn_bwd")
    return dq, dk, dv


def groupnorm_bwd(x, gamma, beta, dy, *, rows_per_group, eps, silu, groups=32):
    _chk2d(x, "groupnorm_bwd.x")
    _chk_grad(dy, "groupnorm_bwd.dy")
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    scratch = _work(capi.lib().me_groupnorm_bwd_scratch_bytes(x.shape[0], rows_per_group, groups), x.device, "gnbwd")
    capi.check(capi.lib().me_groupnorm_bwd(dx.data_ptr(), dx.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), dy.data_ptr(), dy.stride(0),
                                           x.shape[0], rows_per_group, x.shape[1], groups, eps, 1 if silu else 0, scratch.data_ptr(), _stream()), "me_groupnorm_bwd")
    return dx


def layernorm_bwd(x, gamma, dy, *, eps=1e-5):
    _chk2d(x, "layernorm_bwd.x")
    _chk_grad(dy, "layernorm_bwd.dy")
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    capi.check(capi.lib().me_layernorm_bwd(dx.data_ptr(), dx.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(), dy.data_ptr(), dy.stride(0), x.shape[0], x.shape[1],
                                           eps, _stream()), "me_layernorm_bwd")
    return dx


# -- parameter-gradient primitives of the adapter training step (train_adaptor.py:364-368): fp32-accumulating, fp32-output kernels
#    (csrc/train.hip); each ACCUMULATES into `dst`, an fp32 tensor in the packed parameter layout (a view of the trainer's flat bucket).
_scratch = {}


_scratch_retired = []   # outgrown scratch blocks: never freed -- a captured hipGraph may hold their address (me_gemm's split-K `work`)


def _work(nbytes: int, device, tag: str) -> torch.Tensor:
    """Grow-only fp32 scratch per (device, stream, purpose).  A block that a larger request replaces is RETIRED, not freed: its raw pointer may sit in a
    captured step (denoise_step_graphed), whose replays would otherwise write fp32 partial sums into memory the allocator has handed to someone else."""
    key = (str(device), _stream(), tag)
    t = _scratch.get(key)
    if t is None or t.numel() * 4 < nbytes:
        if t is not None:
            _scratch_retired.append(t)
        t = _scratch[key] = torch.empty((nbytes + 3) // 4 + 4, dtype=torch.float32, device=device)
    return t


def scratch_trim() -> int:
    """Free the retired scratch blocks.  Safe only when no captured / recorded step that may hold their addresses is alive -- the caller
    (MotionEditorPipeline.release_plans) has just dropped all of them.  Returns the number of blocks released."""
    n = len(_scratch_retired)
    _scratch_retired.clear()
    return n


def gemm_dw(dy, x, *, dst, taps, K, M, alpha=1.0, conv=None, tconv=None):
    """dst (fp32 [N, taps, K]) += dW of me_gemm's y = alpha * gather(x) @ w^T for dense and TemporalConv layers (the adapter has no 3x3
    convolution): the token axis is the MFMA contraction, per-split fp32 partial tiles are folded in a fixed order."""
    if conv is not None:
        raise NotImplementedError("gemm_dw: 3x3 convolution weights are not trained (the adapter has none)")
    if tconv is not None and len(tconv) > 3:
        raise NotImplementedError("gemm_dw: the frame-sharded TemporalConv is not differentiated")
    N = dy.shape[1]
    if dst.dtype != torch.float32 or not dst.is_contiguous() or tuple(dst.shape) != (N, taps, K):
        raise ValueError(f"gemm_dw: dst must be a contiguous fp32 [{N}, {taps}, {K}] tensor")
    _chk2d(x, "gemm_dw.x")
    a = GemmDwArgs()
    a.dY, a.X, a.dW = dy.data_ptr(), x.data_ptr(), dst.data_ptr()
    a.M, a.N, a.K, a.lddy, a.ldx, a.dy_is_f16 = M, N, K, dy.stride(0), x.stride(0), 1 if dy.dtype == F16 else 0
    a.work = _work(capi.lib().me_gemm_dw_work_bytes(M, N, K), dy.device, "dw").data_ptr()
    a.taps, a.alpha = taps, alpha
    a.gather = capi.GATHER_DENSE
    if tconv is not None:
        a.gather = capi.GATHER_TCONV
        a.frames, a.npix, a.chunk = tconv
    for tap in range(taps):
        a.tap = tap
        capi.check(capi.lib().me_gemm_dw(C.byref(a), _stream()), "me_gemm_dw")
    return dst


def colsum_grad(dy, *, dst, alpha=1.0):
    """dst (fp32 [N]) += alpha * column sums of a gradient (a bias gradient)."""
    M, N = dy.shape
    capi.check(capi.lib().me_colsum(dst.data_ptr(), dy.data_ptr(), dy.stride(0), 1 if dy.dtype == F16 else 0, M, N, float(alpha),
                                    _work(capi.lib().me_colsum_work_bytes(N), dy.device, "colsum").data_ptr(), _stream()), "me_colsum")
    return dst


def relu_bwd(dy, out):
    _chk_grad(dy, "relu_bwd.dy")
    _chk2d(out, "relu_bwd.out")
    dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
    capi.check(capi.lib().me_relu_bwd(dx.data_ptr(), dx.stride(0), dy.data_ptr(), dy.stride(0), out.data_ptr(), out.stride(0), dy.shape[0], dy.shape[1], _stream()),
               "me_relu_bwd")
    return dx


def layernorm_bwd_params(x, dy, *, dgamma=None, dbeta=None, eps=1e-5):
    """dgamma (fp32 [C]) += sum_m dy xhat, dbeta += sum_m dy (either may be None)."""
    _chk2d(x, "layernorm_bwd_params.x")
    _chk_grad(dy, "layernorm_bwd_params.dy")
    M, Cc = x.shape
    capi.check(capi.lib().me_layernorm_bwd_params(_p(dgamma), _p(dbeta), x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), M, Cc, eps, 1.0,
                                                  _work(capi.lib().me_layernorm_bwd_params_work_bytes(M, Cc), x.device, "lnp").data_ptr(), _stream()), "me_layernorm_bwd_params")


# -- loss, norms, optimiser (csrc/train.hip) --
def sumsq_absmax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [2] device tensor {sum x^2, max |x|} of a contiguous fp32 tensor (no host sync)."""
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("sumsq_absmax: contiguous fp32 input")
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=x.device)
    capi.check(capi.lib().me_sumsq_absmax(out.data_ptr(), x.data_ptr(), x.numel(), _work(capi.lib().me_sumsq_work_bytes(), x.device, "sumsq").data_ptr(), _stream()),
               "me_sumsq_absmax")
    return out


def adamw(p, m, v, g, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step: int, gnorm_sq: Optional[torch.Tensor] = None, max_grad_norm: float = 0.0,
          grad_scale: float = 1.0) -> None:
    """One AdamW step in place on contiguous fp32 p, m, v with gradient g (torch.optim.AdamW; weight_decay 0 = Adam); gnorm_sq (device scalar)
    switches on clip_grad_norm_(max_grad_norm) over the bucket it was computed on."""
    for t in (p, m, v, g):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adamw: contiguous fp32 tensors of equal size")
    capi.check(capi.lib().me_adamw(p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   1.0 - beta1 ** step, 1.0 - beta2 ** step, _p(gnorm_sq), float(max_grad_norm), float(grad_scale), _stream()), "me_adamw")


def cast_f16(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst fp16 = src fp32, contiguous and of equal size (packed weights from their fp32 masters)."""
    if dst.dtype != F16 or src.dtype != torch.float32 or not dst.is_contiguous() or not src.is_contiguous() or dst.numel() != src.numel():
        raise ValueError("cast_f16: contiguous fp16 <- fp32 of equal size")
    capi.check(capi.lib().me_cast_f16(dst.data_ptr(), src.data_ptr(), src.numel(), _stream()), "me_cast_f16")
    return dst


def mse_seed(eps_u: torch.Tensor, target: torch.Tensor, *, eps_c: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None, guidance: float = 1.0, ca: float = 0.0,
             cb: float = 1.0, coef: float = 1.0):
    """rec = ca x + cb (eps_u + guidance (eps_c - eps_u)), diff = rec - target (fp32 [nb, C, f, h, w]); returns (diff, d_eps rows fp32 [(nb f h w), C] = coef * diff)."""
    nb, Cc, f, h, w = target.shape
    if target.dtype != torch.float32 or not target.is_contiguous() or (x is not None and (x.dtype != torch.float32 or not x.is_contiguous())):
        raise ValueError("mse_seed: contiguous fp32 latents")
    diff = torch.empty_like(target)
    d_eps = torch.empty((nb * f * h * w, Cc), dtype=torch.float32, device=target.device)
    capi.check(capi.lib().me_mse_seed(diff.data_ptr(), d_eps.data_ptr(), d_eps.stride(0), eps_u.data_ptr(), eps_u.stride(0), _p(eps_c), 0 if eps_c is None else eps_c.stride(0),
                                      _p(x), target.data_ptr(), nb, Cc, f, h * w, float(guidance), float(ca), float(cb), float(coef), _stream()), "me_mse_seed")
    return diff, d_eps
