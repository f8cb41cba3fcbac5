"""Host-side launch graphs of the denoising step: UNet3D, ControlAdapter and ControlNet expressed as
sequences of ``libmotioned`` kernel launches on channels-last fp16 activations.

The reference builds the same graphs out of nn.Modules (models/unet_2d_condition.py:363-546,
models/unet_2d_blocks.py, models/attention_2d.py, models/resnet_2d.py, models/controlnet_adapter.py);
this file keeps their dataflow and nothing else: no nn.Module, no autograd, no torch arithmetic.
Row order everywhere is (batch, frame, pixel), so "(b f) c h w", "(b f) (h w) c" and "(b d) f c"
are the same buffer.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from .. import ops, plan, segments
from ..weights import Packed

HEADS = 8
ADAPTER_CHUNK = 8  # hard-coded num_frames=8 in the reference adapter (controlnet_adapter.py:414,438,472)
HEAD_MAJOR_Q = __import__("os").environ.get("ME_HEAD_MAJOR_Q", "1") != "0"     # ... and Q (me_attn_args.hsq, ABI 8); A/B switch
HEAD_MAJOR_KV = __import__("os").environ.get("ME_HEAD_MAJOR_KV", "1") != "0"   # attn1: K and V leave the fused q|k|v projection as per-head [rows, dh] panels (me_gemm_args.C2 / me_attn_args.hsk); A/B switch
SPLIT_SHARDED_TCONV = __import__("os").environ.get("ME_SPLIT_TCONV", "1") != "0"   # frame-sharded TemporalConv: interior frames behind the posted halo exchange, boundary frames after it
INPLACE_SKIPS = __import__("os").environ.get("ME_INPLACE_SKIPS", "1") != "0"   # the down path writes its skips straight into the up path's concat buffers (unet_forward); A/B switch
LN_FOLD = __import__("os").environ.get("ME_LN_FOLD", "1") != "0"   # LayerNorm folded into the projection that consumes it (me_gemm_args.ln_stats, ABI 9); A/B switch
COND_EMBED_CACHE = True   # ControlNet conditioning embedding of an unchanged skeleton tensor is computed once per run (controlnet_forward)


@dataclass
class Act:
    """Activation: rows [(B*f*h*w), C] + its logical shape."""
    t: torch.Tensor
    B: int
    f: int
    h: int
    w: int
    cat: Optional[torch.Tensor] = None   # a skip that was produced in place: the up path's concat buffer [rows, C_hidden + C] whose right columns `t` is

    @property
    def N(self) -> int:
        return self.h * self.w

    @property
    def C(self) -> int:
        return self.t.shape[1]

    def like(self, t: torch.Tensor, h: Optional[int] = None, w: Optional[int] = None) -> "Act":
        return Act(t, self.B, self.f, self.h if h is None else h, self.w if w is None else w)

    def rows_of(self, b: int) -> torch.Tensor:
        n = self.f * self.N
        return self.t[b * n:(b + 1) * n]


@dataclass
class AttnCall:
    """What an attention editor receives instead of pre-gathered (B*H, n, dh) tensors: the un-gathered
    q/k/v row views plus the geometry needed to pick a key-segment table."""
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    B: int
    f: int
    N: int
    dh: int
    nk: int          # keys per kv item (N for self, 77 for text)
    is_cross: bool
    shard: object = None   # parallel.FrameShard: k/v are then the all-gather of every rank's frames (part-major items)
    q_items: int = 0       # > 0: q holds that many query items; item i reads item i % q_items (queries shared by the CFG halves, unet_forward)

    def run(self, seg_item: torch.Tensor, seg_mode: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.attention(self.q, self.k, self.v, heads=HEADS, dh=self.dh, n_items=self.B * self.f, nq=self.N, nk=self.nk,
                             seg_item=seg_item, seg_mode=seg_mode, mask=mask, **({"q_items": self.q_items} if self.q_items else {}))


@dataclass
class TemporalCall:
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    B: int
    f: int
    N: int
    dh: int
    shard: object = None   # parallel.FrameShard: q holds the local frames, k/v all frames (all-gathered, part-major)
    parts: int = 0         # > 1: after the frame<->pixel all-to-all -- q, k, v hold ALL f frames of N pixels, part-major

    def run(self, kv_map: Optional[Sequence[int]] = None) -> torch.Tensor:
        if self.parts > 1:
            return ops.temporal_attention(self.q, self.k, self.v, heads=HEADS, dh=self.dh, batch=self.B, frames=self.f, npix=self.N, kv_map=kv_map,
                                          kv_parts=self.parts, q_parts=self.parts)
        if self.shard is None:
            return ops.temporal_attention(self.q, self.k, self.v, heads=HEADS, dh=self.dh, batch=self.B, frames=self.f, npix=self.N, kv_map=kv_map)
        sh = self.shard
        return ops.temporal_attention(self.q, self.k, self.v, heads=HEADS, dh=self.dh, batch=self.B, frames=sh.f_total, npix=self.N, kv_map=kv_map,
                                      q_frames=sh.f_loc, q_frame0=sh.frame0, kv_parts=sh.world)


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def conv3x3(P: Packed, name: str, x: Act, stride: int = 1, ups: int = 0, **epi) -> Act:
    """InflatedConv3d 3x3 pad 1 (resnet_2d.py:28-36); stride-2 = Downsample2D (:94-125); ups=1 folds
    Upsample2D's nearest-2x (:77) into the gather."""
    hv, wv = x.h << ups, x.w << ups
    ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
    out = ops.gemm(x.t, P.mat(name + ".weight"), M=x.B * x.f * ho * wo, bias=P.vec(name + ".bias"),
                   conv=(x.h, x.w, ho, wo, stride, ups), **epi)
    return x.like(out, ho, wo)


def resnet_block(P: Packed, p: str, x: Act, temb: torch.Tensor, temb_off: int, *, per_frame_stats: bool, eps: float = 1e-5, shard=None,
                 out: Optional[torch.Tensor] = None) -> Act:
    """ResnetBlock2D.forward (resnet_2d.py:199-249).  GroupNorm statistics span all frames of a batch row
    for the 3-D UNet (5-D GroupNorm, :202,230) and one image for the 2-D ControlNet."""
    rpg = x.N if per_frame_stats else x.f * x.N
    cout = P.vec(p + ".conv1.bias").shape[0]
    tv = temb[:, temb_off:temb_off + cout]
    rows_per_vec = x.t.shape[0] if temb.shape[0] == 1 else x.f * x.N
    has_t1 = P.has(p + ".temp_conv1.weight") and not P.is_zero(p + ".temp_conv1.weight", p + ".temp_conv1.bias")
    has_t2 = P.has(p + ".temp_conv2.weight") and not P.is_zero(p + ".temp_conv2.weight", p + ".temp_conv2.bias")

    gn = {}
    if shard is not None and not per_frame_stats:   # 5-D GroupNorm: statistics span ALL frames -> all-reduce over the frame shards
        gn = dict(reduce=shard.allreduce_, rows_per_group_total=shard.f_total * x.N)
    ftot = shard.f_total if shard is not None else x.f
    rows = x.B * x.f * x.N
    h = ops.groupnorm(x.t, P.vec(p + ".norm1.weight"), P.vec(p + ".norm1.bias"), rows_per_group=rpg, eps=eps, silu=True, **gn)
    if has_t1:
        hx = torch.empty((_ext_rows(x, shard), cout), dtype=P.dtype, device=x.t.device)
        h = conv3x3(P, p + ".conv1", x.like(h), out=hx).t
        # h + temp_conv1(h) + temb  (resnet_2d.py:207-228)
        h = _tconv(P, p + ".temp_conv1", hx, x, ftot, shard, rowvec=tv, rows_per_vec=rows_per_vec, res=h)
    else:
        h = conv3x3(P, p + ".conv1", x.like(h), rowvec=tv, rows_per_vec=rows_per_vec).t
    h = ops.groupnorm(h, P.vec(p + ".norm2.weight"), P.vec(p + ".norm2.bias"), rows_per_group=rpg, eps=eps, silu=True, **gn)
    if P.has(p + ".conv_shortcut.weight"):
        sc = ops.gemm(x.t, P.mat(p + ".conv_shortcut.weight"), bias=P.vec(p + ".conv_shortcut.bias"))
    else:
        sc = x.t
    if has_t2:
        hx = torch.empty((_ext_rows(x, shard), cout), dtype=P.dtype, device=x.t.device)
        h = conv3x3(P, p + ".conv2", x.like(h), out=hx).t
        h = _tconv(P, p + ".temp_conv2", hx, x, ftot, shard, res=h, res2=sc, **({} if out is None else {"out": out}))
    else:
        h = conv3x3(P, p + ".conv2", x.like(h), res=sc, **({} if out is None else {"out": out})).t
    return x.like(h)


def _fold() -> bool:
    """LayerNorms are folded into their projections (no normalised tensor, no LayerNorm launch): not under the autodiff tape (its backward rules differentiate
    the LayerNorm launch) and only on backends that implement it."""
    return LN_FOLD and not getattr(ops, "recording", False) and getattr(ops, "LN_FOLD", False)


def _gemm_st(*args, **kw):
    """ops.gemm that also hands back the partial row sums of its output rows -- (out, stats) -- for the LayerNorm-folded projection that reads them next
    (stats = None when LayerNorms are not folded)."""
    if _fold():
        return ops.gemm(*args, ln_out=True, **kw)
    return ops.gemm(*args, **kw), None


class LN:
    """LayerNorm `norm` of the rows x, not yet applied (nn.LayerNorm in front of attn1 / attn2.to_q / ff / attn_temp, attention_2d.py:493-547): a projection
    takes it folded -- gemm(names): x W'^T mapped through rstd (acc - mean colsum) + cvec, W' = W diag(gamma) -- or, where the normalised rows themselves are
    needed (autodiff tape, the frame<->pixel exchange), as a launch of its own -- rows().  stats: the partial row sums the producer of x left (ops.gemm ln_out)."""

    def __init__(self, P: Packed, norm: str, x: torch.Tensor, stats: Optional[torch.Tensor] = None):
        self.P, self.norm, self.x, self.stats, self._rows = P, norm, x, stats, None

    def rows(self) -> torch.Tensor:
        if self._rows is None:
            self._rows = ops.layernorm(self.x, self.P.vec(self.norm + ".weight"), self.P.vec(self.norm + ".bias"))
        return self._rows

    def gemm(self, names: Sequence[str], *, geglu_bias: Optional[str] = None, **kw):
        names = list(names)
        geglu = geglu_bias is not None
        if _fold() and self._rows is None:
            if self.stats is None:
                self.stats = ops.ln_stats(self.x)
            w, cs, cv = self.P.ln_fold(self.norm, names, geglu=geglu, bias=geglu_bias)
            return ops.gemm(self.x, w, ln=(self.stats, cs, cv, 1e-5), geglu=geglu, **kw)
        if geglu:
            return ops.gemm(self.rows(), self.P.geglu_mat(names[0]), bias=self.P.geglu_vec(geglu_bias), geglu=True, **kw)
        return ops.gemm(self.rows(), self.P.fused(names) if len(names) > 1 else self.P.mat(names[0]), **kw)


def feed_forward(P: Packed, p: str, n: "LN", res: torch.Tensor, want_stats: bool = False):
    """diffusers FeedForward(geglu) + residual: GEGLU fused into the first GEMM's epilogue, norm3 / ff_norm folded into it.  Returns (rows, their partial row
    sums or None)."""
    g = n.gemm([p + ".net.0.proj.weight"], geglu_bias=p + ".net.0.proj.bias")
    return (_gemm_st if want_stats else (lambda *a, **k: (ops.gemm(*a, **k), None)))(g, P.mat(p + ".net.2.weight"), bias=P.vec(p + ".net.2.bias"), res=res)


def _ln(P: Packed, p: str, x: torch.Tensor) -> torch.Tensor:
    return ops.layernorm(x, P.vec(p + ".weight"), P.vec(p + ".bias"))


def _qkv(P: Packed, p: str, n: "LN", C: int, shard, B: int = 0, N: int = 0, head_major: bool = False):
    """q, k, v row views of the self-attention projections of the LayerNorm `n` (folded into them).  Unsharded: one fused [rows, 3C] GEMM.  Frame-sharded:
    q stays local, k|v is projected into a contiguous [rows, 2C] tensor and completed by the shard view (all-gather over
    the frame shards, or the previous rank's last frame only for spatial attn1)."""
    names = [p + ".to_q.weight", p + ".to_k.weight", p + ".to_v.weight"]
    if shard is None:
        if head_major and HEAD_MAJOR_KV and not getattr(ops, "recording", False) and getattr(ops, "HEAD_MAJOR_KV", False):
            # K and V leave the projection as one contiguous [rows, dh] panel per head (me_gemm's second output): what the attention kernel's
            # K/V tile fill wants (2.5 x fewer cache lines than dh-wide slices of 3C-wide rows); Q stays a row tensor
            if HEAD_MAJOR_Q:   # (round 5) Q as panels too: with the heads-slowest block order each XCD reads ONE head's queries -- 80-byte slices of 640-byte rows
                _, qkv = n.gemm(names, head_major=(0, C // HEADS))   # cost 2.4 x their bytes in cache lines per XCD, panels 1 x
                return qkv[:HEADS], qkv[HEADS:2 * HEADS], qkv[2 * HEADS:]
            q, kv = n.gemm(names, head_major=(C, C // HEADS))
            return q, kv[:HEADS], kv[HEADS:]
        qkv = n.gemm(names)
        return qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ext, loc = shard.kv_buffer(n.x.shape[0], 2 * C, B, N, n.x)     # the K|V GEMM writes straight into its slot of the exchanged tensor
    n.gemm(names[1:], out=loc)
    pending = shard.start_kv(ext, B, N, ops.copy_rows)          # the exchange (RCCL stream) runs under the query projection
    q = n.gemm(names[:1])
    kv = shard.finish_kv(pending)
    return q, kv[:, :C], kv[:, C:]


def _temporal_attn(P: Packed, p: str, n: "LN", C: int, B: int, f: int, N: int, dh: int, shard, editor=None, place: str = "") -> torch.Tensor:
    """Causal attention over frames of the projections of `n` (the normed stream, rows (b, local frame, pixel)); returns the
    attention output in the same row order.  Frame-sharded: the rows of `n` go through the frame<->pixel all-to-all BEFORE
    the q|k|v projection (a row-wise GEMM commutes with the row exchange, so C columns travel instead of 3C), every rank
    projects and attends over all frames of its pixel slice, and a second all-to-all returns the output rows
    (parallel.FrameShard.to_pixel_shards); or K|V is all-gathered (shard.temporal == "gather")."""
    def go(tc):
        return editor(call=tc, is_cross=False, place_in_unet=place, num_heads=HEADS) if editor is not None else tc.run()
    if shard is not None and shard.pixel_sharded(N):
        r = shard.to_pixel_shards(n.rows(), B * f, N, ops.copy_blocks)     # (the NORMALISED rows travel: the exchange needs a tensor of its own anyway)
        qkv = ops.gemm(r, P.fused([p + ".to_q.weight", p + ".to_k.weight", p + ".to_v.weight"]))
        a = go(TemporalCall(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, shard.f_total, N // shard.world, dh, None, shard.world))
        return shard.to_frame_shards(a, B * f, N, ops.copy_blocks)
    q, k, v = _qkv(P, p, n, C, shard)
    return go(TemporalCall(q, k, v, B, f, N, dh, shard))


def _tconv(P: Packed, name: str, h_ext: torch.Tensor, x: "Act", chunk: int, shard, **epi) -> torch.Tensor:
    """TemporalConv k=3 over frames.  Sharded: h_ext has 2*B*N spare rows for the neighbours' boundary frames."""
    rows = x.B * x.f * x.N
    if shard is None:
        return ops.gemm(h_ext, P.mat(name + ".weight"), M=rows, bias=P.vec(name + ".bias"), tconv=(x.f, x.N, chunk), **epi)
    if SPLIT_SHARDED_TCONV and x.f >= 3 and getattr(ops, "ROW_RANGE", False) and not ops.gemm_splits_k(rows, P.mat(name + ".weight").shape[0], h_ext.shape[1], 3):
        # (launches that me_gemm splits along K -- the 8 x 8-latent level -- keep the one-launch form: a row-range piece is never split, and the
        # summation orders must agree for the sharded step to stay bitwise equal to the plain one)
        # Interior / boundary split: the halo exchange is POSTED, the frames that need no remote data (1 .. f - 2 of every batch entry) are computed
        # while it travels, then the exchange is joined and the boundary frames follow -- consecutive batch entries' (last, first) frames are
        # neighbouring rows and share a launch.  Same arithmetic per output row as the one-launch form (bitwise: the kernels accumulate taps and
        # channel slabs in one order), 2 B + 1 launches instead of 1.
        w, b = P.mat(name + ".weight"), P.vec(name + ".bias")
        out = epi.pop("out", None)
        if out is None:
            out = torch.empty((rows, w.shape[0]), dtype=h_ext.dtype, device=h_ext.device)
        handle = shard.exchange_halos(h_ext, x.B, x.N, ops.copy_rows, defer=True)
        interior = (x.f, x.N, chunk, shard.frame0, shard.f_total, -1, -1)      # (never reads a halo row)
        for bi in range(x.B):
            ops.gemm(h_ext, w, M=rows, bias=b, tconv=interior, out=out, row_range=((bi * x.f + 1) * x.N, (bi * x.f + x.f - 1) * x.N), **epi)
        hp, hn = shard.finish_halos(handle)
        full = (x.f, x.N, chunk, shard.frame0, shard.f_total, hp, hn)
        bounds = [(0, x.N)] + [((bi * x.f - 1) * x.N, (bi * x.f + 1) * x.N) for bi in range(1, x.B)] + [(rows - x.N, rows)]
        for lo, hi in bounds:
            ops.gemm(h_ext, w, M=rows, bias=b, tconv=full, out=out, row_range=(lo, hi), **epi)
        return out
    hp, hn = shard.exchange_halos(h_ext, x.B, x.N, ops.copy_rows)
    return ops.gemm(h_ext, P.mat(name + ".weight"), M=rows, bias=P.vec(name + ".bias"),
                    tconv=(x.f, x.N, chunk, shard.frame0, shard.f_total, hp, hn), **epi)


def _ext_rows(x: "Act", shard) -> int:
    return x.B * x.f * x.N + (2 * x.B * x.N if shard is not None else 0)


def basic_block(P: Packed, p: str, x: Act, text: Optional[torch.Tensor], text_seg, *, spatial, temporal, place: str,
                sc_attn: bool, has_temp: bool, shard=None, text_kv: Optional[torch.Tensor] = None, expand: int = 1, stats: Optional[torch.Tensor] = None) -> Act:
    """BasicTransformerBlock.forward (attention_2d.py:493-547) on rows [(B f N), C].
    expand > 1 (unet_forward's CFG prefix): x holds B / expand distinct batch entries -- entry b + k B / expand of the full batch would be a
    bit-for-bit copy of entry b up to the text cross-attention -- so attn1 runs once per distinct entry, attn2 reads the shared queries
    (q_items) against every entry's own text keys, and its output projection adds the shared residual (res_rows): from there on the batch is full.
    Round 6: the four LayerNorms are folded into the projections behind them (class LN); `stats` / `st` = the partial row sums of the stream, which every
    projection that writes the stream leaves behind for the next fold (ops.gemm ln_out) -- no LayerNorm launch, no normalised tensor."""
    t, C = x.t, x.C
    dh = C // HEADS
    # --- attn1 (MotionFrameAttention / patched closure, attention_2d.py:705-768, fully_control_utils.py:113-161)
    # plain per-frame self-attention (ControlNet, normal_infer) needs no other frames; [prev | cur] needs ONE halo frame
    sh1 = shard.prev_frame_view() if (shard is not None and sc_attn) else None
    q, k, v = _qkv(P, p + ".attn1", LN(P, p + ".norm1", t, stats), C, sh1, x.B, x.N, head_major=True)
    call = AttnCall(q, k, v, x.B, x.f, x.N, dh, x.N, False, sh1)
    if spatial is not None:
        a = spatial(call=call, is_cross=False, place_in_unet=place, num_heads=HEADS)
    elif sc_attn:
        a = call.run(*segments.prev_cur(x.B, x.f, t.device, sh1))
    else:
        a = call.run(*segments.self_items(x.B * x.f, t.device))
    t, st = _gemm_st(a, P.mat(p + ".attn1.to_out.0.weight"), bias=P.vec(p + ".attn1.to_out.0.bias"), res=t)
    # --- attn2 (CrossAttention, attention_2d.py:115-201): K/V projected once per text row
    if expand > 1 and text is None:
        raise ValueError("basic_block: expand needs the text cross-attention (that is where the batch entries start to differ)")
    if text is not None:
        q = LN(P, p + ".norm2", t, st).gemm([p + ".attn2.to_q.weight"])
        # k | v of the text rows: this block's column slice of the one GEMM that projects the text for every layer (text_kv_all), or its own
        kv = text_kv if text_kv is not None else ops.gemm(text, P.fused([p + ".attn2.to_k.weight", p + ".attn2.to_v.weight"]))
        call = AttnCall(q, kv[:, :C], kv[:, C:], x.B * expand, x.f, x.N, dh, 77, True, None, x.B * x.f if expand > 1 else 0)
        if spatial is not None:
            a = spatial(call=call, is_cross=True, place_in_unet=place, num_heads=HEADS, text_seg=text_seg)
        else:
            a = call.run(*text_seg)
        if expand > 1:
            # from here on the launches hold the full batch -- the attention output `a` already does: me_gemm selects their kernels by their own row count again
            # (unet_forward set the scale for the shared sub-batch; the 8-phase and the 128-row kernels leave different partial row sums behind and round a
            # LayerNorm-folded projection differently, so the choice has to be the full launch's.  Round 6: the reset stood BEHIND this projection, which then
            # chose as for twice the batch -- harmless until the 192-row 8-phase kernel took grids that small)
            ops.SELECT_ROWS_SCALE = 1
        t, st = _gemm_st(a, P.mat(p + ".attn2.to_out.0.weight"), bias=P.vec(p + ".attn2.to_out.0.bias"), res=t, **({"res_rows": t.shape[0]} if expand > 1 else {}))
        if expand > 1:
            x = Act(t, x.B * expand, x.f, x.h, x.w)
    # --- feed-forward (attention_2d.py:531)
    t, st = feed_forward(P, p + ".ff", LN(P, p + ".norm3", t, st), t, want_stats=has_temp)
    # --- temporal attention over frames, causal (attention_2d.py:534-545)
    if has_temp:
        a = _temporal_attn(P, p + ".attn_temp", LN(P, p + ".norm_temp", t, st), C, x.B, x.f, x.N, dh, shard, temporal, place)
        t = ops.gemm(a, P.mat(p + ".attn_temp.to_out.0.weight"), bias=P.vec(p + ".attn_temp.to_out.0.bias"), res=t)
    return x.like(t)


def transformer2d(P: Packed, p: str, x: Act, text, text_seg, *, spatial=None, temporal=None, place: str = "", sc_attn: bool = True,
                  has_temp: bool = True, shard=None, out: Optional[torch.Tensor] = None, text_kv: Optional[dict] = None, expand: int = 1) -> Act:
    """Transformer2DModel.forward (attention_2d.py:338-389): per-frame GroupNorm(32, eps 1e-6), 1x1 proj in/out.
    out: where the block's result is written (a column slice of the next skip-concat buffer)."""
    n = ops.groupnorm(x.t, P.vec(p + ".norm.weight"), P.vec(p + ".norm.bias"), rows_per_group=x.N, eps=1e-6, silu=False)
    t, st = _gemm_st(n, P.mat(p + ".proj_in.weight"), bias=P.vec(p + ".proj_in.bias"))
    t = basic_block(P, p + ".transformer_blocks.0", x.like(t), text, text_seg, spatial=spatial, temporal=temporal, place=place,
                    sc_attn=sc_attn, has_temp=has_temp, shard=shard, text_kv=None if text_kv is None else text_kv[p], expand=expand, stats=st).t
    y = ops.gemm(t, P.mat(p + ".proj_out.weight"), bias=P.vec(p + ".proj_out.bias"), res=x.t, **({} if out is None else {"out": out}),
                 **({"res_rows": x.t.shape[0]} if expand > 1 else {}))
    return Act(y, x.B * expand, x.f, x.h, x.w)


# ---------------------------------------------------------------------------------------------
# time embedding: sinusoid -> MLP, then EVERY resnet's time_emb_proj in one GEMM
# ---------------------------------------------------------------------------------------------
def resnet_names(n_up: bool) -> List[str]:
    names = [f"down_blocks.{i}.resnets.{j}" for i in range(4) for j in range(2)] + ["mid_block.resnets.0", "mid_block.resnets.1"]
    if n_up:
        names += [f"up_blocks.{i}.resnets.{j}" for i in range(4) for j in range(3)]
    return names


def time_embedding(P: Packed, t: float, names: List[str], device):
    """Timesteps(320, flip_sin_to_cos, shift 0) -> linear_1 -> SiLU -> linear_2 (unet_2d_condition.py:432-438),
    then SiLU -> concat(time_emb_proj) (resnet_2d.py:211).  One row: every batch entry shares t."""
    e = ops.timestep_embed(1, 320, t, device)
    e = ops.gemm(e, P.mat("time_embedding.linear_1.weight"), bias=P.vec("time_embedding.linear_1.bias"), act=2)
    e = ops.gemm(e, P.mat("time_embedding.linear_2.weight"), bias=P.vec("time_embedding.linear_2.bias"), act=2)  # silu(emb) for the resnets
    w = P.fused([n + ".time_emb_proj.weight" for n in names])
    b = P.fused_vec([n + ".time_emb_proj.bias" for n in names])
    temb = ops.gemm(e, w, bias=b)
    offs, o = {}, 0
    for n in names:
        offs[n] = o
        o += P.vec(n + ".conv1.bias").shape[0]
    return temb, offs


def attention_block_names(n_up: bool) -> List[str]:
    names = [f"down_blocks.{i}.attentions.{j}" for i in range(3) for j in range(2)] + ["mid_block.attentions.0"]
    if n_up:
        names += [f"up_blocks.{i}.attentions.{j}" for i in range(1, 4) for j in range(3)]
    return names


def text_kv_all(P: Packed, text: torch.Tensor, names: List[str]) -> dict:
    """attn2.to_k | attn2.to_v of EVERY transformer block applied to the text rows in one GEMM (the reference projects the text in each
    CrossAttention.forward, attention_2d.py:146-147, once per frame): 16 (UNet) / 7 (ControlNet) launches on [B*77, 768] become one.
    Returns block name -> its [rows, 2C] column slice."""
    ws = []
    for n in names:
        t = n + ".transformer_blocks.0.attn2."
        ws += [t + "to_k.weight", t + "to_v.weight"]
    kv = ops.gemm(text, P.fused(ws))
    out, o = {}, 0
    for n in names:
        c2 = 2 * P.vec(n + ".norm.weight").shape[0]
        out[n] = kv[:, o:o + c2]
        o += c2
    return out


# ---------------------------------------------------------------------------------------------
# ControlAdapter (controlnet_adapter.py:437-565)
# ---------------------------------------------------------------------------------------------
def adapter_block(P: Packed, p: str, x: Act, src, nb: Optional[int] = None, shard=None) -> torch.Tensor:
    """ResnetBlock.forward (controlnet_adapter.py:497-534).  x: ControlNet residual rows [(b t N), C];
    src: UNet edit-branch skip rows [(nb t N), C], or a list of nb row tensors [(t N), C] (the edit rows of the batch-4 skip, read in
    place).  Returns motion residual rows [(nb t N), C].

    nb > x.B (x.B == 1): the ControlNet residual is SHARED by the nb batch entries (the reference feeds the
    ControlNet the same edit latent twice, see pipelines.MotionEditorPipeline.dedup_controlnet).  Everything up to
    the pose cross-attention depends on x only -- temporal convs, sparse-causal self-attention, cross_pose_norm,
    the pose query projection -- so it is computed once and READ by every batch entry: the attention kernel takes the shared
    queries (q_items), the GEMM epilogues the shared residuals (res_rows / res2_rows); nothing is copied."""
    t, C, dev = x.t, x.C, x.t.device
    dh = C // HEADS
    nb = x.B if nb is None else nb
    share = nb != x.B
    if share:
        assert x.B == 1
    rows_x = t.shape[0]
    # conv path: TemporalConv(k=3) -> ReLU -> TemporalConv(k=1) -> + x, on independent chunks of 8 frames
    if shard is None:
        tx = t
    else:   # chunks of 8 global frames may straddle ranks: boundary frames of the conv input come from the neighbours
        tx = torch.empty((_ext_rows(x, shard), C), dtype=P.dtype, device=dev)
        ops.copy_rows(tx[:t.shape[0]], t)
    hc = _tconv(P, p + ".block1", tx, x, ADAPTER_CHUNK, shard, act=1)
    hc = ops.gemm(hc, P.mat(p + ".block2.weight"), bias=P.vec(p + ".block2.bias"), res=t)
    # sparse-causal self attention inside chunks of 8 frames
    sh2 = shard.chunk_view(ADAPTER_CHUNK) if (shard is not None and shard.adapter == "halo") else shard   # two halo frames, not the all-gather
    q_, k_, v_ = _qkv(P, p + ".attn_temp", LN(P, p + ".norm_temp", t), C, sh2, x.B, x.N)
    a = AttnCall(q_, k_, v_, x.B, x.f, x.N, dh, x.N, False, sh2).run(*segments.first_prev_chunked(x.B, x.f, ADAPTER_CHUNK, dev, sh2))
    a = ops.gemm(a, P.mat(p + ".attn_temp.to_out.0.weight"), bias=P.vec(p + ".attn_temp.to_out.0.bias"), res=t)
    a = _ln(P, p + ".cross_pose_norm", a)  # the normed tensor replaces the stream (controlnet_adapter.py:518)
    # pose x UNet-feature cross attention, per frame
    q = ops.gemm(a, P.mat(p + ".attn_pose.to_q.weight"))
    wkv = P.fused([p + ".attn_pose.to_k.weight", p + ".attn_pose.to_v.weight"])
    if isinstance(src, (list, tuple)):   # the edit rows of the UNet skip, projected where they lie: one GEMM per batch entry, no gather copy
        n1 = src[0].shape[0]
        kv = torch.empty((len(src) * n1, 2 * C), dtype=P.dtype, device=dev)
        for k, part in enumerate(src):
            ops.gemm(part, wkv, out=kv[k * n1:(k + 1) * n1])
    else:
        kv = ops.gemm(src, wkv)
    seg = segments.self_items(nb * x.f, dev)
    ap = ops.attention(q, kv[:, :C], kv[:, C:], heads=HEADS, dh=dh, n_items=nb * x.f, nq=x.N, nk=x.N, seg_item=seg[0], seg_mode=seg[1],
                       **({"q_items": x.B * x.f} if share else {}))
    a, st = _gemm_st(ap, P.mat(p + ".attn_pose.to_out.0.weight"), bias=P.vec(p + ".attn_pose.to_out.0.bias"), res=a, **({"res_rows": rows_x} if share else {}))
    a, st = feed_forward(P, p + ".ff", LN(P, p + ".ff_norm", a, st), a, want_stats=True)
    # causal temporal attention over the TRUE frame count
    at = _temporal_attn(P, p + ".attn_self_temp", LN(P, p + ".norm_self_temp", a, st), C, nb, x.f, x.N, dh, shard)
    return ops.gemm(at, P.mat(p + ".attn_self_temp.to_out.0.weight"), bias=P.vec(p + ".attn_self_temp.to_out.0.bias"), res=a, res2=hc,
                    **({"res2_rows": rows_x} if share else {}))


def adapter_pack(P: Packed, prefix: str = "controlnet_adapter.") -> List[str]:
    """Build (in the order and the fusions adapter_block uses them) every packed tensor of the adapter and return their cache keys:
    what util.AdapterTrainer trains.  A gradient that reaches a packed tensor outside this list is reported by the trainer."""
    for i in range(12):
        p = f"{prefix}body.{i}"
        P.mat(p + ".block1.weight"), P.vec(p + ".block1.bias"), P.mat(p + ".block2.weight"), P.vec(p + ".block2.bias")
        for a in ("attn_temp", "attn_self_temp"):
            P.fused([f"{p}.{a}.to_q.weight", f"{p}.{a}.to_k.weight", f"{p}.{a}.to_v.weight"])
        P.mat(p + ".attn_pose.to_q.weight")
        P.fused([p + ".attn_pose.to_k.weight", p + ".attn_pose.to_v.weight"])
        for a in ("attn_temp", "attn_pose", "attn_self_temp"):
            P.mat(f"{p}.{a}.to_out.0.weight"), P.vec(f"{p}.{a}.to_out.0.bias")
        for n in ("norm_temp", "cross_pose_norm", "ff_norm", "norm_self_temp"):
            P.vec(f"{p}.{n}.weight"), P.vec(f"{p}.{n}.bias")
        P.geglu_mat(p + ".ff.net.0.proj.weight"), P.geglu_vec(p + ".ff.net.0.proj.bias"), P.mat(p + ".ff.net.2.weight"), P.vec(p + ".ff.net.2.bias")
    return sorted(P.trainable_ids(prefix).values())


# ---------------------------------------------------------------------------------------------
# UNet3D (unet_2d_condition.py:363-546)
# ---------------------------------------------------------------------------------------------
DOWN_HAS_ATTN = (True, True, True, False)
UP_HAS_ATTN = (False, True, True, True)


def text_rows(ehs: torch.Tensor, dtype=torch.float16) -> torch.Tensor:
    """[B, 77, 768] (any float dtype, device) -> fp16 rows [B*77, 768]."""
    if ehs.is_cuda and dtype == torch.float16 and getattr(ops, "NATIVE", False) and ehs.dtype in (torch.float16, torch.float32):
        return ops.to_f16_rows(ehs)          # library cast / copy: a recorded step (plan.py) holds no torch kernel
    plan.torch_fallback("text_rows")
    return ehs.to(dtype).reshape(-1, ehs.shape[-1]).contiguous()


def unet_forward(P: Packed, sample: torch.Tensor, t: float, ehs: torch.Tensor, *, down_res: Optional[Sequence[torch.Tensor]] = None,
                 mid_res: Optional[torch.Tensor] = None, two_branch: bool = False, spatial=None, temporal=None,
                 taps: Optional[dict] = None, shard=None, normal_infer: bool = False, res_ready=None, side_stream=None, cfg_dup: bool = False) -> Act:
    """sample: fp32 [B,4,f,h,w] (reference layout).  down_res: 12 row tensors [(2 f N_i), C_i] (two_branch,
    ControlNet batch = the two edit rows) or [(B f N_i), C_i]; mid_res rows [(2|B f N_3), 1280].
    Returns eps rows [(B f N), 4] as an Act.

    cfg_dup: the caller guarantees sample[B/2:] is a copy of sample[:B/2] (the pipeline's `torch.cat([latents] * 2)` for classifier-free
    guidance, pipeline_motion_editor.py:605).  The two halves then differ only through the text embedding, which first enters at the
    cross-attention of down_blocks.0.attentions.0: conv_in, the first resnet, that block's GroupNorm / proj_in / self-attention and the
    cross-attention's query projection are bit-for-bit the same work for both halves and are run ONCE (B / 2 entries); the rest of the graph
    sees the full batch.  The outputs are bitwise those of the un-shared graph (every kernel treats batch entries independently)."""
    B, _, f, h, w = sample.shape
    dev = sample.device
    sample = sample.contiguous().float()
    edit_rows = [b for b in range(B) if b % 2 == 1]   # batch = (recon, edit) pairs: 4 rows, or 2 on one CFG-parallel rank
    temb, toff = time_embedding(P, t, resnet_names(True), dev)
    text = text_rows(ehs, P.dtype)
    tseg = segments.cross_text(B, f, dev)
    # normal_infer (DDIM inversion, inference.py:292): attn1 = plain per-frame self-attention (attention_2d.py:770-777)
    kw = dict(spatial=spatial, temporal=temporal, shard=shard, sc_attn=not normal_infer,   # shard: this rank holds f = f_total / world consecutive frames
              text_kv=text_kv_all(P, text, attention_block_names(True)))

    # the CFG prefix (docstring): only when the spatial editor leaves the first self-attention alone (it does for start_layer > 0), un-sharded, not recording
    # (round 6: frame-sharded too -- `frames` mode holds the full batch of 4 on every rank; the prefix's exchanges then run at half the batch)
    share = (cfg_dup and B % 2 == 0 and DOWN_HAS_ATTN[0] and not getattr(ops, "recording", False) and
             (spatial is None or (hasattr(spatial, "edits_next_self_attention") and not spatial.edits_next_self_attention())))
    Bp = B // 2 if share else B
    x = Act(ops.conv_small(sample, P.mat32("conv_in.weight"), P.vec32("conv_in.bias"), n_img=Bp * f, Cin=4, H=h, Wd=w,
                           img_stride=4 * f * h * w, ch_stride=f * h * w, frames=f, frame_stride=h * w), Bp, f, h, w)
    # Adapter block i needs skip i and ControlNet residual i only, and its output is consumed by the UP path: with a side
    # stream (the one ControlNet ran on, so the residuals are ordered) every block is enqueued there as soon as its skip
    # exists and runs beside the rest of the down path and the mid block; the skips themselves are updated after the
    # down path (main may still be reading them), and the up path waits on one event.
    # Frame-sharded (round 6, `--shard-overlap`): the adapter's own exchanges need their own communicator beside the main stream's (parallel.FrameShard.side_shard)
    side = side_stream if (side_stream is not None and down_res is not None and taps is None and sample.is_cuda and
                           (shard is None or getattr(shard, "side_shard", None) is not None)) else None
    ashard = shard.side_shard if (shard is not None and side is not None) else shard     # the shard view the adapter's exchanges go through
    main = torch.cuda.current_stream() if side is not None else None
    motion: List[torch.Tensor] = []

    def adapter_for(i: int, s: Act) -> torch.Tensor:
        r = down_res[i]
        if two_branch:   # adapter sees the edit rows only (unet_2d_condition.py:479-481): read in place, batch entry by batch entry
            n = s.f * s.N
            shared = r.shape[0] == n and len(edit_rows) > 1   # one ControlNet entry shared by all edit rows
            return adapter_block(P, f"controlnet_adapter.body.{i}", Act(r, 1 if shared else len(edit_rows), s.f, s.h, s.w), [s.rows_of(eb) for eb in edit_rows],
                                 len(edit_rows), ashard)
        return adapter_block(P, f"controlnet_adapter.body.{i}", Act(r, s.B, s.f, s.h, s.w), s.t, None, ashard)   # (unet_2d_condition.py:483-485)

    def push_skip(s: Act) -> None:
        skips.append(s)
        if side is not None:
            plan.share(s.t, side)
            plan.wait_stream(side, main)
            with torch.cuda.stream(side):
                motion.append(adapter_for(len(skips) - 1, s))

    skips: List[Act] = []
    # Skips produced in place (exact; -12 copies of 16 ... 252 MB per step): `torch.cat([hidden, skip], 1)` of up resnet (i, j) (unet_2d_blocks.py) is one
    # buffer [rows, C_hidden + C_skip]; its right columns are allocated HERE, on the way down, and the op that produces skip k -- the 12 skips are consumed
    # in reverse order, skip k by up resnet number 11 - k -- writes them directly (a strided `out=`), the next down layer, the adapter and the motion
    # update read / update them where they lie, and the up path later writes the hidden half beside them.  Not while sharded / recording a tape / tapping
    # (other allocation rules), not for skip 0 without the CFG prefix (conv_small has no strided output) and not for the last skip when the adapter
    # updates it (it is also the mid block's un-updated input: its updated clone goes into the slot instead, below).
    inplace_skips = INPLACE_SKIPS and shard is None and taps is None and not getattr(ops, "recording", False)

    def skip_slot(c_skip: int, rows: int):
        """(concat buffer, its right-hand column view) for the skip about to be produced, or (None, None)."""
        k = len(skips)
        if not inplace_skips or (k == 11 and down_res is not None):
            return None, None
        c_all = P.mat(f"up_blocks.{(11 - k) // 3}.resnets.{(11 - k) % 3}.conv1.weight").shape[2]
        buf = torch.empty((rows, c_all), dtype=P.dtype, device=dev)
        return buf, buf[:, c_all - c_skip:]

    if share:   # skip 0 (the up path's last concat and the adapter's first block read all B entries): the shared rows, twice
        buf, full = skip_slot(x.C, B * f * h * w)
        if full is None:
            full = torch.empty((B * f * h * w, x.C), dtype=x.t.dtype, device=dev)
        ops.copy_rows(full[:x.t.shape[0]], x.t)
        ops.copy_rows(full[x.t.shape[0]:], x.t)
        push_skip(Act(full, B, f, h, w, buf))
    else:
        push_skip(x)
    for i in range(4):
        for j in range(2):
            n = f"down_blocks.{i}.resnets.{j}"
            prefix = share and i == 0 and j == 0
            if prefix:   # the shared sub-batch: me_gemm picks its kernels as for the full batch, so that the rows are bitwise those of the duplicated execution
                ops.SELECT_ROWS_SCALE = 2
            try:
                buf, slot = skip_slot(P.vec(n + ".conv1.bias").shape[0], x.t.shape[0] * (2 if prefix else 1))
                x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=False, shard=shard, out=None if DOWN_HAS_ATTN[i] else slot)
                if DOWN_HAS_ATTN[i]:
                    x = transformer2d(P, f"down_blocks.{i}.attentions.{j}", x, text, tseg, place="down", expand=2 if prefix else 1, out=slot, **kw)
            finally:
                if prefix:
                    ops.SELECT_ROWS_SCALE = 1
            x.cat = buf
            push_skip(x)
        if i < 3:
            buf, slot = skip_slot(x.C, x.t.shape[0] // 4)
            x = conv3x3(P, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2, **({} if slot is None else {"out": slot}))
            x.cat = buf
            push_skip(x)
    if taps is not None:
        taps["skips"] = [s.t.clone() for s in skips]

    adapter_done = None
    if down_res is not None:
        if side is not None:
            plan.wait_stream(side, main)         # main is past its last read of every skip: safe to add the motion in place
        elif res_ready is not None:              # ControlNet residuals produced on another stream
            plan.wait_event(torch.cuda.current_stream(), res_ready)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if side is None:
                motion = [adapter_for(i, s) for i, s in enumerate(skips)]
            if taps is not None:
                taps["motion"] = [m.clone() for m in motion]
            new_skips = []
            for i, (s, m) in enumerate(zip(skips, motion)):
                tgt = s
                if i == len(skips) - 1:  # the last skip is also the mid block's input: keep that one un-modified
                    if inplace_skips:    # ... and put its updated clone where the up path's first concat wants it
                        c_all = P.mat("up_blocks.0.resnets.0.conv1.weight").shape[2]
                        buf = torch.empty((s.t.shape[0], c_all), dtype=P.dtype, device=dev)
                        tgt = s.like(ops.copy_rows(buf[:, c_all - s.C:], s.t))
                        tgt.cat = buf
                    else:
                        tgt = s.like(ops.clone_rows(s.t) if getattr(ops, "NATIVE", False) and s.t.is_cuda else s.t.clone())
                if two_branch:           # [0, m0, 0, m1] (unet_2d_condition.py:481)
                    n = s.f * s.N
                    for k, eb in enumerate(edit_rows):
                        ops.axpy_rows(tgt.rows_of(eb), tgt.rows_of(eb), m[k * n:(k + 1) * n])
                elif getattr(ops, "recording", False):   # differentiated run (autodiff tape): single assignment, the un-updated skip stays readable
                    tgt = s.like(ops.axpy_rows(torch.empty_like(s.t), s.t, m))
                else:
                    ops.axpy_rows(tgt.t, tgt.t, m)
                new_skips.append(tgt)
            skips = new_skips
        if side is not None:
            adapter_done = plan.record_event(side)
            plan.share(skips[-1].t, main)        # the cloned last skip was allocated on the side stream

    n = "mid_block.resnets.0"
    x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=False, shard=shard)
    x = transformer2d(P, "mid_block.attentions.0", x, text, tseg, place="mid", **kw)
    n = "mid_block.resnets.1"
    x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=False, shard=shard)
    if mid_res is not None:
        if two_branch:
            nr = x.f * x.N
            for k, eb in enumerate(edit_rows):
                mk = mid_res[:nr] if mid_res.shape[0] == nr else mid_res[k * nr:(k + 1) * nr]   # shared or per-entry residual
                ops.axpy_rows(x.rows_of(eb), x.rows_of(eb), mk)
        elif getattr(ops, "recording", False):
            x = x.like(ops.axpy_rows(torch.empty_like(x.t), x.t, mid_res))
        else:
            ops.axpy_rows(x.t, x.t, mid_res)
    if taps is not None:
        taps["mid"] = x.t.clone()

    if adapter_done is not None:
        plan.wait_event(torch.cuda.current_stream(), adapter_done)
    cat = None   # torch.cat([hidden, res], dim=1) of the coming resnet, when its hidden half has already been written in place
    for i in range(4):
        for j in range(3):
            s = skips.pop()
            if cat is None:     # the first concat: the mid block's output is copied in (a level-3 tensor)
                cat = s.cat if s.cat is not None else torch.empty((x.t.shape[0], x.C + s.C), dtype=P.dtype, device=dev)
                ops.copy_rows(cat[:, :x.C], x.t)
            if s.cat is None:   # a skip that was not produced in place
                ops.copy_rows(cat[:, x.C:], s.t)
            elif s.cat is not cat or cat.shape[1] != x.C + s.C:
                raise RuntimeError("unet_forward: a skip's concat buffer does not match the up path's layout")
            # the LAST op of this block (resnet / transformer / upsampler) writes its result straight into the left columns of
            # the next block's concat buffer: one copy per concat instead of two
            n = f"up_blocks.{i}.resnets.{j}"
            cout = P.vec(n + ".conv1.bias").shape[0]
            ups = j == 2 and i < 3
            nxt = None
            if skips:
                nxt = skips[-1].cat     # the next skip sits in its concat buffer already: this block's result goes into the columns beside it
                if nxt is None:
                    nxt = torch.empty((x.t.shape[0] * (4 if ups else 1), cout + skips[-1].C), dtype=P.dtype, device=dev)
                elif nxt.shape != (x.t.shape[0] * (4 if ups else 1), cout + skips[-1].C):
                    raise RuntimeError("unet_forward: a skip's concat buffer does not match the up path's layout")
            tgt = None if nxt is None else nxt[:, :cout]
            last = "ups" if ups else ("attn" if UP_HAS_ATTN[i] else "res")
            x = resnet_block(P, n, x.like(cat), temb, toff[n], per_frame_stats=False, shard=shard, out=tgt if last == "res" else None)
            if UP_HAS_ATTN[i]:
                x = transformer2d(P, f"up_blocks.{i}.attentions.{j}", x, text, tseg, place="up", out=tgt if last == "attn" else None, **kw)
            if ups:
                x = conv3x3(P, f"up_blocks.{i}.upsamplers.0.conv", x, ups=1, **({} if tgt is None else {"out": tgt}))
            cat = nxt
    gn = dict(reduce=shard.allreduce_, rows_per_group_total=shard.f_total * x.N) if shard is not None else {}
    y = ops.groupnorm(x.t, P.vec("conv_norm_out.weight"), P.vec("conv_norm_out.bias"), rows_per_group=x.f * x.N, eps=1e-5, silu=True, **gn)
    return conv3x3(P, "conv_out", x.like(y))


# ---------------------------------------------------------------------------------------------
# ControlNet (diffusers 0.15.1 ControlNetModel.forward, called pipeline_motion_editor.py:618-625)
# ---------------------------------------------------------------------------------------------
def controlnet_forward(P: Packed, latents: torch.Tensor, lat_index: Sequence[int], t: float, prompt: torch.Tensor, cond: torch.Tensor,
                       scale: float = 1.0, row_offset: int = 0):
    """latents fp32 [nb,4,f,h,w]; lat_index: which latent row each ControlNet batch entry reads (the
    pipeline feeds rows [1,3] of cat([latents]*2), i.e. the edit latent twice); prompt [n_text,77,768]
    with the reference's interleave (row r -> text r % n_text); cond fp32/fp16 [(nbc f),3,8h,8w].
    Returns (12 down residual row tensors [(nbc f N_i), C_i], mid rows), 2-D per-frame semantics."""
    nb, _, f, h, w = latents.shape
    nbc = len(lat_index)
    dev = latents.device
    nimg = nbc * f
    latents = latents.contiguous().float()
    temb, toff = time_embedding(P, t, resnet_names(False), dev)
    text = text_rows(prompt, P.dtype)
    tseg = segments.cross_interleaved(nimg, prompt.shape[0], dev, row_offset)   # row_offset: this rank's first "(b f)" row in the full ControlNet batch

    # conditioning embedding: 3->16 (direct), then 16->16, 16->32 s2, 32->32, 32->96 s2, 96->96, 96->256 s2 (SiLU each), 256->320.
    # Everything up to the last convolution depends on the skeleton images only -- the same tensor in every denoising step of a run
    # (pipeline_motion_editor.py:556-570 prepares it once, before the loop) -- so it is computed at the first step and kept
    # (COND_EMBED_CACHE = False recomputes it every step, as the reference does).
    H8, W8 = cond.shape[-2], cond.shape[-1]
    cond = cond.contiguous()
    ckey = (cond.data_ptr(), cond._version, tuple(cond.shape), cond.dtype, nimg)
    # one entry per conditioning tensor (address, version, shape), a handful at most: a captured step (denoise_step_graphed) bakes in the address of the
    # embedding it was captured with and holds a reference to it, so an entry that a later call with another skeleton pushes out of this table stays
    # alive for exactly as long as a graph can replay against it
    table = P.cache.setdefault("cond_embed", {}) if COND_EMBED_CACHE else None
    hit = table.get(ckey) if table is not None else None
    if hit is not None:
        c = hit[0]
    else:
        c = Act(ops.conv_small(cond, P.mat32("controlnet_cond_embedding.conv_in.weight"), P.vec32("controlnet_cond_embedding.conv_in.bias"),
                               n_img=nimg, Cin=3, H=H8, Wd=W8, img_stride=3 * H8 * W8, ch_stride=H8 * W8, silu=True), nimg, 1, H8, W8)
        for i in range(6):
            c = conv3x3(P, f"controlnet_cond_embedding.blocks.{i}", c, stride=2 if i % 2 == 1 else 1, act=2)
        if table is not None:
            while len(table) >= 4:
                table.pop(next(iter(table)))
            table[ckey] = (c, cond)   # cond kept alive: the key is its address
    # conv_in(sample) per ControlNet batch entry, then + cond embedding in the last cond conv's epilogue
    x0 = torch.empty((nimg * h * w, 320), dtype=P.dtype, device=dev)
    for bi, li in enumerate(lat_index):
        part = ops.conv_small(latents[li], P.mat32("conv_in.weight"), P.vec32("conv_in.bias"), n_img=f, Cin=4, H=h, Wd=w,
                              img_stride=h * w, ch_stride=f * h * w)
        ops.copy_rows(x0[bi * f * h * w:(bi + 1) * f * h * w], part)
    x = conv3x3(P, "controlnet_cond_embedding.conv_out", c, res=x0)
    x = Act(x.t, nimg, 1, h, w)

    tkv = text_kv_all(P, text, attention_block_names(False))
    outs = [x]
    for i in range(4):
        for j in range(2):
            n = f"down_blocks.{i}.resnets.{j}"
            x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=True)
            if DOWN_HAS_ATTN[i]:
                x = transformer2d(P, f"down_blocks.{i}.attentions.{j}", x, text, tseg, sc_attn=False, has_temp=False, text_kv=tkv)
            outs.append(x)
        if i < 3:
            x = conv3x3(P, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            outs.append(x)
    n = "mid_block.resnets.0"
    x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=True)
    x = transformer2d(P, "mid_block.attentions.0", x, text, tseg, sc_attn=False, has_temp=False, text_kv=tkv)
    n = "mid_block.resnets.1"
    x = resnet_block(P, n, x, temb, toff[n], per_frame_stats=True)
    down = [ops.gemm(o.t, P.mat(f"controlnet_down_blocks.{i}.weight"), bias=P.vec(f"controlnet_down_blocks.{i}.bias"), alpha=1.0)
            for i, o in enumerate(outs)]
    mid = ops.gemm(x.t, P.mat("controlnet_mid_block.weight"), bias=P.vec("controlnet_mid_block.bias"))
    if scale != 1.0:
        down = [ops.axpy_rows(d, d, d, scale - 1.0) for d in down]
        mid = ops.axpy_rows(mid, mid, mid, scale - 1.0)
    return down, mid
