"""TEST INFRASTRUCTURE: a torch-CPU fp32 emulation of every ``motioneditor_amd.ops`` entry point, with
the exact argument conventions of the C ABI (packed [N, taps, K] weights, key-segment tables, GEGLU
row interleave, ...).  Two uses:
  * ``-m "not gpu"`` tests monkeypatch it under ``models.graph`` to check the host launch graphs,
    weight packing, segment tables and editors against the oracle WITHOUT a GPU;
  * ``-m gpu`` tests use the same functions as the per-kernel fp32 reference for the HIP kernels.
It is never imported by the product package.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

SEG_PLAIN, SEG_DUAL_CUR, SEG_DUAL_PREV, SEG_DUAL_BIN = 0, 1, 2, 3


def _f(t):
    return None if t is None else t.float()


ROW_RANGE = True   # gemm(row_range=...) is emulated (the frame-sharded TemporalConv split)


def gemm_splits_k(M, N, K, taps=1):
    return False


def empty(rows, cols, like):
    return torch.empty((rows, cols), dtype=like.dtype, device=like.device)


LN_FOLD = True   # this backend implements gemm(ln=..., ln_out=...) / ln_stats (ABI 9)


def _row_parts(y):
    """[P, rows, 2] fp32 partial row sums (sum, sum of squares) over 320-column parts: the me_gemm_args.ln_stats format."""
    rows, C = y.shape
    P = C // 320 if C % 320 == 0 else 1
    yf = y.float().reshape(rows, P, C // P)
    return torch.stack([yf.sum(-1), (yf * yf).sum(-1)], dim=-1).permute(1, 0, 2).contiguous()


def ln_stats(x):
    return _row_parts(x)


def gemm(x, w, *, M=None, out=None, bias=None, rowvec=None, rows_per_vec=0, res=None, res2=None, geglu=False, act=0, alpha=1.0,
         conv=None, tconv=None, res_rows=0, res2_rows=0, row_range=None, ln=None, ln_out=False):
    if ln_out:
        y = gemm(x, w, M=M, out=out, bias=bias, rowvec=rowvec, rows_per_vec=rows_per_vec, res=res, res2=res2, geglu=geglu, act=act, alpha=alpha, conv=conv, tconv=tconv,
                 res_rows=res_rows, res2_rows=res2_rows, row_range=row_range, ln=ln)
        return y, _row_parts(y)
    if row_range is not None:   # rows [lo, hi) of the full launch, written into `out`
        lo, hi = row_range
        full = gemm(x, w, M=M, bias=bias, rowvec=rowvec, rows_per_vec=rows_per_vec, res=res, res2=res2, geglu=geglu, act=act, alpha=alpha, conv=conv, tconv=tconv,
                    res_rows=res_rows, res2_rows=res2_rows, ln=ln)
        out[lo:hi, :full.shape[1]] = full[lo:hi]
        return out[:full.shape[0], :full.shape[1]]
    N, taps, K = w.shape
    xf, wf = x.float()[:, :K], w.float()
    if conv is not None:
        Hin, Win, Hout, Wout, stride, ups = conv[:6]
        pad0 = conv[6] if len(conv) > 6 else 0
        n_img = x.shape[0] // (Hin * Win)
        img = xf.reshape(n_img, Hin, Win, K).permute(0, 3, 1, 2)
        if ups == 1:
            img = img.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        elif ups == 2:   # zero-stuffed 2x: virtual pixel (2y, 2x) = input pixel (y, x), the rest zeros
            z = torch.zeros(n_img, K, 2 * Hin, 2 * Win, dtype=img.dtype)
            z[:, :, ::2, ::2] = img
            img = z
        wk = wf.reshape(N, 3, 3, K).permute(0, 3, 1, 2)
        y = F.conv2d(F.pad(img, (0, 1, 0, 1)), wk, None, stride=stride) if pad0 else F.conv2d(img, wk, None, stride=stride, padding=1)
        assert y.shape[2] == Hout and y.shape[3] == Wout
        acc = y.permute(0, 2, 3, 1).reshape(-1, N)
    elif tconv is not None and len(tconv) > 3:
        # frame-sharded: local frames [frame0, frame0+frames) of frames_total; halo rows appended to x
        frames, npix, chunk, frame0, ftot, hp, hn = tconv
        Ml = M if M is not None else x.shape[0]
        nb = Ml // (frames * npix)
        loc = xf[:Ml].reshape(nb, frames, npix, K)
        acc = torch.zeros(nb, frames, npix, N)
        for fr in range(frames):
            g = frame0 + fr
            for tap in range(3):
                gs = g + tap - 1
                if gs < 0 or gs >= ftot or gs // chunk != g // chunk:
                    continue
                ls = fr + tap - 1
                if 0 <= ls < frames:
                    src = loc[:, ls]
                else:
                    h0 = hp if ls < 0 else hn
                    if h0 < 0:      # no halo block given (me_gemm: the tap contributes nothing): a row-range launch over interior frames never needs one
                        continue
                    src = xf[h0:h0 + nb * npix].reshape(nb, npix, K)
                acc[:, fr] += src @ wf[:, tap].t()
        acc = acc.reshape(-1, N)
    elif tconv is not None:
        frames, npix, chunk = tconv
        nb = x.shape[0] // (frames * npix)
        t = xf.reshape(nb, frames // chunk, chunk, npix, K).permute(0, 1, 3, 4, 2).reshape(-1, K, chunk)   # (b ch p) c t
        y = F.conv1d(t, wf.permute(0, 2, 1), None, padding=1)
        acc = y.reshape(nb, frames // chunk, npix, N, chunk).permute(0, 1, 4, 2, 3).reshape(-1, N)
    else:
        acc = xf @ wf[:, 0].t()
    if M is None:
        M = acc.shape[0]
    acc = acc[:M] * alpha
    if ln is not None:   # LayerNorm folded into the projection: rstd (acc - mean colsum) + cvec from the partial row sums (me_gemm_args.ln_stats)
        st, colsum, cvec, eps = ln
        assert conv is None and tconv is None and bias is None and alpha == 1.0 and rowvec is None and res is None and res2 is None and act == 0
        S = st.float().sum(dim=0)[:M]
        mean = S[:, 0] / K
        rstd = torch.rsqrt((S[:, 1] / K - mean * mean).clamp_min(0) + eps)
        acc = rstd[:, None] * (acc - mean[:, None] * colsum.float()[None]) + cvec.float()[None]
    if geglu:
        if bias is not None:
            acc = acc + bias.float()
        q = acc.reshape(M, N // 32, 2, 16)
        val = q[:, :, 0] * F.gelu(q[:, :, 1])
        y = val.reshape(M, N // 2)
    else:
        y = acc
        if bias is not None:
            y = y + bias.float()
        if rowvec is not None:
            idx = torch.arange(M) // rows_per_vec
            y = y + rowvec.float()[idx][:, :N]
        if act == 1:
            y = F.relu(y)
        elif act == 2:
            y = F.silu(y)
        if res is not None:
            y = y + res.float()[(torch.arange(M) % res_rows) if res_rows else slice(0, M), :N]
        if res2 is not None:
            y = y + res2.float()[(torch.arange(M) % res2_rows) if res2_rows else slice(0, M), :N]
    y = y.to(x.dtype)
    if out is not None:
        out[:M, :y.shape[1]] = y
        return out[:M, :y.shape[1]]
    return y


def conv_small(inp, w, bias, *, n_img, Cin, H, Wd, img_stride, ch_stride, frames=0, frame_stride=0, silu=False):
    flat = inp.reshape(-1).float()
    imgs = []
    for i in range(n_img):
        base = (i // frames) * img_stride + (i % frames) * frame_stride if frames > 0 else i * img_stride
        imgs.append(torch.stack([flat[base + c * ch_stride: base + c * ch_stride + H * Wd].reshape(H, Wd) for c in range(Cin)]))
    x = torch.stack(imgs)
    Cout = w.shape[0]
    y = F.conv2d(x, w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), _f(bias), padding=1)
    if silu:
        y = F.silu(y)
    return y.permute(0, 2, 3, 1).reshape(-1, Cout).to(_EMU_DTYPE if w.dtype == torch.float32 else w.dtype)


def attention(q, k, v, *, heads, dh, n_items, nq, nk, seg_item, seg_mode, mask=None, scale=None, out=None, q_items=0, lse=None):
    scale = dh ** -0.5 if scale is None else scale
    C = heads * dh
    qf, kf, vf = q.float()[:, :C], k.float()[:, :C], v.float()[:, :C]
    res = torch.empty((n_items * nq, C), dtype=torch.float32)
    si, sm = seg_item.tolist(), seg_mode.tolist()
    mk = _f(mask)
    for it in range(n_items):
        iq = it % q_items if q_items else it
        qi = qf[iq * nq:(iq + 1) * nq].reshape(nq, heads, dh).permute(1, 0, 2)   # [H, nq, dh]
        logits, vals = [], []
        for s, kit in enumerate(si[it]):
            if kit < 0:
                break
            ks = kf[kit * nk:(kit + 1) * nk].reshape(nk, heads, dh).permute(1, 0, 2)
            vs = vf[kit * nk:(kit + 1) * nk].reshape(nk, heads, dh).permute(1, 0, 2)
            sc = torch.einsum("hqd,hkd->hqk", qi, ks) * scale
            mode = sm[it][s]
            if mode == SEG_PLAIN:
                logits.append(sc)
                vals.append(vs)
            elif mode == SEG_DUAL_BIN:   # binary mask: one copy keeps the key, the other is a zero vector (logit 0)
                logits += [sc, torch.zeros_like(sc)]
                vals += [vs, vs]
            else:
                planes = torch.tensor([(h if mode == SEG_DUAL_CUR else max(h - 1, 0)) for h in range(heads)])
                m = mk[planes][:, None, :]                     # [H, 1, nk]
                logits += [sc * m, sc * (1 - m)]
                vals += [vs, vs]
        lg = torch.cat(logits, dim=-1)
        if lse is not None:   # log2-domain log-sum-exp [nq, heads], what the HIP forward stashes for its backward
            lse[it * nq:(it + 1) * nq] = (torch.logsumexp(lg.detach(), dim=-1) * 1.4426950408889634).t()
        p = lg.softmax(dim=-1)
        o = torch.einsum("hqk,hkd->hqd", p, torch.cat(vals, dim=1))
        res[it * nq:(it + 1) * nq] = o.permute(1, 0, 2).reshape(nq, C)
    res = res.to(q.dtype)
    if out is not None:
        out[:, :C] = res
        return out
    return res


def temporal_attention(q, k, v, *, heads, dh, batch, frames, npix, kv_map=None, scale=None, q_frames=0, q_frame0=0, kv_parts=1, q_parts=1):
    scale = dh ** -0.5 if scale is None else scale
    C = heads * dh
    km = list(kv_map) if kv_map is not None else list(range(batch))
    qf = q_frames or frames
    fpp = frames // max(kv_parts, 1)

    def kv_shp(t):  # part-major rows (part b fl p) -> [b, p, H, f, dh]
        t = t.float()[:, :C].reshape(max(kv_parts, 1), batch, fpp, npix, heads, dh).permute(1, 0, 2, 3, 4, 5).reshape(batch, frames, npix, heads, dh)
        return t.permute(0, 2, 3, 1, 4)

    qq = kv_shp(q) if q_parts > 1 else q.float()[:, :C].reshape(batch, qf, npix, heads, dh).permute(0, 2, 3, 1, 4)
    kk, vv = kv_shp(k)[km], kv_shp(v)[km]
    s = torch.einsum("bphid,bphjd->bphij", qq, kk) * scale
    gi = (q_frame0 if q_frames else 0) + torch.arange(qf)
    s = s + (torch.arange(frames)[None, :] > gi[:, None]).float() * -10000.0
    o = torch.einsum("bphij,bphjd->bphid", s.softmax(-1), vv)
    if q_parts > 1:   # part-major output rows (part b fl p)
        o = o.permute(0, 3, 1, 2, 4).reshape(batch, q_parts, fpp, npix, C).permute(1, 0, 2, 3, 4)
        return o.reshape(batch * frames * npix, C).to(q.dtype)
    return o.permute(0, 3, 1, 2, 4).reshape(batch * qf * npix, C).to(q.dtype)


def groupnorm(x, gamma, beta, *, rows_per_group, eps, silu, groups=32, out=None, reduce=None, rows_per_group_total=None):
    rows, C = x.shape
    t = x.float().reshape(rows // rows_per_group, rows_per_group, groups, C // groups)
    if reduce is not None:   # frame-sharded: (sum, sumsq) all-reduced over the ranks, global count
        st = torch.stack([t.sum(dim=(1, 3)), (t * t).sum(dim=(1, 3))], dim=-1).reshape(-1).contiguous()
        reduce(st)
        st = st.reshape(rows // rows_per_group, groups, 2)
        cnt = float((rows_per_group_total or rows_per_group) * (C // groups))
        mean = (st[..., 0] / cnt)[:, None, :, None]
        var = (st[..., 1] / cnt)[:, None, :, None] - mean * mean
    else:
        mean = t.mean(dim=(1, 3), keepdim=True)
        var = t.var(dim=(1, 3), unbiased=False, keepdim=True)
    y = ((t - mean) / torch.sqrt(var + eps)).reshape(rows, C) * gamma.float() + beta.float()
    if silu:
        y = F.silu(y)
    return y.to(x.dtype)


def layernorm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps).to(x.dtype)


def axpy_rows(y, x, a_, alpha=1.0):
    y.copy_((x.float() + alpha * a_.float()).to(y.dtype))
    return y


def copy_rows(y, x):
    y.copy_(x)
    return y


def copy_blocks(y, x, n0, n1, rows, *, ys0, ys1, xs0, xs1):
    for i in range(n0):
        for j in range(n1):
            y[i * ys0 + j * ys1:i * ys0 + j * ys1 + rows, :x.shape[1]] = x[i * xs0 + j * xs1:i * xs0 + j * xs1 + rows]
    return y


def silu(x):
    return F.silu(x.float()).to(x.dtype)


def relu(x):
    return F.relu(x.float()).to(x.dtype)


_EMU_DTYPE = torch.float32


def timestep_embed(rows, dim, t, device):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = float(t) * freqs
    return torch.cat([torch.cos(ang), torch.sin(ang)])[None].repeat(rows, 1).to(_EMU_DTYPE)


def softmax_rows(x, out=None):
    y = torch.softmax(x.float(), dim=-1).to(_EMU_DTYPE)
    if out is not None:
        out.copy_(y)
        return out
    return y


def cfg_ddim(latents, eps_rows, *, guidance, ca, cb):
    nb, C, f, h, w = latents.shape
    e = eps_rows.float()[:, :C].reshape(2 * nb, f, h * w, C).permute(0, 3, 1, 2).reshape(2 * nb, C, f, h, w)
    eu, ec = e[:nb], e[nb:]
    return ca * latents + cb * (eu + guidance * (ec - eu))


def gaussian_sample(moments, noise, n_img, npix, scale=1.0):
    m = moments.float()[:, :8].reshape(n_img, npix, 8).permute(0, 2, 1)
    return (m[:, :4] + torch.exp(0.5 * m[:, 4:].clamp(-30.0, 20.0)) * noise.float().reshape(n_img, 4, npix)) * scale


def nchw5_to_rows(x):
    B, C, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(B * f * h * w, C).to(_EMU_DTYPE).contiguous()


def rows_to_nchw5(rows, B, C, f, h, w):
    return rows.float()[:, :C].reshape(B, f, h, w, C).permute(0, 4, 1, 2, 3).contiguous()


def rows_to_nchw(x, n_img, C, npix):
    return x.float()[:, :C].reshape(n_img, npix, C).permute(0, 2, 1).contiguous()


def nchw_to_rows(x, n_img, C, npix, img_stride, ch_stride):
    flat = x.reshape(-1).float()
    out = torch.empty((n_img * npix, C), dtype=_EMU_DTYPE)
    for i in range(n_img):
        for c in range(C):
            out[i * npix:(i + 1) * npix, c] = flat[i * img_stride + c * ch_stride: i * img_stride + c * ch_stride + npix]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# backward primitives (motioneditor_amd/autodiff.py): each is the vector-Jacobian product of its forward emulation above,
# taken with torch autograd -- the CPU statement of what the backward kernels have to compute.  Same contract as ops.py:
# entries with `dst` / `dq, dk, dv` ACCUMULATE into those fp32 views.
# ---------------------------------------------------------------------------------------------------------------------
def _leaf(t):
    return t.detach().float().clone().requires_grad_(True)


SELECT_ROWS_SCALE = 1   # (kernel-selection hint of the HIP backend: no meaning here)


def grad_acc(dst, src, alpha=1.0, pool=None, store=False):
    if store:   # the buffer was never written: whatever it holds (NaN under ME_GRAD_POISON) must not be read
        dst.zero_()
    s_ = src.float()
    if pool is not None:
        H, W = pool
        s_ = s_.reshape(-1, H, 2, W, 2, s_.shape[-1]).sum(dim=(2, 4)).reshape(-1, s_.shape[-1])
    if dst.dim() == 2:
        dst.add_(alpha * s_[:dst.shape[0], :dst.shape[1]])
    else:
        dst.add_(alpha * s_.reshape(dst.shape))
    return dst


def invalidate_transposed(tensors=None):
    pass


def gemm_dx(dy, w, *, dst, M, alpha=1.0, conv=None, tconv=None, store=False):
    """dst += dX of y = alpha * gather(x) @ w^T (any gather mode): linear in x, so the point of linearisation is irrelevant."""
    if store:
        dst.zero_()
    K = w.shape[2]
    x0 = torch.zeros((dst.shape[0], K), dtype=torch.float32, requires_grad=True)
    y = gemm(x0, w.float(), M=M, alpha=alpha, conv=conv, tconv=tconv)
    dst.add_(torch.autograd.grad(y, x0, dy.float()[:y.shape[0], :y.shape[1]])[0])
    return dst


def geglu_bwd(pre, dy):
    """pre: biased pre-activation [M, N] in the packed (16 value | 16 gate) column order; dy [M, N/2] -> d pre [M, N]."""
    p0 = _leaf(pre)
    M, N = p0.shape
    q = p0.reshape(M, N // 32, 2, 16)
    y = (q[:, :, 0] * F.gelu(q[:, :, 1])).reshape(M, N // 2)
    return torch.autograd.grad(y, p0, dy.float())[0]


def attention_bwd(q, k, v, out, dout, *, dq, dk, dv, lse=None, **kw):
    q0, k0, v0 = _leaf(q), _leaf(k), _leaf(v)
    y = attention(q0, k0, v0, **kw)
    gq, gk, gv = torch.autograd.grad(y, (q0, k0, v0), dout.float())
    dq.add_(gq[:, :dq.shape[1]])
    dk.add_(gk[:, :dk.shape[1]])
    dv.add_(gv[:, :dv.shape[1]])


def temporal_attention_bwd(q, k, v, out, dout, **kw):
    q0, k0, v0 = _leaf(q), _leaf(k), _leaf(v)
    y = temporal_attention(q0, k0, v0, **kw)
    return torch.autograd.grad(y, (q0, k0, v0), dout.float())


def groupnorm_bwd(x, gamma, beta, dy, *, rows_per_group, eps, silu, groups=32):
    x0 = _leaf(x)
    y = groupnorm(x0, gamma, beta, rows_per_group=rows_per_group, eps=eps, silu=silu, groups=groups)
    return torch.autograd.grad(y, x0, dy.float())[0]


def layernorm_bwd(x, gamma, dy, *, eps=1e-5):
    x0 = _leaf(x)
    y = layernorm(x0, gamma, torch.zeros_like(gamma), eps)
    return torch.autograd.grad(y, x0, dy.float())[0]


def gemm_dw(dy, x, *, dst, taps, K, M, alpha=1.0, conv=None, tconv=None):
    """dst += dW [N, taps, K] of y = alpha * gather(x) @ w^T: linear in w."""
    N = dy.shape[1]
    w0 = torch.zeros((N, taps, K), dtype=torch.float32, requires_grad=True)
    y = gemm(x.float(), w0, M=M, alpha=alpha, conv=conv, tconv=tconv)
    dst.add_(torch.autograd.grad(y, w0, dy.float()[:y.shape[0], :y.shape[1]])[0])
    return dst


def colsum_grad(dy, *, dst, alpha=1.0):
    dst.add_(alpha * dy.float().sum(dim=0))
    return dst


def relu_bwd(dy, out):
    return dy.float() * (out.float() > 0).float()


def layernorm_bwd_params(x, dy, *, dgamma=None, dbeta=None, eps=1e-5):
    g0 = torch.ones(x.shape[1], requires_grad=True)
    b0 = torch.zeros(x.shape[1], requires_grad=True)
    y = layernorm(x.float(), g0, b0, eps)
    dg, db = torch.autograd.grad(y, (g0, b0), dy.float())
    if dgamma is not None:
        dgamma.add_(dg)
    if dbeta is not None:
        dbeta.add_(db)


def sumsq_absmax(x, out=None):
    r = torch.stack([(x.double() ** 2).sum().float(), x.abs().max().float()])
    if out is not None:
        out.copy_(r)
        return out
    return r


def adamw(p, m, v, g, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step, gnorm_sq=None, max_grad_norm=0.0, grad_scale=1.0):
    """torch.optim.AdamW's update written out (weight_decay 0 = Adam), with clip_grad_norm_ folded in as in me_adamw."""
    gs = grad_scale
    if gnorm_sq is not None:
        total = float(gnorm_sq.reshape(-1)[0]) ** 0.5 * grad_scale
        gs *= min(1.0, max_grad_norm / (total + 1e-6))
    gi = g * gs
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(gi, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1.0 - beta2)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)


def cast_f16(dst, src):
    dst.copy_(src.to(dst.dtype))
    return dst


def mse_seed(eps_u, target, *, eps_c=None, x=None, guidance=1.0, ca=0.0, cb=1.0, coef=1.0):
    nb, C, f, h, w = target.shape
    to5 = lambda r: r.float()[:, :C].reshape(nb, f, h * w, C).permute(0, 3, 1, 2).reshape(nb, C, f, h, w)   # noqa: E731
    e = to5(eps_u)
    if eps_c is not None:
        e = e + guidance * (to5(eps_c) - e)
    rec = cb * e + (ca * x if x is not None else 0.0)
    diff = rec - target
    return diff, (coef * diff).permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
