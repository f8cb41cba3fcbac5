"""Per-kernel parity on a real MI355X: every libmotioned entry point (through the C ABI / ctypes) against
the fp32 torch emulation in tests/emu_ops.py on the same seeded fp16 inputs.

Tolerance (fp16 storage, fp32 accumulate): rel-L2 <= 2e-3 per kernel and
max-abs error <= 2e-2 x mean-abs of the reference."""
import math

import numpy as np
import pytest
import torch

import emu_ops as emu

pytestmark = pytest.mark.gpu

REL_L2 = 2e-3
MAX_REL = 2e-2


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the HIP library is the only compute path")
    from motioneditor_amd import capi, ops as _ops
    capi.lib()  # fails loudly when libmotioned.so is missing
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


def check(got, want, name="", rel=REL_L2, mx=MAX_REL):
    got, want = got.detach().float().cpu().double(), want.detach().float().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got).all(), name
    r = float((got - want).norm() / want.norm().clamp_min(1e-30))
    m = float((got - want).abs().max() / want.abs().mean().clamp_min(1e-30))
    assert r <= rel and m <= mx, f"{name}: rel-L2 {r:.3e} (<= {rel}), max/mean {m:.3e} (<= {mx})"


def cu(t):
    return t.cuda() if t is not None else None


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (77 * 2, 640, 768), (1000, 1280, 1280), (37, 4, 320), (520, 960, 64), (8, 64, 8)])
def test_gemm_dense_plain(ops, M, N, K):
    x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
    check(ops.gemm(cu(x), cu(w)), emu.gemm(x, w), f"gemm {M}x{N}x{K}")


def test_gemm_dense_epilogue_all(ops):
    M, N, K = 520, 320, 640
    x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
    bias, res, res2 = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    rowvec = rnd(4, 2048, seed=6)[:, 128:128 + N]
    for act in (0, 1, 2):
        got = ops.gemm(cu(x), cu(w), bias=cu(bias), rowvec=cu(rnd(4, 2048, seed=6))[:, 128:128 + N], rows_per_vec=130, res=cu(res), res2=cu(res2), act=act, alpha=0.5)
        want = emu.gemm(x, w, bias=bias, rowvec=rowvec, rows_per_vec=130, res=res, res2=res2, act=act, alpha=0.5)
        check(got, want, f"gemm epilogue act={act}")


@pytest.mark.parametrize("terms", ["none", "bias", "rowvec", "res", "res+res2", "bias+rowvec", "bias+res", "bias+rowvec+res", "inplace"])
@pytest.mark.parametrize("M,N,K", [(256 * 520, 320, 64), (1000, 640, 128), (300, 128, 64)])
def test_gemm_epilogue_term_combinations(ops, terms, M, N, K):
    """Each specialised epilogue body (epilogue_rows<F>) and the generic one, on the 256x320 tile (>= 512 tiles),
    the 128x160 tile and the 128x128 tile; bias enters through the accumulator init."""
    x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
    bias = rnd(N, seed=3) if "bias" in terms or terms == "inplace" else None
    rowvec = rnd(8, N, seed=4) if "rowvec" in terms else None
    res = rnd(M, N, seed=5) if "res" in terms or terms == "inplace" else None
    res2 = rnd(M, N, seed=6) if "res2" in terms else None
    rpv = (M + 7) // 8
    want = emu.gemm(x, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv if rowvec is not None else 0, res=res, res2=res2)
    if terms == "inplace":
        out = cu(res).clone()
        ops.gemm(cu(x), cu(w), bias=cu(bias), res=out, out=out)
    else:
        out = ops.gemm(cu(x), cu(w), bias=cu(bias), rowvec=cu(rowvec), rows_per_vec=rpv if rowvec is not None else 0, res=cu(res), res2=cu(res2))
    check(out, want, f"gemm {M}x{N}x{K} terms={terms}")


def test_gemm_inplace_residual_and_strided_views(ops):
    M, C = 260, 320
    x, w = rnd(M, C, seed=1), rnd(C, 1, C, seed=2, scale=C ** -0.5)
    t = rnd(M, C, seed=3)
    tg = cu(t).clone()
    ops.gemm(cu(x), cu(w), res=tg, out=tg)            # out aliases res
    check(tg, emu.gemm(x, w, res=t), "gemm in-place residual")
    big = rnd(M, 3 * C, seed=4)                         # strided X view (ldx = 3C) and strided out
    outbuf = torch.zeros(M, 2 * C, dtype=torch.float16, device="cuda")
    ops.gemm(cu(big)[:, C:2 * C], cu(w), out=outbuf[:, C:])
    check(outbuf[:, C:], emu.gemm(big[:, C:2 * C], w), "gemm strided")
    assert float(outbuf[:, :C].abs().max()) == 0.0


@pytest.mark.parametrize("C,M", [(320, 300), (640, 130), (1280, 64)])
def test_gemm_geglu(ops, C, M):
    from motioneditor_amd.weights import Packed
    W = (torch.randn(8 * C, C, generator=torch.Generator().manual_seed(1)) * C ** -0.5)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(2)) * 0.1
    P = Packed({"w": W, "b": b}, "cuda")
    x = rnd(M, C, seed=3)
    got = ops.gemm(cu(x), P.geglu_mat("w"), bias=P.geglu_vec("b"), geglu=True)
    h = x.float() @ W.half().float().t() + b.half().float()
    want = h[:, :4 * C] * torch.nn.functional.gelu(h[:, 4 * C:])
    check(got, want, f"geglu C={C}")
    Pc = Packed({"w": W, "b": b}, "cpu")
    check(got, emu.gemm(x, Pc.geglu_mat("w"), bias=Pc.geglu_vec("b"), geglu=True), "geglu vs emu packing")


@pytest.mark.parametrize("Cin,Cout,H,W,stride,ups,nimg", [(320, 320, 8, 8, 1, 0, 5), (320, 640, 8, 8, 2, 0, 3), (640, 640, 4, 4, 1, 1, 3), (16, 32, 16, 16, 2, 0, 2),
                                                          (96, 96, 8, 8, 1, 0, 2), (320, 4, 8, 8, 1, 0, 4), (1280, 1280, 2, 2, 1, 0, 8), (1280, 1280, 1, 1, 1, 0, 16),
                                                          (2560, 1280, 2, 2, 1, 0, 4), (640, 320, 16, 12, 1, 0, 2),
                                                          (16, 16, 24, 20, 1, 0, 3), (32, 32, 16, 16, 1, 0, 2), (32, 96, 16, 16, 2, 0, 2), (8, 16, 8, 8, 1, 0, 2)])
def test_gemm_conv3x3(ops, Cin, Cout, H, W, stride, ups, nimg):
    x = rnd(nimg * H * W, Cin, seed=1)
    w = rnd(Cout, 9, Cin, seed=2, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=3)
    hv, wv = H << ups, W << ups
    ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
    conv = (H, W, ho, wo, stride, ups)
    got = ops.gemm(cu(x), cu(w), M=nimg * ho * wo, bias=cu(bias), conv=conv)
    check(got, emu.gemm(x, w, M=nimg * ho * wo, bias=bias, conv=conv), f"conv {Cin}->{Cout} {H}x{W} s{stride} u{ups}")


@pytest.mark.parametrize("Cin,Cout,H,W,nimg", [(128, 320, 64, 48, 43), (64, 640, 32, 32, 64), (192, 320, 16, 16, 520)])
def test_gemm_conv3x3_halo_tile(ops, Cin, Cout, H, W, nimg):
    """Grids of >= 512 (16x16 patch, 320 channel) tiles take the LDS-halo kernel; full epilogue (bias, per-image
    row vector, residual, SiLU) and image-border / patch-border taps."""
    assert nimg * (H // 16) * (W // 16) * (Cout // 320) >= 512
    M = nimg * H * W
    x = rnd(M, Cin, seed=1)
    w = rnd(Cout, 9, Cin, seed=2, scale=(9 * Cin) ** -0.5)
    bias, res, rowvec = rnd(Cout, seed=3), rnd(M, Cout, seed=4), rnd(nimg, Cout, seed=5)
    conv = (H, W, H, W, 1, 0)
    got = ops.gemm(cu(x), cu(w), M=M, bias=cu(bias), conv=conv, rowvec=cu(rowvec), rows_per_vec=H * W, res=cu(res), act=2)
    want = emu.gemm(x, w, M=M, bias=bias, conv=conv, rowvec=rowvec, rows_per_vec=H * W, res=res, act=2)
    check(got, want, f"halo conv {Cin}->{Cout} {H}x{W} x{nimg}")


@pytest.mark.parametrize("C,frames,npix,chunk,nb", [(320, 16, 4, 16, 2), (320, 16, 9, 8, 2), (640, 24, 4, 8, 1), (1280, 8, 1, 8, 4)])
def test_gemm_tconv(ops, C, frames, npix, chunk, nb):
    x = rnd(nb * frames * npix, C, seed=1)
    w = rnd(C, 3, C, seed=2, scale=(3 * C) ** -0.5)
    bias, res = rnd(C, seed=3), rnd(nb * frames * npix, C, seed=4)
    got = ops.gemm(cu(x), cu(w), bias=cu(bias), tconv=(frames, npix, chunk), res=cu(res))
    check(got, emu.gemm(x, w, bias=bias, tconv=(frames, npix, chunk), res=res), f"tconv C={C} f={frames} chunk={chunk}")


@pytest.mark.parametrize("case", ["dense k320 bias+res", "dense tail k128 res+res2", "dense generic epilogue", "dense one k tile", "geglu", "conv stride 2",
                                  "conv upsampled", "tconv", "dense 192-row tiles", "dense 192-row tiles tail", "conv 192-row tiles", "tconv 192-row tiles"])
def test_gemm_8phase_kernel(ops, case, monkeypatch):
    """The 8-phase ping-pong kernel (grids of >= 512 tiles of 256 rows; 192-row tiles for grids of 256 .. 511 such tiles): against the
    emulation, BITWISE against the one-barrier-per-slab kernels (same MFMA order per accumulator, same epilogue code) and bitwise across
    repeated runs -- a staging race would show as a tile that differs on some run."""
    kw, ekw = {}, {}
    geglu = False
    if case.startswith("dense"):
        M, N, K = {"dense k320 bias+res": (256 * 520, 320, 320), "dense tail k128 res+res2": (66000, 640, 128),
                   "dense generic epilogue": (256 * 260, 640, 192), "dense one k tile": (256 * 520, 320, 64),
                   "dense 192-row tiles": (24576, 1280, 640), "dense 192-row tiles tail": (24000, 1280, 512)}[case]
        x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
        ekw["bias"] = rnd(N, seed=3)
        ekw["res"] = rnd(M, N, seed=4)
        if "res2" in case or "generic" in case:
            ekw["res2"] = rnd(M, N, seed=5)
        if "generic" in case:
            ekw["rowvec"], ekw["rows_per_vec"], ekw["act"] = rnd(8, N, seed=6), (M + 7) // 8, 2
    elif case == "geglu":
        M, C = 33000, 320
        from motioneditor_amd.weights import Packed
        Pc = Packed({"w": torch.randn(8 * C, C, generator=torch.Generator().manual_seed(1)) * C ** -0.5,
                     "b": torch.randn(8 * C, generator=torch.Generator().manual_seed(2)) * 0.1}, "cpu")
        x, w = rnd(M, C, seed=3), Pc.geglu_mat("w")
        ekw["bias"] = Pc.geglu_vec("b")
        geglu = True
    elif case.startswith("conv"):
        Cin, Cout, H, W, stride, ups, nimg = ((128, 320, 32, 32, 2, 0, 512) if "stride" in case else (64, 1280, 16, 16, 1, 0, 96) if "192" in case
                                              else (64, 320, 8, 8, 1, 1, 512))
        ho, wo = ((H << ups) - 1) // stride + 1, ((W << ups) - 1) // stride + 1
        M = nimg * ho * wo
        x, w = rnd(nimg * H * W, Cin, seed=1), rnd(Cout, 9, Cin, seed=2, scale=(9 * Cin) ** -0.5)
        ekw.update(M=M, conv=(H, W, ho, wo, stride, ups), bias=rnd(Cout, seed=3), res=rnd(M, Cout, seed=4))
    else:
        C, frames, npix, chunk, nb = (320, 16, 64, 8, 128) if "192" not in case else (1280, 24, 256, 8, 4)
        M = nb * frames * npix
        x, w = rnd(M, C, seed=1), rnd(C, 3, C, seed=2, scale=(3 * C) ** -0.5)
        ekw.update(tconv=(frames, npix, chunk), bias=rnd(C, seed=3), rowvec=rnd(nb, C, seed=5), rows_per_vec=frames * npix, res=rnd(M, C, seed=4))
    for k, v in ekw.items():
        kw[k] = cu(v) if isinstance(v, torch.Tensor) else v
    xg, wg = cu(x), cu(w)
    monkeypatch.setenv("ME_GEMM_8P", "0")
    ref = ops.gemm(xg, wg, geglu=geglu, **kw)
    assert ops._last_kernel().startswith("gemm_kernel<"), ops._last_kernel()
    monkeypatch.setenv("ME_GEMM_8P", "1")
    for rep in range(4):
        got = ops.gemm(xg, wg, geglu=geglu, **kw)
        assert ops._last_kernel().startswith("gemm8p_kernel<192" if "192" in case else "gemm8p_kernel<256"), ops._last_kernel()
        assert torch.equal(got, ref), f"{case}: run {rep} differs from the one-barrier kernel in {int((got != ref).sum())} elements"
    sub = slice(0, None, 7)   # every 7th row against the emulation (the bitwise check above covers the rest)
    want = emu.gemm(x, w, geglu=geglu, **ekw)
    check(got[sub], want[sub], f"8-phase gemm {case}")


@pytest.mark.parametrize("terms", ["none", "head_major", "rowvec", "res", "rowvec+res", "res+res2", "res shared small", "res shared wrap", "rowvec short", "alpha", "geglu"])
@pytest.mark.parametrize("shape", ["256-row", "256-row tail", "192-row"])
def test_gemm_row_contiguous_epilogue_equals_the_direct_epilogue(ops, terms, shape, monkeypatch):
    """epilogue_rowpass (round 5: the 8-phase kernels park their fp16 tile in wave-private LDS and store / read terms as 16-byte pieces of
    160-byte row segments) against the direct epilogue (ME_GEMM_ROWEPI=0): BITWISE for every specialised term set -- no terms, the head-major
    second output, rowvec, res, rowvec + res, res + res2 -- with shared residual rows shorter than a wave tile (general modulo path) and wrapping
    inside a tile, a row-vector period shorter than a wave tile, alpha != 1, ragged last row tiles (M % 256 != 0) and the 192-row tiles."""
    M, N, K = {"256-row": (256 * 520, 320, 128), "256-row tail": (256 * 519 + 77, 640, 128), "192-row": (24576 - 40, 1280, 512)}[shape]
    if terms == "head_major" and shape == "192-row":
        pytest.skip("the q|k|v widths (3 C) never land on the 192-row tiles")
    if terms == "geglu":      # the 256 x 256 GEGLU tile: one LDS tile per block, 256-byte rows (epilogue_geglu_rowpass)
        if shape == "192-row":
            pytest.skip("the GEGLU kernel has the 256-row form only")
        N = 2560
    x, w = rnd(M, K, seed=1).cuda(), rnd(N, 1, K, seed=2, scale=K ** -0.5).cuda()
    kw = dict(bias=rnd(N, seed=3).cuda())
    if terms == "geglu":
        kw["geglu"] = True
    if terms == "head_major":
        if N % 960:
            N = 960
            w = rnd(N, 1, K, seed=2, scale=K ** -0.5).cuda()
            kw = dict(bias=rnd(N, seed=3).cuda())
        kw["head_major"] = (N // 3, N // 24)
    if "rowvec" in terms:
        rpv = 100 if terms == "rowvec short" else (M + 5) // 6 + 1
        kw.update(rowvec=rnd((M + rpv - 1) // rpv, N, seed=6).cuda(), rows_per_vec=rpv)
    if terms in ("res", "rowvec+res", "res+res2", "alpha"):
        kw["res"] = rnd(M, N, seed=4).cuda()
    if terms == "res+res2":
        kw["res2"] = rnd(M, N, seed=5).cuda()
    if terms.startswith("res shared"):
        rr = 96 if "small" in terms else 256 * 3 + 50
        kw.update(res=rnd(rr, N, seed=4).cuda(), res_rows=rr)
    if terms == "alpha":
        kw["alpha"] = 0.125 * 3
    monkeypatch.setenv("ME_GEMM_ROWEPI", "0")
    ref = ops.gemm(x, w, **kw)
    want_kernel = "gemm8p_kernel<192" if shape == "192-row" else "gemm8p_kernel<256"
    assert ops._last_kernel().startswith(want_kernel), ops._last_kernel()
    monkeypatch.setenv("ME_GEMM_ROWEPI", "1")
    got = ops.gemm(x, w, **kw)
    assert ops._last_kernel().startswith(want_kernel), ops._last_kernel()
    if terms == "head_major":
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    else:
        assert torch.equal(got, ref), f"{int((got != ref).sum())} elements differ"
    # in place: the residual aliases the output
    if terms == "res":
        buf = kw["res"].clone()
        ops.gemm(x, w, bias=kw["bias"], res=buf, out=buf)
        assert torch.equal(buf, ref)


def test_gemm_8phase_kernel_views_inplace_and_shared_residual(ops, monkeypatch):
    """The 8-phase kernel on what the launch graph actually hands it: X as a column slice of a wider tensor (ldx > K), the output as a column
    slice, the residual aliased to the output (in place), and a residual shared by several batch entries (res_rows)."""
    monkeypatch.setenv("ME_GEMM_8P", "1")
    M, C = 256 * 520, 320
    big = rnd(M, 3 * C, seed=1)
    w, bias = rnd(C, 1, C, seed=2, scale=C ** -0.5), rnd(C, seed=3)
    t = rnd(M, 2 * C, seed=4)
    tg = cu(t).clone()
    ops.gemm(cu(big)[:, C:2 * C], cu(w), bias=cu(bias), res=tg[:, C:], out=tg[:, C:])          # strided X, strided out, out aliases res
    assert ops._last_kernel().startswith("gemm8p_kernel"), ops._last_kernel()
    want = emu.gemm(big[:, C:2 * C], w, bias=bias, res=t[:, C:])
    check(tg[::5, C:], want[::5], "8-phase gemm strided views, in-place residual")
    assert torch.equal(tg[:, :C].cpu(), t[:, :C])                                                # the other columns are untouched
    shared = rnd(M // 4, C, seed=5)
    got = ops.gemm(cu(big)[:, :C], cu(w), res=cu(shared), res_rows=M // 4)
    assert ops._last_kernel().startswith("gemm8p_kernel"), ops._last_kernel()
    check(got[::5], emu.gemm(big[:, :C], w, res=shared, res_rows=M // 4)[::5], "8-phase gemm shared residual rows")


@pytest.mark.parametrize("case", ["conv 8x8 bias+res", "dense rowvec+res+res2", "dense silu generic", "tconv", "conv stride 2"])
def test_gemm_split_k_small_grids(ops, case, monkeypatch):
    """Grids too small to fill the chip (the 8 x 8-latent level, batch-1 backward GEMMs) are split along K: fp32 partial sums in a scratch,
    a second kernel adds them in a fixed order and applies the epilogue.  Against the emulation, against the un-split launch (same values up
    to fp32 summation order) and bitwise run to run."""
    kw, ekw = {}, {}
    if case == "conv 8x8 bias+res":
        Cin, Cout, H, W, nimg = 1280, 1280, 8, 8, 24
        M = nimg * H * W
        x, w = rnd(M, Cin, seed=1), rnd(Cout, 9, Cin, seed=2, scale=(9 * Cin) ** -0.5)
        ekw.update(M=M, conv=(H, W, H, W, 1, 0), bias=rnd(Cout, seed=3), res=rnd(M, Cout, seed=4))
    elif case == "conv stride 2":
        Cin, Cout, H, W, nimg = 640, 1280, 16, 16, 24
        M = nimg * 8 * 8
        x, w = rnd(nimg * H * W, Cin, seed=1), rnd(Cout, 9, Cin, seed=2, scale=(9 * Cin) ** -0.5)
        ekw.update(M=M, conv=(H, W, 8, 8, 2, 0), bias=rnd(Cout, seed=3))
    elif case == "tconv":
        C, frames, npix, chunk, nb = 1280, 24, 64, 8, 1
        M = nb * frames * npix
        x, w = rnd(M, C, seed=1), rnd(C, 3, C, seed=2, scale=(3 * C) ** -0.5)
        ekw.update(tconv=(frames, npix, chunk), bias=rnd(C, seed=3), res=rnd(M, C, seed=4))
    else:
        M, N, K = (3072, 1280, 2048) if "rowvec" in case else (3000, 1280, 2560)
        x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
        ekw.update(bias=rnd(N, seed=3), rowvec=rnd(8, N, seed=6), rows_per_vec=(M + 7) // 8, res=rnd(M, N, seed=4), res2=rnd(M, N, seed=5))
        if "silu" in case:
            ekw["act"] = 2
    for k, v in ekw.items():
        kw[k] = cu(v) if isinstance(v, torch.Tensor) else v
    xg, wg = cu(x), cu(w)
    monkeypatch.setenv("ME_GEMM_SPLITK", "0")
    ref = ops.gemm(xg, wg, **kw)
    assert "splitk" not in ops._last_kernel(), ops._last_kernel()
    monkeypatch.delenv("ME_GEMM_SPLITK")
    got = ops.gemm(xg, wg, **kw)
    assert ops._last_kernel().endswith("+splitk"), ops._last_kernel()
    assert torch.equal(got, ops.gemm(xg, wg, **kw))                       # fixed-order reduction: run to run bitwise
    check(got, ref, f"split-K vs one pass {case}", rel=5e-4, mx=1e-2)     # same products, another fp32 summation order (+ one fp16 rounding)
    check(got, emu.gemm(x, w, **ekw), f"split-K gemm {case}")


@pytest.mark.parametrize("f,N", [(24, 4096), (48, 9216)])
def test_full_size_properties_of_the_level0_kernels(ops, f, N):
    """(48, 9216): the same at BASELINE configs[4]'s level-0 size -- 48 frames x 96 x 96 latents, batch 4: 1 769 472 rows, 18 query blocks per (item, head).
    Size-independent properties at the benchmarked geometry (24 frames x 64 x 64 latents, batch 4: 393216 rows), where no reference
    can be computed in the test: (1) a GEMM is linear in its activations and exact on one-hot activations; (2) an attention whose values are
    all equal returns that value whatever the scores, and whose V is the one-hot of the key index returns rows that sum to 1 (the softmax
    weights); (3) LayerNorm output rows have zero mean and unit variance; (4) GroupNorm output groups likewise."""
    from motioneditor_amd import segments
    M, C = 4 * f * N, 320
    g = torch.Generator(device="cuda").manual_seed(3)
    x1 = (torch.randn(M, C, device="cuda", generator=g) * 0.5).half()
    # (1a) exactness on one-hot rows: row m selects column (m % C) of W^T
    w = (torch.randn(3 * C, 1, C, device="cuda", generator=g) * C ** -0.5).half()
    eye = torch.zeros(M, C, dtype=torch.float16, device="cuda")
    eye[torch.arange(M, device="cuda"), torch.arange(M, device="cuda") % C] = 1.0
    y = ops.gemm(eye, w)
    wt = w[:, 0, :].t().contiguous()
    rows = torch.cat([torch.arange(0, 2 * C), torch.arange(M // 2 - C, M // 2 + C), torch.arange(M - 2 * C, M)]).cuda()   # first, middle and last tiles
    assert torch.equal(y[rows], wt[rows % C])
    # (1b) linearity with an exactly representable scaling: gemm(2 x) == 2 gemm(x) bitwise wherever the result is a normal fp16 number (a power of
    # two commutes with every rounding except the one into fp16's subnormals)
    y1, y2 = ops.gemm(x1, w), ops.gemm(x1 * 2, w)
    normal = y1.abs() >= 2.0 ** -13
    assert torch.equal(y2[normal], (y1 * 2)[normal]) and float((y2.float() - 2 * y1.float()).abs().max()) <= 2.0 ** -23
    # (2) attention, [prev | cur] segments, dh = 40
    B, dh = 4, 40
    qkv = (torch.randn(M, 3 * C, device="cuda", generator=g) * 0.7).half()
    si, sm = segments.prev_cur(B, f, "cuda")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=N, nk=N, seg_item=si, seg_mode=sm)
    vconst = torch.full((M, C), 0.75, dtype=torch.float16, device="cuda")
    o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], vconst, **args)
    assert float((o.float() - 0.75).abs().max()) <= 1e-3          # sum_j p_j * 0.75 with sum_j p_j = 1 (fp16 P, fp32 accumulation)
    # (3) LayerNorm rows
    ln = ops.layernorm(x1, torch.ones(C, dtype=torch.float16, device="cuda"), torch.zeros(C, dtype=torch.float16, device="cuda")).float()
    assert float(ln.mean(1).abs().max()) < 2e-3 and float((ln.var(1, unbiased=False) - 1).abs().max()) < 5e-3
    # (4) GroupNorm over (frames x pixels x 10 channels) per batch entry
    gn = ops.groupnorm(x1, torch.ones(C, dtype=torch.float16, device="cuda"), torch.zeros(C, dtype=torch.float16, device="cuda"), rows_per_group=f * N, eps=1e-5,
                       silu=False).float().reshape(4, f * N, 32, 10)
    assert float(gn.mean((1, 3)).abs().max()) < 1e-3 and float((gn.var((1, 3), unbiased=False) - 1).abs().max()) < 2e-3


@pytest.mark.parametrize("dh,N,kind", [(40, 9216, "pc"), (40, 9216, "ed_bin"), (80, 2304, "pc"), (80, 2304, "ed_bin"), (160, 576, "pc")])
def test_attention_at_the_96x96_latent_geometry(ops, dh, N, kind):
    """BASELINE configs[4]'s spatial geometry (768^2 images): 9216 tokens at level 0 (18 query blocks of 512, 36 stages of 256 keys per segment),
    2304 at level 1 (dh = 80: 18 blocks of 128 queries, 36 key tiles per segment), 576 at level 2 -- non-power-of-two block / tile counts that the
    64x64-latent tests never see.  (1) constant values come back whatever the scores; (2) the output matches the fp32 reference on a random subset
    of 160 query rows per item; (3) the head-major K | V form is bitwise the row-major one."""
    from motioneditor_amd import segments
    C = 8 * dh
    if kind == "pc":
        B, f = 1, 2
        si, sm = segments.prev_cur(B, f, "cpu")
        n_items, mask = B * f, None
    else:
        B, f = 2, 1
        si, sm = segments.edited_spatial(f, "cpu", binary_mask=True, B=B)
        mask = (torch.rand(8, N, generator=torch.Generator().manual_seed(11)) > 0.5).half()
        n_items = B * f
    qkv = rnd(n_items * N, 3 * C, seed=3)
    qkv[N // 2 + 5, C:2 * C] *= 4.0
    qkv[N - 3, C:2 * C] *= 6.0
    args = dict(heads=8, dh=dh, n_items=n_items, nk=N)
    dq = cu(qkv)
    kw = dict(seg_item=cu(si), seg_mode=cu(sm), mask=None if mask is None else cu(mask), nq=N, **args)
    got = ops.attention(dq[:, :C], dq[:, C:2 * C], dq[:, 2 * C:], **kw)
    vconst = torch.full((n_items * N, C), 0.75, dtype=torch.float16, device="cuda")
    o = ops.attention(dq[:, :C], dq[:, C:2 * C], vconst, **kw)
    assert float((o.float() - 0.75).abs().max()) <= 1e-3
    kv = dq[:, C:].reshape(n_items * N, 16, dh).permute(1, 0, 2).contiguous()
    assert torch.equal(ops.attention(dq[:, :C], kv[:8], kv[8:], **kw), got)
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:160].sort().values
    rows = torch.cat([it * N + idx for it in range(n_items)])
    want = emu.attention(qkv[rows, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, mask=mask, nq=idx.numel(), **args)
    peak = float(want.float().abs().max() / want.float().abs().mean())
    check(got.cpu()[rows], want, f"attn 96x96 geometry dh={dh} N={N} {kind}", mx=2e-3 * peak)


def test_gemm_rejects_bad_arguments(ops):
    x, w = cu(rnd(16, 12)), cu(rnd(8, 1, 12))
    with pytest.raises(ValueError):
        ops.gemm(x, w)  # K = 12 is not a multiple of 8


# ------------------------------------------------------------------ conv_small
def test_conv_small_5d_latents_and_images(ops):
    B, f, h, w = 2, 3, 6, 5
    lat = torch.randn(B, 4, f, h, w, generator=torch.Generator().manual_seed(1))
    W = rnd(320, 9, 4, seed=2, scale=1 / 6).float()
    b = rnd(320, seed=3).float()
    kw = dict(n_img=B * f, Cin=4, H=h, Wd=w, img_stride=4 * f * h * w, ch_stride=f * h * w, frames=f, frame_stride=h * w)
    check(ops.conv_small(cu(lat), cu(W), cu(b), **kw), emu.conv_small(lat, W, b, **kw), "conv_in 5-D")
    img = torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(4))
    W = rnd(16, 9, 3, seed=5, scale=1 / 5).float()
    kw = dict(n_img=4, Cin=3, H=16, Wd=16, img_stride=3 * 256, ch_stride=256, silu=True)
    check(ops.conv_small(cu(img), cu(W), cu(rnd(16, seed=6).float()), **kw), emu.conv_small(img, W, rnd(16, seed=6), **kw), "cond conv_in")


# ------------------------------------------------------------------ attention
def seg(rows, modes):
    return torch.tensor(rows, dtype=torch.int32), torch.tensor(modes, dtype=torch.int32)


@pytest.mark.parametrize("dh", [40, 80, 160])
@pytest.mark.parametrize("nq", [1, 16, 64, 100, 256])
def test_attention_prev_cur(ops, dh, nq):
    from motioneditor_amd import segments
    B, f, C = 2, 3, 8 * dh
    qkv = rnd(B * f * nq, 3 * C, seed=1)
    si, sm = segments.prev_cur(B, f, "cpu")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=nq, nk=nq)
    got = ops.attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), **args)
    want = emu.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, **args)
    check(got, want, f"attn prev|cur dh={dh} nq={nq}")


@pytest.mark.parametrize("dh,nq", [(40, 64), (80, 16), (160, 4), (40, 256)])
def test_attention_cross_text_77_keys(ops, dh, nq):
    from motioneditor_amd import segments
    B, f, C = 4, 2, 8 * dh
    q, kv = rnd(B * f * nq, C, seed=1), rnd(B * 77, 2 * C, seed=2)
    si, sm = segments.cross_text(B, f, "cpu")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=nq, nk=77)
    got = ops.attention(cu(q), cu(kv)[:, :C], cu(kv)[:, C:], seg_item=cu(si), seg_mode=cu(sm), **args)
    check(got, emu.attention(q, kv[:, :C], kv[:, C:], seg_item=si, seg_mode=sm, **args), f"attn cross dh={dh}")


@pytest.mark.parametrize("dh,nq,nk,B,f,il", [(40, 4096, 77, 2, 3, False), (40, 1024, 77, 4, 2, True), (40, 600, 80, 2, 2, False), (40, 512, 65, 1, 2, False),
                                             (80, 1024, 77, 2, 3, False), (80, 320, 70, 2, 2, True), (80, 256, 77, 4, 24, False),
                                             (160, 256, 77, 4, 6, False), (160, 576, 77, 2, 3, True), (160, 300, 66, 2, 2, False)])
def test_attention_keys_resident_across_query_blocks(ops, dh, nq, nk, B, f, il):
    """Round 6: the text cross-attention (attention_2d.py:343; one segment of 65 .. 80 keys) stages its K | V once per block, walks several query blocks over
    it and takes all five 16-key tiles of a query in one pass (attn2_kernel<..., KVRES>: plain softmax, no running rescale).  Against the fp32 reference and
    against the per-query-block launch (ME_ATTN_KVRES=0: the online softmax over two 64-key tiles -- another summation order, so close, not bitwise);
    bitwise among walks of different lengths (ME_ATTN_KVRES=n: n query blocks per block, a divisor of the item's block count or not).  Cases: the level-0 /
    level-1 shapes, a ragged last query block, 80 and 65 keys, the ControlNet's interleaved text rows (pipeline_motion_editor.py:615,621), enough items
    that the launcher's own choice walks only part of an item."""
    import os
    from motioneditor_amd import segments
    C = 8 * dh
    q, kv = rnd(B * f * nq, C, seed=1), rnd(B * nk, 2 * C, seed=2)
    si, sm = segments.cross_interleaved(B * f, B, "cpu") if il else segments.cross_text(B, f, "cpu")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=nq, nk=nk)
    cq, ck, cv, csi, csm = cu(q), cu(kv)[:, :C], cu(kv)[:, C:], cu(si), cu(sm)
    lse = torch.empty(B * f * nq, 8, dtype=torch.float32, device="cuda")   # (the log-sum-exp the backward reads: written by both forms)
    got = ops.attention(cq, ck, cv, seg_item=csi, seg_mode=csm, lse=lse, **args)   # (the launcher's own choice of the walk)
    assert "kvres" in ops._last_kernel()
    try:
        os.environ["ME_ATTN_KVRES"] = "0"
        lse_ref = torch.empty_like(lse)
        ref = ops.attention(cq, ck, cv, seg_item=csi, seg_mode=csm, lse=lse_ref, **args)
        assert "kvres" not in ops._last_kernel()
        assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * float(ref.float().abs().max())
        assert float((lse - lse_ref).abs().max()) <= 1e-4 * max(1.0, float(lse_ref.abs().max()))
        for force in ("2", "3", "16"):
            os.environ["ME_ATTN_KVRES"] = force
            walk = ops.attention(cq, ck, cv, seg_item=csi, seg_mode=csm, **args)
            assert "kvres" in ops._last_kernel()
            assert torch.equal(walk, got), force
    finally:
        os.environ.pop("ME_ATTN_KVRES", None)
    check(got, emu.attention(q, kv[:, :C], kv[:, C:], seg_item=si, seg_mode=sm, **args), f"attn cross kvres dh={dh} nq={nq} nk={nk}")


@pytest.mark.parametrize("kind,N", [("pc", 1024), ("ed_bin", 1024), ("pc", 300)])
def test_attention_dh80_32_queries_per_wave(ops, kind, N):
    """Round 6: the dh = 80 launches with whole 256-query blocks take 8 waves x 32 queries and 128-key stages (every K / V fragment read feeds two MFMAs).
    Same tile arithmetic per query as the 16-queries-per-wave kernel (ME_ATTN_80_QT2=0) -- only the re-basing points of the fixed-offset softmax move with the
    stage length, so the two agree to rounding (bitwise on data that never re-bases) -- and both against the fp32 reference."""
    import os
    from motioneditor_amd import segments
    dh, f = 80, 3
    C = 8 * dh
    B = 4 if kind == "ed_bin" else 2
    qkv = rnd(B * f * N, 3 * C, seed=5)
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(8, N, generator=g) > 0.5).half() if kind == "ed_bin" else None
    si, sm = segments.edited_spatial(f, "cpu", True) if kind == "ed_bin" else segments.prev_cur(B, f, "cpu")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=N, nk=N)
    c = cu(qkv)
    run = lambda: ops.attention(c[:, :C], c[:, C:2 * C], c[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), mask=None if mask is None else cu(mask), **args)
    got = run()
    assert ops._last_kernel() == "attn2_kernel<80,2,8,fold>"
    try:
        os.environ["ME_ATTN_80_QT2"] = "0"
        ref = run()
        assert ops._last_kernel() == "attn2_kernel<80,1,8,fold>"
    finally:
        os.environ.pop("ME_ATTN_80_QT2", None)
    assert float((got.float() - ref.float()).abs().max()) <= 1e-3 * float(ref.float().abs().max())
    want = emu.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, mask=mask, **args)
    check(got, want, f"attn dh=80 QT=2 {kind} N={N}")


@pytest.mark.parametrize("dh,N,binary", [(40, 64, True), (80, 16, True), (40, 256, True), (40, 64, False), (80, 144, True)])
def test_attention_edited_dual_mask_5N_keys(ops, dh, N, binary):
    """The spatial editor's masked attention: recon rows [prev|cur], edit rows [src prev dual | src cur dual | own cur]
    with head-indexed mask planes (fully_control.py:372-447)."""
    from motioneditor_amd import segments
    f, C = 8, 8 * dh
    qkv = rnd(4 * f * N, 3 * C, seed=1)
    g = torch.Generator().manual_seed(7)
    mask = (torch.rand(8, N, generator=g) > 0.5).half() if binary else torch.rand(8, N, generator=g).half()
    si, sm = segments.edited_spatial(f, "cpu")
    args = dict(heads=8, dh=dh, n_items=4 * f, nq=N, nk=N)
    got = ops.attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), mask=cu(mask), **args)
    want = emu.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, mask=mask, **args)
    check(got, want, f"attn edited dh={dh} N={N} binary={binary}")


@pytest.mark.parametrize("dh,N", [(40, 64), (80, 100), (40, 256)])
def test_attention_edited_binary_fast_path_equals_general_dual(ops, dh, N):
    """DUAL_BIN (no mask read) must equal the general dual-mask formula for any BINARY mask."""
    from motioneditor_amd import segments
    f, C = 8, 8 * dh
    qkv = rnd(4 * f * N, 3 * C, seed=1)
    mask = (torch.rand(8, N, generator=torch.Generator().manual_seed(7)) > 0.4).half()
    si, sm = segments.edited_spatial(f, "cpu", binary_mask=True)
    sg, smg = segments.edited_spatial(f, "cpu", binary_mask=False)
    args = dict(heads=8, dh=dh, n_items=4 * f, nq=N, nk=N)
    got = ops.attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), mask=cu(mask), **args)
    want = emu.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=sg, seg_mode=smg, mask=mask, **args)
    check(got, want, f"attn edited binary fast path dh={dh} N={N}")


@pytest.mark.parametrize("dh,N,kind", [(40, 4096, "pc"), (40, 4096, "ed_bin"), (40, 4096, "ed_gen"), (80, 1024, "pc"), (80, 1024, "ed_bin"), (80, 1024, "ed_gen")])
def test_attention_production_size(ops, dh, N, kind):
    """The benchmarked launch geometry (level 0: 4096 queries x 2 / 3 segments of 4096 keys = 128 / 192 key tiles, 16 query
    blocks of 256; level 1: 1024 x 1024): every query block and every rescale of the online softmax is exercised; the fp32
    reference is evaluated on a random subset of 192 query rows per item (the full score matrix of one edited item is 2.7 GB)."""
    from motioneditor_amd import segments
    C = 8 * dh
    if kind == "pc":
        B, f = 1, 2
        si, sm = segments.prev_cur(B, f, "cpu")
        n_items, mask = B * f, None
    else:
        B, f = 2, 1                                    # one (recon, edit) pair, one frame: the edit item attends 3 segments (5N reference keys)
        si, sm = segments.edited_spatial(f, "cpu", binary_mask=(kind == "ed_bin"), B=B)
        g = torch.Generator().manual_seed(11)
        mask = (torch.rand(8, N, generator=g) > 0.5).half() if kind == "ed_bin" else torch.rand(8, N, generator=g).half()
        n_items = B * f
    qkv = rnd(n_items * N, 3 * C, seed=3)
    qkv[N // 2 + 5, C:2 * C] *= 4.0                    # a few dominant keys: late jumps of the running max
    qkv[N - 3, C:2 * C] *= 6.0
    args = dict(heads=8, dh=dh, n_items=n_items, nk=N)
    got = ops.attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), mask=None if mask is None else cu(mask), nq=N, **args)
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:192].sort().values
    rows = torch.cat([it * N + idx for it in range(n_items)])
    want = emu.attention(qkv[rows, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, mask=mask, nq=idx.numel(), **args)
    # averaging over thousands of keys makes the typical output tiny next to the rows a dominant key pulls to |V| ~ 0.5: bound the
    # largest error by the largest output (fp16 rounding of the output is 5e-4 of it) rather than by the mean
    peak = float(want.float().abs().max() / want.float().abs().mean())
    check(got.cpu()[rows], want, f"attn production size dh={dh} N={N} {kind}", mx=2e-3 * peak)


@pytest.mark.parametrize("dh,nq,nk,where", [(40, 512, 1024, "late"), (40, 512, 1024, "some"), (40, 96, 640, "late"), (80, 256, 512, "late"), (80, 64, 512, "some"),
                                            (40, 512, 1024, "edge")])
def test_attention_fixed_offset_overflow_falls_back_to_running_max(ops, dh, nq, nk, where):
    """The fast path of attn2 fixes every query's softmax offset from the FIRST 64-key tile and checks once per block that no
    P reached 2^15; blocks that did are recomputed in-kernel with the classic running maximum.  'late': every query meets keys
    ~ e^35 heavier than its first tile (all blocks fall back); 'some': only the queries of one 16-row group do (a single block
    falls back, the others keep the fast result); 'edge': the heavy key sits 14.5 binades up -- just inside fp16's range, no
    fallback, P up to 2^14.5 next to P ~ 1."""
    from motioneditor_amd import segments
    C, n_items = 8 * dh, 2
    g = torch.Generator().manual_seed(21)
    q = torch.randn(n_items * nq, C, generator=g) * 0.5
    k = torch.randn(n_items * nk, C, generator=g) * 0.5
    v = torch.randn(n_items * nk, C, generator=g)
    scale = dh ** -0.5
    hot = nk - 70                                       # a key of the last-but-one tile
    for it in range(n_items):
        rows = range(nq) if where != "some" else range(32, 48)
        for h in range(8):
            qs = q[it * nq:(it + 1) * nq, h * dh:(h + 1) * dh]
            # make key `hot` parallel to the mean query direction of the chosen rows: logit = |q| |k| cos ~ target nats
            d = qs[list(rows)].mean(0)
            d = d / d.norm()
            target = 10.0 if where == "edge" else 35.0   # nats above the typical logit (|.| < 2)
            qs[list(rows)] += d * 3.0                   # give those queries a common component of length 3
            k[it * nk + hot, h * dh:(h + 1) * dh] = d * (target / (3.0 * scale))
    q, k, v = q.half(), k.half(), v.half()
    si, sm = segments.self_items(n_items, "cpu")
    args = dict(heads=8, dh=dh, n_items=n_items, nq=nq, nk=nk, seg_item=si, seg_mode=sm)
    got = ops.attention(cu(q), cu(k), cu(v), **{**args, "seg_item": cu(si), "seg_mode": cu(sm)})
    want = emu.attention(q, k, v, **args)
    assert torch.isfinite(got).all()
    check(got, want, f"attn fixed-offset fallback dh={dh} nq={nq} {where}", rel=2e-3, mx=2e-2)
    ops.attention_fallback_blocks(reset=True)
    assert torch.equal(got, ops.attention(cu(q), cu(k), cu(v), **{**args, "seg_item": cu(si), "seg_mode": cu(sm)}))   # run to run
    nfb = ops.attention_fallback_blocks()
    assert where == "edge" or nfb > 0, (where, nfb)   # the sudden 2^50 jump is what phase B is for ("edge": 2^14.4 +- the queries' own spread: either path)


@pytest.mark.parametrize("dh,nq,nk", [(40, 512, 2048), (80, 256, 1024), (40, 96, 640)])
def test_attention_fixed_offset_rebases_under_gradual_growth(ops, dh, nq, nk):
    """Keys get steadily heavier along the key order (a common component of the keys grows linearly: +0.03 nats per key, i.e. 5.5 binades
    per 128-key stage, 61 nats = 88 binades over 2048 keys): the first queries' probe tiles promise far lighter keys than the sweep then
    meets.  The per-stage re-basing of the fixed offset (denominator > 2^9 -> offset += d = ceil(log2) - 5, accumulators x 2^-d) must keep every
    block on the fast path -- no block may fall back to the running-maximum sweep -- and the result must match the reference softmax."""
    from motioneditor_amd import segments
    C, n_items = 8 * dh, 2
    g = torch.Generator().manual_seed(23)
    q = torch.randn(n_items * nq, C, generator=g) * 0.5
    k = torch.randn(n_items * nk, C, generator=g) * 0.5
    v = torch.randn(n_items * nk, C, generator=g)
    scale = dh ** -0.5
    ramp = 0.03 * torch.arange(nk, dtype=torch.float32)   # nats added to the logit of key j for every query
    for it in range(n_items):
        for h in range(8):
            qs = q[it * nq:(it + 1) * nq, h * dh:(h + 1) * dh]
            d = torch.randn(dh, generator=g)
            d = d / d.norm()
            qs -= (qs @ d)[:, None] * d                 # every query gets the SAME component 3 along d ...
            qs += 3.0 * d
            ks = k[it * nk:(it + 1) * nk, h * dh:(h + 1) * dh]
            ks -= (ks @ d)[:, None] * d                 # ... and key j the component ramp_j / (3 scale): logit += ramp_j
            ks += (ramp / (3.0 * scale))[:, None] * d
    q, k, v = q.half(), k.half(), v.half()
    si, sm = segments.self_items(n_items, "cpu")
    args = dict(heads=8, dh=dh, n_items=n_items, nq=nq, nk=nk, seg_item=si, seg_mode=sm)
    ops.attention_fallback_blocks(reset=True)
    got = ops.attention(cu(q), cu(k), cu(v), **{**args, "seg_item": cu(si), "seg_mode": cu(sm)})
    nfb = ops.attention_fallback_blocks()
    want = emu.attention(q, k, v, **args)
    assert torch.isfinite(got).all()
    check(got, want, f"attn fixed-offset re-base dh={dh} nq={nq}", rel=2e-3, mx=2e-2)
    assert nfb == 0, f"{nfb} blocks fell back to the running-maximum sweep"


def test_attention_large_logits_online_softmax_rescale(ops):
    """Force the running max to jump late (a spiked key in the LAST tile) so the rescale path matters."""
    dh, nq, nk, C = 40, 64, 300, 320
    q, k, v = rnd(nq, C, seed=1), rnd(nk, C, seed=2), rnd(nk, C, seed=3)
    k[-3] = q[5] * 6.0
    k[10] = q[7] * 4.0
    si, sm = seg([[0]], [[0]])
    args = dict(heads=8, dh=dh, n_items=1, nq=nq, nk=nk)
    check(ops.attention(cu(q), cu(k), cu(v), seg_item=cu(si), seg_mode=cu(sm), **args), emu.attention(q, k, v, seg_item=si, seg_mode=sm, **args), "attn spike")


def test_attention_adapter_chunked_first_prev(ops):
    from motioneditor_amd import segments
    dh, N, B, f = 80, 16, 2, 16
    C = 8 * dh
    qkv = rnd(B * f * N, 3 * C, seed=1)
    si, sm = segments.first_prev_chunked(B, f, 8, "cpu")
    args = dict(heads=8, dh=dh, n_items=B * f, nq=N, nk=N)
    got = ops.attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], seg_item=cu(si), seg_mode=cu(sm), **args)
    check(got, emu.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], seg_item=si, seg_mode=sm, **args), "attn first|prev chunk 8")


# ------------------------------------------------------------------ temporal attention
@pytest.mark.parametrize("F,dh,npix,kv_map", [(8, 40, 5, None), (16, 80, 3, [0, 0, 2, 2]), (24, 40, 4, [0, 0, 2, 2]), (24, 160, 1, None), (48, 40, 2, None)])
def test_temporal_attention(ops, F, dh, npix, kv_map):
    B, C = 4, 8 * dh
    qkv = rnd(B * F * npix, 3 * C, seed=1)
    args = dict(heads=8, dh=dh, batch=B, frames=F, npix=npix, kv_map=kv_map)
    got = ops.temporal_attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], **args)
    check(got, emu.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], **args), f"tattn F={F} dh={dh}")


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("C,rows,rpg,silu", [(320, 4 * 16 * 64, 16 * 64, True), (640, 96, 8, False), (960, 2 * 300, 300, True), (1280, 64, 16, True),
                                             (1920, 128, 64, True), (2560, 8 * 33, 33, True), (320, 6 * 1000, 1000, False)])
def test_groupnorm(ops, C, rows, rpg, silu):
    x = (rnd(rows, C, seed=1) * 2 + 0.7).half()
    gm, bt = (1 + 0.1 * rnd(C, seed=2)).half(), (0.1 * rnd(C, seed=3)).half()
    got = ops.groupnorm(cu(x), cu(gm), cu(bt), rows_per_group=rpg, eps=1e-5, silu=silu)
    check(got, emu.groupnorm(x, gm, bt, rows_per_group=rpg, eps=1e-5, silu=silu), f"groupnorm C={C}")


@pytest.mark.parametrize("C,rows,rpg,mean", [(320, 4096, 2048, 100.0), (640, 6 * 1024, 6 * 1024, 300.0), (1280, 512, 64, 40.0)])
def test_groupnorm_large_mean_small_spread(ops, C, rows, rpg, mean):
    """Activations whose group mean dwarfs their spread (SD feature maps: |x| ~ 1e2 with sub-unit variation): E[x^2] - mean^2
    cancels catastrophically in fp32; the kernel keeps the statistics in fp64."""
    x = (mean + 0.25 * rnd(rows, C, seed=1)).half()     # fp16 resolution at 300 is 0.25: the data itself is coarse, the reference sees the same
    gm, bt = (1 + 0.1 * rnd(C, seed=2)).half(), (0.1 * rnd(C, seed=3)).half()
    got = ops.groupnorm(cu(x), cu(gm), cu(bt), rows_per_group=rpg, eps=1e-5, silu=False)
    want = emu.groupnorm(x.double(), gm.double(), bt.double(), rows_per_group=rpg, eps=1e-5, silu=False)
    check(got, want.float(), f"groupnorm large mean C={C} mean={mean}")


def test_groupnorm_is_bitwise_reproducible(ops):
    x = cu((rnd(4 * 96 * 64, 320, seed=1) * 2 + 0.7).half())
    gm, bt = cu((1 + 0.1 * rnd(320, seed=2)).half()), cu((0.1 * rnd(320, seed=3)).half())
    outs = [ops.groupnorm(x, gm, bt, rows_per_group=96 * 64, eps=1e-5, silu=True).clone() for _ in range(5)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("C,rows", [(320, 1000), (320, 4096 * 3 + 5), (320, 70001), (640, 77), (1280, 130)])   # >= 4096 rows of 320: the half-wave-per-row kernel, odd tails
def test_layernorm(ops, C, rows):
    x = (rnd(rows, C, seed=1) * 3 - 0.4).half()
    gm, bt = (1 + 0.1 * rnd(C, seed=2)).half(), (0.1 * rnd(C, seed=3)).half()
    check(ops.layernorm(cu(x), cu(gm), cu(bt)), emu.layernorm(x, gm, bt), f"layernorm C={C}")


@pytest.mark.parametrize("rows,cols", [(64, 64), (130, 1024), (77, 4096), (9, 8192), (33, 72)])
def test_softmax_rows(ops, rows, cols):
    # peaked rows: the largest probability carries the fp16 rounding error, so the max-abs criterion is relative to
    # the output's largest value, not its mean
    x = (rnd(rows, cols, seed=1) * 4).half()
    mx = MAX_REL * cols
    check(ops.softmax_rows(cu(x)), emu.softmax_rows(x), f"softmax {rows}x{cols}", mx=mx)
    xg = cu(x).clone()
    ops.softmax_rows(xg, out=xg)   # in place
    check(xg, emu.softmax_rows(x), "softmax in place", mx=mx)
    view = cu(rnd(rows, cols + 8, seed=2))[:, 8:]   # strided view (ldx != cols) keeps 16-byte alignment
    check(ops.softmax_rows(view), emu.softmax_rows(view.cpu()), "softmax strided", mx=mx)
    assert float((ops.softmax_rows(cu(x)).float().sum(1) - 1).abs().max()) < 2e-3


# ------------------------------------------------------------------ element-wise
def test_elementwise_family(ops):
    x, a = rnd(300, 640, seed=1), rnd(300, 640, seed=2)
    y = cu(x).clone()
    ops.axpy_rows(y[100:200], y[100:200], cu(a)[:100], 0.5)
    want = x.clone().float()
    want[100:200] += 0.5 * a[:100].float()
    check(y, want, "axpy_rows")
    dst = torch.zeros(300, 1280, dtype=torch.float16, device="cuda")
    ops.copy_rows(dst[:, 640:], cu(x))
    assert torch.equal(dst[:, 640:].cpu(), x) and float(dst[:, :640].abs().max()) == 0
    check(ops.silu(cu(x)), emu.silu(x), "silu")
    check(ops.relu(cu(x)), emu.relu(x), "relu")
    check(ops.timestep_embed(3, 320, 981.0, "cuda"), emu.timestep_embed(3, 320, 981.0, "cpu"), "timestep_embed", rel=2e-3)
    lat = torch.randn(2, 4, 3, 5, 7, generator=torch.Generator().manual_seed(3))
    eps = rnd(4 * 3 * 35, 4, seed=4)
    check(ops.cfg_ddim(cu(lat), cu(eps), guidance=7.5, ca=1.01, cb=-0.07), emu.cfg_ddim(lat, eps, guidance=7.5, ca=1.01, cb=-0.07), "cfg_ddim", rel=1e-5, mx=1e-4)
    t5 = torch.randn(2, 8, 3, 4, 5, generator=torch.Generator().manual_seed(5))
    rows = ops.nchw5_to_rows(cu(t5))
    check(rows, emu.nchw5_to_rows(t5), "nchw5_to_rows")
    back = ops.rows_to_nchw5(rows, 2, 8, 3, 4, 5)
    check(back, t5.half().float(), "rows_to_nchw5", rel=1e-6, mx=1e-6)


def test_ddim_matches_reference_golden(ops):
    """me_cfg_ddim + DDIMScheduler coefficients against the reference's own prev_step vectors
    (tests/golden/ddim.npz, p2p/null_text_optimization.py:26-36)."""
    from conftest import GOLD
    from motioneditor_amd.schedulers import DDIMScheduler
    g = np.load(GOLD / "ddim.npz")
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps == g["timesteps"].tolist()
    x = torch.from_numpy(g["x"]).reshape(2, 4, 1, 8, 8)
    e = torch.from_numpy(g["eps"]).reshape(2, 4, 1, 8, 8)
    for t in (981, 501, 21, 1):
        out = s.step(cu(e), t, cu(x)).prev_sample
        check(out, torch.from_numpy(g[f"prev_{t}"]).reshape(2, 4, 1, 8, 8), f"ddim t={t}", rel=1e-3, mx=5e-3)  # eps passes through fp16 rows


# ------------------------------------------------------------------ frame-sharding features of the kernels
@pytest.mark.parametrize("F,parts,qf,q0,dh", [(16, 2, 8, 8, 40), (24, 3, 8, 8, 80), (24, 4, 6, 18, 40), (16, 1, 16, 0, 160)])
def test_temporal_attention_sharded_queries_part_major_kv(ops, F, parts, qf, q0, dh):
    B, npix, C = 4, 3, 8 * dh
    q = rnd(B * qf * npix, C, seed=1)
    kv = rnd(parts * B * (F // parts) * npix, 2 * C, seed=2)      # all-gathered K|V, part-major
    args = dict(heads=8, dh=dh, batch=B, frames=F, npix=npix, kv_map=[0, 0, 2, 2], q_frames=qf, q_frame0=q0, kv_parts=parts)
    got = ops.temporal_attention(cu(q), cu(kv)[:, :C], cu(kv)[:, C:], **args)
    check(got, emu.temporal_attention(q, kv[:, :C], kv[:, C:], **args), f"tattn sharded F={F} parts={parts}")


@pytest.mark.parametrize("F,parts,dh,npix", [(16, 2, 40, 5), (24, 4, 80, 3), (24, 8, 160, 2)])
def test_temporal_attention_pixel_sharded_part_major_q_and_kv(ops, F, parts, dh, npix):
    """After the frame<->pixel all-to-all q, k, v and the output are all part-major (q_parts = kv_parts); the result must be
    the plain temporal attention of the same rows put back in (b, frame, pixel) order."""
    B, C, fpp = 4, 8 * dh, F // parts
    qkv = rnd(parts * B * fpp * npix, 3 * C, seed=3)
    args = dict(heads=8, dh=dh, batch=B, frames=F, npix=npix, kv_map=[0, 0, 2, 2])
    got = ops.temporal_attention(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], kv_parts=parts, q_parts=parts, **args)
    check(got, emu.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], kv_parts=parts, q_parts=parts, **args), f"tattn a2a F={F} parts={parts}")
    # same numbers as the unsharded kernel on the rows in (b, frame, pixel) order
    plain = qkv.reshape(parts, B, fpp, npix, 3 * C).permute(1, 0, 2, 3, 4).reshape(B * F * npix, 3 * C).contiguous()
    ref = ops.temporal_attention(cu(plain)[:, :C], cu(plain)[:, C:2 * C], cu(plain)[:, 2 * C:], **args)
    ref = ref.reshape(B, parts, fpp, npix, C).permute(1, 0, 2, 3, 4).reshape(-1, C)
    assert torch.equal(got, ref)


def test_copy_blocks_is_the_frame_pixel_reorder(ops):
    R, BF, Ns, W = 4, 6, 16, 48
    x = rnd(BF * R * Ns, W + 8, seed=4)
    y = torch.zeros(R * BF * Ns, W, dtype=torch.float16, device="cuda")
    ops.copy_blocks(y, cu(x)[:, :W], R, BF, Ns, ys0=BF * Ns, ys1=Ns, xs0=Ns, xs1=R * Ns)
    want = x[:, :W].reshape(BF, R, Ns, W).permute(1, 0, 2, 3).reshape(-1, W)
    assert torch.equal(y.cpu(), want)
    back = torch.zeros(BF * R * Ns, W, dtype=torch.float16, device="cuda")
    ops.copy_blocks(back, y, R, BF, Ns, ys0=Ns, ys1=R * Ns, xs0=BF * Ns, xs1=Ns)
    assert torch.equal(back.cpu(), x[:, :W])
    with pytest.raises(ValueError):
        ops.copy_blocks(back, y, R + 1, BF, Ns, ys0=Ns, ys1=R * Ns, xs0=BF * Ns, xs1=Ns)


@pytest.mark.parametrize("C,f_loc,f_tot,frame0,chunk,npix,nb", [(320, 8, 24, 8, 24, 4, 2), (320, 12, 24, 12, 8, 4, 1), (640, 6, 24, 0, 8, 2, 2), (320, 6, 24, 18, 24, 3, 4)])
def test_gemm_tconv_sharded_with_halos(ops, C, f_loc, f_tot, frame0, chunk, npix, nb):
    rows = nb * f_loc * npix
    hb = nb * npix
    x = rnd(rows + 2 * hb, C, seed=1)                              # [local | prev halo | next halo]
    w = rnd(C, 3, C, seed=2, scale=(3 * C) ** -0.5)
    bias, res = rnd(C, seed=3), rnd(rows, C, seed=4)
    hp = rows if frame0 > 0 else -1
    hn = rows + hb if frame0 + f_loc < f_tot else -1
    tc = (f_loc, npix, chunk, frame0, f_tot, hp, hn)
    got = ops.gemm(cu(x), cu(w), M=rows, bias=cu(bias), tconv=tc, res=cu(res))
    check(got, emu.gemm(x, w, M=rows, bias=bias, tconv=tc, res=res), f"tconv sharded f_loc={f_loc} frame0={frame0} chunk={chunk}")


@pytest.mark.parametrize("C,f_loc,f_tot,frame0,npix,nb", [(320, 6, 24, 6, 4096, 2), (640, 3, 24, 21, 1024, 4), (1280, 12, 24, 0, 256, 2), (320, 4, 16, 4, 100, 3), (1280, 6, 24, 12, 64, 2),
                                                   # shapes whose FULL launch takes the 192-row 8-phase tiles / 128 x 128 tiles / 128 x 64 tiles while a piece by itself
                                                   # would count fewer tiles (every tile-count heuristic of me_gemm now selects by sel_rows, round-4 advisor finding)
                                                   (640, 6, 24, 6, 4096, 2), (640, 6, 24, 6, 256, 4), (640, 3, 24, 3, 256, 4)])
def test_gemm_row_range_pieces_of_a_sharded_tconv_equal_the_one_launch_form(ops, C, f_loc, f_tot, frame0, npix, nb):
    """me_gemm_args.m_off: the interior launches (frames 1 .. f_loc - 2 of every batch entry, issued while the halo exchange travels) and the boundary
    launches (first frame | (last, first) pairs of neighbouring batch entries | last frame) of a frame-sharded TemporalConv must reproduce the one-launch
    result BIT FOR BIT -- with the time-embedding row vector and residual terms of temp_conv1 (absolute row indices) -- at the level-0 ... level-3
    geometries of 4- and 8-way frame sharding."""
    rows, hb = nb * f_loc * npix, nb * npix
    x = rnd(rows + 2 * hb, C, seed=1).cuda()
    w = (rnd(C, 3, C, seed=2, scale=(3 * C) ** -0.5)).cuda()
    bias, res, rv = rnd(C, seed=3).cuda(), rnd(rows, C, seed=4).cuda(), rnd(nb, C, seed=5).cuda()
    hp = rows if frame0 > 0 else -1
    hn = rows + hb if frame0 + f_loc < f_tot else -1
    epi = dict(bias=bias, res=res, rowvec=rv, rows_per_vec=f_loc * npix)
    want = ops.gemm(x, w, M=rows, tconv=(f_loc, npix, f_tot, frame0, f_tot, hp, hn), **epi)
    if ops.gemm_splits_k(rows, C, C, 3):   # the one-launch form MAY be split along K (another summation order): graph._tconv keeps it whole then
        return
    out = torch.full((rows, C), float("nan"), dtype=torch.float16, device="cuda")
    for bi in range(nb):
        ops.gemm(x, w, M=rows, tconv=(f_loc, npix, f_tot, frame0, f_tot, -1, -1), out=out, row_range=((bi * f_loc + 1) * npix, (bi * f_loc + f_loc - 1) * npix), **epi)
    bounds = [(0, npix)] + [((bi * f_loc - 1) * npix, (bi * f_loc + 1) * npix) for bi in range(1, nb)] + [(rows - npix, rows)]
    for lo, hi in bounds:
        ops.gemm(x, w, M=rows, tconv=(f_loc, npix, f_tot, frame0, f_tot, hp, hn), out=out, row_range=(lo, hi), **epi)
    assert torch.equal(out, want)
    with pytest.raises(Exception):
        ops.gemm(x, w, M=rows, tconv=(f_loc, npix, f_tot, frame0, f_tot, hp, hn), out=out, row_range=(rows, rows + 1))


def test_groupnorm_with_cross_rank_statistic_reduction_hook(ops):
    """stats -> reduce hook -> apply with the global count: doubling the statistics and the count must be the identity."""
    C, rows, rpg = 640, 4 * 96, 96
    x = (rnd(rows, C, seed=1) * 2 + 0.7).half()
    gm, bt = (1 + 0.1 * rnd(C, seed=2)).half(), (0.1 * rnd(C, seed=3)).half()
    got = ops.groupnorm(cu(x), cu(gm), cu(bt), rows_per_group=rpg, eps=1e-5, silu=True, reduce=lambda st: st.mul_(2.0), rows_per_group_total=2 * rpg)
    check(got, emu.groupnorm(x, gm, bt, rows_per_group=rpg, eps=1e-5, silu=True), "groupnorm split stats/apply")


# ------------------------------------------------------------------ backward primitives (motioneditor_amd/autodiff.py)
# Contract: entries that take `dst` / `dq, dk, dv` ACCUMULATE into those fp32 views -- every test starts them from a non-zero tensor.
def _acc0(shape, seed=99, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("mode", ["dense", "dense_n4", "conv", "conv_s2", "conv_ups", "tconv", "geglu_w"])
def test_gemm_dx_matches_the_vjp_of_the_forward_emulation(ops, mode):
    """Input gradient of me_gemm in every gather mode the UNet uses, computed by me_gemm itself on transposed / tap-reversed weights
    (the stride-2 case through the zero-stuffed gather mode ups = 2, the upsampled case with me_grad_acc's 2 x 2 pooling) vs torch autograd
    through the forward emulation; accumulated into a strided view of a wider gradient buffer."""
    g = torch.Generator().manual_seed(5)
    conv = tconv = None
    if mode in ("dense", "geglu_w"):
        M, N, K, taps, xr = 384, 320 if mode == "dense" else 640, 192, 1, 384
    elif mode == "dense_n4":
        M, N, K, taps, xr = 256, 4, 320, 1, 256
    elif mode == "conv":
        M, N, K, taps, xr, conv = 2 * 16 * 16, 128, 64, 9, 2 * 16 * 16, (16, 16, 16, 16, 1, 0)
    elif mode == "conv_s2":
        M, N, K, taps, xr, conv = 2 * 8 * 8, 128, 64, 9, 2 * 16 * 16, (16, 16, 8, 8, 2, 0)
    elif mode == "conv_ups":
        M, N, K, taps, xr, conv = 2 * 16 * 16, 128, 64, 9, 2 * 8 * 8, (8, 8, 16, 16, 1, 1)
    else:
        M, N, K, taps, xr, tconv = 2 * 8 * 12, 128, 64, 3, 2 * 8 * 12, (8, 12, 8)
    w = (torch.randn(N, taps, K, generator=g) * (taps * K) ** -0.5).half()
    dy = torch.randn(M, N, generator=g)
    base = _acc0((xr, K + 8))
    want = base.clone()
    emu.gemm_dx(dy.half().float(), w, dst=want[:, 4:4 + K], M=M, conv=conv, tconv=tconv)
    got = cu(base)
    ops.gemm_dx(cu(dy), cu(w), dst=got[:, 4:4 + K], M=M, conv=conv, tconv=tconv)
    check(got, want, f"gemm_dx {mode}")
    assert torch.equal(got[:, :4].cpu(), base[:, :4]) and torch.equal(got[:, 4 + K:].cpu(), base[:, 4 + K:])   # columns outside the view untouched


@pytest.mark.parametrize("geom", [(16, 16, 1), (8, 12, 2), (20, 8, 3)])
def test_gemm_conv_zero_stuffed_gather(ops, geom):
    """Gather mode ups = 2 on its own: a 3x3 stride-1 convolution over the zero-stuffed 2x upsample of the input."""
    H, W, n_img = geom
    K, N = 64, 128
    x, w = rnd(n_img * H * W, K, seed=1), rnd(N, 9, K, seed=2, scale=(9 * K) ** -0.5)
    conv = (H, W, 2 * H, 2 * W, 1, 2)
    check(ops.gemm(cu(x), cu(w), M=n_img * 4 * H * W, conv=conv), emu.gemm(x, w, M=n_img * 4 * H * W, conv=conv), f"conv zero-stuffed {geom}")


@pytest.mark.parametrize("M,N,K", [(3 * 200, 320, 64), (2 * 256 * 260, 320, 64), (4 * 96, 128, 128)])
def test_gemm_shared_residual_rows(ops, M, N, K):
    """res_rows / res2_rows: residuals that hold one batch entry's rows and are read by every entry (row m reads row m % rows), on the
    specialised and the generic epilogues, 256x320 and 128-wide tiles."""
    nb = 3 if M == 600 else (2 if M > 1000 else 4)
    rr = M // nb
    x, w, bias = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    res, res2 = rnd(rr, N, seed=4), rnd(rr, N, seed=5)
    full = rnd(M, N, seed=6)
    for kw_gpu, kw_emu, name in ((dict(res=cu(res), res_rows=rr), dict(res=res, res_rows=rr), "res shared"),
                                 (dict(res=cu(full), res2=cu(res2), res2_rows=rr), dict(res=full, res2=res2, res2_rows=rr), "res full + res2 shared"),
                                 (dict(res=cu(res), res_rows=rr, act=1), dict(res=res, res_rows=rr, act=1), "generic epilogue")):
        check(ops.gemm(cu(x), cu(w), bias=cu(bias), **kw_gpu), emu.gemm(x, w, bias=bias, **kw_emu), f"gemm {name} M={M}")


def test_geglu_bwd(ops):
    g = torch.Generator().manual_seed(6)
    M, N = 300, 640
    pre = (torch.randn(M, N, generator=g) * 1.5).half()
    dy = torch.randn(M, N // 2, generator=g)
    check(ops.geglu_bwd(cu(pre), cu(dy)), emu.geglu_bwd(pre, dy), "geglu_bwd")
    big = torch.zeros(M, N // 2 + 8)                       # a strided gradient view, as the tape hands them over
    big[:, :N // 2] = dy
    check(ops.geglu_bwd(cu(pre), cu(big)[:, :N // 2]), emu.geglu_bwd(pre, dy), "geglu_bwd strided")


@pytest.mark.parametrize("C,rows", [(320, 257), (640, 64), (1280, 33)])
def test_layernorm_bwd(ops, C, rows):
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).half()
    gm = (1 + 0.2 * torch.randn(C, generator=g)).half()
    dy = torch.randn(rows, C, generator=g)
    check(ops.layernorm_bwd(cu(x), cu(gm), cu(dy), eps=1e-5), emu.layernorm_bwd(x, gm, dy, eps=1e-5), f"layernorm_bwd C={C}")


@pytest.mark.parametrize("C,rpg,nsg,silu", [(320, 96, 2, True), (640, 50, 3, False), (1920, 24, 1, True), (320, 6144, 2, True), (2560, 700, 1, False)])
def test_groupnorm_bwd(ops, C, rpg, nsg, silu):
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(nsg * rpg, C, generator=g) * 1.5 + 0.5).half()
    gm = (1 + 0.2 * torch.randn(C, generator=g)).half()
    bt = (0.2 * torch.randn(C, generator=g)).half()
    dy = torch.randn(nsg * rpg, C, generator=g)
    want = emu.groupnorm_bwd(x, gm, bt, dy, rows_per_group=rpg, eps=1e-5, silu=silu)
    got = ops.groupnorm_bwd(cu(x), cu(gm), cu(bt), cu(dy), rows_per_group=rpg, eps=1e-5, silu=silu)
    check(got, want, f"groupnorm_bwd C={C} silu={silu}")
    assert torch.equal(got, ops.groupnorm_bwd(cu(x), cu(gm), cu(bt), cu(dy), rows_per_group=rpg, eps=1e-5, silu=silu))   # fixed-order sums: bitwise reproducible


@pytest.mark.parametrize("F,dh,npix", [(24, 40, 5), (8, 80, 3), (16, 160, 2), (24, 160, 2), (48, 160, 1), (7, 40, 4), (32, 80, 2), (1, 40, 3), (24, 80, 70)])   # F <= 32: the lane-parallel kernel; 48: the first version
def test_temporal_attention_bwd(ops, F, dh, npix):
    g = torch.Generator().manual_seed(9)
    B, C = 2, 8 * dh
    qkv = (torch.randn(B * F * npix, 3 * C, generator=g) * 0.7).half()
    dout = torch.randn(B * F * npix, C, generator=g)
    args = dict(heads=8, dh=dh, batch=B, frames=F, npix=npix)
    want = emu.temporal_attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], None, dout, **args)
    got = ops.temporal_attention_bwd(cu(qkv)[:, :C], cu(qkv)[:, C:2 * C], cu(qkv)[:, 2 * C:], None, cu(dout), **args)
    for a, b, n in zip(got, want, "qkv"):
        check(a, b, f"tattn_bwd d{n} F={F} dh={dh}")


def _attn_case(kind, f, device="cpu"):
    from motioneditor_amd import segments
    if kind == "pc":
        si, sm = segments.prev_cur(1, f, device)
        return si, sm, f, f
    if kind == "self":
        si, sm = segments.self_items(2, device)
        return si, sm, 2, 2
    if kind == "fp":      # adapter [first | prev] inside chunks of 8 frames: the first frame of a chunk is listed by up to nine query items
        si, sm = segments.first_prev_chunked(1, f, 8, device)
        return si, sm, f, f
    si, sm = segments.cross_text(1, f, device)
    return si, sm, f, 1


@pytest.mark.parametrize("kind,dh,nq,nk,f", [("pc", 40, 128, 128, 3), ("pc", 80, 96, 96, 3), ("cross", 40, 64, 77, 3), ("self", 160, 64, 64, 2), ("fp", 40, 100, 100, 10),
                                             ("pc", 40, 1024, 1024, 3), ("pc", 80, 1024, 1024, 2), ("self", 160, 256, 256, 2), ("cross", 80, 1000, 77, 4),
                                             ("pc", 160, 40, 40, 3)])
def test_attention_bwd_plain_segments(ops, kind, dh, nq, nk, f):
    """me_attn_bwd (flash-style, two kernels) for the tables the un-edited UNet and the adapter use -- [prev | cur] (a kv item is named by
    two query items: its gradients add up), self, text (one kv item named by every frame), [first | prev] -- vs the vjp of the forward
    emulation.  P is rebuilt from the log-sum-exp the HIP forward stashes; q | k | v and their gradients are column slices of fused
    [rows, 3C] tensors (strided views), every output accumulates onto what the buffer held; nq / nk with tails (77, 100, 40 keys)."""
    g = torch.Generator().manual_seed(10)
    C = 8 * dh
    si, sm, n_items, n_kv = _attn_case(kind, f)
    fused = kind in ("pc", "self", "fp") and nq == nk
    q = (torch.randn(n_items * nq, C, generator=g) * 0.7).half()
    k = (torch.randn(n_kv * nk, C, generator=g) * 0.7).half()
    v = torch.randn(n_kv * nk, C, generator=g).half()
    dout = torch.randn(n_items * nq, C, generator=g)
    args = dict(heads=8, dh=dh, n_items=n_items, nq=nq, nk=nk)
    if fused:      # one [rows, 3C] allocation as the q|k|v GEMM writes it, gradients likewise
        qkv = cu(torch.cat([q, k, v], dim=1))
        qc, kc, vc = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        g0 = _acc0((n_items * nq, 3 * C))
        gg = cu(g0)
        dq, dk, dv = gg[:, :C], gg[:, C:2 * C], gg[:, 2 * C:]
        w0 = (g0[:, :C].clone(), g0[:, C:2 * C].clone(), g0[:, 2 * C:].clone())
    else:
        qc, kc, vc = cu(q), cu(k), cu(v)
        w0 = (_acc0((n_items * nq, C), 1), _acc0((n_kv * nk, C), 2), _acc0((n_kv * nk, C), 3))
        dq, dk, dv = cu(w0[0]), cu(w0[1]), cu(w0[2])
    sic, smc = _attn_case(kind, f, "cuda")[:2]
    lse = torch.empty((n_items * nq, 8), dtype=torch.float32, device="cuda")
    out = ops.attention(qc, kc, vc, seg_item=sic, seg_mode=smc, lse=lse, **args)
    lse_want = torch.empty((n_items * nq, 8))
    emu.attention(q, k, v, seg_item=si, seg_mode=sm, lse=lse_want, **args)
    assert float((lse.cpu() - lse_want).abs().max()) < 2e-2, "log-sum-exp stashed by the forward"
    want = [w.clone() for w in w0]
    emu.attention_bwd(q, k, v, None, dout, dq=want[0], dk=want[1], dv=want[2], seg_item=si, seg_mode=sm, **args)
    ops.attention_bwd(qc, kc, vc, out, cu(dout), dq=dq, dk=dk, dv=dv, lse=lse, seg_item=sic, seg_mode=smc, **args)
    for a, b, z, n in zip((dq, dk, dv), want, w0, "qkv"):
        check(a - cu(z), b - z, f"attention_bwd d{n} {kind} dh={dh} nq={nq} nk={nk}", rel=4e-3, mx=6e-2)


def test_attention_shared_query_items(ops):
    """q_items: the adapter's pose queries computed once and read by every batch entry (item i reads query item i % q_items)."""
    from motioneditor_amd import segments
    g = torch.Generator().manual_seed(11)
    dh, nq, f, nb = 40, 256, 3, 2
    C = 8 * dh
    q = (torch.randn(f * nq, C, generator=g) * 0.7).half()
    kv = (torch.randn(nb * f * nq, 2 * C, generator=g) * 0.7).half()
    si, sm = segments.self_items(nb * f, "cpu")
    sic, smc = segments.self_items(nb * f, "cuda")
    args = dict(heads=8, dh=dh, n_items=nb * f, nq=nq, nk=nq, q_items=f)
    got = ops.attention(cu(q), cu(kv)[:, :C], cu(kv)[:, C:], seg_item=sic, seg_mode=smc, **args)
    check(got, emu.attention(q, kv[:, :C], kv[:, C:], seg_item=si, seg_mode=sm, **args), "attention q_items")
    check(got, emu.attention(torch.cat([q] * nb), kv[:, :C], kv[:, C:], seg_item=si, seg_mode=sm, heads=8, dh=dh, n_items=nb * f, nq=nq, nk=nq), "attention q_items == replicated q")


@pytest.mark.parametrize("mode", ["dense", "tconv", "geglu_n", "dense_f16", "big"])
def test_gemm_dw_and_bias_gradients(ops, mode):
    """me_gemm_dw / me_colsum: fp32-accumulating, fp32-output weight and bias gradients of the adapter's Linear and TemporalConv layers (the
    token axis is the MFMA contraction; fixed-order split partials) vs the vjp of the forward emulation; accumulated onto existing content."""
    g = torch.Generator().manual_seed(12)
    tconv = None
    if mode in ("dense", "dense_f16"):
        M, N, K, taps = 300, 320, 192, 1
    elif mode == "geglu_n":
        M, N, K, taps = 200, 2560, 320, 1
    elif mode == "big":
        M, N, K, taps = 24 * 1024, 640, 640, 1
    else:
        M, N, K, taps, tconv = 2 * 16 * 12, 128, 64, 3, (16, 12, 8)
    x = (torch.randn(M, K, generator=g)).half()
    dy = torch.randn(M, N, generator=g)
    dyq = dy.half().float()
    base = _acc0((N, taps, K))
    want = emu.gemm_dw(dyq, x, dst=base.clone(), taps=taps, K=K, M=M, tconv=tconv)
    got = ops.gemm_dw(cu(dy.half()) if mode == "dense_f16" else cu(dy), cu(x), dst=cu(base), taps=taps, K=K, M=M, tconv=tconv)
    check(got - cu(base), want - base, f"gemm_dw {mode}", rel=4e-3, mx=6e-2)
    b0 = _acc0((N,))
    check(ops.colsum_grad(cu(dy), dst=cu(b0)) - cu(b0), emu.colsum_grad(dy, dst=b0.clone()) - b0, f"colsum_grad {mode}", rel=1e-4, mx=1e-3)


def test_relu_bwd_and_layernorm_param_gradients(ops):
    g = torch.Generator().manual_seed(13)
    rows, C = 1000, 320
    out = torch.randn(rows, C, generator=g).half()
    dy = torch.randn(rows, C, generator=g)
    assert torch.equal(ops.relu_bwd(cu(dy), cu(out)).cpu(), emu.relu_bwd(dy, out))
    for C in (320, 1280):
        x = (torch.randn(rows, C, generator=g) * 2 + 0.3).half()
        dy = torch.randn(rows, C, generator=g)
        g0, b0 = _acc0((C,), 1), _acc0((C,), 2)
        dg, db = cu(g0), cu(b0)
        ops.layernorm_bwd_params(cu(x), cu(dy), dgamma=dg, dbeta=db, eps=1e-5)
        wg, wb = g0.clone(), b0.clone()
        emu.layernorm_bwd_params(x, dy, dgamma=wg, dbeta=wb, eps=1e-5)
        check(dg - cu(g0), wg - g0, f"layernorm d gamma C={C}", rel=1e-3, mx=2e-2)
        check(db - cu(b0), wb - b0, f"layernorm d beta C={C}", rel=1e-4, mx=1e-3)


def test_grad_acc_plain_strided_fp16_and_pooled(ops):
    g = torch.Generator().manual_seed(14)
    rows, cols = 300, 64
    dst0 = _acc0((rows, cols + 16))
    src = torch.randn(rows + 5, cols + 8, generator=g)
    for s_, name in ((src, "fp32"), (src.half(), "fp16")):
        got = cu(dst0)
        ops.grad_acc(got[:, 8:8 + cols], cu(s_)[:, :cols], 0.5)
        want = dst0.clone()
        emu.grad_acc(want[:, 8:8 + cols], s_[:, :cols], 0.5)
        check(got, want, f"grad_acc {name}", rel=1e-6 if name == "fp32" else 1e-3, mx=1e-2)
    n_img, H, W = 3, 6, 10
    big = torch.randn(n_img * 4 * H * W, cols, generator=g).half()
    d0 = _acc0((n_img * H * W, cols))
    got = cu(d0)
    ops.grad_acc(got, cu(big), 1.0, pool=(H, W))
    check(got, emu.grad_acc(d0.clone(), big, 1.0, pool=(H, W)), "grad_acc 2x2 pooling", rel=1e-3, mx=1e-2)
    flat, add = _acc0((77 * 768,)), _acc0((77 * 768,), 5)
    got = cu(flat)
    ops.grad_acc(got, cu(add), 2.0)
    check(got, flat + 2.0 * add, "grad_acc 1-D", rel=1e-6, mx=1e-5)


def test_sumsq_absmax_is_deterministic_and_exact_enough(ops):
    g = torch.Generator().manual_seed(15)
    for n in (5, 1000, 3_000_001):
        x = torch.randn(n, generator=g) * 3
        x[n // 2] = -40.0
        a, b = ops.sumsq_absmax(cu(x)).cpu(), ops.sumsq_absmax(cu(x)).cpu()
        assert torch.equal(a, b)
        assert abs(float(a[0]) / float((x.double() ** 2).sum()) - 1) < 1e-5 and float(a[1]) == 40.0


@pytest.mark.parametrize("wd,clip", [(1e-2, True), (0.0, False)])
def test_adamw_kernel_matches_torch_optim(ops, wd, clip):
    """me_adamw over three steps vs torch.optim.AdamW (weight decay, bias correction) with torch.nn.utils.clip_grad_norm_ in front: the clip
    factor comes from the device-side sum of squares, the loss scale is divided out by grad_scale."""
    g = torch.Generator().manual_seed(16)
    n, ls = 10_000, 256.0
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    p, m, v = cu(p0.clone()), cu(torch.zeros(n)), cu(torch.zeros(n))
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * (0.05 if step == 2 else 3.0)     # step 2 stays below the clipping norm
        ref.grad = gr.clone()
        if clip:
            torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gs = cu(gr * ls)
        ops.adamw(p, m, v, gs, lr=1e-3, weight_decay=wd, step=step, gnorm_sq=ops.sumsq_absmax(gs) if clip else None, max_grad_norm=1.0, grad_scale=1.0 / ls)
    check(p, ref.detach(), f"adamw wd={wd} clip={clip}", rel=1e-5, mx=1e-4)


def test_cast_kernels_and_mse_seed(ops):
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1003, generator=g) * 10
    out = torch.empty(1003, dtype=torch.float16, device="cuda")
    assert torch.equal(ops.cast_f16(out, cu(x)).cpu(), x.half())
    wide = torch.randn(50, 40, generator=g)
    got = ops._f16(cu(wide)[:, 4:24], 24)                      # a strided fp32 view, 20 columns padded to 24
    assert torch.equal(got[:, :20].cpu(), wide[:, 4:24].half()) and float(got[:, 20:].abs().max()) == 0.0
    nb, C, f, h, w = 1, 4, 3, 4, 5
    eu, ec = (torch.randn(nb * f * h * w, 8, generator=g)).half(), (torch.randn(nb * f * h * w, 8, generator=g)).half()
    xl, tg = torch.randn(nb, C, f, h, w, generator=g), torch.randn(nb, C, f, h, w, generator=g)
    for kw in (dict(guidance=7.5, ca=1.01, cb=-0.2, coef=0.3), dict(coef=2.0 / tg.numel())):
        full = "guidance" in kw
        d1, r1 = ops.mse_seed(cu(eu), cu(tg), eps_c=cu(ec) if full else None, x=cu(xl) if full else None, **kw)
        d0, r0 = emu.mse_seed(eu, tg, eps_c=ec if full else None, x=xl if full else None, **kw)
        check(d1, d0, "mse_seed diff", rel=1e-6, mx=1e-5)
        check(r1, r0, "mse_seed seed rows", rel=1e-6, mx=1e-5)


@pytest.mark.parametrize("Cin,Cout,n_img,H,W,frames", [(4, 320, 4, 16, 16, 2), (3, 128, 2, 24, 40, 0), (4, 320, 3, 7, 9, 0)])
def test_conv_small_wide_outputs_tile_kernel(ops, Cin, Cout, n_img, H, W, frames):
    """conv_in (4 -> 320, 5-D latents) and the VAE encoder's 3 -> 128 through the 64-pixel tile kernel (row-contiguous stores), incl. a
    pixel count that is not a multiple of 64."""
    g = torch.Generator().manual_seed(18)
    wt = torch.randn(Cout, 9, Cin, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g) * 0.1
    if frames:
        B = n_img // frames
        x = torch.randn(B, Cin, frames, H, W, generator=g)
        kw = dict(n_img=n_img, Cin=Cin, H=H, Wd=W, img_stride=Cin * frames * H * W, ch_stride=frames * H * W, frames=frames, frame_stride=H * W)
    else:
        x = torch.randn(n_img, Cin, H, W, generator=g)
        kw = dict(n_img=n_img, Cin=Cin, H=H, Wd=W, img_stride=Cin * H * W, ch_stride=H * W)
    check(ops.conv_small(cu(x), cu(wt), cu(bias), **kw), emu.conv_small(x, wt, bias, **kw), f"conv_small tile {Cin}->{Cout}")


# ------------------------------------------------------------------ head-major K | V (ABI 6: me_gemm_args.C2, me_attn_args.hsk / hsv)
@pytest.mark.parametrize("M,C,K", [(512, 320, 320), (1000, 320, 320), (131072, 320, 320), (24576, 640, 640), (6144, 1280, 1280), (300, 1280, 1280), (196608, 320, 320)])
def test_gemm_head_major_second_output_is_the_same_numbers_elsewhere(ops, M, C, K):
    """q | k | v projection with K and V leaving as [16, M, dh] panels: bit for bit the column slices of the one-tensor launch -- through the
    8-phase 256 x 320 kernel (wide 16-byte stores + the odd fifth column tile), the 128-row kernels and ragged last row tiles."""
    dh = C // 8
    x, w = rnd(M, K, seed=1).cuda(), rnd(3 * C, 1, K, seed=2, scale=K ** -0.5).cuda()
    ref = ops.gemm(x, w)
    q, kv = ops.gemm(x, w, head_major=(C, dh))
    assert q.shape == (M, C) and kv.shape == (16, M, dh) and kv.is_contiguous()
    assert torch.equal(q, ref[:, :C])
    assert torch.equal(kv.permute(1, 0, 2).reshape(M, 2 * C), ref[:, C:])
    b = rnd(3 * C, seed=3).cuda()
    q2, kv2 = ops.gemm(x, w, bias=b, head_major=(C, dh))
    ref2 = ops.gemm(x, w, bias=b)
    assert torch.equal(q2, ref2[:, :C]) and torch.equal(kv2.permute(1, 0, 2).reshape(M, 2 * C), ref2[:, C:])
    # every column as panels (q | k | v all head-major, c2_col0 = 0): no row tensor comes back
    none, qkv_p = ops.gemm(x, w, bias=b, head_major=(0, dh))
    assert none is None and qkv_p.shape == (24, M, dh) and torch.equal(qkv_p.permute(1, 0, 2).reshape(M, 3 * C), ref2)
    with pytest.raises(Exception):
        ops.gemm(x, w, res=ref, head_major=(C, dh))
    with pytest.raises(Exception):
        ops.gemm(x, w, head_major=(C, dh + 4))


@pytest.mark.parametrize("dh,N,kind", [(40, 4096, "pc"), (40, 4096, "ed_bin"), (40, 4096, "ed_gen"), (80, 1024, "pc"), (80, 1024, "ed_bin"), (160, 256, "pc"), (40, 100, "pc"),
                                       (40, 576, "ed_bin"), (80, 144, "ed_gen")])
def test_attention_head_major_kv_equals_row_major(ops, dh, N, kind):
    """The same launch with K and V as per-head [rows, dh] panels (what the head-major projection writes) must give bitwise the output of the
    row-major launch: only addresses change (K/V tile fill, column sums of V for the binary dual segments), no arithmetic."""
    from motioneditor_amd import segments
    C = 8 * dh
    if kind == "pc":
        B, f = 2, 2
        si, sm = segments.prev_cur(B, f, "cpu")
        n_items, mask = B * f, None
    else:
        B, f = 2, 1
        si, sm = segments.edited_spatial(f, "cpu", binary_mask=(kind == "ed_bin"), B=B)
        g = torch.Generator().manual_seed(11)
        mask = ((torch.rand(8, N, generator=g) > 0.5).half() if kind == "ed_bin" else torch.rand(8, N, generator=g).half()).cuda()
        n_items = B * f
    qkv = rnd(n_items * N, 3 * C, seed=3).cuda()
    args = dict(heads=8, dh=dh, n_items=n_items, nq=N, nk=N, seg_item=cu(si), seg_mode=cu(sm), mask=mask)
    want = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], **args)
    kv = qkv[:, C:].reshape(n_items * N, 16, dh).permute(1, 0, 2).contiguous()
    got = ops.attention(qkv[:, :C], kv[:8], kv[8:], **args)
    assert torch.equal(got, want)
    # ... and with Q as panels too (ABI 8, me_attn_args.hsq) -- the general-dual (mask-reading) kernel included (round 6: the default graph hands it
    # head-major Q whenever the source masks are not binary)
    qp = qkv[:, :C].reshape(n_items * N, 8, dh).permute(1, 0, 2).contiguous()
    assert torch.equal(ops.attention(qp, kv[:8], kv[8:], **args), want)


# ------------------------------------------------------------------ LayerNorm folded into the projection (ABI 9)
def _ln_fold_pack(w, gamma, beta, bias=None):
    """weights.Packed.ln_fold on raw tensors: W' = W diag(gamma) rounded to fp16, colsum over the ROUNDED W', W beta + b."""
    wq = (w.float() * gamma.float()[None, None, :]).half()
    cs = wq.float().sum(dim=(1, 2))
    cv = w.float()[:, 0, :] @ beta.float()
    if bias is not None:
        cv = cv + bias.float()
    return wq, cs, cv


@pytest.mark.parametrize("M,C,N,kind", [(256 * 520, 320, 960, "hm"), (256 * 520, 320, 320, "plain"), (256 * 514 + 77, 320, 2560, "geglu"), (256 * 260, 640, 1920, "hm"),
                                        (192 * 130 + 5, 1280, 3840, "plain"), (6144, 1280, 3840, "plain"), (1000, 1280, 10240, "geglu"), (300, 320, 960, "plain"),
                                        (1536, 640, 640, "plain")])
def test_gemm_layernorm_fold(ops, M, C, N, kind):
    """LN(x) W^T (+ b, GEGLU) computed as rstd (x W'^T - mean colsum) + cvec from partial row sums (me_gemm_args.ln_stats) against LayerNorm-then-GEMM in fp32:
    the 8-phase kernels (row-contiguous F = 0 epilogue with and without head-major panels, the GEGLU row pass), the 192-row tiles, the 128-row kernels; rows
    with a large common offset (mean / std ~ 8); ragged last tiles."""
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, C, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + 4.0 * torch.randn(M, 1, generator=g)).half()
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
    w = rnd(N, 1, C, seed=2, scale=C ** -0.5)
    bias = rnd(N, seed=3, scale=0.1) if kind == "geglu" else None
    wq, cs, cv = _ln_fold_pack(w, gamma, beta, bias)
    n = emu.layernorm(x.float(), gamma, beta)
    want = emu.gemm(n, w.float(), bias=None if bias is None else bias.float(), geglu=kind == "geglu")
    st = ops.ln_stats(cu(x))
    P = C // 320
    assert st.shape == (P, M, 2)
    xf = x.float().reshape(M, P, 320)
    check(st[:, :, 0].t(), xf.sum(-1), "ln_stats sum", rel=1e-5, mx=1e-3)
    check(st[:, :, 1].t(), (xf * xf).sum(-1), "ln_stats sumsq", rel=1e-5, mx=1e-3)
    ln = (st, cu(cs), cu(cv), 1e-5)
    if kind == "hm":
        _, panels = ops.gemm(cu(x), cu(wq), ln=ln, head_major=(0, C // 8))
        got = panels.permute(1, 0, 2).reshape(M, N)
    else:
        got = ops.gemm(cu(x), cu(wq), ln=ln, geglu=kind == "geglu")
    check(got, want, f"ln-folded gemm {M}x{N}x{C} {kind}", mx=(4e-2 if kind == "geglu" else MAX_REL))   # (a * gelu(g): heavy-tailed products, max / mean-abs is a loose yardstick)
    # the emulation of the same call (what the CPU graph tests run) agrees too
    check(emu.gemm(x.float(), wq.float(), ln=(emu.ln_stats(x), cs, cv, 1e-5), geglu=kind == "geglu"), want, "emulated fold", rel=1e-3)
    with pytest.raises(Exception):
        ops.gemm(cu(x), cu(wq), ln=ln, bias=cu(rnd(N, seed=9)))


@pytest.mark.parametrize("M,N,K,terms", [(256 * 520, 320, 320, "res"), (256 * 520 + 31, 320, 1280, "bias+res"), (256 * 260, 640, 640, "bias"), (192 * 130, 1280, 1280, "bias+res"),
                                         (256 * 520, 320, 64, "res+res2"), (6144, 1280, 1280, "bias+res"), (300, 320, 320, "res")])
def test_gemm_row_sums_of_the_output_for_the_next_layernorm_fold(ops, M, N, K, terms, monkeypatch):
    """ops.gemm(ln_out=True): the partial row sums (sum y, sum y^2) per 320-column part of the rows a projection WRITES -- from the row-contiguous epilogue of the
    8-phase kernels (256- and 192-row tiles, every term set), from me_ln_stats behind the launch otherwise -- equal the sums of the fp16 output it stored; the
    output itself is bitwise what the launch writes without ln_out; and the two producers of the statistics agree to fp32 rounding."""
    x, w = rnd(M, K, seed=1), rnd(N, 1, K, seed=2, scale=K ** -0.5)
    bias = rnd(N, seed=3) if "bias" in terms else None
    res = rnd(M, N, seed=5) if "res" in terms else None
    res2 = rnd(M, N, seed=6) if "res2" in terms else None
    kw = dict(bias=cu(bias), res=cu(res), res2=cu(res2))
    plain = ops.gemm(cu(x), cu(w), **kw)
    y, st = ops.gemm(cu(x), cu(w), ln_out=True, **kw)
    assert torch.equal(y, plain)
    P = N // 320
    assert st.shape == (P, M, 2)
    yf = y.float().reshape(M, P, 320)
    check(st[:, :, 0].t(), yf.sum(-1), "row sums", rel=1e-5, mx=1e-3)
    check(st[:, :, 1].t(), (yf * yf).sum(-1), "row sums of squares", rel=1e-5, mx=1e-3)
    monkeypatch.setenv("ME_GEMM_ROWEPI", "0")     # the direct epilogue has no row sums: me_gemm appends the read-only pass
    y2, st2 = ops.gemm(cu(x), cu(w), ln_out=True, **kw)
    assert torch.equal(y2, plain)
    check(st2, st, "fused vs appended statistics", rel=2e-5, mx=1e-3)
    assert torch.equal(ops.ln_stats(y), st2)
