"""Kernel micro-benchmarks on the GPU box: python tools/kbench.py [gemm] [attn] [misc]
Times representative config-3 (24f x 512^2, B=4) launches with HIP events (median of N reps)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from motioneditor_amd import ops, segments  # noqa: E402

dev = "cuda"
REPS = 7


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).half()


def bench_gemm():
    B, f = 4, 24
    L = [(64, 320), (32, 640), (16, 1280), (8, 1280)]
    rows = []
    for li, (hw, C) in enumerate(L):
        M = B * f * hw * hw
        rows += [(f"L{li} qkv", M, 3 * C, C, None, None, False), (f"L{li} out", M, C, C, None, None, False),
                 (f"L{li} ff1 geglu", M, 8 * C, C, None, None, True), (f"L{li} ff2", M, C, 4 * C, None, None, False),
                 (f"L{li} conv3x3", M, C, C, (hw, hw, hw, hw, 1, 0), None, False), (f"L{li} tconv", M, C, C, None, (f, hw * hw, f), False)]
    rows += [("L0 conv 960->320", B * f * 4096, 320, 960, (64, 64, 64, 64, 1, 0), None, False),
             ("L1 conv 1920->640", B * f * 1024, 640, 1920, (32, 32, 32, 32, 1, 0), None, False),
             ("L0 conv 640->320", B * f * 4096, 320, 640, (64, 64, 64, 64, 1, 0), None, False),
             ("L1 conv 1280->640", B * f * 1024, 640, 1280, (32, 32, 32, 32, 1, 0), None, False),
             ("L2 conv 2560->1280", B * f * 256, 1280, 2560, (16, 16, 16, 16, 1, 0), None, False),
             ("L1->L0 ups conv 640", B * f * 4096, 640, 640, (32, 32, 64, 64, 1, 1), None, False),
             ("L2->L1 ups conv 1280", B * f * 1024, 1280, 1280, (16, 16, 32, 32, 1, 1), None, False),
             ("cn L0 conv3x3", 2 * f * 4096, 320, 320, (64, 64, 64, 64, 1, 0), None, False),
             ("L0 conv_out 320->4", B * f * 4096, 4, 320, (64, 64, 64, 64, 1, 0), None, False),
             ("cond 16->16 @512", 48 * 512 * 512, 16, 16, (512, 512, 512, 512, 1, 0), None, False),
             ("cond 96->256 s2", 48 * 64 * 64, 256, 96, (128, 128, 64, 64, 2, 0), None, False)]
    print(f"{'gemm':24s} {'M':>8s} {'N':>6s} {'K':>6s} {'ms':>8s} {'TF/s':>7s}")
    tot = 0
    for name, M, N, K, conv, tconv, geglu in rows:
        taps = 9 if conv else (3 if tconv else 1)
        rows_in = M if not conv else (M // (conv[2] * conv[3])) * conv[0] * conv[1]
        x, w = rnd(rows_in, K), rnd(N, taps, K)
        ms = timeit(lambda: ops.gemm(x, w, M=M, conv=conv, tconv=tconv, geglu=geglu))
        tot += ms
        print(f"{name:24s} {M:8d} {N:6d} {K*taps:6d} {ms:8.3f} {2.0*M*N*K*taps/ms/1e9:7.1f}")
        del x, w
    print("sum ms", round(tot, 2))


def bench_lnfold():
    """LayerNorm folded into the projection (round 6, ABI 9) against LayerNorm + projection, and the projection that produces the row sums against the plain one:
    same process, same data.  `fold` = me_gemm with ln_stats given (statistics from the producer: no extra launch); `ln+gemm` = me_layernorm + me_gemm."""
    B, f = 4, 24
    print(f"{'LN fold':22s} {'M':>8s} {'N':>6s} {'K':>6s} {'ln ms':>7s} {'gemm ms':>8s} {'fold ms':>8s} {'x':>6s} {'fold TF/s':>9s}")
    for li, (hw, C) in enumerate([(64, 320), (32, 640), (16, 1280)]):
        M = B * f * hw * hw
        x = rnd(M, C)
        gamma, beta = (1 + 0.1 * torch.randn(C, device=dev)).half(), (0.1 * torch.randn(C, device=dev)).half()
        st = ops.ln_stats(x)
        for name, N, geglu, hm in ((f"L{li} qkv (panels)", 3 * C, False, True), (f"L{li} to_q", C, False, False), (f"L{li} ff1 geglu", 8 * C, True, False)):
            w = rnd(N, 1, C)
            cs, cv = torch.randn(N, device=dev), torch.randn(N, device=dev)
            kw = dict(head_major=(0, C // 8)) if hm else {}
            t_ln = timeit(lambda: ops.layernorm(x, gamma, beta))
            t_g = timeit(lambda: ops.gemm(x, w, geglu=geglu, **kw))
            t_f = timeit(lambda: ops.gemm(x, w, geglu=geglu, ln=(st, cs, cv, 1e-5), **kw))
            print(f"{name:22s} {M:8d} {N:6d} {C:6d} {t_ln:7.3f} {t_g:8.3f} {t_f:8.3f} {(t_ln + t_g) / t_f:6.2f} {2.0 * M * N * C / t_f / 1e9:9.1f}")
            del w
        w, res = rnd(C, 1, C), rnd(M, C)
        t_p = timeit(lambda: ops.gemm(x, w, res=res))
        t_s = timeit(lambda: ops.gemm(x, w, res=res, ln_out=True))
        t_k = timeit(lambda: ops.ln_stats(x))
        print(f"L{li} out + res: plain {t_p:.3f} ms, with row sums {t_s:.3f} ms (+{(t_s / t_p - 1) * 100:.1f} %); me_ln_stats alone {t_k:.3f} ms")
        del x, w, res


def bench_gemm_8p():
    """8-phase ping-pong kernel vs the one-barrier-per-slab kernel on the shapes that take the 256-row tile (ME_GEMM_8P flips per call):
    times both, and checks the 8-phase result BITWISE against the other kernel's (same MFMA order per accumulator, same epilogue) on
    every one of several runs (a staging race shows up as a tile that differs on some run)."""
    import os
    B, f = 4, 24
    cases = []
    for li, (hw, C) in enumerate([(64, 320), (32, 640), (16, 1280)]):
        M = B * f * hw * hw
        cases += [(f"L{li} qkv", M, 3 * C, C, None, None, False, ""), (f"L{li} out +b+res", M, C, C, None, None, False, "br"),
                  (f"L{li} ff1 geglu", M, 8 * C, C, None, None, True, "b"), (f"L{li} ff2 +b+res", M, C, 4 * C, None, None, False, "br")]
        if li < 2:
            cases += [(f"L{li} tconv +b+rv+res", M, C, C, None, (f, hw * hw, f), False, "bvr")]
    cases += [("L2 conv3x3", B * f * 256, 1280, 1280, (16, 16, 16, 16, 1, 0), None, False, "b"), ("L2 conv 2560->1280", B * f * 256, 1280, 2560, (16, 16, 16, 16, 1, 0), None, False, "b"),
              ("L2 tconv +b+rv+res", B * f * 256, 1280, 1280, None, (f, 256, f), False, "bvr"), ("cn L1 out +b+res", 2 * f * 1024 * 2, 320, 320, None, None, False, "br"),
              ("L1->L0 ups conv 640", B * f * 4096, 640, 640, (32, 32, 64, 64, 1, 1), None, False, "b"),
              ("L2->L1 ups conv 1280", B * f * 1024, 1280, 1280, (16, 16, 32, 32, 1, 1), None, False, "b"),
              ("L0->L1 s2 conv 320", B * f * 1024, 320, 320, (64, 64, 32, 32, 2, 0), None, False, "b"),
              ("L2 conv 2560->1280 @32", B * f * 1024, 1280, 2560, (32, 32, 32, 32, 1, 0), None, False, "b"),
              ("big 8192 x 8320 K4096", 8192, 8320, 4096, None, None, False, ""),
              ("big 8192 x 8192 K4096 geglu", 8192, 8192, 4096, None, None, True, "b"),
              ("tail M=100000 N=960 K=320", 100000, 960, 320, None, None, False, "br")]
    print(f"{'gemm':30s} {'M':>8s} {'N':>6s} {'K':>6s} {'old ms':>8s} {'TF/s':>7s} {'8p ms':>8s} {'TF/s':>7s} {'x':>5s}  bitwise", flush=True)
    for name, M, N, K, conv, tconv, geglu, terms in cases:
        taps = 9 if conv else (3 if tconv else 1)
        rows_in = M if not conv else (M // (conv[2] * conv[3])) * conv[0] * conv[1]
        x, w = rnd(rows_in, K), rnd(N, taps, K) * (0.05 if K * taps > 2000 else 0.2)
        kw = dict(M=M, conv=conv, tconv=tconv, geglu=geglu)
        if "b" in terms:
            kw["bias"] = rnd(N)
        if "v" in terms:
            kw["rowvec"], kw["rows_per_vec"] = rnd(B, N), M // B
        if "r" in terms:
            kw["res"] = rnd(M, N)
        os.environ["ME_GEMM_8P"] = "0"
        ref = ops.gemm(x, w, **kw)
        k_old = ops._last_kernel()
        t_old = timeit(lambda: ops.gemm(x, w, **kw))
        os.environ["ME_GEMM_8P"] = "1"
        ok = True
        for _ in range(6):
            y = ops.gemm(x, w, **kw)
            ok = ok and torch.equal(y, ref)
        k_new = ops._last_kernel()
        t_new = timeit(lambda: ops.gemm(x, w, **kw))
        fl = 2.0 * M * N * K * taps
        print(f"{name:30s} {M:8d} {N:6d} {K*taps:6d} {t_old:8.3f} {fl/t_old/1e9:7.1f} {t_new:8.3f} {fl/t_new/1e9:7.1f} {t_old/t_new:5.2f}  {'equal' if ok else 'DIFFERENT'}  {k_old} -> {k_new}", flush=True)
        del x, w, ref, y, kw
    os.environ.pop("ME_GEMM_8P", None)


def bench_gemm_rowepi():
    """Row-contiguous epilogue (epilogue_rowpass, round 5) vs the direct epilogue on the launches that take the 8-phase 256/192 x 320 tiles
    (ME_GEMM_ROWEPI flips per call): same process, same data, bitwise check."""
    import os
    B, f = 4, 24
    cases = []
    for li, (hw, C) in enumerate([(64, 320), (32, 640), (16, 1280)]):
        M = B * f * hw * hw
        cases += [(f"L{li} qkv", M, 3 * C, C, None, None, "", None), (f"L{li} qkv head-major", M, 3 * C, C, None, None, "", (C, C // 8)), (f"L{li} proj +b", M, C, C, None, None, "b", None),
                  (f"L{li} out +b+res", M, C, C, None, None, "br", None), (f"L{li} ff2 +b+res", M, C, 4 * C, None, None, "br", None),
                  (f"L{li} tconv +b+rv+res", M, C, C, None, (f, hw * hw, f), "bvr", None), (f"L{li} tconv +b+res+res2", M, C, C, None, (f, hw * hw, f), "brs", None)]
    cases += [("L0 ff1 geglu +b", B * f * 4096, 2560, 320, None, None, "bg", None), ("L1 ff1 geglu +b", B * f * 1024, 5120, 640, None, None, "bg", None),
              ("L2 ff1 geglu +b", B * f * 256, 10240, 1280, None, None, "bg", None)]
    cases += [("L1->L0 ups conv 640 +b", B * f * 4096, 640, 640, (32, 32, 64, 64, 1, 1), None, "b", None), ("L0->L1 s2 conv 320 +b", B * f * 1024, 320, 320, (64, 64, 32, 32, 2, 0), None, "b", None),
              ("L2 conv 2560->1280 +b+rv", B * f * 256, 1280, 2560, (16, 16, 16, 16, 1, 0), None, "bv", None), ("cn L0 out +b+res", 2 * f * 4096, 320, 320, None, None, "br", None)]
    print(f"{'gemm':30s} {'M':>8s} {'N':>6s} {'K':>6s} {'direct ms':>9s} {'TF/s':>7s} {'rows ms':>8s} {'TF/s':>7s} {'x':>5s}  bitwise", flush=True)
    tot = [0.0, 0.0]
    for name, M, N, K, conv, tconv, terms, hm in cases:
        taps = 9 if conv else (3 if tconv else 1)
        rows_in = M if not conv else (M // (conv[2] * conv[3])) * conv[0] * conv[1]
        x, w = rnd(rows_in, K), rnd(N, taps, K) * (0.05 if K * taps > 2000 else 0.2)
        kw = dict(M=M, conv=conv, tconv=tconv)
        if "b" in terms:
            kw["bias"] = rnd(N)
        if "v" in terms:
            kw["rowvec"], kw["rows_per_vec"] = rnd(B, N), M // B
        if "r" in terms:
            kw["res"] = rnd(M, N)
        if "s" in terms:
            kw["res2"] = rnd(M, N)
        if "g" in terms:
            kw["geglu"] = True
        if hm:
            kw["head_major"] = hm
        os.environ["ME_GEMM_ROWEPI"] = "0"
        ref = ops.gemm(x, w, **kw)
        k_old = ops._last_kernel()
        t_old = timeit(lambda: ops.gemm(x, w, **kw))
        os.environ["ME_GEMM_ROWEPI"] = "1"
        y = ops.gemm(x, w, **kw)
        ok = all(torch.equal(a_, b_) for a_, b_ in zip(y, ref)) if hm else torch.equal(y, ref)
        t_new = timeit(lambda: ops.gemm(x, w, **kw))
        fl = 2.0 * M * N * K * taps
        tot[0] += t_old
        tot[1] += t_new
        print(f"{name:30s} {M:8d} {N:6d} {K*taps:6d} {t_old:9.3f} {fl/t_old/1e9:7.1f} {t_new:8.3f} {fl/t_new/1e9:7.1f} {t_old/t_new:5.2f}  {'equal' if ok else 'DIFFERENT'}  {k_old}", flush=True)
        del x, w, ref, y, kw
    print(f"sum: direct {tot[0]:.3f} ms, row-contiguous {tot[1]:.3f} ms ({tot[0] / tot[1]:.3f} x)")
    os.environ.pop("ME_GEMM_ROWEPI", None)


def bench_gemm_tileorder():
    """Experiment (round 5): order of the tiles inside an XCD's contiguous run of the 8-phase kernels -- column tiles fastest (default) vs row blocks fastest
    (ME_GEMM_TILE_ORDER=1: 32 row blocks of one column tile at a time share its weight slabs in L2).  Same process, alternating, bitwise check."""
    import os
    B, f = 4, 24
    cases = []
    for li, (hw, C) in enumerate([(64, 320), (32, 640), (16, 1280)]):
        M = B * f * hw * hw
        cases += [(f"L{li} qkv", M, 3 * C, C, None, False), (f"L{li} ff1 geglu", M, 8 * C, C, None, True), (f"L{li} ff2", M, C, 4 * C, None, False)]
    cases += [("L2 conv3x3", B * f * 256, 1280, 1280, (16, 16, 16, 16, 1, 0), False), ("L2 conv 2560->1280", B * f * 256, 1280, 2560, (16, 16, 16, 16, 1, 0), False),
              ("L1->L0 ups conv 640", B * f * 4096, 640, 640, (32, 32, 64, 64, 1, 1), False), ("L2->L1 ups conv 1280", B * f * 1024, 1280, 1280, (16, 16, 32, 32, 1, 1), False),
              ("L3 ff1 geglu", B * f * 64, 10240, 1280, None, True), ("big 8192 x 8320 K4096", 8192, 8320, 4096, None, False)]
    print(f"{'gemm':26s} {'M':>8s} {'N':>6s} {'K':>6s} {'cols-fastest ms':>16s} {'rows-fastest ms':>16s} {'x':>6s}  bitwise", flush=True)
    for name, M, N, K, conv, geglu in cases:
        taps = 9 if conv else 1
        rows_in = M if not conv else (M // (conv[2] * conv[3])) * conv[0] * conv[1]
        x, w = rnd(rows_in, K), rnd(N, taps, K) * (0.05 if K * taps > 2000 else 0.2)
        kw = dict(M=M, conv=conv, geglu=geglu, bias=rnd(N))
        res = {"0": [], "1": []}
        outs = {}
        for rep in range(2):
            for o in ("0", "1"):
                os.environ["ME_GEMM_TILE_ORDER"] = o
                outs[o] = ops.gemm(x, w, **kw)
                res[o].append(timeit(lambda: ops.gemm(x, w, **kw)))
        t0, t1 = min(res["0"]), min(res["1"])
        print(f"{name:26s} {M:8d} {N:6d} {K*taps:6d} {t0:16.3f} {t1:16.3f} {t0 / t1:6.3f}  {'equal' if torch.equal(outs['0'], outs['1']) else 'DIFFERENT'}  {ops._last_kernel()}", flush=True)
        del x, w, outs, kw
    os.environ.pop("ME_GEMM_TILE_ORDER", None)


def bench_gemm_cached():
    """Same dense shapes with every X row aliased to row 0 (stride-0 view): X comes from L2, only the output streams.
    The gap to the normal run = what HBM latency / bandwidth on the activation stream costs."""
    B, f = 4, 24
    print(f"{'gemm':24s} {'M':>8s} {'N':>6s} {'K':>6s} {'ms':>8s} {'ms X-cached':>12s}")
    for name, M, N, K in [("L0 qkv", B * f * 4096, 960, 320), ("L0 out", B * f * 4096, 320, 320), ("L0 ff2", B * f * 4096, 320, 1280),
                          ("L1 qkv", B * f * 1024, 1920, 640), ("L1 ff2", B * f * 1024, 640, 2560), ("L2 ff2", B * f * 256, 1280, 5120)]:
        x, w = rnd(M, K), rnd(N, 1, K)
        xc = x[:256].repeat(1, 1).as_strided((M, K), (0, 1))
        ms = timeit(lambda: ops.gemm(x, w))
        msc = timeit(lambda: ops.gemm(xc, w))
        print(f"{name:24s} {M:8d} {N:6d} {K:6d} {ms:8.3f} {msc:12.3f}")


def bench_gemm_ksweep():
    """Fixed M, N; K sweep: slope = time per 64-wide K slab, intercept = per-tile prologue + epilogue."""
    M = 4 * 24 * 4096
    print(f"{'N':>6s} {'K':>6s} {'ms':>8s} {'us/tile':>8s}")
    for N in (320, 960):
        for K in (64, 128, 320, 640, 1280):
            x, w = rnd(M, K), rnd(N, 1, K)
            ms = timeit(lambda: ops.gemm(x, w))
            tiles = (M // 256) * (N // 320)
            print(f"{N:6d} {K:6d} {ms:8.3f} {ms * 1e3 * 256 / tiles:8.2f}")


def bench_gemm_epi():
    """Epilogue terms on the bandwidth-bound L0 / L1 square projections."""
    print(f"{'shape':28s} {'plain':>8s} {'bias':>8s} {'res':>8s} {'bias+res':>9s} {'inplace':>8s}")
    for M, C in ((4 * 24 * 4096, 320), (4 * 24 * 1024, 640), (4 * 24 * 256, 1280)):
        x, w, b, r = rnd(M, C), rnd(C, 1, C), rnd(C), rnd(M, C)
        o = torch.empty_like(r)
        t = [timeit(lambda: ops.gemm(x, w, out=o)), timeit(lambda: ops.gemm(x, w, bias=b, out=o)), timeit(lambda: ops.gemm(x, w, res=r, out=o)),
             timeit(lambda: ops.gemm(x, w, bias=b, res=r, out=o)), timeit(lambda: ops.gemm(x, w, bias=b, res=r, out=r))]
        print(f"M{M} N{C} K{C}".ljust(28) + " ".join(f"{v:8.3f}" for v in t))


def bench_gemm_small():
    """Small-M layers (level 3 / mid / ControlNet): few tiles per CU."""
    print(f"{'shape':36s} {'ms':>8s} {'TF/s':>7s}")
    for M, N, K, conv in [(1536, 1280, 1280, (8, 8, 8, 8, 1, 0)), (1536, 1280, 1280, None), (3072, 1280, 1280, (8, 8, 8, 8, 1, 0)), (6144, 1280, 1280, (8, 8, 8, 8, 1, 0)),
                          (6144, 1280, 2560, (8, 8, 8, 8, 1, 0)), (6144, 1280, 1280, None), (6144, 3840, 1280, None), (24576, 640, 640, None), (24576, 1280, 1280, None),
                          (98304, 320, 320, None), (98304, 320, 320, (64, 64, 64, 64, 1, 0))]:
        taps = 9 if conv else 1
        x, w = rnd(M, K), rnd(N, taps, K)
        ms = timeit(lambda: ops.gemm(x, w, M=M, conv=conv))
        print(f"M{M} N{N} K{K} taps{taps}".ljust(36) + f" {ms:8.3f} {2.0*M*N*K*taps/ms/1e9:7.1f}")


def bench_gemm_abl():
    """A few big-tile shapes for the ME_GEMM_ABL main-loop ablations."""
    B, f = 4, 24
    print(f"{'shape':28s} {'ms':>8s} {'TF/s':>7s}")
    for name, M, N, K in [("L0 qkv", B * f * 4096, 960, 320), ("L0 ff1", B * f * 4096, 2560, 320), ("L0 ff2", B * f * 4096, 320, 1280),
                          ("L1 qkv", B * f * 1024, 1920, 640), ("L1 ff2", B * f * 1024, 640, 2560), ("L2 ff1", B * f * 256, 10240, 1280), ("big 8192^2 K4096", 8192, 8320, 4096)]:
        x, w = rnd(M, K), rnd(N, 1, K)
        ms = timeit(lambda: ops.gemm(x, w))
        print(f"{name:28s} {ms:8.3f} {2.0*M*N*K/ms/1e9:7.1f}")
        del x, w


def bench_attn(only_first=False):
    B, f = 4, 24
    print(f"{'attn':28s} {'ms':>8s} {'TF/s(ref)':>9s}")
    for name, dh, N, seg, items, mask in [("L0 prev|cur", 40, 4096, "pc", B * f, False)] if only_first else [("L0 prev|cur", 40, 4096, "pc", B * f, False), ("L0 edited", 40, 4096, "ed", B * f, True),
                                          ("L0 self (cn)", 40, 4096, "self", 2 * f, False), ("L0 cross 77", 40, 4096, "cross", B * f, False),
                                          ("L1 prev|cur", 80, 1024, "pc", B * f, False), ("L1 edited", 80, 1024, "ed", B * f, True),
                                          ("L2 prev|cur", 160, 256, "pc", B * f, False)]:
        C = 8 * dh
        nk = 77 if seg == "cross" else N
        q = rnd(items * N, 3 * C)
        kv = rnd(B * 77, 2 * C) if seg == "cross" else None
        si, sm = {"pc": lambda: segments.prev_cur(B, f, dev), "ed": lambda: segments.edited_spatial(f, dev, True),
                  "self": lambda: segments.self_items(items, dev), "cross": lambda: segments.cross_text(B, f, dev)}[seg]()
        mk = (torch.rand(8, N, device=dev) > 0.5).half() if mask else None
        if seg == "cross":
            fn = lambda: ops.attention(q[:, :C], kv[:, :C], kv[:, C:], heads=8, dh=dh, n_items=items, nq=N, nk=nk, seg_item=si, seg_mode=sm)
        else:
            fn = lambda: ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], heads=8, dh=dh, n_items=items, nq=N, nk=nk, seg_item=si, seg_mode=sm, mask=mk)
        ms = timeit(fn)
        units = segments.KEY_UNITS[si.data_ptr()]
        print(f"{name:28s} {ms:8.3f} {4.0*8*N*nk*units*dh/ms/1e9:9.1f}")
        if name == "L0 prev|cur" and not only_first:
            # worst case of the speculative fixed-offset softmax: a key far heavier than the first tile late in every row ->
            # every block discards phase A and re-runs the classic sweep
            for it in range(items):
                q[it * N + N - 100, C:2 * C] *= 40.0
            ms = timeit(fn)
            print(f"{name + ' (all blocks fall back)':28s} {ms:8.3f} {4.0*8*N*nk*units*dh/ms/1e9:9.1f}")
            # trained-model-like: every query's heaviest key is its own position (k_i parallel to q_i, ~ +30 nats over the rest);
            # the probe takes the offset from there, so nothing falls back
            q = rnd(items * N, 3 * C)
            q[:, C:2 * C] = q[:, :C] * 4.0
            ms = timeit(fn)
            print(f"{name + ' (diagonal-dominant)':28s} {ms:8.3f} {4.0*8*N*nk*units*dh/ms/1e9:9.1f}")
            # gradual growth: every query's logits rise by 60 nats from the first key of a frame to the last (a common component of q against a
            # ramp in k): without the per-stage re-basing every block of the early queries falls back, with it none does
            q = rnd(items * N, 3 * C)
            ramp = torch.linspace(0.0, 60.0, N, device=dev).repeat(items)
            for h in range(8):
                q[:, h * dh] = 3.0                                           # query component along e_0 of the head
                q[:, C + h * dh] = (ramp / (3.0 * dh ** -0.5)).half()        # key component: logit += ramp_j
            ops.attention_fallback_blocks(reset=True)
            ms = timeit(fn)
            print(f"{name + ' (keys grow 60 nats)':28s} {ms:8.3f} {4.0*8*N*nk*units*dh/ms/1e9:9.1f}   blocks that fell back: {ops.attention_fallback_blocks()}")
        del q


def bench_attn_order():
    """Block order of the multi-segment attention launches (ME_ATTN_ORDER flips per call): heads slowest (round 5: an XCD's run is one head over all items,
    the `cur` frame of item f is still in its L2 when item f + 1 reads it as `prev`) vs items slowest.  Same process, alternating, bitwise check.
    Head-major K | V panels as the model's q|k|v projection writes them."""
    import os
    B, f = 4, 24
    print(f"{'attn':28s} {'items-slowest ms':>17s} {'heads-slowest ms':>17s} {'x':>6s}  bitwise")
    for name, dh, N, seg, mask in [("L0 prev|cur", 40, 4096, "pc", False), ("L0 edited", 40, 4096, "ed", True), ("L1 prev|cur", 80, 1024, "pc", False), ("L1 edited", 80, 1024, "ed", True)]:
        C = 8 * dh
        items = B * f
        q = rnd(items * N, C)
        kv = rnd(16, items * N, dh)          # head-major panels: K heads 0..7, V heads 8..15
        si, sm = {"pc": lambda: segments.prev_cur(B, f, dev), "ed": lambda: segments.edited_spatial(f, dev, True)}[seg]()
        mk = (torch.rand(8, N, device=dev) > 0.5).half() if mask else None
        fn = lambda: ops.attention(q, kv[:8], kv[8:], heads=8, dh=dh, n_items=items, nq=N, nk=N, seg_item=si, seg_mode=sm, mask=mk)
        res = {}
        for rep in range(2):
            for order in ("0", "1"):
                os.environ["ME_ATTN_ORDER"] = order
                out = fn()
                res.setdefault(order, []).append(timeit(fn))
                res["out" + order] = out
        t0, t1 = min(res["0"]), min(res["1"])
        print(f"{name:28s} {t0:17.3f} {t1:17.3f} {t0 / t1:6.3f}  {'equal' if torch.equal(res['out0'], res['out1']) else 'DIFFERENT'}", flush=True)
        del q, kv
    os.environ.pop("ME_ATTN_ORDER", None)


def bench_attn_kvres():
    """The 77-key text cross-attention with K | V resident across query blocks (round 6, attn2_kernel<..., KVRES>; ME_ATTN_KVRES flips per call): same
    process, alternating, bitwise check.  GB/s = Q read + O written (the launch's algorithmic bytes; K | V are 4 x 77 rows)."""
    import os
    B, f = 4, 24
    print(f"{'cross-attention, 77 keys':28s} {'per-block ms':>13s} {'resident ms':>12s} {'x':>6s} {'GB/s':>7s}  vs per-block")
    for name, dh, N, items, il in [("L0 UNet", 40, 4096, B * f, False), ("L0 ControlNet", 40, 4096, 2 * f, True), ("L1 UNet", 80, 1024, B * f, False), ("L1 ControlNet", 80, 1024, 2 * f, True)]:
        C = 8 * dh
        q = rnd(items * N, C)
        kv = rnd(B * 77, 2 * C)
        si, sm = segments.cross_interleaved(items, 2, dev) if il else segments.cross_text(B, f, dev)
        fn = lambda: ops.attention(q, kv[:, :C], kv[:, C:], heads=8, dh=dh, n_items=items, nq=N, nk=77, seg_item=si, seg_mode=sm)
        res = {}
        for rep in range(2):
            for sw in ("0", "1"):
                os.environ["ME_ATTN_KVRES"] = sw
                out = fn()
                res.setdefault(sw, []).append(timeit(fn))
                res["out" + sw] = out
        t0, t1 = min(res["0"]), min(res["1"])
        print(f"{name:28s} {t0:13.4f} {t1:12.4f} {t0 / t1:6.3f} {2 * items * N * C * 2 / t1 / 1e6:7.0f}  max |diff| {float((res['out0'].float() - res['out1'].float()).abs().max()):.2e}", flush=True)
        del q, kv
    os.environ.pop("ME_ATTN_KVRES", None)


def bench_attn_headmajor():
    """Experiment: the L0 [prev|cur] launch with K/V (and Q) stored head-major (contiguous 80-byte rows per head): heads folded into the batch axis."""
    B, f, dh, N = 4, 24, 40, 4096
    q = rnd(B * 8 * f * N, 3 * dh)
    k = rnd(B * 8 * f * N, dh)
    v = rnd(B * 8 * f * N, dh)
    si, sm = segments.prev_cur(B * 8, f, dev)
    for name, kk, vv in (("rows of 3*dh", q[:, dh:2 * dh], q[:, 2 * dh:]), ("contiguous dh", k, v)):
        ms = timeit(lambda: ops.attention(q[:, :dh], kk, vv, heads=1, dh=dh, n_items=B * 8 * f, nq=N, nk=N, seg_item=si, seg_mode=sm))
        print(f"L0 prev|cur head-major K/V {name:16s} {ms:8.3f} ms {4.0*8*N*N*2*B*f*dh/ms/1e9:9.1f} TF/s(ref)")


def bench_attn_headmajor_panels():
    """The production form of head-major K | V: the fused q|k|v projection writes K and V as [16, rows, dh] panels (me_gemm_args.C2), the attention reads
    them through head strides -- against the row-major form, projection and attention timed separately (level 0 and level 1, [prev | cur] and edited)."""
    B, f = 4, 24
    for dh, N in ((40, 4096), (80, 1024)):
        C = 8 * dh
        x, w = rnd(B * f * N, C), rnd(3 * C, 1, C) * (C ** -0.5)
        t_rm = timeit(lambda: ops.gemm(x, w))
        t_hm = timeit(lambda: ops.gemm(x, w, head_major=(C, dh)))
        qkv = ops.gemm(x, w)
        q, kv = ops.gemm(x, w, head_major=(C, dh))
        print(f"dh {dh}: q|k|v projection row-major {t_rm:.3f} ms, head-major K|V {t_hm:.3f} ms")
        for seg in ("pc", "ed"):
            si, sm = segments.prev_cur(B, f, dev) if seg == "pc" else segments.edited_spatial(f, dev, True)
            mk = (torch.rand(8, N, device=dev) > 0.5).half() if seg == "ed" else None
            kw = dict(heads=8, dh=dh, n_items=B * f, nq=N, nk=N, seg_item=si, seg_mode=sm, mask=mk)
            a_rm = timeit(lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], **kw))
            a_hm = timeit(lambda: ops.attention(q, kv[:8], kv[8:], **kw))
            print(f"   attention {seg}: row-major {a_rm:.3f} ms, head-major {a_hm:.3f} ms ({100 * (a_hm / a_rm - 1):+.1f} %)")
        del x, w, qkv, q, kv


def bench_misc():
    B, f = 4, 24
    for C, hw in [(320, 64), (640, 32), (1280, 16)]:
        M = B * f * hw * hw
        x = rnd(M, 3 * C)
        ms = timeit(lambda: ops.temporal_attention(x[:, :C], x[:, C:2 * C], x[:, 2 * C:], heads=8, dh=C // 8, batch=B, frames=f, npix=hw * hw))
        print(f"tattn C={C} {ms:.3f} ms  {8.0*M*C/ms/1e6:.0f} GB/s")
        y, g, b = rnd(M, C), rnd(C), rnd(C)
        ms = timeit(lambda: ops.groupnorm(y, g, b, rows_per_group=f * hw * hw, eps=1e-5, silu=True))
        print(f"groupnorm C={C} {ms:.3f} ms  {4.0*M*C/ms/1e6:.0f} GB/s (algorithmic)")
        ms = timeit(lambda: ops.layernorm(y, g, b))
        print(f"layernorm C={C} {ms:.3f} ms  {4.0*M*C/ms/1e6:.0f} GB/s")
        del x, y


def bench_gn():
    """GroupNorm at the step's shapes (B = 4, 24 frames): the resnets' 5-D form (statistics across all frames of a batch row) on plain and skip-concat
    widths, the transformers' per-frame form; statistics and apply passes timed separately as well."""
    B, f = 4, 24
    tot = 0.0
    for C, hw, per_frame, count in [(320, 64, False, 7), (640, 64, False, 2), (960, 64, False, 1), (320, 64, True, 5), (640, 32, False, 5), (1280, 32, False, 1),
                                    (1920, 32, False, 1), (640, 32, True, 5), (1280, 16, False, 5), (2560, 16, False, 2), (1280, 16, True, 5), (1280, 8, False, 8)]:
        M = B * f * hw * hw
        y, g, b = rnd(M, C), rnd(C), rnd(C)
        rpg = hw * hw if per_frame else f * hw * hw
        ms = timeit(lambda: ops.groupnorm(y, g, b, rows_per_group=rpg, eps=1e-5, silu=True))
        tot += ms * count
        print(f"groupnorm C={C:4d} {hw}x{hw} {'per-frame' if per_frame else '5-D      '} {ms:.3f} ms  {4.0*M*C/ms/1e6:.0f} GB/s (algorithmic, in + out)")
        del y
    print(f"groupnorm weighted sum (~ launches of a step) {tot:.3f} ms")


def bench_bwd():
    """Backward kernels at the null-text / adapter-training geometry (batch 1, 24 frames x 512^2)."""
    f = 24
    print(f"{'backward':40s} {'ms':>8s} {'TF/s':>8s}")
    for name, dh, N in [("attn_bwd L0 prev|cur", 40, 4096), ("attn_bwd L1 prev|cur", 80, 1024), ("attn_bwd L2 prev|cur", 160, 256)]:
        C = 8 * dh
        qkv = rnd(f * N, 3 * C)
        si, sm = segments.prev_cur(1, f, dev)
        lse = torch.empty((f * N, 8), dtype=torch.float32, device=dev)
        out = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads=8, dh=dh, n_items=f, nq=N, nk=N, seg_item=si, seg_mode=sm, lse=lse)
        dout = torch.randn(f * N, C, device=dev)
        g = torch.zeros(f * N, 3 * C, device=dev)
        fn = lambda: ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, dout, dq=g[:, :C], dk=g[:, C:2 * C], dv=g[:, 2 * C:], lse=lse, heads=8, dh=dh,  # noqa: E731
                                       n_items=f, nq=N, nk=N, seg_item=si, seg_mode=sm)
        ms = timeit(fn)
        units = segments.KEY_UNITS[si.data_ptr()]
        print(f"{name:40s} {ms:8.3f} {10.0 * 8 * N * N * units * dh / ms / 1e9:8.1f}   (5 matrix products of the flash backward; 7 executed)")
        ms_f = timeit(lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads=8, dh=dh, n_items=f, nq=N, nk=N, seg_item=si, seg_mode=sm))
        print(f"{'   forward of the same launch':40s} {ms_f:8.3f} {4.0 * 8 * N * N * units * dh / ms_f / 1e9:8.1f}")
    for name, M, N, K in [("gemm_dw L0 qkv", f * 4096, 960, 320), ("gemm_dw L0 ff1", f * 4096, 2560, 320), ("gemm_dw L2 ff2", f * 256, 1280, 5120)]:
        x, dy = rnd(M, K), torch.randn(M, N, device=dev)
        dst = torch.zeros(N, 1, K, device=dev)
        ms = timeit(lambda: ops.gemm_dw(dy, x, dst=dst, taps=1, K=K, M=M))
        print(f"{name:40s} {ms:8.3f} {2.0 * M * N * K / ms / 1e9:8.1f}")
    for C, hw in [(320, 64), (1280, 16)]:
        M = f * hw * hw
        x = rnd(M, 3 * C)
        dout = torch.randn(M, C, device=dev)
        ms = timeit(lambda: ops.temporal_attention_bwd(x[:, :C], x[:, C:2 * C], x[:, 2 * C:], None, dout, heads=8, dh=C // 8, batch=1, frames=f, npix=hw * hw))
        print(f"{'tattn_bwd C=' + str(C):40s} {ms:8.3f}")
        y, gm, bt = rnd(M, C), rnd(C), rnd(C)
        dy = torch.randn(M, C, device=dev)
        ms = timeit(lambda: ops.groupnorm_bwd(y, gm, bt, dy, rows_per_group=f * hw * hw, eps=1e-5, silu=True))
        print(f"{'groupnorm_bwd C=' + str(C):40s} {ms:8.3f}")
        ms = timeit(lambda: ops.layernorm_bwd(y, gm, dy))
        print(f"{'layernorm_bwd C=' + str(C):40s} {ms:8.3f}")


if __name__ == "__main__" and "lnfold" in sys.argv[1:]:
    bench_lnfold()
    sys.exit(0)

if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn", "misc"]
    if "gemm" in what:
        bench_gemm()
    if "attn" in what:
        bench_attn()
    if "gemms" in what:
        bench_gemm_small()
    if "gemme" in what:
        bench_gemm_epi()
    if "gemmk" in what:
        bench_gemm_ksweep()
    if "gemm8p" in what:
        bench_gemm_8p()
    if "gemmabl" in what:
        bench_gemm_abl()
    if "rowepi" in what:
        bench_gemm_rowepi()
    if "tileorder" in what:
        bench_gemm_tileorder()
    if "gemmc" in what:
        bench_gemm_cached()
    if "attnorder" in what:
        bench_attn_order()
    if "attnkvres" in what:
        bench_attn_kvres()
    if "attnhm" in what:
        bench_attn_headmajor()
    if "attnhmp" in what:
        bench_attn_headmajor_panels()
    if "attn1" in what:
        bench_attn(True)
    if "misc" in what:
        bench_misc()
    if "gn" in what:
        bench_gn()
    if "bwd" in what:
        bench_bwd()
