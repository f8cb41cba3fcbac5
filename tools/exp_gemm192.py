"""Experiment (round 6): which grids should take the 192-row 8-phase kernel.  (a) ME_GEMM_192_MINK: fewest K tiles (was 8: the [98304 x 320] K = 320
projections stayed on 128 x 160 tiles); (b) ME_GEMM_8P_192: smallest grid in 192 x 320 tiles (448).  Same process, alternating; bitwise check."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, dev
what = sys.argv[1] if len(sys.argv) > 1 else "mink"
if what == "mink":
    var, vals, shapes = "ME_GEMM_192_MINK", ("8", "5", "4"), [(98304, 320, 320), (49152, 640, 320), (98304, 320, 256)]
elif what == "small":
    var, vals = "ME_GEMM_8P_192", ("448", "160", "128", "96", "64")
    shapes = [(6144, 1280, 1280), (6144, 1280, 5120), (6144, 1280, 2560), (3072, 1280, 1280), (3072, 1280, 5120), (12288, 640, 640), (12288, 640, 2560), (6144, 1920, 640), (6144, 640, 640), (1536, 3840, 1280), (1536, 1280, 1280)]
else:
    var, vals = "ME_GEMM_8P_192", ("448", "384", "256", "192")
    shapes = [(24576, 640, 640), (24576, 640, 2560), (12288, 1280, 1280), (6144, 3840, 1280), (24576, 960, 320), (49152, 320, 320), (12288, 1280, 5120), (6144, 2560, 1280), (3072, 3840, 1280), (49152, 320, 1280)]
for M, N, K in shapes:
    x, w, b, r = rnd(M, K), rnd(N, 1, K), rnd(N), rnd(M, N)
    for name, kw in [("plain", {}), ("+b +res", dict(bias=b, res=r)), ("+b +res lnout", dict(bias=b, res=r, ln_out=True))]:
        if name == "+b +res lnout" and N > 1536: continue
        res, outs, kn = {}, {}, {}
        for rep in range(2):
            for sw in vals:
                os.environ[var] = sw
                o = ops.gemm(x, w, **kw)
                outs[sw] = o[0] if isinstance(o, tuple) else o
                res.setdefault(sw, []).append(timeit(lambda: ops.gemm(x, w, **kw)))
                kn[sw] = ops._last_kernel().replace("_kernel", "")
        os.environ.pop(var, None)
        print(f"M{M} N{N} K{K} {name:14s}", {k_: round(min(v), 4) for k_, v in res.items()}, kn[vals[0]], "->", kn[vals[-1]], "max diff", max(float((outs[vals[0]].float() - outs[v_].float()).abs().max()) for v_ in vals[1:]), flush=True)
