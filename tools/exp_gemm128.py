"""Experiment (round 6): 128-row tiles of the 8-phase kernel (gemm8p_kernel<128,320,.>) for grids below 192 tiles of 192 x 320: ME_GEMM_8P_128 = smallest grid
in 128 x 320 tiles (0 / unset = never).  Same process, alternating; bitwise check against the default dispatch."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, dev
var, vals = "ME_GEMM_8P_128", ("0", "192", "128", "96")
for M, N, K, hw in [(6144, 1280, 1280, 0), (6144, 1280, 5120, 0), (6144, 1280, 1280, 16), (6144, 1280, 2560, 16), (6144, 1280, 640, 16), (12288, 640, 640, 0), (12288, 640, 2560, 0), (24576, 320, 320, 0), (24576, 320, 320, 32),
                    (3072, 1280, 1280, 0), (3072, 1280, 1280, 8), (6144, 640, 640, 16), (49152, 320, 320, 0)]:
    taps = 9 if hw else 1
    x, w, b, r = rnd(M, K), rnd(N, taps, K), rnd(N), rnd(M, N)
    conv = dict(M=M, conv=(hw, hw, hw, hw, 1, 0)) if hw else {}
    for name, kw in [("+b", dict(bias=b)), ("+b +res", dict(bias=b, res=r))] + ([] if hw else [("+b +res lnout", dict(bias=b, res=r, ln_out=True))]):
        res, outs, kn = {}, {}, {}
        for rep in range(2):
            for sw in vals:
                os.environ[var] = sw
                o = ops.gemm(x, w, **conv, **kw)
                outs[sw] = o[0] if isinstance(o, tuple) else o
                res.setdefault(sw, []).append(timeit(lambda: ops.gemm(x, w, **conv, **kw)))
                kn[sw] = ops._last_kernel().replace("_kernel", "")
        os.environ.pop(var, None)
        print(f"M{M} N{N} K{K} taps{taps} {name:14s}", {k_: round(min(v), 4) for k_, v in res.items()}, kn[vals[0]], "->", kn[vals[-1]], "max diff", max(float((outs[vals[0]].float() - outs[v_].float()).abs().max()) for v_ in vals[1:]), flush=True)
