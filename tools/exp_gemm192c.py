"""Experiment (round 6): 3 x 3 convolutions (gather form) on grids of 192 .. 447 tiles of 192 x 320: ME_GEMM_8P_192 = 448 (old default) against 192."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, dev
var, vals = "ME_GEMM_8P_192", ("448", "192")
for M, N, K, hw in [(24576, 640, 640, 32), (24576, 640, 320, 32), (24576, 640, 1280, 32), (24576, 320, 320, 32), (6144, 1280, 1280, 16), (6144, 1920, 640, 16), (98304, 320, 320, 64), (49152, 320, 320, 64), (12288, 1280, 1280, 16)]:
    x, w, b, r = rnd(M, K), rnd(N, 9, K), rnd(N), rnd(M, N)
    for name, kw in [("+b", dict(bias=b)), ("+b +res", dict(bias=b, res=r))]:
        res, outs, kn = {}, {}, {}
        for rep in range(2):
            for sw in vals:
                os.environ[var] = sw
                outs[sw] = ops.gemm(x, w, M=M, conv=(hw, hw, hw, hw, 1, 0), **kw)
                res.setdefault(sw, []).append(timeit(lambda: ops.gemm(x, w, M=M, conv=(hw, hw, hw, hw, 1, 0), **kw)))
                kn[sw] = ops._last_kernel().replace("_kernel", "")
        os.environ.pop(var, None)
        print(f"M{M} N{N} K{K} taps9 {name:8s}", {k_: round(min(v), 4) for k_, v in res.items()}, kn[vals[0]], "->", kn[vals[-1]], "max diff", float((outs[vals[0]].float() - outs[vals[-1]].float()).abs().max()), flush=True)
