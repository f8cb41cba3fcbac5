"""Experiment (round 6): GEGLU projections of the small levels on the 256-wide 8-phase kernel: ME_GEMM_GEGLU_MIN (tiles of 256 x 256) 640 (old) / 480 / 240 / 120."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, dev
var, vals = "ME_GEMM_GEGLU_MIN", ("640", "480", "240", "120")
for M, N, K in [(3072, 10240, 1280), (1536, 10240, 1280), (12288, 5120, 640), (6144, 5120, 640), (24576, 2560, 320), (12288, 2560, 320)]:
    x, w, b = rnd(M, K), rnd(N, 1, K), rnd(N)
    res, outs, kn = {}, {}, {}
    for rep in range(2):
        for sw in vals:
            os.environ[var] = sw
            outs[sw] = ops.gemm(x, w, bias=b, geglu=True)
            res.setdefault(sw, []).append(timeit(lambda: ops.gemm(x, w, bias=b, geglu=True)))
            kn[sw] = ops._last_kernel().replace("_kernel", "")
    os.environ.pop(var, None)
    print(f"M{M} N{N} K{K} geglu", {k_: round(min(v), 4) for k_, v in res.items()}, [kn[v_] for v_ in vals], "max diff", max(float((outs[vals[0]].float() - outs[v_].float()).abs().max()) for v_ in vals[1:]), flush=True)
