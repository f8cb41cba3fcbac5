"""debug: run-to-run and sub-batch bitwise equality of the LayerNorm-folded projections (round 6)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from motioneditor_amd import ops

torch.manual_seed(0)
dev = "cuda"
for (M, C, N, kind) in [(131072, 320, 960, "hm"), (131072, 320, 320, "plain"), (131072, 320, 2560, "geglu"), (32768, 640, 1920, "hm"), (32768, 640, 5120, "geglu")]:
    x = (torch.randn(M // 2, C, device=dev) * 0.7 + 0.3).half()
    x = torch.cat([x, x]).contiguous()
    w = (torch.randn(N, 1, C, device=dev) * C ** -0.5).half()
    cs, cv = torch.randn(N, device=dev), torch.randn(N, device=dev)
    st = ops.ln_stats(x)
    kw = dict(head_major=(0, C // 8)) if kind == "hm" else {}

    def run(xx, ss, sel=1):
        ops.SELECT_ROWS_SCALE = sel
        try:
            o = ops.gemm(xx, w, geglu=kind == "geglu", ln=(ss, cs, cv, 1e-5), **kw)
        finally:
            ops.SELECT_ROWS_SCALE = 1
        return (o[1] if kind == "hm" else o).clone()
    a = run(x, st)
    b = run(x, st)
    h = run(x[:M // 2], st[:, :M // 2].contiguous(), 2)
    if kind == "hm":
        top, bot, hh = a[:, :M // 2], a[:, M // 2:], h
    else:
        top, bot, hh = a[:M // 2], a[M // 2:], h
    print(M, C, N, kind, "run-to-run equal:", torch.equal(a, b), " duplicate rows equal:", torch.equal(top, bot), " half-batch equal:", torch.equal(top, hh),
          " max diff half:", float((top.float() - hh.float()).abs().max()), flush=True)
    # producer stats: fused vs half
    res = torch.randn(M, C, device=dev).half() if N == C else None
