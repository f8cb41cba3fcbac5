"""Round 6: dh = 80 attention with 32 queries per wave and 128-key stages (the default for whole 256-query blocks) against the 16-queries-per-wave form
(ME_ATTN_80_QT2=0), level-1 launches, head-major K | V and Q panels as the model's projection writes them.  Same process, alternating."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, segments, dev
B, f, dh, N = 4, 24, 80, 1024
C = 8 * dh; items = B * f
qkv = rnd(24, items * N, dh)
for name, seg, mask in [("L1 prev|cur", "pc", False), ("L1 edited", "ed", True)]:
    si, sm = {"pc": lambda: segments.prev_cur(B, f, dev), "ed": lambda: segments.edited_spatial(f, dev, True)}[seg]()
    mk = (torch.rand(8, N, device=dev) > 0.5).half() if mask else None
    fn = lambda: ops.attention(qkv[:8], qkv[8:16], qkv[16:], heads=8, dh=dh, n_items=items, nq=N, nk=N, seg_item=si, seg_mode=sm, mask=mk)
    res, outs = {}, {}
    for rep in range(2):
        for sw in ("", "0"):
            if sw: os.environ["ME_ATTN_80_QT2"] = sw
            else: os.environ.pop("ME_ATTN_80_QT2", None)
            outs[sw] = fn()
            res.setdefault(sw, []).append(timeit(fn))
            k = ops._last_kernel()
    os.environ.pop("ME_ATTN_80_QT2", None)
    print(name, {("32 q/wave" if not k else "16 q/wave"): round(min(v), 4) for k, v in res.items()}, "max diff", float((outs[""].float() - outs["0"].float()).abs().max()), flush=True)
