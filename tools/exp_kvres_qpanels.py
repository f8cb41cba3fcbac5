"""Experiment (round 6): the resident-key text cross-attention at level 0 / 1 -- walk length sweep (ME_ATTN_KVRES=n) and Q as head-major panels."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, segments, dev
B, f = 4, 24
for name, dh, N in [("L0", 40, 4096), ("L1", 80, 1024)]:
    C = 8 * dh; items = B * f
    q = rnd(items * N, C)
    qp = rnd(8, items * N, dh)
    kvp = rnd(16, B * 77, dh)
    si, sm = segments.cross_text(B, f, dev)
    a = dict(heads=8, dh=dh, n_items=items, nq=N, nk=77, seg_item=si, seg_mode=sm)
    out = torch.empty(items * N, C, dtype=torch.float16, device=dev)
    for n in ("1", "2", "4", "8", "16"):
        os.environ["ME_ATTN_KVRES"] = n
        t_row = timeit(lambda: ops.attention(q, kvp[:8], kvp[8:], out=out, **a))
        t_pan = timeit(lambda: ops.attention(qp, kvp[:8], kvp[8:], out=out, **a))
        print(name, "walk", n, "Q rows", round(t_row, 4), "Q panels", round(t_pan, 4), ops._last_kernel(), flush=True)
