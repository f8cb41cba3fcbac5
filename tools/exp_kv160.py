"""Experiment (round 6): the resident-key cross-attention form at dh = 160 (level 2: 256 queries per item, 77 keys) -- ME_LIB = build with the <160,1,8,...,KVRES> dispatch."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import kbench
from kbench import rnd, timeit, ops, segments, dev
sys.path.insert(0, "/root/repo/tests")
B, f, dh = 4, 24, 160
C = 8 * dh
for N in (256, 1024):
    items = B * f
    q, kv = rnd(items * N, C), rnd(B * 77, 2 * C)
    si, sm = segments.cross_text(B, f, dev)
    fn = lambda: ops.attention(q, kv[:, :C], kv[:, C:], heads=8, dh=dh, n_items=items, nq=N, nk=77, seg_item=si, seg_mode=sm)
    out = fn(); k = ops._last_kernel()
    os.environ["ME_ATTN_KVRES"] = "0"
    ref = fn(); k0 = ops._last_kernel(); t0 = timeit(fn)
    os.environ.pop("ME_ATTN_KVRES")
    t1 = timeit(fn)
    print(f"dh160 nq={N}: {k0} {t0:.4f} ms -> {k} {t1:.4f} ms  max diff {float((out.float() - ref.float()).abs().max()):.2e} of {float(ref.float().abs().max()):.2e}", flush=True)
