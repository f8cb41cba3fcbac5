"""Where the fixed cost of the dh = 40 attention launch sits: t = L + rounds x (p + nseg x s) fitted over [prev | cur] (2 segments) and self (1 segment)
launches of 24 ... 96 items (6 ... 24 block rounds of the 256 CUs).  python tools/exp_attn_prologue.py"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from kbench import timeit, rnd, dev  # noqa: E402
from motioneditor_amd import ops, segments  # noqa: E402

dh, N, C = 40, 4096, 320
rows = []
for items_b, f in ((1, 24), (2, 24), (4, 24), (4, 12)):
    for kind in ("pc", "self"):
        items = items_b * f
        q = rnd(items * N, 3 * C)
        si, sm = segments.prev_cur(items_b, f, dev) if kind == "pc" else segments.self_items(items, dev)
        ms = timeit(lambda: ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], heads=8, dh=dh, n_items=items, nq=N, nk=N, seg_item=si, seg_mode=sm))
        nseg = 2 if kind == "pc" else 1
        rounds = items * 8 * (N // 512) / 256
        # frame 0 of every batch row of a [prev | cur] table carries ONE segment (its duplicate is collapsed): segments per item on average
        segs = (2 * items - items_b) / items if kind == "pc" else 1.0
        rows.append((rounds, rounds * segs, ms))
        print(f"{kind:5s} items {items:3d} rounds {rounds:5.1f} segments/item {segs:.3f}  {ms:7.3f} ms")
        del q
A = np.array([[1.0, r, rs] for r, rs, _ in rows])
y = np.array([m for _, _, m in rows])
(L, p, s), res, *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"fit: per launch L = {L * 1e3:.1f} us, per block round p = {p * 1e3:.1f} us, per (round, segment) s = {s * 1e3:.1f} us; residual {np.abs(A @ [L, p, s] - y).max() * 1e3:.1f} us")
