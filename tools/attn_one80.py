"""The level-1 [prev | cur] attention launch of config 3 (dh = 80, 1024 queries / keys per item, head-major q | k | v panels), a few times -- the workload of the SQ-counter
passes in tools/exp_pmc_attn80.sh.  usage: python tools/attn_one80.py [reps]   (ME_ATTN_80_QT2=0: the 16-queries-per-wave kernel)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from motioneditor_amd import ops, segments  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, f, N, dh = 4, 24, 1024, 80
items = B * f
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(24, items * N, dh, device="cuda", generator=g) * 0.5).half()
si, sm = segments.prev_cur(B, f, "cuda")
for _ in range(reps):
    ops.attention(qkv[:8], qkv[8:16], qkv[16:], heads=8, dh=dh, n_items=items, nq=N, nk=N, seg_item=si, seg_mode=sm)
torch.cuda.synchronize()
print(ops._last_kernel())
