"""The level-0 [prev | cur] and edited attention launches of config 3 (head-major K | V, as the model issues them), a few times each -- the workload of the
PMC passes in tools/exp_attn_order.sh.  usage: python tools/attn_one.py [pc|ed] [reps] [qhm]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from motioneditor_amd import ops, segments  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pc"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, f, N, dh = 4, 24, 4096, 40
C, items = 8 * dh, B * f
g = torch.Generator(device="cuda").manual_seed(1)
q = (torch.randn(items * N, C, device="cuda", generator=g) * 0.5).half()
kv = (torch.randn(16, items * N, dh, device="cuda", generator=g) * 0.5).half()
si, sm = segments.prev_cur(B, f, "cuda") if kind == "pc" else segments.edited_spatial(f, "cuda", True)
mk = (torch.rand(8, N, device="cuda", generator=g) > 0.5).half() if kind == "ed" else None
if len(sys.argv) > 3 and sys.argv[3] == "qhm":      # Q as per-head panels too (ABI 8)
    q = q.reshape(items * N, 8, dh).permute(1, 0, 2).contiguous()
for _ in range(reps):
    ops.attention(q, kv[:8], kv[8:], heads=8, dh=dh, n_items=items, nq=N, nk=N, seg_item=si, seg_mode=sm, mask=mk)
torch.cuda.synchronize()
