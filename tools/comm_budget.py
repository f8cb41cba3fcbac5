"""Analytic data-path exchange budget of one frame-sharded denoising step.
Bytes a rank RECEIVES per step for the temporal-attention and adapter exchanges, old (all-gather) vs new
(frame<->pixel all-to-all, two-frame halo).  python tools/comm_budget.py [frames] [latent_side]"""
import sys

f = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = [S * S >> (2 * i) for i in range(4)]
UNET_T = [(320, N[0])] * 2 + [(640, N[1])] * 2 + [(1280, N[2])] * 2 + [(1280, N[3])] + [(1280, N[2])] * 3 + [(640, N[1])] * 3 + [(320, N[0])] * 3
ADAPT = [(320, N[0])] * 3 + [(320, N[1])] + [(640, N[1])] * 2 + [(640, N[2])] + [(1280, N[2])] * 2 + [(1280, N[3])] * 3
ATTN1 = UNET_T   # one spatial attn1 per transformer block, same (C, N)

print(f"{f} frames, {S}x{S} latents; MB received per rank and step")
print(f"{'layout':<22}{'fl':>4}{'attn1 halo':>12}{'temporal AG':>13}{'temporal A2A':>14}{'adapter AG':>12}{'adapter halo':>14}{'tconv halo':>12}{'total old':>11}{'total new':>11}")
for name, R, B in (("cfg2 x frames2 (4)", 2, 2), ("cfg2 x frames4 (8)", 4, 2), ("frames4 (4)", 4, 4), ("frames8 (8)", 8, 4)):
    fl = f // R
    nb = B // 2                       # edit rows on the rank (adapter temporal attention / motion residual batch)
    halo1 = sum(B * n * 2 * c * 2 for c, n in ATTN1)
    t_ag = (R - 1) * (sum(B * fl * n * 2 * c * 2 for c, n in UNET_T) + sum(nb * fl * n * 2 * c * 2 for c, n in ADAPT))
    t_a2a = (R - 1) / R * (sum(B * fl * n * 2 * c * 2 for c, n in UNET_T) + sum(nb * fl * n * 2 * c * 2 for c, n in ADAPT))   # in (C) + out (C)
    a_ag = (R - 1) * sum(fl * n * 2 * c * 2 for c, n in ADAPT)
    a_halo = sum(2 * n * 2 * c * 2 for c, n in ADAPT) if fl % 8 else 0   # <= two remote frames (a rank that starts on a chunk boundary: none)
    # TemporalConv k=3 halos: 4 per resnet (2 convs x 2 neighbours), 22 resnets + adapter 12 blocks x 2 -- measured by bench; rough: rows of one frame
    MB = 1e-6
    print(f"{name:<22}{fl:>4}{halo1*MB:>12.0f}{t_ag*MB:>13.0f}{t_a2a*MB:>14.0f}{a_ag*MB:>12.0f}{a_halo*MB:>14.0f}{'':>12}{(halo1+t_ag+a_ag)*MB:>11.0f}{(halo1+t_a2a+a_halo)*MB:>11.0f}")
