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


# ---- predicted ms / step per rank: to be checked against the first SCALE_rNN.json the driver can produce -------------------------
# Inputs: the 1-GPU kernel-family table of a bench run (profiles/r04_bench_c3.json, event pass, single stream) and the exchange budget above.
# Model:  t(N) = compute(N) + bytes(N) / (links x LINK_GBS) + n_exchanges x LAT_US          (exchanges are NOT overlapped in the eager path)
#   compute(N): the per-rank share of the batch-4 step.  The ControlNet and the classifier-free-guidance prefix are computed once per step on one GPU
#   (dedup); under a CFG split every pair member computes both again, under frame sharding they shard with the frames.  Small grids lose efficiency:
#   EFF[batch rows per rank] from the measured secondary workloads (8 f x 32^2: 9.3 TFLOP in 19.9 ms = 0.47 PF/s vs 0.91 at config 3).
#   links: a 2-rank exchange uses ONE xGMI link (153 GB/s per direction, MI355X guide: 7 links x ~153 GB/s per GPU); an all-to-all among R ranks R - 1 links.
def predict(one_gpu_ms=152.0, cn_ms=11.5, prefix_ms=3.8, link_gbs=153.0, lat_us=35.0):   # round 5: 145.7 ... 155.2 ms/step on seven boxes, 152 in the middle
    rows = []
    base = one_gpu_ms - cn_ms - prefix_ms          # the part that splits along the CFG axis
    n_ops = {1: 0, 2: 185, 4: 185, 8: 185}         # exchanges per step and rank in a frame-sharded step: 16 halos + 2 x 28 all-to-all + 12 + 56 + 45
    for n, mode, cfg, R, mb_recv, tconv_mb in ((1, "single", 1, 1, 0, 0), (2, "cfg", 2, 1, 1.5, 0), (4, "cfg2 x frames2", 2, 2, 859, 219), (8, "cfg2 x frames4", 2, 4, 681, 219),
                                                 (4, "frames4", 1, 4, 1308, 412), (8, "frames8", 1, 8, 862, 412)):
        eff = {1: 1.0, 2: 0.97, 4: 0.93, 8: 0.85}[cfg * R]          # grid-size efficiency of the per-rank sub-problem
        comp = (base / cfg + (cn_ms + prefix_ms) * (1.0 if cfg == 2 else 1.0)) / R / eff if cfg * R > 1 else one_gpu_ms
        links = max(R - 1, 1)
        comm = (mb_recv / (links * link_gbs) + tconv_mb / link_gbs) if R > 1 else (mb_recv / link_gbs)
        lat = (n_ops[R * cfg] if R > 1 else (1 if cfg == 2 else 0)) * lat_us * 1e-3
        rows.append((n, mode, comp, comm, lat, comp + comm + lat, one_gpu_ms / (comp + comm + lat)))
    print(f"\npredicted ms / step per rank from a {one_gpu_ms} ms one-GPU step (ControlNet {cn_ms} ms, shared CFG prefix {prefix_ms} ms), {link_gbs} GB/s per xGMI link, {lat_us} us per exchange:")
    print(f"{'GPUs':>4} {'mode':<18}{'compute':>9}{'bytes':>8}{'latency':>9}{'total':>8}{'speed-up':>10}")
    for n, mode, comp, comm, lat, tot, sp in rows:
        print(f"{n:>4} {mode:<18}{comp:>9.1f}{comm:>8.1f}{lat:>9.1f}{tot:>8.1f}{sp:>10.2f}")


if f == 24 and S == 64:
    predict()
