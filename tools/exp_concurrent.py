"""Experiment: how much do two INDEPENDENT denoising steps overlap when enqueued on two HIP streams of one GPU?
(upper bound for pipelining the source and edit halves of one step against each other; results are not checked -- the two
pipelines share ops' scratch buffers, so this is a timing experiment only.)"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from motioneditor_amd import synth  # noqa: E402

dev = torch.device("cuda")
f, h, w = 24, 64, 64
usd = synth.synth_state_dict(synth.unet_schema())
csd = synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")
pipes = []
for c in range(2):
    x = bench.build_inputs(f, h, w, seed=33 + c)
    pipe, sed, ted = bench.make_pipeline(dev, usd, csd, x["masks"])
    pipe.overlap_controlnet = pipe.overlap_adapter = False
    sed.cur_step = ted.cur_step = 4
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * w).to(dev)
    lat = x["latents"].to(dev)
    cond = x["cond"].to(dev)
    unc = [u.to(dev) for u in x["uncond"]]
    pipes.append((pipe, images, lat, cond, unc))
ts = pipes[0][0].scheduler.timesteps


def step(c, i, lat):
    pipe, images, _, cond, unc = pipes[c]
    emb = torch.cat([unc[i].expand(2, 77, 768), cond])
    return pipe.denoise_step(lat, ts[i], emb, images, 7.5)


lats = [p[2] for p in pipes]
for c in range(2):
    for i in range(2):
        lats[c] = step(c, 4 + i, lats[c])
torch.cuda.synchronize()
N = 4
t0 = time.perf_counter()
for c in range(2):
    for i in range(N):
        lats[c] = step(c, 6 + i, lats[c])
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
t0 = time.perf_counter()
for i in range(N):
    for c in range(2):
        with torch.cuda.stream(streams[c]):
            lats[c] = step(c, 10 + i, lats[c])
torch.cuda.synchronize()
t_con = time.perf_counter() - t0
print(f"2 x {N} steps: one stream {t_seq * 1e3 / (2 * N):.1f} ms/step, two streams {t_con * 1e3 / (2 * N):.1f} ms/step  (ratio {t_con / t_seq:.3f})")
