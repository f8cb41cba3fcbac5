"""cProfile of the host side of the denoising step (who spends the enqueue time): python tools/host_profile.py [steps]"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from motioneditor_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
f, h, w = 24, 64, 64
usd = synth.synth_state_dict(synth.unet_schema())
csd = synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")
x = bench.build_inputs(f, h, w)
dev = torch.device("cuda:0")
pipe, sed, ted = bench.make_pipeline(dev, usd, csd, x["masks"], None)
pipe.scheduler.set_timesteps(50)
lat = x["latents"].to(dev)
emb = torch.cat([x["uncond"][0].expand(2, 77, 768), x["cond"]]).to(dev)
img = torch.cat([x["skeleton"].reshape(f, 3, 8 * h, 8 * w)] * 2).to(dev)
sed.cur_step = ted.cur_step = 4


def step(i, lat):
    return pipe.denoise_step(lat, pipe.scheduler.timesteps[i], emb, img, 7.5)


for i in range(2):
    lat = step(4 + i, lat)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    lat = step(6 + i, lat)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
