"""fp16 calibration of the parity tolerances ("to be calibrated against PyTorch fp16 eager of the restatement").

TEST TOOLING.  Runs the oracle (oracle/ref_cpu.py) on ONE two-branch denoising step twice -- fp32, and with every weight and input cast to
fp16 (torch eager, fp16 storage between ops) -- and prints the rel-L2 distance of the ControlNet residuals, the CFG-amplified noise
prediction and the updated latents.  That distance is what "an fp16 implementation of the same arithmetic" costs; the HIP path (fp16
storage, fp32 accumulation and softmax / normalisation statistics in fp32 or fp64) has to sit at or below it.
    python tools/calibrate_fp16.py [--frames 8 --latent 8] [--device cpu|cuda]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--latent", type=int, default=8)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--step", type=int, default=4)
    a = ap.parse_args()
    from motioneditor_amd import synth
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs(f=a.frames, h=a.latent, w=a.latent)
    usd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.unet_schema()).items()}
    csd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.").items()}
    f = a.frames
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * a.latent, 8 * a.latent)
    ddim = ref_cpu.DDIM()
    t = ddim.timesteps[a.step]
    outs = {}
    for name, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        dev = a.device if name == "fp16" else "cpu"
        c = lambda v: v.to(dev, dt)   # noqa: E731
        sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
        sp.cur_step = tp.cur_step = a.step
        taps = {}
        t0 = time.time()
        with torch.no_grad():
            lat = ref_cpu.denoise_step({k: c(v) for k, v in usd.items()}, {k: c(v) for k, v in csd.items()}, ddim, c(x["latents"]), t, c(x["uncond"]), c(x["cond"]),
                                       c(images), sp, tp, 7.5, taps=taps)
        outs[name] = dict(latents=lat.float().cpu(), noise_pred=taps["noise_pred"].float().cpu(), cn_down0=taps["cn_down"][0].float().cpu(),
                          cn_down11=taps["cn_down"][11].float().cpu(), skip11=taps["skips"][11].float().cpu())
        print(f"{name}: {time.time() - t0:.1f} s", flush=True)
    rel = lambda k: float((outs["fp16"][k] - outs["fp32"][k]).norm() / outs["fp32"][k].norm())   # noqa: E731
    print(f"fp16 eager vs fp32 oracle, {a.frames} frames x {a.latent}x{a.latent} latents, step {a.step} (editors active):")
    for k in ("cn_down0", "cn_down11", "skip11", "noise_pred", "latents"):
        print(f"  rel-L2 {k:12s} {rel(k):.3e}")


if __name__ == "__main__":
    main()
