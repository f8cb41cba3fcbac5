"""One optimisation step of the content-aware motion adapter as the reference takes it (train_adaptor.py:318-385), with synthetic
tensors in place of the dataset, CLIP and the checkpoints (none exist offline):

    pixel_values [1, f, 3, H, W] --vae.encode(...).latent_dist.sample() * 0.18215--> latents [1, 4, f, h, w]     (:325-330)
    noise ~ N(0, 1); t ~ U{0..999}; noisy = sqrt(a_t) latents + sqrt(1 - a_t) noise  (DDPMScheduler.add_noise)   (:333-337)
    ControlNet(noisy "(b f) c h w", t, ehs.repeat(f), skeleton) -> 12 down + 1 mid residual, "(b f) .. -> b c f .." (:346-362)
    model_pred = unet(noisy, t, ehs, down_block_additional_residuals, mid_block_additional_residual).sample      (:364)
    loss = mse(model_pred, noise); backward; clip_grad_norm_(1.0); AdamW.step()  -- only controlnet_adapter.* trains (:365-384)

    python examples/train_adapter.py [--frames 8 --size 128 --steps 2]
Across GPUs: run under torchrun; util.AdapterTrainer averages the adapter gradients over the ranks in one all-reduced bucket.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def alphas_cumprod() -> torch.Tensor:   # SD-1.5 scheduler config: scaled-linear betas 0.00085 -> 0.012 over 1000 steps
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def training_batch(f: int, H: int, W: int, seed: int = 7) -> dict:
    from motioneditor_amd import synth
    T = torch.from_numpy
    return dict(pixel_values=T(np.tanh(synth.synth_normal("train.pixels", (1, f, 3, H, W), seed)).astype(np.float32)),
                skeleton=T(np.clip(synth.synth_normal("train.skel", (1, f, 3, H, W), seed, 0.5) + 0.5, 0, 1).astype(np.float32)),
                ehs=T(synth.synth_normal("train.ehs", (1, 77, 768), seed, 0.3)),
                encode_noise=T(synth.synth_normal("train.vae_noise", (f, 4, H // 8, W // 8), seed)),
                noise=T(synth.synth_normal("train.noise", (1, 4, f, H // 8, W // 8), seed)))


def step(trainer, vae, controlnet, batch: dict, t: int) -> float:
    """train_adaptor.py:318-385 for one clip."""
    pv = batch["pixel_values"]
    f, H, W = pv.shape[1], pv.shape[3], pv.shape[4]
    h, w = H // 8, W // 8
    lat = vae.encode(pv.reshape(f, 3, H, W)).latent_dist.sample(noise=batch["encode_noise"])                    # (:325-326)
    lat = lat.reshape(1, f, 4, h, w).permute(0, 2, 1, 3, 4).contiguous().float().cpu() * 0.18215                  # (:328-330)
    a = float(alphas_cumprod()[t])
    noise = batch["noise"]
    noisy = a ** 0.5 * lat + (1.0 - a) ** 0.5 * noise                                                             # (:337)
    images = batch["skeleton"].reshape(f, 3, H, W)                                                                 # prepare_image + "b f c h w -> (b f) c h w" (:348-358)
    cn_in = noisy.permute(0, 2, 1, 3, 4).reshape(f, 4, h, w)                                                       # (:359-360)
    down, mid = controlnet(cn_in, t, batch["ehs"].repeat(f, 1, 1), images, conditioning_scale=1.0)                # (:361-368)
    down = [d.reshape(1, f, d.shape[1], d.shape[2], d.shape[3]).permute(0, 2, 1, 3, 4).contiguous() for d in down]  # (:369)
    mid = mid.reshape(1, f, 1280, mid.shape[2], mid.shape[3]).permute(0, 2, 1, 3, 4).contiguous()                  # (:370)
    return trainer.step(noisy, t, batch["ehs"], down, mid, target=noise)                                          # (:372-384)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--lr", type=float, default=3e-5)
    args = ap.parse_args()
    from motioneditor_amd import util
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.models.vae import AutoencoderKL
    dev = "cuda"
    vae, unet, cn = AutoencoderKL.from_synthetic(dev), UNet2DConditionModel.from_synthetic(dev), ControlNetModel.from_synthetic(dev)
    trainer = util.AdapterTrainer(unet, lr=args.lr)
    import time
    g = torch.Generator().manual_seed(0)
    for i in range(args.steps):
        t = int(torch.randint(0, 1000, (1,), generator=g))                                                         # (:335)
        batch = training_batch(args.frames, args.size, args.size, seed=7 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step(trainer, vae, cn, batch, t)
        torch.cuda.synchronize()
        print(f"step {i}: t = {t}, loss = {loss:.5f}   ({(time.perf_counter() - t0) * 1e3:.0f} ms: VAE encode, ControlNet, UNet + adapter forward / backward, clip, AdamW)")


if __name__ == "__main__":
    main()
