"""The reference harness sequence (inference.py:255-323) on motioneditor_amd, with synthetic tensors in place of the dataset,
CLIP and the checkpoints (none exist offline):

    pixels [1, f, 3, H, W]  --vae.encode(...).latent_dist.sample() * 0.18215-->  latents [1, 4, f, h, w]      (:260-265)
    ddim_inversion(pipeline, scheduler, latents, num_inv_steps, prompt="", normal_infer=True)[-1]               (:288-293)
    ddim_inv_latent.repeat(2, 1, 1, 1, 1)                                                                       (:296)
    prompts = [source prompt, target prompt]; skeleton = cat([0, target, 0, target])                            (:298-302)
    TemporalSelfAttentionControl(4, 10) + FullySelfAttentionControlMask(4, 10, source_masks=...) registered     (:307-313)
    pipeline(prompts, generator, latents=ddim_inv_latent, uncond_embeddings=None, skeleton=skeleton, ...)       (:315-323)
    sample_inv, sample_gen = sample.chunk(2)                                                                    (:326)

    python examples/run_edit.py [--frames 8 --size 128 --steps 10 --inv-steps 10]      (defaults finish in seconds on one MI355X)
    python examples/run_edit.py --frames 24 --size 512 --steps 50 --inv-steps 50        (the reference's case-1 geometry)
Without a text encoder the prompt embeddings enter as tensors (`text_embeddings`, `negative_text_embeddings`).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def harness_inputs(f: int, H: int, W: int, seed: int = 33) -> dict:
    """Synthetic stand-ins for one batch of the reference's dataset (data/dataset.py): seeded, layout and value ranges as
    inference.py consumes them."""
    from motioneditor_amd import synth
    T = torch.from_numpy
    return dict(pixel_values=T(np.tanh(synth.synth_normal("harness.pixels", (1, f, 3, H, W), seed)).astype(np.float32)),             # [-1, 1]
                target_skeleton=T(np.clip(synth.synth_normal("harness.skel", (1, f, 3, H, W), seed, 0.5) + 0.5, 0, 1).astype(np.float32)),  # openposefull / 255
                source_masks=T(synth.synth_masks(f, H, W)),                                                                              # [1, f, 1, H, W] in {0, 1}
                text_embeddings=T(synth.synth_normal("harness.cond", (2, 77, 768), seed, 0.3)),        # CLIP(source prompt), CLIP(target prompt)
                negative_text_embeddings=T(synth.synth_normal("harness.uncond", (1, 77, 768), seed, 0.3)),  # CLIP("")
                encode_noise=T(synth.synth_normal("harness.vae_noise", (f, 4, H // 8, W // 8), seed)))


def run(pipe, x: dict, *, steps: int, inv_steps: int, guidance: float = 7.5, output_type: str = "tensor", graphed: bool = False):
    """inference.py:259-326 for one (source prompt, target prompt) pair.  Returns (sample_inv, sample_gen, ddim_inv_latent)."""
    from motioneditor_amd import util
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
    from motioneditor_amd.schedulers import DDIMScheduler
    pv = x["pixel_values"]
    f, H, W = pv.shape[1], pv.shape[3], pv.shape[4]
    latents = pipe.vae.encode(pv.reshape(f, 3, H, W)).latent_dist.sample(noise=x["encode_noise"])       # "b f c h w -> (b f) c h w" (:261-262)
    latents = latents.reshape(1, f, 4, H // 8, W // 8).permute(0, 2, 1, 3, 4).contiguous() * 0.18215      # (:264-265)
    inv_sched = DDIMScheduler()
    inv_sched.set_timesteps(inv_steps)
    ddim_inv_latent = util.ddim_inversion(pipe, inv_sched, latents, inv_steps, prompt="", normal_infer=True,
                                          text_embeddings=x["negative_text_embeddings"])[-1]             # (:288-293; prompt "" = the empty-prompt embedding)
    ddim_inv_latent = ddim_inv_latent.repeat(2, 1, 1, 1, 1)                                                # (:296)
    tgt = x["target_skeleton"]
    skeleton = torch.cat([torch.zeros_like(tgt), tgt, torch.zeros_like(tgt), tgt], dim=0)                  # (:300-302)
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=x["source_masks"], target_masks=None, rectangle_source_masks=None)
    regiter_fully_attention_editor_diffusers(pipe, sed)
    sample = pipe(["a source prompt", "a target prompt"], video_length=f, height=H, width=W, num_inference_steps=steps, guidance_scale=guidance,
                  latents=ddim_inv_latent, uncond_embeddings=None, skeleton=skeleton, source_masks=None, target_masks=None,
                  rectangle_source_masks=None, background_latents=None, output_type=output_type,
                  text_embeddings=x["text_embeddings"], negative_text_embeddings=x["negative_text_embeddings"]).images
    assert sample.shape[0] == 2
    sample_inv, sample_gen = sample.chunk(2)
    return sample_inv, sample_gen, ddim_inv_latent


def build_pipeline(device: str = "cuda"):
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.models.vae import AutoencoderKL
    from motioneditor_amd.pipelines import MotionEditorPipeline
    return MotionEditorPipeline(vae=AutoencoderKL.from_synthetic(device), unet=UNet2DConditionModel.from_synthetic(device),
                                controlnet=ControlNetModel.from_synthetic(device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--inv-steps", type=int, default=10)
    ap.add_argument("--executor", choices=["eager", "plan"], default="plan",
                    help="who issues the ~1100 launches of a denoising step: 'plan' = one me_denoise_step call per step (the launch list is recorded at the first step of "
                         "each editor gating, csrc/plan.hip), 'eager' = Python, launch by launch; the results are bitwise the same")
    a = ap.parse_args()
    pipe = build_pipeline()
    pipe.step_executor = a.executor
    x = {k: v.cuda() for k, v in harness_inputs(a.frames, a.size, a.size).items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inv, gen, _ = run(pipe, x, steps=a.steps, inv_steps=a.inv_steps)
    torch.cuda.synchronize()
    print(f"{a.frames} frames {a.size}x{a.size}: encode + {a.inv_steps} inversion steps + {a.steps} denoising steps + decode in {time.perf_counter() - t0:.2f} s; "
          f"reconstruction {tuple(inv.shape)}, edit {tuple(gen.shape)}, range [{float(gen.min()):.3f}, {float(gen.max()):.3f}]")


if __name__ == "__main__":
    main()
