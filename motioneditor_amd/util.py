"""DDIM inversion, the step in front of the denoising loop (reference `motion_editor/util.py:77-130`, called from
`inference.py:289-293` with `normal_infer=True`).  Same signatures; the prompt may be given as ready text embeddings
(`text_embeddings=[1,77,768]`) when the pipeline carries no text encoder.  The null-text optimisation that can follow it
(`p2p/null_text_optimization.py`) needs autograd through the UNet and is out of scope."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops


def next_coeffs(ddim_scheduler, timestep: int):
    """next_sample = ca * sample + cb * model_output  (util.py:77-87 collapsed to one linear update)."""
    n_train = ddim_scheduler.config.num_train_timesteps
    cur_t = min(int(timestep) - n_train // ddim_scheduler.num_inference_steps, 999)
    a_c = float(ddim_scheduler.alphas_cumprod[cur_t]) if cur_t >= 0 else float(ddim_scheduler.final_alpha_cumprod)
    a_n = float(ddim_scheduler.alphas_cumprod[int(timestep)])
    return (a_n / a_c) ** 0.5, (1 - a_n) ** 0.5 - (a_n * (1 - a_c) / a_c) ** 0.5


def next_step(model_output: torch.Tensor, timestep: int, sample: torch.Tensor, ddim_scheduler) -> torch.Tensor:
    """util.py:77-87 on reference-layout tensors [B,4,f,h,w] (fp32 latents, any float model output)."""
    ca, cb = next_coeffs(ddim_scheduler, timestep)
    return ca * sample + cb * model_output.to(sample.dtype)


def _context(pipeline, prompt, text_embeddings: Optional[torch.Tensor]) -> torch.Tensor:
    if text_embeddings is not None:
        return text_embeddings
    if getattr(pipeline, "text_encoder", None) is None:
        raise ValueError("ddim_loop needs text_embeddings= when the pipeline has no text encoder")
    tok = pipeline.tokenizer([prompt], padding="max_length", max_length=pipeline.tokenizer.model_max_length, truncation=True, return_tensors="pt")
    return pipeline.text_encoder(tok.input_ids.to(pipeline.device))[0]      # util.py:64-71 (conditional half of init_prompt)


@torch.no_grad()
def ddim_loop(pipeline, ddim_scheduler, latent: torch.Tensor, num_inv_steps: int, prompt: str = "", normal_infer: bool = False,
              text_embeddings: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """util.py:111-124: one single-branch UNet forward on the conditional embedding per step, walking the timesteps
    upwards.  The linear update runs in the fused `me_cfg_ddim` kernel (guidance 1 on a duplicated noise prediction
    selects it unchanged)."""
    cond = _context(pipeline, prompt, text_embeddings)
    unet = pipeline.unet
    latent = latent.to(unet.device, torch.float32).contiguous()
    all_latent = [latent]
    for i in range(num_inv_steps):
        t = ddim_scheduler.timesteps[len(ddim_scheduler.timesteps) - i - 1]
        ehs = cond if cond.shape[0] == latent.shape[0] else cond.repeat(latent.shape[0], 1, 1)   # util.py:91-93
        eps = unet.forward_rows(latent, t, ehs, normal_infer=normal_infer).t
        ca, cb = next_coeffs(ddim_scheduler, int(t))
        latent = ops.cfg_ddim(latent, torch.cat([eps, eps]), guidance=1.0, ca=ca, cb=cb)
        all_latent.append(latent)
    return all_latent


@torch.no_grad()
def ddim_inversion(pipeline, ddim_scheduler, video_latent: torch.Tensor, num_inv_steps: int, prompt: str = "", normal_infer: bool = False,
                   text_embeddings: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """util.py:127-130."""
    return ddim_loop(pipeline, ddim_scheduler, video_latent, num_inv_steps, prompt, normal_infer=normal_infer, text_embeddings=text_embeddings)
