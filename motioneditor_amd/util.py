"""DDIM inversion, the step in front of the denoising loop (reference `motion_editor/util.py:77-130`, called from
`inference.py:289-293` with `normal_infer=True`).  Same signatures; the prompt may be given as ready text embeddings
(`text_embeddings=[1,77,768]`) when the pipeline carries no text encoder.  The null-text optimisation that can follow it
(`p2p/null_text_optimization.py`) needs autograd through the UNet and is out of scope."""
from __future__ import annotations

import math
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


# ---------------------------------------------------------------------------------------------------------------------
# Null-text optimisation (p2p/null_text_optimization.py:133-166; inference.py:277-287 runs it before the editing loop)
# ---------------------------------------------------------------------------------------------------------------------
def null_optimization(pipeline, ddim_scheduler, latents, context: torch.Tensor, null_inner_steps: int = 10, epsilon: float = 1e-5,
                      num_ddim_steps: Optional[int] = None, guidance_scale: float = 7.5, grads: Optional[list] = None) -> List[torch.Tensor]:
    """MyNullInversion.null_optimization: for each DDIM step a fresh Adam (lr 1e-2 (1 - i / 100)) moves the unconditional text
    embedding so that the guided prev_step of the current latent reproduces the inversion latent one step earlier; early stop at
    loss < epsilon + 2e-5 i; the latent then advances with the optimised embedding.  latents = the DDIM inversion trajectory
    (util.ddim_inversion), context = [uncond, cond] (2, 77, 768).  Returns the list of optimised [1, 77, 768] embeddings
    (what the pipeline takes as `uncond_embeddings`).

    The reference differentiates with torch autograd; here the forward is the same launch graph as everywhere else and the
    gradient comes from motioneditor_amd.autodiff (a tape over the C-ABI operators and their backward primitives).  As in the
    reference (`:49-51` hard-codes it) the UNet runs with normal_infer=False -- sparse-causal attn1 -- and without editors."""
    from . import autodiff
    from .models import graph
    unet = pipeline.unet
    dev = unet.device
    n = len(ddim_scheduler.timesteps) if num_ddim_steps is None else num_ddim_steps
    uncond, cond = context.to(dev).float().chunk(2)
    out: List[torch.Tensor] = []
    latent_cur = latents[-1].to(dev).float()

    def eps_of(lat, emb, tape_on=False):
        # the rows the UNet projects to K / V: handed over as OUR allocation, so that the gradient store can be asked for it
        rows = emb.to(unet.P.dtype).reshape(-1, emb.shape[-1]).contiguous().clone()
        if not tape_on:
            return graph.unet_forward(unet.P, lat, float(t), rows), None, None
        with autodiff.record(graph) as tape:
            act = graph.unet_forward(unet.P, lat, float(t), rows)
        return act, tape, rows

    B, _, f, h, w = latent_cur.shape
    for i in range(n):
        uncond = uncond.clone().detach().requires_grad_(True)
        opt = torch.optim.Adam([uncond], lr=1e-2 * (1.0 - i / 100.0))
        latent_prev = latents[len(latents) - i - 2].to(dev).float()
        t = ddim_scheduler.timesteps[i]
        ca, cb = ddim_scheduler.coeffs(int(t))
        eps_c = graph.ops.rows_to_nchw5(eps_of(latent_cur, cond)[0].t, B, 4, f, h, w).float()
        for _ in range(null_inner_steps):
            act, tape, text = eps_of(latent_cur, uncond.detach(), tape_on=True)
            eps_u = graph.ops.rows_to_nchw5(act.t, B, 4, f, h, w).float()
            rec = ca * latent_cur + cb * (eps_u + guidance_scale * (eps_c - eps_u))          # prev_step (:26-36)
            diff = rec - latent_prev
            loss = float((diff * diff).mean())
            # d loss / d eps_u, back in the row layout of the UNet output: [(b f h w), 4]
            d_eps = (2.0 / diff.numel()) * diff * (cb * (1.0 - guidance_scale))
            d_rows = d_eps.permute(0, 2, 3, 4, 1).reshape(-1, 4)
            # loss scaling: the backward kernels carry gradients between layers in fp16 (like every activation); a power of two
            # brings the seed's largest element to ~64 and is divided out of the result
            amax = float(d_rows.abs().max())
            ls = 2.0 ** math.floor(math.log2(64.0 / amax)) if amax > 0.0 else 1.0
            G = autodiff.backward(tape, [(act.t, d_rows * ls)])
            g = (G.view(text).reshape(uncond.shape) / ls).to(uncond.dtype)
            if grads is not None:
                grads.append(g.clone())
            opt.zero_grad()
            uncond.grad = g
            opt.step()
            del tape, G
            if loss < epsilon + i * 2e-5:
                break
        out.append(uncond[:1].detach().clone())
        both = graph.unet_forward(unet.P, torch.cat([latent_cur] * 2), float(t), torch.cat([uncond.detach(), cond]))
        e2 = graph.ops.rows_to_nchw5(both.t, 2 * B, 4, f, h, w).float()
        eu, ec = e2.chunk(2)
        latent_cur = ca * latent_cur + cb * (eu + guidance_scale * (ec - eu))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Adapter training step, the arithmetic of train_adaptor.py:364-368
# ---------------------------------------------------------------------------------------------------------------------
def adapter_training_grads(unet, noisy_latents: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, down_block_res_samples, mid_block_res_sample,
                           target: torch.Tensor, prefix: str = "controlnet_adapter."):
    """loss = mse(unet(noisy, t, ehs, down_block_additional_residuals, mid_block_additional_residual), target) on ONE clip and its gradient
    w.r.t. every parameter under `prefix`, keyed by the reference's parameter names and in the reference's layouts -- what
    `accelerator.backward(loss)` leaves in `.grad` of the adapter (the reference trains nothing else: train_adaptor.py freezes the
    rest).  Residuals in the reference layout [b, C, f, h', w'] (ControlNet outputs, no gradient).  The forward is the ordinary
    launch graph on an autodiff tape; the caller owns the optimiser step (AdamW + clip_grad_norm in the reference) and, across
    GPUs, the gradient all-reduce.  Parameter-gradient kernels do not exist yet: on a GPU the primitives raise."""
    from . import autodiff
    from .models import graph
    P = unet.P
    dev = unet.device
    B, _, f, h, w = noisy_latents.shape
    rows = lambda r: graph.ops.nchw5_to_rows(r.to(dev))   # noqa: E731
    down = [rows(r) for r in down_block_res_samples]
    mid = rows(mid_block_res_sample)
    ehs = encoder_hidden_states.to(dev).to(P.dtype).reshape(-1, encoder_hidden_states.shape[-1]).contiguous().clone()
    t = float(timestep.item() if torch.is_tensor(timestep) else timestep)
    with autodiff.record(graph) as tape:
        act = graph.unet_forward(P, noisy_latents.to(dev), t, ehs, down_res=down, mid_res=mid, two_branch=False)
    pred = graph.ops.rows_to_nchw5(act.t, B, 4, f, h, w).float()
    diff = pred - target.to(dev).float()
    loss = float((diff * diff).mean())
    d_rows = ((2.0 / diff.numel()) * diff).permute(0, 2, 3, 4, 1).reshape(-1, 4)
    amax = float(d_rows.abs().max())
    ls = 2.0 ** math.floor(math.log2(64.0 / amax)) if amax > 0.0 else 1.0     # loss scaling, as in null_optimization
    G = autodiff.backward(tape, [(act.t, d_rows * ls)], trainable=P.trainable_ids(prefix))
    grads = {}
    for key, g in G.params.items():
        for name, gn in P.unpack_grad(key, g / ls).items():
            grads[name] = grads[name] + gn if name in grads else gn
    return loss, grads


class AdapterTrainer:
    """One optimisation step of the content-aware motion adapter as train_adaptor.py:364-385 takes it: loss and gradients from
    `adapter_training_grads`, gradient averaging over the data-parallel ranks (what accelerate's DDP does for the reference; one flat
    bucket, RCCL-friendly), `clip_grad_norm_(max_grad_norm)` over the adapter's parameters, AdamW on fp32 master copies (the
    reference's defaults: lr 3e-5, betas (0.9, 0.999), weight decay 1e-2, eps 1e-8), and the updated parameters written back into
    the UNet's weight store so that the next forward packs them afresh.  The loss is gathered for logging as `:377` does."""

    def __init__(self, unet, lr: float = 3e-5, betas=(0.9, 0.999), weight_decay: float = 1e-2, eps: float = 1e-8, max_grad_norm: float = 1.0,
                 prefix: str = "controlnet_adapter.", group=None):
        self.unet, self.prefix, self.group, self.max_grad_norm = unet, prefix, group, max_grad_norm
        P = unet.P
        P.state = dict(P.state)          # a private, mutable weight store: updates must not leak into the caller's mapping
        self.names = sorted(k[len(P.prefix):] for k in P.state if k.startswith(P.prefix + prefix))
        self.master = {n: torch.nn.Parameter(P.raw(n).clone()) for n in self.names}
        self.opt = torch.optim.AdamW(list(self.master.values()), lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)

    def step(self, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target) -> float:
        import torch.distributed as dist
        loss, grads = adapter_training_grads(self.unet, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target, self.prefix)
        missing = [n for n in self.names if n not in grads]
        if missing:
            raise RuntimeError(f"no gradient reached {missing[:3]} ...")
        flat = torch.cat([grads[n].reshape(-1).float().cpu() for n in self.names])
        loss_t = torch.tensor([loss], dtype=torch.float64)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            world = dist.get_world_size(self.group)
            dev = self.unet.device if str(dist.get_backend(self.group)) == "nccl" else torch.device("cpu")
            flat = flat.to(dev)
            dist.all_reduce(flat, group=self.group)                    # DP gradient average: ONE bucket of every adapter gradient
            flat = (flat / world).cpu()
            loss_t = loss_t.to(dev)
            dist.all_reduce(loss_t, group=self.group)                  # train_adaptor.py:377 (accelerator.gather(loss).mean())
            loss_t = (loss_t / world).cpu()
        total = float(flat.norm())
        clip = min(1.0, self.max_grad_norm / (total + 1e-6))            # torch.nn.utils.clip_grad_norm_
        o = 0
        for n in self.names:
            p = self.master[n]
            p.grad = (flat[o:o + p.numel()] * clip).reshape(p.shape).clone()
            o += p.numel()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        P = self.unet.P
        for n in self.names:
            P.update(n, self.master[n].detach())
        return float(loss_t[0])
