"""DDIM inversion, the step in front of the denoising loop (reference `motion_editor/util.py:77-130`, called from
`inference.py:289-293` with `normal_infer=True`).  Same signatures; the prompt may be given as ready text embeddings
(`text_embeddings=[1,77,768]`) when the pipeline carries no text encoder.  Also here: the null-text optimisation that follows it
(`p2p/null_text_optimization.py:133-166`) and the adapter training step (`train_adaptor.py:364-385`), both on the reverse-mode tape of
`motioneditor_amd/autodiff.py` with loss, gradient norm and the Adam / AdamW update as device kernels (csrc/train.hip)."""
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
def _loss_scale(amax: float) -> float:
    """The backward kernels carry gradients between layers in fp16 (like every activation): a power of two brings the seed's largest
    element to ~64; it is divided out again by the optimiser kernel (grad_scale)."""
    return 2.0 ** math.floor(math.log2(64.0 / amax)) if amax > 0.0 and math.isfinite(amax) else 1.0


def null_optimization(pipeline, ddim_scheduler, latents, context: torch.Tensor, null_inner_steps: int = 10, epsilon: float = 1e-5,
                      num_ddim_steps: Optional[int] = None, guidance_scale: float = 7.5, grads: Optional[list] = None) -> List[torch.Tensor]:
    """MyNullInversion.null_optimization: for each DDIM step a fresh Adam (lr 1e-2 (1 - i / 100)) moves the unconditional text
    embedding so that the guided prev_step of the current latent reproduces the inversion latent one step earlier; early stop at
    loss < epsilon + 2e-5 i; the latent then advances with the optimised embedding.  latents = the DDIM inversion trajectory
    (util.ddim_inversion), context = [uncond, cond] (2, 77, 768).  Returns the list of optimised [1, 77, 768] embeddings
    (what the pipeline takes as `uncond_embeddings`).

    The reference differentiates with torch autograd; here the forward is the same launch graph as everywhere else, the gradient
    comes from motioneditor_amd.autodiff (a tape over the C-ABI operators and their backward kernels) and the loss (me_mse_seed +
    me_sumsq_absmax), the loss-scale selection and Adam (me_adamw on the fp32 embedding) are device kernels too: the host reads two
    floats per inner step (the loss for the early stop, the seed's largest element for the loss scale).  As in the reference
    (`:49-51` hard-codes it) the UNet runs with normal_infer=False -- sparse-causal attn1 -- and without editors."""
    from . import autodiff
    from .models import graph
    B_ = graph.ops
    unet = pipeline.unet
    P = unet.P
    dev = unet.device
    n = len(ddim_scheduler.timesteps) if num_ddim_steps is None else num_ddim_steps
    uncond0, cond = context.to(dev).float().chunk(2)
    cond_rows = graph.text_rows(cond, P.dtype)
    out: List[torch.Tensor] = []
    latent_cur = latents[-1].to(dev).float().contiguous()
    nel = latent_cur.numel()
    uncond = uncond0.reshape(-1, uncond0.shape[-1]).contiguous().clone()          # fp32 master [77, 768]; Adam updates it in place

    def text_of(u32):   # the rows the UNet projects to K / V: OUR allocation, so that the gradient store can be asked for it
        rows = torch.empty(u32.shape, dtype=P.dtype, device=dev)
        return B_.cast_f16(rows, u32)

    for i in range(n):
        m_, v_ = torch.zeros_like(uncond), torch.zeros_like(uncond)              # a fresh Adam per DDIM step (:141)
        lr = 1e-2 * (1.0 - i / 100.0)
        latent_prev = latents[len(latents) - i - 2].to(dev).float().contiguous()
        t = ddim_scheduler.timesteps[i]
        ca, cb = ddim_scheduler.coeffs(int(t))
        eps_c = graph.unet_forward(P, latent_cur, float(t), cond_rows).t          # rows [(f h w), 4]
        for k in range(null_inner_steps):
            text = text_of(uncond)
            with autodiff.record(graph) as tape:
                act = graph.unet_forward(P, latent_cur, float(t), text)
            # rec = prev_step(eps_u + g (eps_c - eps_u)) (:26-36), loss = mse(rec, latent_prev), d loss / d eps_u in the UNet's row layout
            diff, d_rows = B_.mse_seed(act.t, latent_prev, eps_c=eps_c, x=latent_cur, guidance=guidance_scale, ca=ca, cb=cb,
                                       coef=(2.0 / nel) * cb * (1.0 - guidance_scale))
            st = torch.cat([B_.sumsq_absmax(diff), B_.sumsq_absmax(d_rows)]).tolist()     # one host read: loss, seed magnitude
            loss = st[0] / nel
            ls = _loss_scale(st[3])
            G = autodiff.backward(tape, [(act.t, d_rows)], seed_scale=ls, wrt=[text])   # only what lies downstream of the text rows is walked
            g = G.view(text)
            if grads is not None:      # test hook (not part of the reference): the un-scaled gradient of this inner step
                grads.append((g / ls).reshape(uncond0.shape).clone())
            B_.adamw(uncond, m_, v_, g.contiguous(), lr=lr, step=k + 1, grad_scale=1.0 / ls)
            del tape, G
            if loss < epsilon + i * 2e-5:
                break
        out.append(uncond.reshape(uncond0.shape).clone())
        both = graph.unet_forward(P, torch.cat([latent_cur] * 2), float(t), torch.cat([text_of(uncond), cond_rows]), cfg_dup=True)   # the two rows are copies up to the text
        latent_cur = B_.cfg_ddim(latent_cur, both.t, guidance=guidance_scale, ca=ca, cb=cb)       # get_noise_pred + prev_step (:53-65)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Adapter training step, the arithmetic of train_adaptor.py:364-368
# ---------------------------------------------------------------------------------------------------------------------
def _adapter_backward(unet, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target, prefix, param_buffers=None,
                      sync_amax=None):
    """Forward on the tape, loss, backward.  -> (loss, loss scale, gradient store); the parameter gradients (packed layouts, times the loss
    scale) are accumulated into `param_buffers[key]` when given.  sync_amax: callable(tensor [2]) that makes the seed magnitude -- and with it
    the loss scale -- the same on every data-parallel rank (a MAX all-reduce), so that the gradient buckets can be summed."""
    from . import autodiff
    from .models import graph
    B_ = graph.ops
    P = unet.P
    dev = unet.device
    rows = lambda r: r if r.dim() == 2 else B_.nchw5_to_rows(r.to(dev))   # noqa: E731
    down = [rows(r) for r in down_block_res_samples]
    mid = rows(mid_block_res_sample)
    ehs = graph.text_rows(encoder_hidden_states.to(dev), P.dtype).clone()
    t = float(timestep.item() if torch.is_tensor(timestep) else timestep)
    with autodiff.record(graph) as tape:
        act = graph.unet_forward(P, noisy_latents.to(dev), t, ehs, down_res=down, mid_res=mid, two_branch=False)
    tgt = target.to(dev).float().contiguous()
    diff, d_rows = B_.mse_seed(act.t, tgt, coef=2.0 / tgt.numel())          # loss = mse(model_pred, target) (:368)
    amax = B_.sumsq_absmax(d_rows)
    if sync_amax is not None:
        sync_amax(amax)
    st = torch.cat([B_.sumsq_absmax(diff), amax]).tolist()
    loss, ls = st[0] / tgt.numel(), _loss_scale(st[3])
    G = autodiff.backward(tape, [(act.t, d_rows)], trainable=P.trainable_ids(prefix), seed_scale=ls, param_buffers=param_buffers)
    return loss, ls, G


def adapter_training_grads(unet, noisy_latents: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, down_block_res_samples, mid_block_res_sample,
                           target: torch.Tensor, prefix: str = "controlnet_adapter."):
    """loss = mse(unet(noisy, t, ehs, down_block_additional_residuals, mid_block_additional_residual), target) on ONE clip and its gradient
    w.r.t. every parameter under `prefix`, keyed by the reference's parameter names and in the reference's layouts (host tensors) -- what
    `accelerator.backward(loss)` leaves in `.grad` of the adapter (the reference trains nothing else: train_adaptor.py freezes the
    rest).  Residuals in the reference layout [b, C, f, h', w'] (ControlNet outputs, no gradient).  The forward is the ordinary
    launch graph on an autodiff tape.  This is the inspection / export form; the training step itself (AdapterTrainer) keeps the gradients
    on the device in the packed layouts."""
    loss, ls, G = _adapter_backward(unet, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target, prefix)
    grads = {}
    for key, g in G.params.items():
        for name, gn in unet.P.unpack_grad(key, g).items():    # host copy in the reference layout
            gn = gn / ls
            grads[name] = grads[name] + gn if name in grads else gn
    return loss, grads


class AdapterTrainer:
    """One optimisation step of the content-aware motion adapter as train_adaptor.py:364-385 takes it, on the device: loss and gradients
    from the tape (every adapter parameter gradient accumulates straight into ONE flat fp32 bucket, in the packed layouts the kernels
    read), the data-parallel gradient average as one all-reduce of that bucket (RCCL on the GPU; what accelerate's DDP does for the
    reference), `clip_grad_norm_(max_grad_norm)` from a device reduction of the bucket, AdamW (the reference's defaults: lr 3e-5, betas
    (0.9, 0.999), weight decay 1e-2, eps 1e-8) on fp32 masters kept in the same packed layout, and the fp16 weights the forward reads
    refreshed IN PLACE from the masters by one cast kernel -- no gradient, master or optimiser state ever visits the host.  The loss is
    averaged over the ranks for logging as `:377` does.  `export_state_dict()` returns the trained parameters under the reference's names
    and layouts (what train_adaptor.py saves as the adapter checkpoint)."""

    def __init__(self, unet, lr: float = 3e-5, betas=(0.9, 0.999), weight_decay: float = 1e-2, eps: float = 1e-8, max_grad_norm: float = 1.0,
                 prefix: str = "controlnet_adapter.", group=None):
        from .models import graph
        self.unet, self.prefix, self.group, self.max_grad_norm = unet, prefix, group, max_grad_norm
        self.lr, self.betas, self.weight_decay, self.eps = lr, betas, weight_decay, eps
        self.steps = 0
        self.skipped_steps = 0           # optimisation steps dropped because the gradient norm was not finite (fp16 overflow)
        P = unet.P
        P.make_private()                 # a private, mutable weight store
        self.names = sorted(k[len(P.prefix):] for k in P.state if k.startswith(P.prefix + prefix))
        self.keys = graph.adapter_pack(P, prefix)
        covered = sorted(n for k in self.keys for n in k.partition(":")[2].split("|"))
        if covered != self.names:
            raise RuntimeError("adapter_pack does not cover the adapter's parameters exactly once")
        dev = unet.device
        sizes = [P.cache[k].numel() for k in self.keys]
        total = (sum(sizes) + 3) // 4 * 4
        self.off = {}
        o = 0
        for k, n in zip(self.keys, sizes):
            self.off[k] = (o, n)
            o += n
        self.master = torch.zeros(total, dtype=torch.float32, device=dev)           # fp32 masters, packed layouts
        self.m, self.v, self.grad = torch.zeros_like(self.master), torch.zeros_like(self.master), torch.zeros_like(self.master)
        self.weights = torch.zeros(total, dtype=P.dtype, device=dev)                # what the forward reads (fp16 on the GPU)
        self.views = {}
        for k in self.keys:
            o, n = self.off[k]
            self.master[o:o + n].copy_(P.packed_f32(k).reshape(-1))                # exact fp32 values, not the fp16-rounded packing
            self.views[k] = P.rehome(k, self.weights[o:o + n])
        self.param_buffers = {k: self.grad[self.off[k][0]:self.off[k][0] + self.off[k][1]].view(self.views[k].shape) for k in self.keys}

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def step(self, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target) -> float:
        import torch.distributed as dist
        from .models import graph
        B_ = graph.ops
        world = self._world()
        dist_on = dist.is_available() and dist.is_initialized()
        self.grad.zero_()
        sync = None
        if dist_on:
            sync = lambda a: dist.all_reduce(a, op=dist.ReduceOp.MAX, group=self.group)   # noqa: E731  (one loss scale for every rank's bucket)
        loss, ls, G = _adapter_backward(self.unet, noisy_latents, timestep, encoder_hidden_states, down_block_res_samples, mid_block_res_sample, target, self.prefix,
                                        param_buffers=self.param_buffers, sync_amax=sync)
        stray = [k for k in G.params if k not in self.param_buffers]
        if stray:
            raise RuntimeError(f"a gradient reached packed tensors the trainer does not own: {stray[:3]}")
        if dist_on:
            dist.all_reduce(self.grad, group=self.group)                 # DP gradient sum: ONE bucket of every adapter gradient (averaged by grad_scale below)
            lt = torch.tensor([loss], dtype=torch.float32, device=self.grad.device)
            dist.all_reduce(lt, group=self.group)                        # train_adaptor.py:377 (accelerator.gather(loss).mean())
            loss = float(lt[0]) / world
        gn = B_.sumsq_absmax(self.grad)                                   # device scalar: sum of squares of the (scaled, summed) bucket
        if not bool(torch.isfinite(gn.reshape(-1)[0])):
            # an fp16 overflow somewhere downstream of the seed (the loss scale is chosen from the seed's magnitude): an inf / NaN gradient would
            # poison the masters and Adam's moments for good.  Skip the update, as accelerate's GradScaler does for the reference's fp16 runs.
            self.skipped_steps += 1
            return loss
        self.steps += 1
        B_.adamw(self.master, self.m, self.v, self.grad, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay,
                 step=self.steps, gnorm_sq=gn, max_grad_norm=self.max_grad_norm, grad_scale=1.0 / (ls * world))
        B_.cast_f16(self.weights, self.master)                            # the packed weights of the next forward, in place
        B_.invalidate_transposed([v for v in self.views.values() if v.dim() == 3])
        return loss

    def export_state_dict(self):
        """{reference parameter name: fp32 host tensor in the reference layout} of the trained adapter parameters."""
        P = self.unet.P
        out = {}
        for k in self.keys:
            o, n = self.off[k]
            out.update(P.unpack_grad(k, self.master[o:o + n].view(self.views[k].shape)))
        for name, val in out.items():    # the model's own state follows the training: state_dict() / named_parameters() / .to(device) see the trained values
            P.state[P.prefix + name] = val.detach().cpu().clone()
        return out
