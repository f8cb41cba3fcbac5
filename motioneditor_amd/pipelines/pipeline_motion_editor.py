"""Drop-in for the reference ``motion_editor.pipelines.pipeline_motion_editor.MotionEditorPipeline``
(constructor :70-91, ``__call__`` :505-666).  The denoising loop (:597-654) -- ControlNet on the edit
rows, one batch-4 UNet3D forward with the content-aware motion adapter and both attention editors,
classifier-free guidance and the DDIM update -- runs entirely on libmotioned HIP kernels.

CLIP text encoding is out of scope: with no ``text_encoder`` the prompt embeddings are
passed as ``text_embeddings=[2,77,768]`` (an extension keyword swallowed by the reference's ``**kwargs``).
The VAE (``models/vae.py``) is optional: with no ``vae`` use ``output_type="latent"``.  DDIM inversion and
null-text optimisation live in ``motioneditor_amd/util.py``.
"""
from __future__ import annotations

from dataclasses import dataclass
import collections
import weakref
from typing import Callable, List, Optional, Union

import torch

from .. import ops, plan
from ..schedulers import DDIMScheduler


_LIVE_PIPELINES = weakref.WeakSet()   # every pipeline of this process: release_plans() of one must not free scratch another's recorded steps point into


@dataclass
class MotionEditorPipelineOutput:
    images: torch.Tensor


class MotionEditorPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, safety_checker=None, controlnet=None,
                 feature_extractor=None):
        if unet is None:
            raise ValueError("unet is required")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.controlnet = unet, controlnet
        if scheduler is None:
            scheduler = DDIMScheduler()
        elif not isinstance(scheduler, DDIMScheduler):
            # a diffusers DDIMScheduler as inference.py:187-197 passes it: adopt its configuration (clip_sample is forced off,
            # reference :108-119); schedulers this path cannot reproduce are rejected here, not deep inside the loop
            if not hasattr(scheduler, "config"):
                raise TypeError(f"scheduler must be a DDIMScheduler or carry a DDIM `.config`, got {type(scheduler).__name__}")
            scheduler = DDIMScheduler.from_config(scheduler.config)
        self.scheduler = scheduler
        if getattr(self.scheduler.config, "clip_sample", False):  # reference forces clip_sample False (:108-119)
            self.scheduler.config["clip_sample"] = False
        if getattr(self.scheduler.config, "steps_offset", 1) != 1:  # ... and steps_offset 1 on the PIPELINE's scheduler whatever the config says (:94-105)
            self.scheduler.config["steps_offset"] = 1
            if self.scheduler.num_inference_steps:
                self.scheduler.set_timesteps(self.scheduler.num_inference_steps)
        self.vae_scale_factor = 8
        self.device = unet.device
        # The reference feeds ControlNet rows [1, 3] of cat([latents]*2) -- the SAME edit latent twice -- with prompts
        # tiled as row r -> text r % 2 (pipeline :613-621).  For an even frame count both batch entries therefore see
        # identical latents, images and prompt pattern and produce identical residuals: compute one, use it twice.
        # Set False to execute the redundant second entry exactly as the reference does.
        self.dedup_controlnet = True
        # `torch.cat([latents] * 2)` (:605) makes the conditional half of the UNet batch a copy of the unconditional half until the text enters
        # (the cross-attention of the first transformer block): that prefix is computed once (graph.unet_forward, cfg_dup).  Exact -- the
        # step's output is bitwise the same; False executes the duplicated prefix as the reference does.
        self.dedup_cfg_prefix = True
        # ControlNet feeds the UNet only after its down path (the adapter consumes the residuals, unet_2d_condition.py:477-494):
        # run it on a second HIP stream beside the UNet's down blocks; its small grids (24-frame batch) fill CUs the
        # UNet's tile tails leave idle.  The adapter then runs on the same side stream beside the mid block (tiny grids),
        # and the up path waits on an event.
        self.overlap_controlnet = True
        self.overlap_adapter = True     # needs overlap_controlnet (same side stream, so the residuals are already ordered)
        self.shard_overlap = False      # the same two-stream overlap inside the frame-sharded step (denoise_step_frame_sharded; bench.py --shard-overlap)
        self._side_stream = None
        self.side_stream_priority = 0   # HIP priority of the side stream (-1 = high); A/B switch, measured in DESIGN.md section 3.1
        # denoise_step_graphed / _planned: (shapes, conditioning tensor, editor gating) -> captured / recorded step.  Every entry pins one step's
        # activations (a private memory pool: GBs at 24 f x 64^2), so both tables are LRU-bounded (`max_cached_steps`; two gatings -- editors inactive /
        # active -- of one clip are the working set of a 50-step run) and `release_plans()` returns everything.
        self._graphs = collections.OrderedDict()
        self._plans = collections.OrderedDict()
        _LIVE_PIPELINES.add(self)
        self.max_cached_steps = 4
        # who issues the launches of a step inside __call__'s loop: "eager" = this Python process, launch by launch (denoise_step);
        # "plan" = one me_denoise_step call per step on CUDA inputs (denoise_step_planned: bitwise the eager result, ~4 x less host time)
        self.step_executor = "eager"

    @property
    def _execution_device(self):
        return self.device

    def _cache_get(self, table, key):
        ent = table.get(key)
        if ent is not None:
            table.move_to_end(key)
        return ent

    def _cache_put(self, table, key, ent):
        """Insert as most recent; evict the least recently used entries beyond `max_cached_steps` (their plan handle / graph and private pool are released)."""
        table[key] = ent
        while len(table) > max(1, int(self.max_cached_steps)):
            _, old = table.popitem(last=False)
            self._release_entry(old)
        return ent

    @staticmethod
    def _release_entry(ent):
        pl = ent.pop("plan", None)
        if pl is not None:
            pl.close()
        ent.clear()          # graph, static inputs / outputs, the conditioning tensor and its embedding: dropped with their pool

    def release_plans(self):
        """Drop every recorded launch plan and captured graph (and the memory pools they pin).  The next planned / graphed step records again."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()     # a replay still in flight reads the pools dropped below (round-5 advisor finding)
        for table in (self._plans, self._graphs):
            while table:
                _, old = table.popitem(last=False)
                self._release_entry(old)
        # the retired split-K scratch blocks are process-global: another pipeline's recorded / captured steps may hold their addresses -- trim only when
        # no pipeline of this process has a live plan or graph left (round-5 advisor finding)
        if not any(p_._plans or p_._graphs for p_ in _LIVE_PIPELINES):
            ops.scratch_trim()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    def enable_vae_slicing(self):        # reference pipeline :93-94 (inference.py:197); the VAE here decodes per image already
        if self.vae is not None and hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def disable_vae_slicing(self):       # reference pipeline :96-97
        if self.vae is not None and hasattr(self.vae, "disable_slicing"):
            self.vae.disable_slicing()

    def to(self, device=None, *a, **k):
        """diffusers DiffusionPipeline.to: move every component that can move."""
        for m in (self.unet, self.controlnet, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to") and device is not None:
                m.to(device)
        if device is not None:
            self.device = torch.device(device)
        return self

    # ---- reference :374-387 ----
    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    # ---- reference :389-416 ----
    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            gdev = lambda g: g.device if g is not None else "cpu"  # noqa: E731  (the reference draws on a CUDA generator, inference.py:272)
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=generator[i], device=gdev(generator[i]), dtype=torch.float32) for i in range(batch_size)], dim=0)
            else:
                latents = torch.randn(shape, generator=generator, device=gdev(generator), dtype=torch.float32)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
        return (latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma).contiguous()

    # ---- reference :418-459: a tensor, a PIL image, or a list of either ----
    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype, do_classifier_free_guidance):
        if not isinstance(image, torch.Tensor):
            import numpy as np
            import PIL.Image
            if isinstance(image, PIL.Image.Image):
                image = [image]
            if isinstance(image[0], PIL.Image.Image):     # RGB, Lanczos resize to (width, height), [0, 1], NCHW (:423-440)
                lanczos = PIL.Image.Resampling.LANCZOS if hasattr(PIL.Image, "Resampling") else PIL.Image.LANCZOS
                arr = np.stack([np.asarray(im.convert("RGB").resize((width, height), resample=lanczos)) for im in image])
                image = torch.from_numpy((arr.astype(np.float32) / 255.0).transpose(0, 3, 1, 2).copy())
            elif isinstance(image[0], torch.Tensor):
                image = torch.cat(list(image), dim=0)
            else:
                raise TypeError(f"prepare_image: unsupported image type {type(image[0])}")
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        image = image.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)
        if do_classifier_free_guidance:
            image = torch.cat([image] * 2)
        return image

    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_cfg, negative_prompt, text_embeddings=None, negative_text_embeddings=None):
        """Reference :270-333: with classifier-free guidance and no per-step `uncond_embeddings`, the result is
        cat([uncond, text]) = 2 * len(prompt) rows.  Without a text encoder the embeddings come in as tensors:
        text_embeddings [len(prompt),77,768] (+ negative_text_embeddings [1 or len(prompt),77,768]), or text_embeddings
        that already hold all 2 * len(prompt) rows."""
        n = 1 if isinstance(prompt, str) else len(prompt)
        if text_embeddings is not None:
            cond = text_embeddings.to(device)
            if not do_cfg or cond.shape[0] == 2 * n:
                return cond
            if negative_text_embeddings is None:
                raise ValueError("classifier-free guidance without `uncond_embeddings` needs the unconditional rows: pass "
                                 "negative_text_embeddings=[1 or len(prompt),77,768] (the encoding of the negative / empty prompt) or "
                                 f"text_embeddings with {2 * n} rows = cat([uncond, text]) (reference :296-333)")
            unc = negative_text_embeddings.to(device)
            return torch.cat([unc.expand(cond.shape[0], -1, -1) if unc.shape[0] == 1 else unc, cond])
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError("no text_encoder/tokenizer: pass text_embeddings=[len(prompt),77,768] (CLIP encoding is out of scope)")

        def enc(texts):
            ids = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt").input_ids
            return self.text_encoder(ids.to(self.text_encoder.device))[0].to(device)

        cond = enc(prompt)
        if not do_cfg:
            return cond
        if negative_prompt is None:
            neg = [""] * n
        elif isinstance(negative_prompt, str):
            neg = [negative_prompt] * n
        else:
            if len(negative_prompt) != n:
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {n}")   # reference :309-314
            neg = list(negative_prompt)
        return torch.cat([enc(neg), cond])

    def decode_latents(self, latents):
        if self.vae is None:
            raise ValueError("no vae: call with output_type='latent', or construct the pipeline with motioneditor_amd.models.vae.AutoencoderKL")
        b, c, f, h, w = latents.shape
        x = (1 / 0.18215) * latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = self.vae.decode(x).sample
        video = video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        return ((video / 2 + 0.5).clamp(0, 1)).cpu().float().numpy()

    @torch.no_grad()
    def denoise_step_frame_sharded(self, latents: torch.Tensor, t: int, text_embeddings_input: torch.Tensor, images: Optional[torch.Tensor],
                                   guidance_scale: float, shard, controlnet_conditioning_scale: float = 1.0, cfg_group=None) -> torch.Tensor:
        """The same step with the FRAME axis sharded over the ranks of `shard` (parallel.FrameShard; SURVEY.md 8e, BASELINE
        config 4).  latents fp32 [2,4,f_loc,h,w] and images [2*f_loc,3,H,W] (or [f_loc,...]) hold this rank's frames only.
        Exchanges per layer: one-frame K|V halo for attn1, all-gather of K|V for adapter sparse-causal / temporal attention,
        one-frame halos for the temporal convolutions, all-reduce of the 5-D GroupNorm statistics.  ControlNet, CFG and DDIM
        are rank-local.

        cfg_group (a 2-rank process group of the ranks that hold the SAME frames): hybrid CFG x frame sharding -- this rank
        additionally runs only one classifier-free-guidance half (rank 0 of the pair: the unconditional (recon, edit) rows,
        rank 1: the conditional ones), which halves every frame-shard exchange (batch 2 instead of 4); the pair trades its
        4-channel noise predictions with one all-gather before the fused CFG + DDIM update."""
        if shard.f_total % 2:
            raise NotImplementedError("frame sharding relies on the ControlNet batch-entry identity, which needs an even frame count")
        f = latents.shape[2]
        assert f == shard.f_loc, (f, shard.f_loc)
        r = 0
        if cfg_group is None:
            x = torch.cat([latents] * 2)
            emb = text_embeddings_input
        else:
            from .. import parallel
            cx = parallel.exchange(cfg_group)             # a torch process group, or an exchange adapter (direct RCCL for graph capture)
            assert cx.world == 2, "CFG parallelism is a 2-way split"
            r = cx.rank
            x = latents                                   # both CFG halves see the same [recon, edit] latents (:605)
            emb = text_embeddings_input[2 * r:2 * r + 2]
        down = mid = ready = None
        two = False
        # shard_overlap (round 6, bench.py --shard-overlap; off by default until a multi-GPU run has measured it): the ControlNet -- rank-local, no exchange --
        # runs on the side stream beside the UNet's down path, and the adapter follows it there when the shard has a second communicator (FrameShard.side_shard)
        side_on = bool(self.shard_overlap and x.is_cuda and not torch.cuda.is_current_stream_capturing())
        if self.controlnet is not None and images is not None:
            prompt = text_embeddings_input[1::2]
            img = images[:f] if images.shape[0] == 2 * f else images

            def run_controlnet():
                # row of the full "(b f)" ControlNet batch = entry * f_total + global frame, and it reads prompt row % 2
                # (pipeline :615,621): this rank's first row is entry r (its CFG half), global frame frame0
                return self.controlnet.forward_rows(x, [1], t, prompt, img, controlnet_conditioning_scale, row_offset=r * shard.f_total + shard.frame0)

            if side_on:
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(priority=self.side_stream_priority)
                main = torch.cuda.current_stream()
                plan.wait_stream(self._side_stream, main)
                with torch.cuda.stream(self._side_stream):
                    down, mid = run_controlnet()
                ready = plan.record_event(self._side_stream)
                for r_ in list(down) + [mid]:
                    plan.share(r_, main)
            else:
                down, mid = run_controlnet()
            two = True
        eps = self.unet.forward_rows(x, t, emb, down, mid, two, shard=shard, res_ready=ready,
                                     side_stream=self._side_stream if (ready is not None and getattr(shard, "side_shard", None) is not None) else None,
                                     cfg_dup=self.dedup_cfg_prefix and cfg_group is None).t
        if cfg_group is not None:
            both = torch.empty((2 * eps.shape[0], eps.shape[1]), dtype=eps.dtype, device=eps.device)
            parallel._count("all_gather(noise prediction, CFG pair)", eps)
            cx.all_gather_into(both, eps.contiguous())   # rows: [uncond (rec, edit) | cond (rec, edit)]
            eps = both
        ca, cb = self.scheduler.coeffs(int(t))
        return ops.cfg_ddim(latents, eps, guidance=guidance_scale, ca=ca, cb=cb)

    @torch.no_grad()
    def denoise_step_cfg_parallel(self, latents: torch.Tensor, t: int, text_embeddings_input: torch.Tensor, images: Optional[torch.Tensor],
                                  guidance_scale: float, group=None, controlnet_conditioning_scale: float = 1.0) -> torch.Tensor:
        """The same step split over a 2-rank process group along the classifier-free-guidance axis: rank 0 runs the
        unconditional (recon, edit) pair, rank 1 the conditional pair.  Nothing couples the two halves inside ControlNet /
        UNet / adapter / editors (GroupNorm statistics, K/V injection and the adapter all stay inside a pair), so the only
        exchange is ONE all-gather of the 4-channel noise prediction (RCCL over xGMI on MI355X; 2 x [2,4,f,h,w] fp16) before
        the fused CFG + DDIM update, which every rank then applies to its own copy of the latents."""
        from .. import parallel
        cx = parallel.exchange(group)                      # a torch process group, or an exchange adapter (direct RCCL for graph capture)
        r = cx.rank
        assert cx.world == 2, "CFG parallelism is a 2-way split"
        f = latents.shape[2]
        x2 = latents                                                       # both CFG halves see the same [recon, edit] latents (:605)
        emb = text_embeddings_input[2 * r:2 * r + 2]
        down = mid = None
        two = False
        if self.controlnet is not None and images is not None:
            prompt = text_embeddings_input[1::2]                         # the interleave r % 2 needs BOTH prompts on every rank (:615,621)
            nimg = images.shape[0] // 2
            down, mid = self.controlnet.forward_rows(x2, [1], t, prompt, images[r * nimg:(r + 1) * nimg], controlnet_conditioning_scale, row_offset=r * f)
            two = True
        eps = self.unet.forward_rows(x2, t, emb, down, mid, two).t       # [(2 f N), 4]
        both = torch.empty((2 * eps.shape[0], eps.shape[1]), dtype=eps.dtype, device=eps.device)
        parallel._count("all_gather(noise prediction, CFG pair)", eps)
        cx.all_gather_into(both, eps.contiguous())   # rows: [uncond (rec, edit) | cond (rec, edit)]
        ca, cb = self.scheduler.coeffs(int(t))
        return ops.cfg_ddim(latents, both, guidance=guidance_scale, ca=ca, cb=cb)

    @torch.no_grad()
    def denoise_step(self, latents: torch.Tensor, t: int, text_embeddings_input: torch.Tensor, images: Optional[torch.Tensor],
                     guidance_scale: float, controlnet_conditioning_scale: float = 1.0, taps: Optional[dict] = None) -> torch.Tensor:
        """One iteration of the reference loop body (:603-648).  latents fp32 [2,4,f,h,w] = [recon, edit];
        text_embeddings_input [4,77,768] = [uncond, uncond, cond_recon, cond_edit]."""
        nb = latents.shape[0]
        # :605 (scale_model_input is the identity for DDIM); on the GPU the duplication is two library copies, so that a recorded step holds no torch kernel
        native = latents.is_cuda and getattr(ops, "NATIVE", False) and latents.dtype == torch.float32 and latents.is_contiguous()
        if not native:
            plan.torch_fallback("torch.cat([latents] * 2)")
        x4 = ops.repeat_batch(latents, 2) if native else torch.cat([latents] * 2)
        down = mid = ready = None
        two = False
        if self.controlnet is not None and images is not None:
            prompt = text_embeddings_input[1::2]                         # :615; .repeat(f,1,1) on "(b f)" rows -> row r reads r % 2 (:621)
            f = latents.shape[2]

            def run_controlnet():
                if self.dedup_controlnet and f % 2 == 0:
                    # one entry; the UNet graph broadcasts it to both edit rows (and shares the adapter's x-only half)
                    return self.controlnet.forward_rows(x4, [1], t, prompt, images[:images.shape[0] // 2], controlnet_conditioning_scale)
                return self.controlnet.forward_rows(x4, [1, 3], t, prompt, images, controlnet_conditioning_scale)   # :613-625

            if self.overlap_controlnet and x4.is_cuda and taps is None:
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(priority=self.side_stream_priority)
                main = torch.cuda.current_stream()
                plan.wait_stream(self._side_stream, main)
                with torch.cuda.stream(self._side_stream):
                    down, mid = run_controlnet()
                ready = plan.record_event(self._side_stream)
                for r in list(down) + [mid]:
                    plan.share(r, main)
            else:
                down, mid = run_controlnet()
            two = True                                                     # mid residual scattered as [0, m0, 0, m1] (:628-629)
            if taps is not None:
                taps["cn_down"], taps["cn_mid"] = [d.clone() for d in down], mid.clone()
        eps = self.unet.forward_rows(x4, t, text_embeddings_input, down, mid, two, taps, res_ready=ready,
                                     side_stream=self._side_stream if (ready is not None and self.overlap_adapter) else None,
                                     cfg_dup=self.dedup_cfg_prefix and nb * 2 == x4.shape[0])   # :632-640
        if taps is not None:
            taps["eps_rows"] = eps.t.clone()
        ca, cb = self.scheduler.coeffs(int(t))
        return ops.cfg_ddim(latents, eps.t, guidance=guidance_scale, ca=ca, cb=cb)         # :643-648

    # ---- hipGraph replay of the step -------------------------------------------------------------------------------------
    def _editor_gate(self):
        """What the editors will do during the coming step: a pure function of their step counters (fully_control.py:434,
        temporal_control.py:74), so two captured graphs (editors inactive / active) cover a whole run."""
        sed, ted = self.unet.spatial_editor, self.unet.temporal_editor
        return (None if sed is None else sed.cur_step in sed.step_idx, None if ted is None else ted.cur_step in ted.step_idx)

    @torch.no_grad()
    def denoise_step_graphed(self, latents: torch.Tensor, t: int, text_embeddings_input: torch.Tensor, images: Optional[torch.Tensor],
                             guidance_scale: float, controlnet_conditioning_scale: float = 1.0, *, shard=None, cfg_group=None, cfg_parallel_group=None) -> torch.Tensor:
        """A denoising step with its ~1100 kernel launches (both HIP streams, and -- in the sharded modes -- every RCCL exchange) captured ONCE
        into a hipGraph and replayed: shapes and key-segment tables are static, the editors' gating is a function of the step index, and the
        per-step scalars (timestep, guidance, DDIM coefficients) live in device memory (ops.STEP_PARAMS), so a replay needs four host writes
        and one launch.  The first call per (mode, shape, gating) runs the step eagerly (warm-up: allocations, tables, function attributes,
        communicator set-up) and captures it; editors' counters advance exactly as in the eager step.

        Which step is captured:  plain `denoise_step` by default; `shard=` (+ optional `cfg_group=`) -> `denoise_step_frame_sharded`;
        `cfg_parallel_group=` -> `denoise_step_cfg_parallel`.  Collectives are captured as graph nodes (torch.distributed's NCCL/RCCL
        process group enqueues on its streams, which fork from and join the capturing stream); every rank of the groups involved must
        capture and replay in step."""
        from .. import parallel
        sed, ted = self.unet.spatial_editor, self.unet.temporal_editor
        if shard is not None:
            mode = ("frames", id(shard), None if cfg_group is None else id(cfg_group))
            step = lambda lat, emb: self.denoise_step_frame_sharded(lat, t, emb, images, guidance_scale, shard, controlnet_conditioning_scale, cfg_group=cfg_group)   # noqa: E731
        elif cfg_parallel_group is not None:
            mode = ("cfg", id(cfg_parallel_group))
            step = lambda lat, emb: self.denoise_step_cfg_parallel(lat, t, emb, images, guidance_scale, group=cfg_parallel_group,   # noqa: E731
                                                                   controlnet_conditioning_scale=controlnet_conditioning_scale)
        else:
            mode = ("single",)
            step = lambda lat, emb: self.denoise_step(lat, t, emb, images, guidance_scale, controlnet_conditioning_scale)   # noqa: E731
        # the conditioning images enter the key with their address AND version: the ControlNet's conditioning embedding is computed once per such
        # tensor and the captured graph reads that result -- an in-place rewrite of `images` is a new tensor as far as a replay is concerned
        key = (mode, tuple(latents.shape), tuple(text_embeddings_input.shape), None if images is None else (tuple(images.shape), images.data_ptr(), images._version),
               self._editor_gate(), float(controlnet_conditioning_scale), self.dedup_controlnet, self.dedup_cfg_prefix, self.overlap_controlnet, self.overlap_adapter)
        ca, cb = self.scheduler.coeffs(int(t))
        host = torch.tensor([float(t), float(guidance_scale), ca, cb], dtype=torch.float32)
        ent = self._cache_get(self._graphs, key)
        if ent is None:
            editors = [e for e in (sed, ted) if e is not None]
            counters = [(e.cur_step, e.cur_att_layer) for e in editors]

            def rewind():
                for e, (cs, cl) in zip(editors, counters):
                    e.cur_step, e.cur_att_layer = cs, cl

            st = dict(lat=latents.clone(), emb=text_embeddings_input.clone(), params=host.to(latents.device))
            before = {k: list(v) for k, v in parallel.STATS.items()}
            try:
                step(st["lat"], st["emb"])                      # warm-up, eager
                torch.cuda.synchronize()
                rewind()
                before = {k: list(v) for k, v in parallel.STATS.items()}
                g = torch.cuda.CUDAGraph()
                ops.STEP_PARAMS = st["params"]
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    st["out"] = step(st["lat"], st["emb"])
            except BaseException:
                rewind()        # a failed warm-up / capture must leave the editors where the caller had them: its fallback step is THIS step
                raise
            finally:
                ops.STEP_PARAMS = None
            # the exchanges of the captured step, for bench.py's `comm` accounting: added again at every replay
            st["comm"] = {k: [v[0] - before.get(k, [0, 0])[0], v[1] - before.get(k, [0, 0])[1]] for k, v in parallel.STATS.items()}
            for k, v in st["comm"].items():     # the capture pass itself moved nothing
                parallel.STATS[k][0] -= v[0]
                parallel.STATS[k][1] -= v[1]
            st["graph"] = g
            st["images"] = images      # keep the captured conditioning tensor alive
            if self.controlnet is not None:   # ... and the conditioning embeddings the graph may have baked in (graph.controlnet_forward's table may drop them later)
                st["cond_embed"] = dict(self.controlnet.P.cache.get("cond_embed", {}))
            rewind()
            ent = self._cache_put(self._graphs, key, st)
        ent["lat"].copy_(latents)
        ent["emb"].copy_(text_embeddings_input)
        ent["params"].copy_(host, non_blocking=False)
        ent["graph"].replay()
        for k, v in ent["comm"].items():
            c = parallel.STATS.setdefault(k, [0, 0])
            c[0] += v[0]
            c[1] += v[1]
        for e in (sed, ted):
            if e is not None:      # what MutualAttentionBase.__call__ does over the step's attention layers
                e.cur_att_layer = 0
                e.cur_step += 1
                e.after_step()
        return ent["out"].clone()

    # ---- the step as one C call: me_plan_* / me_denoise_step ----------------------------------------------------------------
    @torch.no_grad()
    def denoise_step_planned(self, latents: torch.Tensor, t: int, text_embeddings_input: torch.Tensor, images: Optional[torch.Tensor],
                             guidance_scale: float, controlnet_conditioning_scale: float = 1.0) -> torch.Tensor:
        """`denoise_step` with its execution behind the C ABI (csrc/plan.hip, plan.StepPlan): the first call per (shapes, editor gating) runs the step
        eagerly once (warm-up: weight packing, tables, scratch, function attributes), records a second eager pass -- every kernel launch with its
        arguments, both HIP streams, their event dependencies -- and from then on one `me_denoise_step` call re-issues the ~1100 launches of the
        loop body (pipeline_motion_editor.py:603-648) on the live streams: same kernels, same arguments, same order, hence bit-for-bit the eager
        result, with the host's share of a step down from ~11 ms of Python dispatch to the C loop over the launch list.  The per-step scalars
        (timestep, guidance, DDIM coefficients) live in device memory (ops.STEP_PARAMS) exactly as for `denoise_step_graphed`; editors' counters
        advance as in the eager step.  Single-process steps only (the sharded steps' RCCL exchanges are not library launches)."""
        sed, ted = self.unet.spatial_editor, self.unet.temporal_editor
        if not latents.is_cuda:
            raise ValueError("denoise_step_planned: CUDA tensors only")
        latents = latents.contiguous().float()
        emb = text_embeddings_input.contiguous()
        key = (tuple(latents.shape), tuple(emb.shape), emb.dtype, None if images is None else (tuple(images.shape), images.data_ptr(), images._version),
               self._editor_gate(), float(controlnet_conditioning_scale), self.dedup_controlnet, self.dedup_cfg_prefix, self.overlap_controlnet, self.overlap_adapter)
        # (the caller's stream is NOT part of the key: me_denoise_step substitutes the live stream for the recorded main stream)
        ca, cb = self.scheduler.coeffs(int(t))
        ent = self._cache_get(self._plans, key)
        if ent is None:
            editors = [e for e in (sed, ted) if e is not None]
            counters = [(e.cur_step, e.cur_att_layer) for e in editors]

            def rewind():
                for e, (cs, cl) in zip(editors, counters):
                    e.cur_step, e.cur_att_layer = cs, cl

            st = dict(lat=latents.clone(), emb=emb.clone(), params=torch.tensor([float(t), float(guidance_scale), ca, cb], dtype=torch.float32).to(latents.device))
            step = lambda: self.denoise_step(st["lat"], t, st["emb"], images, guidance_scale, controlnet_conditioning_scale)   # noqa: E731
            ops.STEP_PARAMS = st["params"]
            pl = None
            try:
                step()                                   # warm-up, eager
                torch.cuda.synchronize()
                rewind()
                pl = plan.StepPlan()
                with pl.recording():
                    st["out"] = step()
                torch.cuda.synchronize()
                pl.bind(st["lat"], st["emb"], st["params"], st["out"])
            except BaseException:
                # the warm-up or the recording pass raised part-way through a step: the editors' (cur_step, cur_att_layer) counters have advanced by a
                # partial step.  Put them back where the caller had them -- whoever catches this and retries eagerly (bench.planned_or_eager,
                # run_edit) must compute THIS step, not one with shifted layer gating -- and drop the half-built plan with its pool.
                rewind()
                if pl is not None:
                    pl.close()
                raise
            finally:
                ops.STEP_PARAMS = None
            rewind()
            st["plan"] = pl
            st["images"] = images          # keep the conditioning tensor the recorded launches read alive ...
            if self.controlnet is not None:   # ... and the conditioning embedding computed from it (graph.controlnet_forward's table may drop it later)
                st["cond_embed"] = dict(self.controlnet.P.cache.get("cond_embed", {}))
            ent = self._cache_put(self._plans, key, st)
        out = ent["plan"].step(latents, emb, float(t), float(guidance_scale), ca, cb)
        for e in (sed, ted):
            if e is not None:      # what MutualAttentionBase.__call__ does over the step's attention layers
                e.cur_att_layer = 0
                e.cur_step += 1
                e.after_step()
        return out

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor",
                 return_dict: bool = True, callback: Optional[Callable] = None, callback_steps: Optional[int] = 1,
                 uncond_embeddings: torch.Tensor = None, null_uncond_ratio: float = 1.0, skeleton=None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, num_images_per_prompt: Optional[int] = 1,
                 source_masks=None, target_masks=None, rectangle_source_masks=None, background_latents=None, **kwargs):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        if not do_cfg or batch_size != 2:
            raise NotImplementedError("the two-branch hot path needs guidance_scale > 1 and prompts = [source, target] (inference.py:296-323)")
        with_uncond = do_cfg if uncond_embeddings is None else False
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, with_uncond, negative_prompt, kwargs.get("text_embeddings"),
                                              kwargs.get("negative_text_embeddings"))
        if video_length is not None and self.controlnet is not None and video_length % 8:
            raise ValueError(f"video_length must be a multiple of 8 with the motion adapter (controlnet_adapter.py:414 hard-codes chunks of 8 frames), got {video_length}")

        images = None
        if self.controlnet is not None:
            if skeleton is None:
                raise ValueError("skeleton is required with a ControlNet (pipeline :556)")
            target = torch.unsqueeze(skeleton[-1], dim=0)                   # :556
            images = self.prepare_image(target, width, height, 1 * num_images_per_prompt, num_images_per_prompt, device, torch.float32, do_cfg)
            images = images.reshape(-1, *images.shape[2:]).contiguous()    # "b f c h w -> (b f) c h w" (:570)

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, self.unet.in_channels, video_length, height, width,
                                       torch.float32, device, generator, latents)
        if uncond_embeddings is not None:
            assert len(timesteps) == 50 or len(uncond_embeddings) >= len(timesteps)   # :601-602 (start_time = 50)
        for i, t in enumerate(timesteps):
            if uncond_embeddings is not None:
                emb = torch.cat([uncond_embeddings[i].to(device).expand(*text_embeddings.shape), text_embeddings])   # :608-609
            else:
                emb = text_embeddings
            if emb.shape[0] != 2 * batch_size:
                raise ValueError(f"expected {2 * batch_size} text-embedding rows [uncond x {batch_size}, cond x {batch_size}], got {emb.shape[0]}")
            step = self.denoise_step_planned if (self.step_executor == "plan" and latents.is_cuda) else self.denoise_step
            latents = step(latents, t, emb, images, guidance_scale, 1.0)   # the loop hard-codes scale 1.0 (:616)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if output_type == "latent":
            out = latents
        else:
            out = self.decode_latents(latents)
            if output_type == "tensor":
                out = torch.from_numpy(out)
        return MotionEditorPipelineOutput(images=out) if return_dict else out
