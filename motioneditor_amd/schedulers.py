"""DDIM scheduler with the SD-1.5 ``scheduler_config.json`` values (the reference loads diffusers'
``DDIMScheduler`` from the checkpoint, inference.py:187-197; update formula pinned in-tree at
motion_editor/util.py:77-87 and p2p/null_text_optimization.py:26-36).  eta = 0, clip_sample False
(forced by pipeline_motion_editor.py:108-119), set_alpha_to_one False, steps_offset 1.
The per-step update is ``prev = ca(t) * x + cb(t) * eps`` and is applied by the fused
``me_cfg_ddim`` kernel; ``step`` is the API-compatible entry."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
import torch

from . import ops


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset)
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps: List[int] = []

    @classmethod
    def from_config(cls, config) -> "DDIMScheduler":
        """Adopt a foreign scheduler's configuration (diffusers' DDIMScheduler as inference.py:187-197 builds it: `.config` is a
        FrozenDict).  Keys the configuration does not carry take DIFFUSERS' OWN defaults (steps_offset 0, set_alpha_to_one True, linear
        betas 1e-4 .. 2e-2 -- not the SD-1.5 values this class's constructor defaults to), so that a foreign config means here what it means
        there; what this path cannot reproduce (anything but the epsilon-prediction, leading-spaced, eta = 0 DDIM with scaled-linear betas
        the reference runs) raises instead of being ignored."""
        get = (lambda k, d: config.get(k, d)) if hasattr(config, "get") else (lambda k, d: getattr(config, k, d))
        if get("prediction_type", "epsilon") != "epsilon":
            raise NotImplementedError(f"prediction_type {get('prediction_type', None)!r}")
        if get("trained_betas", None) is not None:
            raise NotImplementedError("trained_betas")
        for flag in ("thresholding", "rescale_betas_zero_snr"):
            if get(flag, False):
                raise NotImplementedError(f"{flag}=True")
        if get("timestep_spacing", "leading") != "leading":
            raise NotImplementedError(f"timestep_spacing {get('timestep_spacing', None)!r} (the reference's SD-1.5 scheduler is 'leading')")
        return cls(num_train_timesteps=get("num_train_timesteps", 1000), beta_start=get("beta_start", 0.0001), beta_end=get("beta_end", 0.02),
                   beta_schedule=get("beta_schedule", "linear"), clip_sample=False, set_alpha_to_one=get("set_alpha_to_one", True),
                   steps_offset=get("steps_offset", 0))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kwargs) -> "DDIMScheduler":
        """diffusers signature for a local directory (inference.py:192,198: DDIMScheduler.from_pretrained(path, subfolder="scheduler")):
        reads `scheduler_config.json`."""
        import json
        from pathlib import Path
        d = Path(pretrained_model_name_or_path) / subfolder if subfolder else Path(pretrained_model_name_or_path)
        f = d / "scheduler_config.json"
        if not f.exists():
            raise FileNotFoundError(f"{f} not found")
        return cls.from_config(json.loads(f.read_text()))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = [int(round(i * ratio)) + self.config.steps_offset for i in range(num_inference_steps)][::-1]

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coeffs(self, t: int) -> Tuple[float, float]:
        prev_t = int(t) - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[int(t)])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        ca = (a_p / a_t) ** 0.5
        cb = (1.0 - a_p) ** 0.5 - (a_p * (1.0 - a_t) / a_t) ** 0.5
        return ca, cb

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0, **kw) -> DDIMSchedulerOutput:
        """model_output / sample: fp32 [nb, C, f, h, w] on the GPU.  (API path; the pipeline uses the fused
        CFG+DDIM kernel directly on the channels-last noise prediction.)"""
        if eta != 0.0:
            raise NotImplementedError("eta != 0 (the reference runs eta = 0)")
        ca, cb = self.coeffs(int(timestep))
        nb, C, f, h, w = sample.shape
        # reuse the fused kernel with guidance 0 on a channels-last view of eps duplicated as [uncond | cond]
        rows = ops.nchw5_to_rows(model_output.float().contiguous())
        rows2 = torch.cat([rows, rows], dim=0)
        return DDIMSchedulerOutput(ops.cfg_ddim(sample.float().contiguous(), rows2, guidance=0.0, ca=ca, cb=cb))
