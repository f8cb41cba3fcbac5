"""Drop-in for diffusers==0.15.1 ``ControlNetModel`` (sd-controlnet-openpose) as the reference calls it
(pipeline_motion_editor.py:618-625): 2-D, every frame an independent image."""
from __future__ import annotations

from typing import Mapping

import torch

from .. import ops, synth
from ..weights import Packed
from . import graph
from .compat import ModuleShims


class ControlNetModel(ModuleShims):
    def __init__(self, state_dict: Mapping[str, object], device="cuda", dtype=torch.float16):
        self.P = Packed(state_dict, device, dtype=dtype)
        self.device = torch.device(device)
        self.dtype = torch.float16
        missing = [k for k in synth.controlnet_schema() if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} ControlNet keys, e.g. {missing[:3]}")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", torch_dtype=None, **kwargs):
        from .. import checkpoint
        return cls(checkpoint.load_controlnet_state_dict(pretrained_model_name_or_path, subfolder), device)

    @classmethod
    def from_synthetic(cls, device="cuda", seed: int = 33):
        return cls(synth.synth_state_dict(synth.controlnet_schema(), seed, salt="controlnet."), device)

    def forward_rows(self, latents, lat_index, timestep, prompt, cond, conditioning_scale: float = 1.0, row_offset: int = 0):
        """Fast path used by the pipeline: latents fp32 [nb,4,f,h,w], ControlNet batch entry i reads
        latents[lat_index[i]]; prompt [n_text,77,768] interleaved over rows (row r -> text r % n_text)."""
        t = float(timestep.item() if torch.is_tensor(timestep) else timestep)
        return graph.controlnet_forward(self.P, latents.to(self.device), list(lat_index), t, prompt.to(self.device), cond.to(self.device), conditioning_scale, row_offset)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale: float = 1.0, return_dict: bool = False):
        """diffusers signature: sample [(b f),4,h,w], encoder_hidden_states [(b f),77,768], controlnet_cond [(b f),3,8h,8w]
        -> (12 down residuals [(b f),C,h',w'], mid residual), fp32."""
        n, _, h, w = sample.shape
        lat = sample.to(self.device).float().reshape(n, 4, 1, h, w)
        down, mid = self.forward_rows(lat, list(range(n)), timestep, encoder_hidden_states, controlnet_cond, conditioning_scale)
        sizes = [(h, w)] * 3 + [(h // 2, w // 2)] * 3 + [(h // 4, w // 4)] * 3 + [(h // 8, w // 8)] * 3   # conv_in, 2 resnets, then each downsampler + 2 resnets
        outs = [ops.rows_to_nchw(d, n, d.shape[1], hh * ww).reshape(n, d.shape[1], hh, ww) for d, (hh, ww) in zip(down, sizes)]
        m = ops.rows_to_nchw(mid, n, 1280, (h // 8) * (w // 8)).reshape(n, 1280, h // 8, w // 8)
        return outs, m

    __call__ = forward
