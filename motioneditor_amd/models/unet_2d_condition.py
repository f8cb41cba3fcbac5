"""Drop-in for the reference ``motion_editor.models.unet_2d_condition.UNet2DConditionModel``
(forward :363-546) running on libmotioned kernels.  Same call signature / output container; weights
come from a state dict with the reference's key schema."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional

import torch

from .. import ops, synth
from ..weights import Packed
from . import graph
from .compat import ModuleShims, _SubModule


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, k):
        return self.sample if k in (0, "sample") else (_ for _ in ()).throw(KeyError(k))


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class UNet2DConditionModel(ModuleShims):
    num_spatial_attention_layers = 32   # 16 transformer blocks x {attn1, attn2} (fully_control_utils.py:220-229)
    num_temporal_attention_layers = 16  # (temporal_control_utils.py:134-144)

    def __init__(self, state_dict: Mapping[str, object], device="cuda", sample_size: int = 64, dtype=torch.float16):
        self.P = Packed(state_dict, device, dtype=dtype)
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.in_channels = 4
        self.config = _Cfg(sample_size=sample_size, in_channels=4, out_channels=4, cross_attention_dim=768, attention_head_dim=8,
                           block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, use_sc_attn=True, use_st_attn=False)
        self.spatial_editor = None
        self.temporal_editor = None
        self.controlnet_adapter = _SubModule(self, "controlnet_adapter.")   # inference.py:240: unet.controlnet_adapter.load_state_dict(...)
        missing = [k for k in synth.unet_schema() if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} UNet keys, e.g. {missing[:3]}")

    @classmethod
    def from_synthetic(cls, device="cuda", seed: int = 33):
        return cls(synth.synth_state_dict(synth.unet_schema(), seed), device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, use_sc_attn=True, use_st_attn=False, st_attn_idx=None,
                        resume_from_checkpoint=None, adapter_weight_path=None, device="cuda", **kwargs):
        """Reference signature (models/unet_2d_condition.py:548-796) for local directories: 2-D SD-1.5 weights are inflated,
        modules the reference adds keep their constructor init (see checkpoint.inflate)."""
        if not use_sc_attn or use_st_attn:
            raise NotImplementedError("only the shipped configuration use_sc_attn=True, use_st_attn=False is on the hot path (eval-motion.yaml:44-46)")
        from .. import checkpoint
        sd, info = checkpoint.load_unet_state_dict(pretrained_model_name_or_path, subfolder, resume_from_checkpoint, adapter_weight_path)
        m = cls(sd, device)
        m.loading_info = info
        return m

    def _residual_rows(self, res, two_branch_hint: bool):
        """Accept the reference layout (5-D [b, C, f, h, w]) or ready channels-last rows."""
        if res is None:
            return None
        if res.dim() == 2:
            return res
        return ops.nchw5_to_rows(res.to(self.device))

    def forward_rows(self, sample, timestep, encoder_hidden_states, down_rows=None, mid_rows=None, two_branch=False, taps=None, shard=None,
                     normal_infer: bool = False, res_ready=None, side_stream=None, cfg_dup: bool = False) -> graph.Act:
        t = float(timestep.item() if torch.is_tensor(timestep) else timestep)
        # normal_infer = the plain SD forward of the inversion (attention_2d.py:770-777 bypasses the patched closures' edits):
        # registered editors neither act nor count there, so a later denoising run starts from un-advanced counters
        spatial = None if normal_infer else self.spatial_editor
        temporal = None if normal_infer else self.temporal_editor
        return graph.unet_forward(self.P, sample.to(self.device), t, encoder_hidden_states.to(self.device), down_res=down_rows, mid_res=mid_rows,
                                  two_branch=two_branch, spatial=spatial, temporal=temporal, taps=taps, shard=shard,
                                  normal_infer=normal_infer, res_ready=res_ready, side_stream=side_stream, cfg_dup=cfg_dup)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None, return_dict: bool = True,
                normal_infer: bool = False, skeleton=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                source_masks=None, target_masks=None, rectangle_source_masks=None, taps=None):
        if class_labels is not None or attention_mask is not None or skeleton is not None:
            raise NotImplementedError("class_labels / attention_mask / skeleton are not on the inference hot path "
                                      "(pipeline_motion_editor.py:632-640 passes none of them)")
        if normal_infer and down_block_additional_residuals is not None:
            raise NotImplementedError("normal_infer is the DDIM-inversion forward (util.py:89-96): single branch, no ControlNet residuals")
        down = mid = None
        two = False
        if down_block_additional_residuals is not None:
            m = mid_block_additional_residual
            if m.dim() == 5:
                two = m.shape[0] == 4  # unet_2d_condition.py:478
                if two:
                    m = m[[1, 3]]      # rows 0, 2 are zeros by construction (pipeline :628-629)
                mid = ops.nchw5_to_rows(m.to(self.device))
            else:
                mid = m
                two = sample.shape[0] == 4 and m.shape[0] * 2 == sample.shape[0] * sample.shape[2] * (sample.shape[3] // 8) * (sample.shape[4] // 8)
            down = [self._residual_rows(r, two) for r in down_block_additional_residuals]
        out = self.forward_rows(sample, timestep, encoder_hidden_states, down, mid, two, taps, normal_infer=normal_infer)
        B, _, f, h, w = sample.shape
        y = ops.rows_to_nchw5(out.t, B, 4, f, h, w)
        return UNet2DConditionOutput(sample=y) if return_dict else (y,)

    __call__ = forward
