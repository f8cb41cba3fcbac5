"""FullySelfAttentionControlMask -- the spatial K/V-injection editor the reference's inference uses
(``motion_editor/attn_control/fully_control.py:331-460``; parents :19-89).  Same constructor,
attributes and gating; the edited attention runs as ONE fused HIP launch over the key-segment table
``segments.edited_spatial`` (recon rows: [prev|cur]; edit rows: [src prev dual | src cur dual | own cur])
instead of four xformers calls on materialised 5N-key tensors.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F

from .. import segments
from .fully_control_utils import MutualAttentionBase


class MutualSelfAttentionControl(MutualAttentionBase):
    MODEL_TYPE = {"SD": 16, "SDXL": 70}

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, model_type="SD"):
        super().__init__()
        self.total_steps = total_steps
        self.total_layers = self.MODEL_TYPE.get(model_type, 16)
        self.start_step = start_step
        self.start_layer = start_layer
        self.layer_idx = layer_idx if layer_idx is not None else list(range(start_layer, self.total_layers))
        self.step_idx = step_idx if step_idx is not None else list(range(start_step, total_steps))


    def edits_next_self_attention(self) -> bool:
        return self.cur_step in self.step_idx and self.cur_att_layer // 2 in self.layer_idx   # the gate of forward() (reference :434)


class FullySelfAttentionControlMask(MutualSelfAttentionControl):
    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, thres=0.1,
                 ref_token_idx=[1], cur_token_idx=[1], mask_save_dir=None, model_type="SD", source_masks=None,
                 target_masks=None, rectangle_source_masks=None):
        super().__init__(start_step, start_layer, layer_idx, step_idx, total_steps, model_type)
        self.thres = thres
        self.ref_token_idx = ref_token_idx
        self.cur_token_idx = cur_token_idx
        self.self_attns = []
        self.cross_attns = []   # the reference appends head-mean 16x16 cross maps here and never reads them (:430-432)
        self.cross_attns_mask = None
        self.self_attns_mask = None
        self.mask_save_dir = mask_save_dir
        if self.mask_save_dir is not None:
            os.makedirs(self.mask_save_dir, exist_ok=True)
        if source_masks is None:
            raise ValueError("FullySelfAttentionControlMask requires source_masks [b, f, 1, H, W] (reference :366-370)")
        if target_masks is not None:
            raise NotImplementedError("target_masks blending (reference :449-457) is dead code at inference (target_masks=None)")
        self.target_masks = None
        self.rectangle_source_masks = None
        self.source_masks = source_masks.permute(0, 2, 1, 3, 4)  # "b f c h w -> b c f h w" (:368)
        self._planes = {}
        m = self.source_masks.detach().float()
        self.binary_masks = bool(((m == 0) | (m == 1)).all())   # man.mask PNGs are 0/255 -> 0/1 (data/dataset.py)

    def mask_planes(self, N: int, device) -> torch.Tensor:
        """fp16 [8, N]: masks nearest-resized to (8, sqrt N, sqrt N) (reference :376-390).  Plane p is used
        by HEAD p ("(b f)" rearrange with hard-coded num_frames=8 on (frame, head)-ordered rows)."""
        key = (N, str(device))
        if key not in self._planes:
            Hs = int(math.isqrt(N))
            if Hs * Hs != N:
                raise ValueError("masked attention needs a square token grid (reference :378)")
            m = F.interpolate(self.source_masks.detach().float().cpu(), (8, Hs, Hs), mode="nearest")  # one-off setup, not hot path
            self._planes[key] = m[0, 0].reshape(8, N).to(torch.float16).contiguous().to(device)
        return self._planes[key]

    def forward(self, q=None, k=None, v=None, sim=None, attn=None, is_cross=None, place_in_unet=None, num_heads=None,
                attention_mask=None, call=None, text_seg=None, **kwargs):
        if is_cross or self.cur_step not in self.step_idx or self.cur_att_layer // 2 not in self.layer_idx:  # reference :434
            return super().forward(is_cross=is_cross, place_in_unet=place_in_unet, num_heads=num_heads, call=call, text_seg=text_seg)
        if call.B not in (2, 4):
            raise ValueError("edited attention expects batch 4 = [uncond.rec, uncond.edit, cond.rec, cond.edit] (reference :439-441) "
                             "or one (rec, edit) pair on a CFG-parallel rank")
        if (num_heads * call.f) % 8:
            raise ValueError("heads * frames must be divisible by 8 (reference :377)")
        return call.run(*segments.edited_spatial(call.f, call.q.device, self.binary_masks, call.B, getattr(call, "shard", None)), mask=self.mask_planes(call.N, call.q.device))
