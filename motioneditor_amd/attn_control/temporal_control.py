"""TemporalSelfAttentionControl (reference ``motion_editor/attn_control/temporal_control.py:26-89``):
on edited layers/steps the edit-branch queries attend the reconstruction branch's K/V (full
replacement), causal mask kept.  In the fused kernel that is just a batch remap kv_map=[0,0,2,2]."""
from __future__ import annotations

from .temporal_control_utils import TemporalAttentionBase


class TemporalSelfAttentionControl(TemporalAttentionBase):
    MODEL_TYPE = {"SD": 16, "SDXL": 70}

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, model_type="SD"):
        super().__init__()
        self.total_steps = total_steps
        self.total_layers = self.MODEL_TYPE.get(model_type, 16)
        self.start_step = start_step
        self.start_layer = start_layer
        self.layer_idx = layer_idx if layer_idx is not None else list(range(start_layer, self.total_layers))
        self.step_idx = step_idx if step_idx is not None else list(range(start_step, total_steps))

    def forward(self, q=None, k=None, v=None, sim=None, attn=None, is_cross=None, place_in_unet=None, num_heads=None,
                attention_mask=None, call=None, **kwargs):
        if is_cross or self.cur_step not in self.step_idx or self.cur_att_layer not in self.layer_idx:  # reference :74
            return super().forward(is_cross=is_cross, place_in_unet=place_in_unet, num_heads=num_heads, call=call)
        if call.B not in (2, 4):
            raise ValueError("edited temporal attention expects batch 4 (reference :77-85) or one (rec, edit) pair")
        return call.run(kv_map=[0, 0, 2, 2][:call.B])
