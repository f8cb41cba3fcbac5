"""Temporal attention-editor base + registration (reference
``motion_editor/attn_control/temporal_control_utils.py:27-70, 77-144``)."""
from __future__ import annotations


class TemporalAttentionBase:
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def after_step(self):
        pass

    def __call__(self, q=None, k=None, v=None, sim=None, attn=None, is_cross=None, place_in_unet=None, num_heads=None,
                 attention_mask=None, **kwargs):
        out = self.forward(q=q, k=k, v=v, is_cross=is_cross, place_in_unet=place_in_unet, num_heads=num_heads,
                           attention_mask=attention_mask, sim=sim, attn=attn, **kwargs)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.after_step()
        return out

    def forward(self, q=None, k=None, v=None, sim=None, attn=None, is_cross=None, place_in_unet=None, num_heads=None,
                attention_mask=None, call=None, **kwargs):
        if call is None:
            raise ValueError("motioneditor_amd editors are driven by the UNet graph with call=TemporalCall(...)")
        if is_cross:
            raise ZeroDivisionError("temporal cross attention is a tripwire in the reference (print(1/0), :66)")
        return call.run()

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


def regiter_temporal_attention_editor_diffusers(model, editor: TemporalAttentionBase):
    """Register `editor` on every TemporalSelfAttention of ``model.unet`` (16 layers; reference :77-144)."""
    unet = model.unet
    unet.temporal_editor = editor
    editor.num_att_layers = unet.num_temporal_attention_layers
    return editor
