"""Spatial attention-editor base + registration, mirroring the reference's
``motion_editor/attn_control/fully_control_utils.py`` (MutualAttentionBase :29-70,
regiter_fully_attention_editor_diffusers :109-229).

Difference by design (MI355X-first): the reference monkey-patches ``forward`` of every
MotionFrameAttention / CrossAttention and hands the editor pre-gathered ``(B*H, n, dh)`` q/k/v
(keys already duplicated to [prev | cur]).  Here the UNet graph calls the registered editor with an
``AttnCall`` (un-gathered row views + geometry) and the editor picks a key-segment table for the
fused HIP kernel -- the gather and the masked-key duplication never exist in memory.  Counters,
gating and attributes are the reference's.
"""
from __future__ import annotations

from .. import segments


class MutualAttentionBase:
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
                attention_mask=None, call=None, text_seg=None, **kwargs):
        """Un-edited attention (reference :48-66): self = [prev | cur] frame keys, cross = text keys."""
        if call is None:
            raise ValueError("motioneditor_amd editors are driven by the UNet graph with call=AttnCall(...)")
        if is_cross:
            return call.run(*text_seg)
        return call.run(*segments.prev_cur(call.B, call.f, call.q.device, getattr(call, "shard", None)))

    def edits_next_self_attention(self) -> bool:
        """Will the NEXT self-attention call be edited (K/V injection)?  The UNet graph asks before it shares the classifier-free-guidance
        prefix between batch entries (an edited layer needs the full (recon, edit) x (uncond, cond) batch).  The base class never edits."""
        return False

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


def regiter_fully_attention_editor_diffusers(model, editor: MutualAttentionBase):
    """Register `editor` on ``model.unet`` (reference :109-229).  The reference counts the patched
    MotionFrameAttention + CrossAttention modules under down/mid/up: 16 transformer blocks x {attn1, attn2}."""
    unet = model.unet
    unet.spatial_editor = editor
    editor.num_att_layers = unet.num_spatial_attention_layers
    return editor
