"""CPU oracle: fp32 PyTorch restatement of MotionEditor's two-branch DDIM denoising step.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module; the product path (``motioneditor_amd``) never does and
fails loudly when its HIP library is missing.

Every function cites the reference file:line it follows (paths relative to ``/root/reference``).
The restatement is *functional*: weights live in a flat ``dict`` keyed by the reference's own
state-dict names (``SURVEY.md §8b``), activations keep the reference layout ``[b, c, f, h, w]``.

Pinning status (``SURVEY.md §8c``):
  * UNet3D, ControlAdapter, both attention editors: pinned against the reference's own modules,
    imported in the build container through ``oracle/shim`` by ``oracle/make_golden.py``; vectors in
    ``tests/golden/``.
  * GEGLU / FeedForward / Timesteps / TimestepEmbedding / DDIMScheduler / ControlNetModel live in
    diffusers==0.15.1 (``requirements.txt:1``), whose source is NOT under ``/root/reference`` and
    which the reference never tests: **parity unpinned** for those rows.  They are restated from
    the published definitions; DDIM is additionally pinned by the reference's in-tree restatement
    ``motion_editor/util.py:77-87`` and ``motion_editor/p2p/null_text_optimization.py:26-36``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
HEADS = 8  # models/unet_2d_condition.py:206 passes attention_head_dim=8 as the head COUNT
SDPA_SCORE_BYTES = 4 << 30  # largest score tensor one sdpa() call materialises (see sdpa)


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _split_heads(t: torch.Tensor, heads: int) -> torch.Tensor:
    """[B, n, H*dh] -> [B*H, n, dh]  (attention_2d.py:95-100)."""
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)


def _merge_heads(t: torch.Tensor, heads: int) -> torch.Tensor:
    """[B*H, n, dh] -> [B, n, H*dh]  (attention_2d.py:102-107)."""
    bh, n, d = t.shape
    return t.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, d * heads)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh) + bias) v on [B*H, n, dh] tensors.

    This is both ``CrossAttention._attention`` (attention_2d.py:172-201, baddbmm/softmax/bmm) and the
    semantics of ``xformers.ops.memory_efficient_attention`` (attention_2d.py:246-253)."""
    scale = q.shape[-1] ** -0.5
    # the score tensor of one call is bounded (SDPA_SCORE_BYTES): larger problems -- config 3's level-0 attention is 768 x 4096 x
    # 8192 fp32 scores = 103 GB -- are walked in batch chunks; every batch entry sees exactly the same baddbmm / softmax / bmm
    step = max(1, SDPA_SCORE_BYTES // max(1, q.shape[1] * k.shape[1] * 4))
    if step < q.shape[0]:
        return torch.cat([sdpa(q[i:i + step], k[i:i + step], v[i:i + step], bias) for i in range(0, q.shape[0], step)], dim=0)
    s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device), q, k.transpose(1, 2), beta=0, alpha=scale)
    if bias is not None:
        s = s + bias
    return torch.bmm(s.softmax(dim=-1), v)


# --------------------------------------------------------------------------------------------
# R2: timestep embedding (diffusers Timesteps + TimestepEmbedding; called unet_2d_condition.py:432-438)
# --------------------------------------------------------------------------------------------
def timestep_sinusoid(t: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """diffusers ``get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def time_embed(sd: SD, p: str, t: torch.Tensor) -> torch.Tensor:
    w = sd[p + "time_embedding.linear_1.weight"]
    e = timestep_sinusoid(t).to(w.device, w.dtype)    # (fp32 everywhere in the pinned runs; the cast serves the fp16 calibration run)
    return _lin(sd, p + "time_embedding.linear_2", F.silu(_lin(sd, p + "time_embedding.linear_1", e)))


# --------------------------------------------------------------------------------------------
# R3/R5/R6: convolutions (resnet_2d.py:10-36, 39-125)
# --------------------------------------------------------------------------------------------
def inflated_conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1) -> torch.Tensor:
    """InflatedConv3d: per-frame Conv2d on "(b f) c h w" (resnet_2d.py:28-36)."""
    b, c, f, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)
    return y.reshape(b, f, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def temporal_conv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """TemporalConv: Conv1d(k=3, pad=1) over f on "(b h w) c f" (resnet_2d.py:10-26)."""
    b, c, f, h, w = x.shape
    y = F.conv1d(x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f), sd[p + ".weight"], sd[p + ".bias"], padding=sd[p + ".weight"].shape[-1] // 2)
    return y.reshape(b, h, w, c, f).permute(0, 3, 4, 1, 2)


def upsample_nearest_2x(x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(scale_factor=[1,2,2], mode="nearest") (resnet_2d.py:77)."""
    return x.repeat_interleave(2, dim=-2).repeat_interleave(2, dim=-1)


# --------------------------------------------------------------------------------------------
# R4: ResnetBlock2D (resnet_2d.py:199-249).  GroupNorm runs on the 5-D tensor, so its statistics
# span (C/32)*f*h*w -- across ALL frames (resnet_2d.py:202,230).
# --------------------------------------------------------------------------------------------
def resnet_block(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor, eps: float = 1e-5, temporal: bool = True) -> torch.Tensor:
    h = F.group_norm(x, 32, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = inflated_conv(sd, p + ".conv1", F.silu(h))
    if temporal and (p + ".temp_conv1.weight") in sd:
        h = h + temporal_conv(sd, p + ".temp_conv1", h)
    h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None, None]
    h = F.group_norm(h, 32, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = inflated_conv(sd, p + ".conv2", F.silu(h))
    if temporal and (p + ".temp_conv2.weight") in sd:
        h = h + temporal_conv(sd, p + ".temp_conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = inflated_conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h  # output_scale_factor == 1.0


# --------------------------------------------------------------------------------------------
# Editors (attn_control/*).  Restated as small state machines + the attention they compute.
# --------------------------------------------------------------------------------------------
class _EditorBase:
    """Layer/step counter of MutualAttentionBase / TemporalAttentionBase
    (fully_control_utils.py:29-46, temporal_control_utils.py:27-44)."""

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, total_layers=16):
        self.cur_step = 0
        self.cur_att_layer = 0
        self.num_att_layers = -1
        self.layer_idx = list(layer_idx) if layer_idx is not None else list(range(start_layer, total_layers))
        self.step_idx = list(step_idx) if step_idx is not None else list(range(start_step, total_steps))

    def _tick(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class SpatialEditor(_EditorBase):
    """FullySelfAttentionControlMask (fully_control.py:331-460) with target_masks=None."""

    def __init__(self, source_masks: torch.Tensor, **kw):
        super().__init__(**kw)
        self.num_att_layers = 32
        # fully_control.py:366-368: "b f c h w -> b c f h w"
        self.source_masks = source_masks.permute(0, 2, 1, 3, 4).float()

    def active(self, is_cross: bool) -> bool:
        # fully_control.py:434
        return (not is_cross) and self.cur_step in self.step_idx and (self.cur_att_layer // 2) in self.layer_idx

    def masked_kv(self, k: torch.Tensor, v: torch.Tensor, N: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """attn_batch(is_mask_attn=True) (fully_control.py:372-413).  k, v: [(f*H), 4N, dh] =
        [src prev | src cur | edit prev | edit cur].  Returns the 5N-key K and V.

        The reference rearranges the (frame, head)-ordered rows with a hard-coded num_frames=8
        (fully_control.py:377,393), so row r is multiplied by mask frame r % 8 == head index."""
        nf = 8
        Hs = int(math.isqrt(N))
        assert Hs * Hs == N and k.shape[0] % nf == 0
        m = F.interpolate(self.source_masks, (nf, Hs, Hs), mode="nearest").to(k.device, k.dtype)  # [1,1,8,Hs,Hs]
        prev_idx = torch.arange(nf) - 1
        prev_idx[0] = 0
        m_prev, m_cur = m[:, :, prev_idx], m

        def mul(part: torch.Tensor, mk: torch.Tensor) -> torch.Tensor:
            rows, _, dh = part.shape
            t = part.reshape(rows // nf, nf, Hs, Hs, dh).permute(0, 4, 1, 2, 3)  # b c f h w
            t = t * mk
            return t.permute(0, 2, 3, 4, 1).reshape(rows, N, dh)

        ks = k[:, : 2 * N]
        k_fg = torch.cat([mul(ks[:, :N], m_prev), mul(ks[:, N:], m_cur)], dim=1)
        k_bg = torch.cat([mul(ks[:, :N], 1 - m_prev), mul(ks[:, N:], 1 - m_cur)], dim=1)
        k5 = torch.cat([k_fg, k_bg, k[:, 3 * N:]], dim=1)
        v5 = torch.cat([v[:, : 2 * N], v[:, : 2 * N], v[:, 3 * N:]], dim=1)
        return k5, v5

    def __call__(self, q, k, v, is_cross: bool) -> torch.Tensor:
        """q: [B*f*H, N, dh]; k, v: [B*f*H, Nk, dh] (already prev|cur gathered for self).  Returns [B*f, N, C]."""
        if not self.active(is_cross):
            out = _merge_heads(sdpa(q, k, v), HEADS)  # fully_control_utils.py:48-66
        else:
            N = q.shape[1]
            qs, ks, vs = q.chunk(4), k.chunk(4), v.chunk(4)  # [u.rec, u.edit, c.rec, c.edit]
            outs = []
            for i in range(4):
                if i % 2 == 0:  # reconstruction rows: own keys (fully_control.py:442-443)
                    outs.append(_merge_heads(sdpa(qs[i], ks[i], vs[i]), HEADS))
                else:           # editing rows: [src | edit] -> 5N masked keys (fully_control.py:444-447)
                    k5, v5 = self.masked_kv(torch.cat([ks[i - 1], ks[i]], 1), torch.cat([vs[i - 1], vs[i]], 1), N)
                    outs.append(_merge_heads(sdpa(qs[i], k5, v5), HEADS))
            out = torch.cat(outs, dim=0)
        self._tick()
        return out


class TemporalEditor(_EditorBase):
    """TemporalSelfAttentionControl (temporal_control.py:26-89)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.num_att_layers = 16

    def active(self) -> bool:
        return self.cur_step in self.step_idx and self.cur_att_layer in self.layer_idx  # temporal_control.py:74

    def __call__(self, q, k, v, bias) -> torch.Tensor:
        """q,k,v: [(B*N*H), f, dh], rows ordered (branch-batch, pixel, head)."""
        if self.active():
            qs, ks, vs = q.chunk(4), list(k.chunk(4)), list(v.chunk(4))
            ks[1], vs[1], ks[3], vs[3] = ks[0], vs[0], ks[2], vs[2]  # edit Q attends recon K,V (temporal_control.py:82-85)
            out = torch.cat([sdpa(qs[i], ks[i], vs[i], bias) for i in range(4)], dim=0)
        else:
            out = sdpa(q, k, v, bias)
        self._tick()
        return _merge_heads(out, HEADS)


# --------------------------------------------------------------------------------------------
# R7-R12: Transformer2DModel / BasicTransformerBlock (attention_2d.py:338-389, 493-547)
# --------------------------------------------------------------------------------------------
def _prev_cur_gather(t: torch.Tensor, f: int) -> torch.Tensor:
    """[B*f, N, C] -> [B*f, 2N, C] keys = [frame max(i-1,0) | frame i] (attention_2d.py:732-740)."""
    bf, n, c = t.shape
    t = t.reshape(bf // f, f, n, c)
    prev = torch.arange(f) - 1
    prev[0] = 0
    return torch.cat([t[:, prev], t], dim=2).reshape(bf, 2 * n, c)


def _first_prev_gather(t: torch.Tensor, f: int) -> torch.Tensor:
    """keys = [frame 0 | frame max(i-1,0)] within chunks of f frames (controlnet_adapter.py:352-361)."""
    bf, n, c = t.shape
    t = t.reshape(bf // f, f, n, c)
    prev = torch.arange(f) - 1
    prev[0] = 0
    return torch.cat([t[:, [0] * f], t[:, prev]], dim=2).reshape(bf, 2 * n, c)


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """diffusers FeedForward(geglu): Linear(C,8C) -> a*gelu_erf(g) -> Linear(4C,C)."""
    a, g = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(g))


def causal_bias(f: int, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(1 - tril) * -10000, shape [1,f,f] (attention_2d.py:542-543)."""
    b = (1.0 - torch.tril(torch.ones(f, f)))[None] * -10000.0
    return b if like is None else b.to(like.device, like.dtype)


def basic_block(sd: SD, p: str, x: torch.Tensor, ehs: Optional[torch.Tensor], f: int,
                spatial: Optional[SpatialEditor], temporal: Optional[TemporalEditor],
                sc_attn: bool = True, has_temp: bool = True) -> torch.Tensor:
    """x: [B*f, N, C]; ehs: [B*f, 77, 768] (already repeated per frame, attention_2d.py:343)."""
    # attn1: MotionFrameAttention.forward_sc_attn (attention_2d.py:705-768) or the patched closure
    # (fully_control_utils.py:113-161)
    n1 = F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    q = _split_heads(_lin(sd, p + ".attn1.to_q", n1), HEADS)
    k, v = _lin(sd, p + ".attn1.to_k", n1), _lin(sd, p + ".attn1.to_v", n1)
    if sc_attn:
        k, v = _prev_cur_gather(k, f), _prev_cur_gather(v, f)
    k, v = _split_heads(k, HEADS), _split_heads(v, HEADS)
    a = spatial(q, k, v, False) if spatial is not None else _merge_heads(sdpa(q, k, v), HEADS)
    x = _lin(sd, p + ".attn1.to_out.0", a) + x
    # attn2: CrossAttention (attention_2d.py:115-201) / closure (fully_control_utils.py:162-206)
    if ehs is not None:
        n2 = F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
        q = _split_heads(_lin(sd, p + ".attn2.to_q", n2), HEADS)
        k = _split_heads(_lin(sd, p + ".attn2.to_k", ehs), HEADS)
        v = _split_heads(_lin(sd, p + ".attn2.to_v", ehs), HEADS)
        a = spatial(q, k, v, True) if spatial is not None else _merge_heads(sdpa(q, k, v), HEADS)
        x = _lin(sd, p + ".attn2.to_out.0", a) + x
    # feed-forward (attention_2d.py:531)
    x = feed_forward(sd, p + ".ff", F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])) + x
    # temporal attention (attention_2d.py:534-545; TemporalSelfAttention temporal_attn.py:95-181)
    if has_temp:
        bf, n, c = x.shape
        xt = x.reshape(bf // f, f, n, c).permute(0, 2, 1, 3).reshape(bf // f * n, f, c)  # "(b f) d c -> (b d) f c"
        nt = F.layer_norm(xt, (c,), sd[p + ".norm_temp.weight"], sd[p + ".norm_temp.bias"])
        q = _split_heads(_lin(sd, p + ".attn_temp.to_q", nt), HEADS)
        k = _split_heads(_lin(sd, p + ".attn_temp.to_k", nt), HEADS)
        v = _split_heads(_lin(sd, p + ".attn_temp.to_v", nt), HEADS)
        a = temporal(q, k, v, causal_bias(f, q)) if temporal is not None else _merge_heads(sdpa(q, k, v, causal_bias(f, q)), HEADS)
        xt = _lin(sd, p + ".attn_temp.to_out.0", a) + xt
        x = xt.reshape(bf // f, n, f, c).permute(0, 2, 1, 3).reshape(bf, n, c)
    return x


def transformer2d(sd: SD, p: str, x: torch.Tensor, ehs: Optional[torch.Tensor],
                  spatial=None, temporal=None, sc_attn: bool = True, has_temp: bool = True) -> torch.Tensor:
    """Transformer2DModel.forward (attention_2d.py:338-389): per-frame GN(32, eps 1e-6), 1x1 proj."""
    b, c, f, h, w = x.shape
    xi = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    e = ehs.repeat_interleave(f, dim=0) if ehs is not None else None  # 'b n c -> (b f) n c'
    t = F.group_norm(xi, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    t = F.conv2d(t, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    t = t.permute(0, 2, 3, 1).reshape(b * f, h * w, c)
    t = basic_block(sd, p + ".transformer_blocks.0", t, e, f, spatial, temporal, sc_attn, has_temp)
    t = t.reshape(b * f, h, w, c).permute(0, 3, 1, 2)
    t = F.conv2d(t, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]) + xi
    return t.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------------------------
# R15: ControlAdapter (controlnet_adapter.py:437-565)
# --------------------------------------------------------------------------------------------
ADAPTER_CHUNK = 8  # hard-coded num_frames=8 (controlnet_adapter.py:414,438,472)


def adapter_block(sd: SD, p: str, x: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward (controlnet_adapter.py:497-534).  x: ControlNet residual [b,C,t,h,w];
    src: UNet edit-branch skip [b,C,t,h,w].  Returns [(b t), C, h, w] reshaped to [b,C,t,h,w]."""
    b, c, t, hh, ww = x.shape
    assert t % ADAPTER_CHUNK == 0
    n = hh * ww
    # conv path: block1 (TemporalConv k=3) -> ReLU -> block2 (TemporalConv k=1) -> + x, on chunks of 8 frames
    xc = x.permute(0, 2, 1, 3, 4).reshape(b * t // ADAPTER_CHUNK, ADAPTER_CHUNK, c, hh, ww).permute(0, 2, 1, 3, 4)
    hc = temporal_conv(sd, p + ".block1", xc)
    hc = temporal_conv(sd, p + ".block2", F.relu(hc)) + xc
    hc = hc.permute(0, 2, 1, 3, 4).reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
    # attention path
    xa = x.permute(0, 2, 3, 4, 1).reshape(b * t, n, c)  # "(b f) (h w) c"
    na = F.layer_norm(xa, (c,), sd[p + ".norm_temp.weight"], sd[p + ".norm_temp.bias"])
    q = _split_heads(_lin(sd, p + ".attn_temp.to_q", na), HEADS)
    k = _split_heads(_first_prev_gather(_lin(sd, p + ".attn_temp.to_k", na), ADAPTER_CHUNK), HEADS)
    v = _split_heads(_first_prev_gather(_lin(sd, p + ".attn_temp.to_v", na), ADAPTER_CHUNK), HEADS)
    a = _lin(sd, p + ".attn_temp.to_out.0", _merge_heads(sdpa(q, k, v), HEADS)) + xa
    a = F.layer_norm(a, (c,), sd[p + ".cross_pose_norm.weight"], sd[p + ".cross_pose_norm.bias"])  # replaces the stream (:518)
    s = src.permute(0, 2, 3, 4, 1).reshape(b * t, n, c)
    q = _split_heads(_lin(sd, p + ".attn_pose.to_q", a), HEADS)
    k = _split_heads(_lin(sd, p + ".attn_pose.to_k", s), HEADS)
    v = _split_heads(_lin(sd, p + ".attn_pose.to_v", s), HEADS)
    a = _lin(sd, p + ".attn_pose.to_out.0", _merge_heads(sdpa(q, k, v), HEADS)) + a
    a = feed_forward(sd, p + ".ff", F.layer_norm(a, (c,), sd[p + ".ff_norm.weight"], sd[p + ".ff_norm.bias"])) + a
    at = a.reshape(b, t, n, c).permute(0, 2, 1, 3).reshape(b * n, t, c)  # "(b f) d c -> (b d) f c" with the TRUE t
    nt = F.layer_norm(at, (c,), sd[p + ".norm_self_temp.weight"], sd[p + ".norm_self_temp.bias"])
    q = _split_heads(_lin(sd, p + ".attn_self_temp.to_q", nt), HEADS)
    k = _split_heads(_lin(sd, p + ".attn_self_temp.to_k", nt), HEADS)
    v = _split_heads(_lin(sd, p + ".attn_self_temp.to_v", nt), HEADS)
    at = _lin(sd, p + ".attn_self_temp.to_out.0", _merge_heads(sdpa(q, k, v, causal_bias(t, q)), HEADS)) + at
    a = at.reshape(b, n, t, c).permute(0, 2, 1, 3)  # b t n c
    a = a.reshape(b, t, hh, ww, c).permute(0, 4, 1, 2, 3)
    return a + hc


def adapter_forward(sd: SD, p: str, ctrl: Sequence[torch.Tensor], src: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """ControlAdapter.forward (controlnet_adapter.py:554-565)."""
    return [adapter_block(sd, f"{p}body.{i}", ctrl[i], src[i]) for i in range(12)]


# --------------------------------------------------------------------------------------------
# R1: UNet2DConditionModel.forward (unet_2d_condition.py:363-546)
# --------------------------------------------------------------------------------------------
DOWN_HAS_ATTN = (True, True, True, False)
UP_HAS_ATTN = (False, True, True, True)


def unet_forward(sd: SD, sample: torch.Tensor, t, ehs: torch.Tensor,
                 down_res: Optional[Sequence[torch.Tensor]] = None, mid_res: Optional[torch.Tensor] = None,
                 spatial: Optional[SpatialEditor] = None, temporal: Optional[TemporalEditor] = None,
                 taps: Optional[dict] = None, normal_infer: bool = False) -> torch.Tensor:
    """normal_infer=True (DDIM inversion, inference.py:292): attn1 is plain per-frame self-attention
    (attention_2d.py:770-777 -> CrossAttention.forward); everything else, temporal attention included, unchanged."""
    sc = not normal_infer
    B = sample.shape[0]
    tt = torch.as_tensor(t).reshape(-1).expand(B)
    emb = time_embed(sd, "", tt)
    x = inflated_conv(sd, "conv_in", sample)
    skips = [x]
    for i in range(4):
        for j in range(2):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb)
            if DOWN_HAS_ATTN[i]:
                x = transformer2d(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, spatial, temporal, sc_attn=sc)
            skips.append(x)
        if i < 3:
            x = inflated_conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
    if taps is not None:
        taps["skips"] = [s.clone() for s in skips]
    if down_res is not None:
        if mid_res.shape[0] == 4:  # two-branch (unet_2d_condition.py:478-481)
            src = [s[[1, 3]] for s in skips]
            mot = adapter_forward(sd, "controlnet_adapter.", down_res, src)
            add = []
            for m in mot:
                z = torch.zeros_like(m[:1])
                add.append(torch.cat([z, m[0:1], z, m[1:2]], dim=0))
        else:                      # (unet_2d_condition.py:482-485)
            add = adapter_forward(sd, "controlnet_adapter.", down_res, skips)
        if taps is not None:
            taps["motion"] = [a.clone() for a in add]
        skips = [s + a for s, a in zip(skips, add)]
    x = resnet_block(sd, "mid_block.resnets.0", x, emb)
    x = transformer2d(sd, "mid_block.attentions.0", x, ehs, spatial, temporal, sc_attn=sc)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb)
    if mid_res is not None:
        x = x + mid_res
    if taps is not None:
        taps["mid"] = x.clone()
    for i in range(4):
        for j in range(3):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, emb)
            if UP_HAS_ATTN[i]:
                x = transformer2d(sd, f"up_blocks.{i}.attentions.{j}", x, ehs, spatial, temporal, sc_attn=sc)
        if i < 3:
            x = inflated_conv(sd, f"up_blocks.{i}.upsamplers.0.conv", upsample_nearest_2x(x))
    x = F.silu(F.group_norm(x, 32, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5))
    return inflated_conv(sd, "conv_out", x)


# --------------------------------------------------------------------------------------------
# R16: ControlNetModel.forward (diffusers 0.15.1 -- NOT in the reference tree; parity unpinned).
# Restated from the published SD-1.5 ControlNet architecture.  It is 2-D:
# every frame is an independent image, so it reuses the blocks above with f=1, no temporal layers
# and plain N-key self attention.
# --------------------------------------------------------------------------------------------
def controlnet_forward(sd: SD, sample: torch.Tensor, t, ehs: torch.Tensor, cond: torch.Tensor,
                       scale: float = 1.0, taps: Optional[dict] = None) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """sample [n,4,h,w], ehs [n,77,768], cond [n,3,8h,8w] -> 12 down residuals + mid residual (4-D).
    taps (test infrastructure): the conditioning embedding and the trunk's tensors in front of the 1x1 zero-convolutions --
    what oracle/make_golden.py --only-controlnet compares with the reference's own 2-D blocks."""
    n = sample.shape[0]
    emb = time_embed(sd, "", torch.as_tensor(t).reshape(-1).expand(n))
    c = F.silu(F.conv2d(cond, sd["controlnet_cond_embedding.conv_in.weight"], sd["controlnet_cond_embedding.conv_in.bias"], padding=1))
    for i in range(6):
        c = F.silu(F.conv2d(c, sd[f"controlnet_cond_embedding.blocks.{i}.weight"], sd[f"controlnet_cond_embedding.blocks.{i}.bias"],
                            padding=1, stride=2 if i % 2 == 1 else 1))
    c = F.conv2d(c, sd["controlnet_cond_embedding.conv_out.weight"], sd["controlnet_cond_embedding.conv_out.bias"], padding=1)
    x = (F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1) + c)[:, :, None]  # [n,C,1,h,w]
    outs = [x]
    for i in range(4):
        for j in range(2):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, temporal=False)
            if DOWN_HAS_ATTN[i]:
                x = transformer2d(sd, f"down_blocks.{i}.attentions.{j}", x, ehs, None, None, sc_attn=False, has_temp=False)
            outs.append(x)
        if i < 3:
            x = inflated_conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
            outs.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, temporal=False)
    x = transformer2d(sd, "mid_block.attentions.0", x, ehs, None, None, sc_attn=False, has_temp=False)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, temporal=False)
    if taps is not None:
        taps["cond_emb"], taps["outs"], taps["mid"] = c, [o[:, :, 0] for o in outs], x[:, :, 0]
    down = [inflated_conv(sd, f"controlnet_down_blocks.{i}", o, padding=0)[:, :, 0] * scale for i, o in enumerate(outs)]
    mid = inflated_conv(sd, "controlnet_mid_block", x, padding=0)[:, :, 0] * scale
    return down, mid


# --------------------------------------------------------------------------------------------
# R17: DDIM (diffusers DDIMScheduler with the SD-1.5 scheduler_config; formula pinned in-tree by
# util.py:77-87 and p2p/null_text_optimization.py:26-36)
# --------------------------------------------------------------------------------------------
@dataclass
class DDIM:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    steps_offset: int = 1
    num_inference_steps: int = 50
    alphas_cumprod: torch.Tensor = field(init=False)
    timesteps: List[int] = field(init=False)

    def __post_init__(self):
        betas = torch.linspace(self.beta_start ** 0.5, self.beta_end ** 0.5, self.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.set_timesteps(self.num_inference_steps)

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = [int(i * ratio) + self.steps_offset for i in range(n)][::-1]

    def coeffs(self, t: int) -> Tuple[float, float]:
        """prev = ca * x + cb * eps  (eta = 0, clip_sample False, set_alpha_to_one False)."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.alphas_cumprod[0])
        ca = (a_p / a_t) ** 0.5
        cb = (1 - a_p) ** 0.5 - (a_p * (1 - a_t) / a_t) ** 0.5
        return ca, cb

    def next_coeffs(self, t: int) -> Tuple[float, float]:
        """DDIM inversion, next = ca * x + cb * eps (util.py:77-87): the model output at `t` is applied between
        t - 20 (clamped to 999; alpha = final_alpha_cumprod = alphas_cumprod[0] below 0) and t."""
        cur_t = min(t - self.num_train_timesteps // self.num_inference_steps, 999)
        a_c = float(self.alphas_cumprod[cur_t]) if cur_t >= 0 else float(self.alphas_cumprod[0])
        a_n = float(self.alphas_cumprod[t])
        ca = (a_n / a_c) ** 0.5
        cb = (1 - a_n) ** 0.5 - (a_n * (1 - a_c) / a_c) ** 0.5
        return ca, cb

    def next_step(self, eps: torch.Tensor, t: int, x: torch.Tensor) -> torch.Tensor:
        """util.py:77-87, term by term."""
        cur_t = min(t - self.num_train_timesteps // self.num_inference_steps, 999)
        a_c = self.alphas_cumprod[cur_t] if cur_t >= 0 else self.alphas_cumprod[0]
        a_n = self.alphas_cumprod[t]
        x0 = (x - (1 - a_c) ** 0.5 * eps) / a_c ** 0.5
        return a_n ** 0.5 * x0 + (1 - a_n) ** 0.5 * eps

    def step(self, eps: torch.Tensor, t: int, x: torch.Tensor) -> torch.Tensor:
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.alphas_cumprod[0]
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


def ddim_loop(unet_sd: SD, ddim: DDIM, latent: torch.Tensor, num_inv_steps: int, cond: torch.Tensor,
              normal_infer: bool = True) -> List[torch.Tensor]:
    """DDIM inversion (util.py:111-124, called with normal_infer=True from inference.py:289-293): single-branch UNet on
    the conditional embedding only (no CFG), timesteps walked upwards; returns every intermediate latent."""
    out = [latent]
    for i in range(num_inv_steps):
        t = ddim.timesteps[len(ddim.timesteps) - i - 1]
        ehs = cond if cond.shape[0] == latent.shape[0] else cond.repeat(latent.shape[0], 1, 1)   # util.py:91-93
        eps = unet_forward(unet_sd, latent, t, ehs, normal_infer=normal_infer)
        latent = ddim.next_step(eps, t, latent)
        out.append(latent)
    return out


# --------------------------------------------------------------------------------------------
# P1: one denoising step (pipeline_motion_editor.py:603-648)
# --------------------------------------------------------------------------------------------
def null_optimization(unet_sd: SD, ddim: DDIM, latents: Sequence[torch.Tensor], context: torch.Tensor, null_inner_steps: int,
                      epsilon: float, num_steps: Optional[int] = None, guidance: float = 7.5,
                      grads: Optional[list] = None) -> List[torch.Tensor]:
    """MyNullInversion.null_optimization (p2p/null_text_optimization.py:133-166): per DDIM step, a fresh Adam
    (lr 1e-2 * (1 - i / 100)) pulls the unconditional embedding so that the guided prev_step of the current latent lands on
    the inversion latent one step earlier; early stop at loss < epsilon + i * 2e-5; the latent then advances with the
    optimised embedding.  The UNet runs with normal_infer=False whatever the caller asked for (:49-51, :54-58) -- sparse-causal
    attn1 on batch 1 / 2.  latents = the DDIM inversion trajectory (x_0 ... x_T), context = [uncond, cond] (2, 77, 768).
    num_steps: NUM_DDIM_STEPS of the reference (50); a shorter sweep uses the first num_steps timesteps.
    grads: if a list, receives the gradient of every inner step (test hook, not part of the reference)."""
    n = ddim.num_inference_steps if num_steps is None else num_steps
    uncond, cond = context.chunk(2)
    out = []
    latent_cur = latents[-1]
    for i in range(n):
        uncond = uncond.clone().detach().requires_grad_(True)
        opt = torch.optim.Adam([uncond], lr=1e-2 * (1.0 - i / 100.0))
        latent_prev = latents[len(latents) - i - 2]
        t = ddim.timesteps[i]
        ca, cb = ddim.coeffs(t)
        with torch.no_grad():
            eps_c = unet_forward(unet_sd, latent_cur, t, cond)
        for _ in range(null_inner_steps):
            eps_u = unet_forward(unet_sd, latent_cur, t, uncond)
            eps = eps_u + guidance * (eps_c - eps_u)
            rec = ca * latent_cur + cb * eps                       # prev_step (:26-36)
            loss = torch.nn.functional.mse_loss(rec, latent_prev)
            opt.zero_grad()
            loss.backward()
            if grads is not None:
                grads.append(uncond.grad.detach().clone())
            opt.step()
            if loss.item() < epsilon + i * 2e-5:
                break
        out.append(uncond[:1].detach())
        with torch.no_grad():                                      # get_noise_pred(latent_cur, t, False, context) (:53-65)
            e2 = unet_forward(unet_sd, torch.cat([latent_cur] * 2), t, torch.cat([uncond, cond]))
            eu, ec = e2.chunk(2)
            latent_cur = ca * latent_cur + cb * (eu + guidance * (ec - eu))
    return out


def denoise_step(unet_sd: SD, cn_sd: Optional[SD], ddim: DDIM, latents: torch.Tensor, t: int,
                 uncond: torch.Tensor, cond: torch.Tensor, ctrl_images: Optional[torch.Tensor],
                 spatial: Optional[SpatialEditor], temporal: Optional[TemporalEditor],
                 guidance: float = 7.5, taps: Optional[dict] = None) -> torch.Tensor:
    """latents [2,4,f,h,w]; uncond [1,77,768] (this step's null embedding); cond [2,77,768];
    ctrl_images [2f,3,8h,8w] (pipeline :418-459,556-570)."""
    f = latents.shape[2]
    x = torch.cat([latents] * 2)                                  # :605
    emb = torch.cat([uncond.expand(*cond.shape), cond])           # :608-609
    down = mid = None
    if cn_sd is not None:
        ci = x[[1, 3]].permute(0, 2, 1, 3, 4).reshape(2 * f, *x.shape[1:2], *x.shape[3:])  # :613-614
        pe = emb[[1, 3]].repeat(f, 1, 1)                          # :615,621 (tiles [e1,e3,e1,e3,...])
        d, m = controlnet_forward(cn_sd, ci, t, pe, ctrl_images)
        down = [r.reshape(2, f, *r.shape[1:]).permute(0, 2, 1, 3, 4) for r in d]           # :626
        m = m.reshape(2, f, *m.shape[1:]).permute(0, 2, 1, 3, 4)
        z = torch.zeros_like(m[:1])
        mid = torch.cat([z, m[0:1], z, m[1:2]], dim=0)            # :628-629
        if taps is not None:
            taps["cn_down"], taps["cn_mid"] = down, mid
    eps = unet_forward(unet_sd, x, t, emb, down, mid, spatial, temporal, taps)             # :632-640
    eu, ec = eps.chunk(2)
    eps = eu + guidance * (ec - eu)                               # :643-645
    if taps is not None:
        taps["noise_pred"] = eps
    return ddim.step(eps, t, latents)                             # :648


# --------------------------------------------------------------------------------------------
# AutoencoderKL.decode (diffusers 0.15.1 -- NOT in the reference tree; PARITY UNPINNED).
# Restated from the published SD-1.5 VAE decoder (diffusers `models/vae.py` Decoder, `unet_2d_blocks.py`
# UNetMidBlock2D / UpDecoderBlock2D, `resnet.py` ResnetBlock2D with temb=None, `attention.py` AttentionBlock);
# call site: pipeline_motion_editor.py:346-355 (`decode_latents`), per frame.
# --------------------------------------------------------------------------------------------
VAE_UP_CH = (512, 512, 256, 128)     # reversed block_out_channels (128, 256, 512, 512)


def _conv2d(sd: SD, p: str, x: torch.Tensor, padding: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def vae_resnet(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock2D, temb None, groups 32, eps 1e-6, output_scale_factor 1."""
    h = F.silu(F.group_norm(x, 32, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6))
    h = _conv2d(sd, p + ".conv1", h)
    h = F.silu(F.group_norm(h, 32, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6))
    h = _conv2d(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv2d(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def vae_attention(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttentionBlock, one head of width C: softmax(q k^T / sqrt(C)) v, proj_attn, + residual."""
    b, c, hh, ww = x.shape
    n = F.group_norm(x, 32, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6).reshape(b, c, hh * ww).transpose(1, 2)
    q, k, v = _lin(sd, p + ".query", n), _lin(sd, p + ".key", n), _lin(sd, p + ".value", n)
    a = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(c)), dim=-1)
    o = _lin(sd, p + ".proj_attn", torch.bmm(a, v))
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def vae_decode(sd: SD, z: torch.Tensor) -> torch.Tensor:
    """z [n,4,h,w] (already divided by 0.18215) -> image [n,3,8h,8w]: post_quant_conv (1x1), Decoder."""
    x = _conv2d(sd, "post_quant_conv", z, padding=0)
    x = _conv2d(sd, "decoder.conv_in", x)
    x = vae_resnet(sd, "decoder.mid_block.resnets.0", x)
    x = vae_attention(sd, "decoder.mid_block.attentions.0", x)
    x = vae_resnet(sd, "decoder.mid_block.resnets.1", x)
    for i in range(4):
        for j in range(3):
            x = vae_resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i < 3:
            x = _conv2d(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    x = F.silu(F.group_norm(x, 32, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return _conv2d(sd, "decoder.conv_out", x)


def vae_encode_moments(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.encode up to the distribution parameters (diffusers 0.15.1 `models/vae.py` Encoder + quant_conv; NOT in
    the reference tree, PARITY UNPINNED; call site inference.py:262): x [n,3,H,W] in [-1,1] -> moments [n,8,H/8,W/8]
    = (mean | logvar).  DownEncoderBlock2D's Downsample2D pads (0,1,0,1) and convolves with stride 2, padding 0."""
    x = _conv2d(sd, "encoder.conv_in", x)
    for i in range(4):
        for j in range(2):
            x = vae_resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i < 3:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    x = vae_resnet(sd, "encoder.mid_block.resnets.0", x)
    x = vae_attention(sd, "encoder.mid_block.attentions.0", x)
    x = vae_resnet(sd, "encoder.mid_block.resnets.1", x)
    x = F.silu(F.group_norm(x, 32, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    x = _conv2d(sd, "encoder.conv_out", x)
    return _conv2d(sd, "quant_conv", x, padding=0)


def vae_encode_sample(sd: SD, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample with the noise supplied: mean + exp(0.5 * clamp(logvar, -30, 20)) * noise."""
    m = vae_encode_moments(sd, x)
    mean, logvar = m[:, :4], m[:, 4:].clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def decode_latents(sd: SD, latents: torch.Tensor) -> torch.Tensor:
    """pipeline_motion_editor.py:346-355: [b,4,f,h,w] latents -> [b,3,f,8h,8w] video in [0,1]."""
    b, c, f, h, w = latents.shape
    x = (1 / 0.18215) * latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    v = vae_decode(sd, x)
    v = v.reshape(b, f, *v.shape[1:]).permute(0, 2, 1, 3, 4)
    return (v / 2 + 0.5).clamp(0, 1)
