"""AutoencoderKL on libmotioned: the steps either side of the denoising loop --
`inference.py:262-265` `vae.encode(pixels).latent_dist.sample() * 0.18215` in front of the inversion, and
`pipeline_motion_editor.py:346-355` -> `vae.decode(latents).sample` per frame behind the loop.

The reference takes the class from diffusers 0.15.1 (`AutoencoderKL`; not in the reference tree -> parity unpinned, the
oracle `oracle/ref_cpu.py::vae_decode` restates the published decoder).  Same constructor contract as the other
models here: a state dict with the diffusers keys (`decoder.*`, `post_quant_conv.*`; encoder keys are ignored).

Launch graph: every 3x3 / 1x1 convolution is `me_gemm` (implicit GEMM, nearest-2x upsample folded into the gather),
GroupNorm(+SiLU) is `me_groupnorm` per image, and the mid block's single-head attention of width 512 -- outside
`me_attn`'s head sizes -- is three GEMMs per frame around `me_softmax_rows`:
    S = Q K^T / sqrt(512)   (K rows are exactly the [N][1][K] weight layout)
    V^T = W_v X^T           (the value projection computed transposed, so no transpose kernel; its bias is added
                             after P V, exact because the rows of P sum to one)
    O = P V^T^T + b_v
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .. import ops
from ..weights import Packed
from .compat import ModuleShims
from .graph import Act, conv3x3

UP_CH = (512, 512, 256, 128)
EPS = 1e-6


@dataclass
class DecoderOutput:
    sample: torch.Tensor


def _gn(P: Packed, p: str, x: Act, silu: bool) -> torch.Tensor:
    return ops.groupnorm(x.t, P.vec(p + ".weight"), P.vec(p + ".bias"), rows_per_group=x.N, eps=EPS, silu=silu)


def resnet(P: Packed, p: str, x: Act) -> Act:
    """diffusers ResnetBlock2D with temb = None, groups 32, eps 1e-6, output_scale_factor 1."""
    h = conv3x3(P, p + ".conv1", x.like(_gn(P, p + ".norm1", x, True)))
    sc = x.t
    if P.has(p + ".conv_shortcut.weight"):
        sc = ops.gemm(x.t, P.mat(p + ".conv_shortcut.weight"), bias=P.vec(p + ".conv_shortcut.bias"))
    return conv3x3(P, p + ".conv2", h.like(_gn(P, p + ".norm2", h, True)), res=sc)


def attention(P: Packed, p: str, x: Act) -> Act:
    """diffusers AttentionBlock: one head of width C over the N = h*w tokens of each image."""
    n_img, N, Cc = x.B * x.f, x.N, x.C
    xn = _gn(P, p + ".group_norm", x, False)
    q = ops.gemm(xn, P.mat(p + ".query.weight"), bias=P.vec(p + ".query.bias"))
    k = ops.gemm(xn, P.mat(p + ".key.weight"), bias=P.vec(p + ".key.bias"))
    wv = P.mat(p + ".value.weight").reshape(Cc, Cc)
    o = torch.empty_like(q)
    for i in range(n_img):
        r = slice(i * N, (i + 1) * N)
        vt = ops.gemm(wv, xn[r].reshape(N, 1, Cc))                                   # [C, N] = V^T without its bias
        s = ops.gemm(q[r], k[r].reshape(N, 1, Cc), alpha=1.0 / math.sqrt(Cc))        # [N, N]
        ops.softmax_rows(s, out=s)
        ops.gemm(s, vt.reshape(Cc, 1, N), bias=P.vec(p + ".value.bias"), out=o[r])
    return x.like(ops.gemm(o, P.mat(p + ".proj_attn.weight"), bias=P.vec(p + ".proj_attn.bias"), res=x.t))


def decode(P: Packed, z: torch.Tensor) -> torch.Tensor:
    """z fp32 [n, 4, h, w] (already divided by the scaling factor) -> fp32 [n, 3, 8h, 8w]."""
    n, c, h, w = z.shape
    if c != 4:
        raise ValueError(f"latent channels must be 4, got {c}")
    z = z.to(P.device, torch.float32).contiguous()
    key = "vae:post_quant3x3"
    if key not in P.cache:   # the 1x1 conv as the centre tap of a 3x3 one (me_conv_small is the C_in = 4 entry point)
        w1 = P.raw("post_quant_conv.weight").reshape(4, 4)
        w3 = torch.zeros(8, 9, 4)          # C_out padded to me_conv_small's multiple of 8
        w3[:4, 4, :] = w1
        b3 = torch.zeros(8)
        b3[:4] = P.raw("post_quant_conv.bias")
        P.cache[key] = (w3.contiguous().to(P.device), b3.to(P.device))
    pq = ops.conv_small(z, P.cache[key][0], P.cache[key][1], n_img=n, Cin=4, H=h, Wd=w, img_stride=4 * h * w, ch_stride=h * w)
    pq = ops.rows_to_nchw(pq, n, 4, h * w)                                           # fp32 planes for conv_in
    x = Act(ops.conv_small(pq, P.mat32("decoder.conv_in.weight"), P.vec32("decoder.conv_in.bias"), n_img=n, Cin=4, H=h, Wd=w,
                           img_stride=4 * h * w, ch_stride=h * w), n, 1, h, w)
    x = resnet(P, "decoder.mid_block.resnets.0", x)
    x = attention(P, "decoder.mid_block.attentions.0", x)
    x = resnet(P, "decoder.mid_block.resnets.1", x)
    for i in range(4):
        for j in range(3):
            x = resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i < 3:
            x = conv3x3(P, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, ups=1)
    y = _gn(P, "decoder.conv_norm_out", x, True)
    key = "vae:conv_out4"
    if key not in P.cache:   # 3 output channels padded to the GEMM's multiple of 4
        wt = torch.zeros(4, 9, UP_CH[-1])
        wt[:3] = Packed._as_taps(P.raw("decoder.conv_out.weight"))
        b = torch.zeros(4)
        b[:3] = P.raw("decoder.conv_out.bias")
        P.cache[key] = (wt.to(P.dtype).contiguous().to(P.device), b.to(P.dtype).to(P.device))
    wt, b = P.cache[key]
    out = ops.gemm(y, wt, M=n * x.N, bias=b, conv=(x.h, x.w, x.h, x.w, 1, 0))
    return ops.rows_to_nchw(out, n, 3, x.N).reshape(n, 3, x.h, x.w)


def encode_moments(P: Packed, x: torch.Tensor) -> Act:
    """x fp32 [n, 3, H, W] in [-1, 1] -> distribution parameters as fp16 rows [(n h w), 8] = (mean | logvar), h = H/8.
    diffusers Encoder: conv_in, four DownEncoderBlock2D (two resnets, Downsample2D(padding=0) = pad (0,1,0,1) + stride-2 conv on
    the first three), mid block (resnet, one-head attention, resnet), GroupNorm + SiLU, conv_out, then quant_conv (1x1)."""
    n, c, H, W = x.shape
    if c != 3 or H % 8 or W % 8:
        raise ValueError(f"expected [n, 3, H, W] with H, W multiples of 8, got {tuple(x.shape)}")
    x = x.to(P.device, torch.float32).contiguous()
    a = Act(ops.conv_small(x, P.mat32("encoder.conv_in.weight"), P.vec32("encoder.conv_in.bias"), n_img=n, Cin=3, H=H, Wd=W,
                           img_stride=3 * H * W, ch_stride=H * W), n, 1, H, W)
    for i in range(4):
        for j in range(2):
            a = resnet(P, f"encoder.down_blocks.{i}.resnets.{j}", a)
        if i < 3:
            name = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            ho, wo = a.h // 2, a.w // 2
            out = ops.gemm(a.t, P.mat(name + ".weight"), M=n * ho * wo, bias=P.vec(name + ".bias"), conv=(a.h, a.w, ho, wo, 2, 0, 1))
            a = a.like(out, ho, wo)
    a = resnet(P, "encoder.mid_block.resnets.0", a)
    a = attention(P, "encoder.mid_block.attentions.0", a)
    a = resnet(P, "encoder.mid_block.resnets.1", a)
    y = _gn(P, "encoder.conv_norm_out", a, True)
    m = conv3x3(P, "encoder.conv_out", a.like(y))
    return a.like(ops.gemm(m.t, P.mat("quant_conv.weight"), bias=P.vec("quant_conv.bias")))


class DiagonalGaussianDistribution:
    """What `AutoencoderKL.encode(x).latent_dist` returns: sample() = mean + std * noise (diffusers' class draws the noise with
    torch.randn; pass `generator` for a reproducible draw, or `noise` to supply it), mode() = mean."""

    def __init__(self, moments: Act):
        self._m = moments

    def _shape(self):
        return (self._m.B, 4, self._m.h, self._m.w)

    def sample(self, generator=None, noise: torch.Tensor = None, scale: float = 1.0) -> torch.Tensor:
        m = self._m
        if noise is None:
            dev = generator.device if generator is not None else m.t.device
            noise = torch.randn(self._shape(), generator=generator, device=dev, dtype=torch.float32)
        noise = noise.to(m.t.device, torch.float32).contiguous()
        return ops.gaussian_sample(m.t, noise, m.B, m.N, scale).reshape(self._shape())

    def mode(self) -> torch.Tensor:
        return ops.rows_to_nchw(self._m.t, self._m.B, 4, self._m.N).reshape(self._shape())


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


class AutoencoderKL(ModuleShims):
    """SD-1.5 VAE: `encode(x).latent_dist.sample()` and `decode(z).sample` like diffusers' class.  A state dict may hold either
    half (or both): the missing half raises on use."""

    scaling_factor = 0.18215

    def __init__(self, state_dict, device: str = "cuda", dtype: torch.dtype = torch.float16):
        self.P = Packed(state_dict, device, dtype=dtype)
        self.device = torch.device(device)
        self.dtype = dtype

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", **kwargs) -> "AutoencoderKL":
        """diffusers signature for a local directory (inference.py:154: AutoencoderKL.from_pretrained(path, subfolder="vae"))."""
        from .. import checkpoint
        return cls(checkpoint.load_file(checkpoint.find_weights(pretrained_model_name_or_path, subfolder)), device)

    def enable_slicing(self):        # diffusers: decode one image at a time to bound memory; every decode here is per image already
        return self

    def disable_slicing(self):
        return self

    @classmethod
    def from_synthetic(cls, device: str = "cuda", seed: int = 33) -> "AutoencoderKL":
        from .. import synth
        sd = synth.synth_state_dict(synth.vae_decoder_schema(), seed, salt="vae.")
        sd.update(synth.synth_state_dict(synth.vae_encoder_schema(), seed, salt="vae."))
        return cls(sd, device)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if not self.P.has("encoder.conv_in.weight"):
            raise KeyError("this AutoencoderKL was built without encoder.* weights")
        d = DiagonalGaussianDistribution(encode_moments(self.P, x))
        return AutoencoderKLOutput(latent_dist=d) if return_dict else (d,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        y = decode(self.P, z)
        return DecoderOutput(sample=y) if return_dict else (y,)
