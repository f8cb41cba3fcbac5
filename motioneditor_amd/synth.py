"""State-dict key schema of the reference models and a deterministic synthetic weight generator.

There are no SD-1.5 / ControlNet / adapter checkpoints on the build or bench machines, so parity and
bench runs use seeded random weights *keyed by the reference's own state-dict names*
(``SURVEY.md §8b``): the same generator runs in the oracle tests and in the product path, so both
sides see bit-identical fp32 tensors.  The schema functions double as the loader contract for real
checkpoints: ``unet_schema()`` must equal the key/shape set of the reference
``UNet2DConditionModel(...).state_dict()`` (checked against ``tests/golden/unet_keys.txt``, dumped
from the reference by ``oracle/make_golden.py``).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

Shape = Tuple[int, ...]
BLOCK_CH = (320, 640, 1280, 1280)
TEMB = 1280
CROSS = 768


def _attn_block(s: Dict[str, Shape], p: str, c: int, temporal: bool) -> None:
    """Transformer2DModel keys (reference models/attention_2d.py:255-336, 392-463)."""
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    s[p + "proj_in.weight"] = (c, c, 1, 1)
    s[p + "proj_in.bias"] = (c,)
    t = p + "transformer_blocks.0."
    for a, kd in (("attn1", c), ("attn2", CROSS)):
        s[t + a + ".to_q.weight"] = (c, c)
        s[t + a + ".to_k.weight"] = (c, kd)
        s[t + a + ".to_v.weight"] = (c, kd)
        s[t + a + ".to_out.0.weight"] = (c, c)
        s[t + a + ".to_out.0.bias"] = (c,)
    s[t + "ff.net.0.proj.weight"] = (8 * c, c)
    s[t + "ff.net.0.proj.bias"] = (8 * c,)
    s[t + "ff.net.2.weight"] = (c, 4 * c)
    s[t + "ff.net.2.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3"):
        s[t + n + ".weight"] = (c,)
        s[t + n + ".bias"] = (c,)
    if temporal:
        s[t + "attn_temp.to_q.weight"] = (c, c)
        s[t + "attn_temp.to_k.weight"] = (c, c)
        s[t + "attn_temp.to_v.weight"] = (c, c)
        s[t + "attn_temp.to_out.0.weight"] = (c, c)
        s[t + "attn_temp.to_out.0.bias"] = (c,)
        s[t + "norm_temp.weight"] = (c,)
        s[t + "norm_temp.bias"] = (c,)
    s[p + "proj_out.weight"] = (c, c, 1, 1)
    s[p + "proj_out.bias"] = (c,)


def _resnet(s: Dict[str, Shape], p: str, cin: int, cout: int, temporal: bool) -> None:
    """ResnetBlock2D keys (reference models/resnet_2d.py:128-197)."""
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    s[p + "time_emb_proj.weight"] = (cout, TEMB)
    s[p + "time_emb_proj.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)
    if temporal:
        for n in ("temp_conv1", "temp_conv2"):
            s[p + n + ".weight"] = (cout, cout, 3)
            s[p + n + ".bias"] = (cout,)


def _encoder(s: Dict[str, Shape], temporal: bool) -> None:
    """conv_in, time embedding, 4 down blocks, mid block -- shared by the UNet and the ControlNet."""
    s["conv_in.weight"] = (320, 4, 3, 3)
    s["conv_in.bias"] = (320,)
    s["time_embedding.linear_1.weight"] = (TEMB, 320)
    s["time_embedding.linear_1.bias"] = (TEMB,)
    s["time_embedding.linear_2.weight"] = (TEMB, TEMB)
    s["time_embedding.linear_2.bias"] = (TEMB,)
    cin = 320
    for i, c in enumerate(BLOCK_CH):
        for j in range(2):
            if i < 3:
                _attn_block(s, f"down_blocks.{i}.attentions.{j}.", c, temporal)
        for j in range(2):
            _resnet(s, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else c, c, temporal)
        if i < 3:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        cin = c
    _attn_block(s, "mid_block.attentions.0.", 1280, temporal)
    _resnet(s, "mid_block.resnets.0.", 1280, 1280, temporal)
    _resnet(s, "mid_block.resnets.1.", 1280, 1280, temporal)


ADAPTER_CH = (320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280)  # controlnet_adapter.py:443-448


def adapter_schema(prefix: str = "controlnet_adapter.") -> "OrderedDict[str, Shape]":
    """ControlAdapter keys (reference models/controlnet_adapter.py:437-552; ksize=1, sk=True)."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    for i, c in enumerate(ADAPTER_CH):
        p = f"{prefix}body.{i}."
        s[p + "block1.weight"] = (c, c, 3)
        s[p + "block1.bias"] = (c,)
        s[p + "block2.weight"] = (c, c, 1)
        s[p + "block2.bias"] = (c,)
        for a in ("attn_temp", "attn_pose", "attn_self_temp"):
            s[p + a + ".to_q.weight"] = (c, c)
            s[p + a + ".to_k.weight"] = (c, c)
            s[p + a + ".to_v.weight"] = (c, c)
            s[p + a + ".to_out.0.weight"] = (c, c)
            s[p + a + ".to_out.0.bias"] = (c,)
        s[p + "ff.net.0.proj.weight"] = (8 * c, c)
        s[p + "ff.net.0.proj.bias"] = (8 * c,)
        s[p + "ff.net.2.weight"] = (c, 4 * c)
        s[p + "ff.net.2.bias"] = (c,)
        for n in ("ff_norm", "norm_temp", "cross_pose_norm", "norm_self_temp"):
            s[p + n + ".weight"] = (c,)
            s[p + n + ".bias"] = (c,)
    return s


def unet_schema(with_adapter: bool = True) -> "OrderedDict[str, Shape]":
    """Key -> shape of the reference ``UNet2DConditionModel`` (SD-1.5 widths, cross_attention_dim 768)."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    _encoder(s, temporal=True)
    prev = 1280
    rev = BLOCK_CH[::-1]
    for i, c in enumerate(rev):
        skip_last = rev[min(i + 1, 3)]
        for j in range(3):
            if i > 0:
                _attn_block(s, f"up_blocks.{i}.attentions.{j}.", c, True)
        for j in range(3):
            skip = skip_last if j == 2 else c
            cin = (prev if j == 0 else c) + skip
            _resnet(s, f"up_blocks.{i}.resnets.{j}.", cin, c, True)
        if i < 3:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
        prev = c
    s["conv_norm_out.weight"] = (320,)
    s["conv_norm_out.bias"] = (320,)
    s["conv_out.weight"] = (4, 320, 3, 3)
    s["conv_out.bias"] = (4,)
    if with_adapter:
        s.update(adapter_schema())
    return s


COND_EMBED_CH = (16, 16, 32, 32, 96, 96, 256)


def controlnet_schema() -> "OrderedDict[str, Shape]":
    """diffusers==0.15.1 ``ControlNetModel`` keys for ``lllyasviel/sd-controlnet-openpose``
    (source not in the reference tree)."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    _encoder(s, temporal=False)
    s["controlnet_cond_embedding.conv_in.weight"] = (16, 3, 3, 3)
    s["controlnet_cond_embedding.conv_in.bias"] = (16,)
    for i in range(6):
        s[f"controlnet_cond_embedding.blocks.{i}.weight"] = (COND_EMBED_CH[i + 1], COND_EMBED_CH[i], 3, 3)
        s[f"controlnet_cond_embedding.blocks.{i}.bias"] = (COND_EMBED_CH[i + 1],)
    s["controlnet_cond_embedding.conv_out.weight"] = (320, 256, 3, 3)
    s["controlnet_cond_embedding.conv_out.bias"] = (320,)
    for i, c in enumerate(ADAPTER_CH):
        s[f"controlnet_down_blocks.{i}.weight"] = (c, c, 1, 1)
        s[f"controlnet_down_blocks.{i}.bias"] = (c,)
    s["controlnet_mid_block.weight"] = (1280, 1280, 1, 1)
    s["controlnet_mid_block.bias"] = (1280,)
    return s


VAE_UP_CH = (512, 512, 256, 128)


def vae_decoder_schema() -> "OrderedDict[str, Shape]":
    """Key -> shape of the decoder half of diffusers 0.15.1 ``AutoencoderKL`` (SD-1.5 VAE: block_out_channels
    (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32) plus ``post_quant_conv``."""
    s: "OrderedDict[str, Shape]" = OrderedDict()

    def conv(p, cout, cin, k):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cout, cin, 1)

    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", 512, 4, 3)
    resnet("decoder.mid_block.resnets.0", 512, 512)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", 512)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{a}.{n}.weight"] = (512, 512)
        s[f"{a}.{n}.bias"] = (512,)
    resnet("decoder.mid_block.resnets.1", 512, 512)
    prev = 512
    for i, c in enumerate(VAE_UP_CH):
        for j in range(3):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < 3:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        prev = c
    norm("decoder.conv_norm_out", 128)
    conv("decoder.conv_out", 3, 128, 3)
    return s


def vae_encoder_schema() -> "OrderedDict[str, Shape]":
    """Key -> shape of the encoder half of diffusers 0.15.1 ``AutoencoderKL`` (SD-1.5 VAE: block_out_channels
    (128, 256, 512, 512), layers_per_block 2, double_z -> 8 output channels) plus ``quant_conv``."""
    s: "OrderedDict[str, Shape]" = OrderedDict()

    def conv(p, cout, cin, k):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cout, cin, 1)

    conv("encoder.conv_in", 128, 3, 3)
    prev = 128
    for i, c in enumerate((128, 256, 512, 512)):
        for j in range(2):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < 3:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        prev = c
    resnet("encoder.mid_block.resnets.0", 512, 512)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", 512)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{a}.{n}.weight"] = (512, 512)
        s[f"{a}.{n}.bias"] = (512,)
    resnet("encoder.mid_block.resnets.1", 512, 512)
    norm("encoder.conv_norm_out", 512)
    conv("encoder.conv_out", 8, 512, 3)
    conv("quant_conv", 8, 8, 1)
    return s


# ---------------------------------------------------------------------------------------------
# deterministic synthetic tensors
# ---------------------------------------------------------------------------------------------
_RESIDUAL_OUT = ("to_out.0.weight", "proj_out.weight", "conv2.weight", "ff.net.2.weight", "temp_conv1.weight", "temp_conv2.weight",
                 "block1.weight", "block2.weight", "conv_shortcut.weight", "proj_attn.weight")


def synth_tensor(name: str, shape: Shape, seed: int = 33) -> np.ndarray:
    """fp32 tensor for state-dict entry `name`.  Norm scales ~ 1 +- 0.1, norm shifts and biases small,
    weights uniform with std gain/sqrt(fan_in) (gain 0.5 on residual-output layers so the 30+
    residual adds keep O(1) activations).  The reference zero-initialises temp_conv*,
    attn_temp.to_out, adapter block1/2 and attn_self_temp.to_out; they are randomised here on
    purpose so the temporal paths are exercised."""
    key = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF
    rng = np.random.Generator(np.random.Philox(key=key))
    n = int(np.prod(shape))
    u = rng.random(n, dtype=np.float32) * 2.0 - 1.0  # uniform(-1, 1), std 1/sqrt(3)
    if name.endswith(".weight") and len(shape) == 1:
        out = 1.0 + 0.1 * u
    elif name.endswith(".bias"):
        out = 0.05 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        gain = 0.5 if name.endswith(_RESIDUAL_OUT) or name.startswith("controlnet_down_blocks") or name.startswith("controlnet_mid_block") else 1.0
        out = u * (gain * (3.0 / fan_in) ** 0.5)
    return out.reshape(shape).astype(np.float32)


def synth_state_dict(schema: Dict[str, Shape], seed: int = 33, salt: str = "") -> "OrderedDict[str, np.ndarray]":
    """The deterministic synthetic weights of a schema.  Generating the 1.3 G parameters of the UNet takes a minute of one core, and the tests / bench /
    multi-process launches of one session each want them: the flat fp32 image is kept in a per-machine cache file (ME_SYNTH_CACHE, default
    <tmp>/me_synth_cache_<uid>, mode 0700; ME_SYNTH_CACHE=0 disables), keyed by a sha1 over the ORDERED schema, the seed, the salt and this
    module's source text; a cached image is used only if this user owns it and its first / last tensors equal what the generator makes now."""
    import hashlib
    import inspect
    import os
    import tempfile
    uid = os.getuid() if hasattr(os, "getuid") else 0
    cache = os.environ.get("ME_SYNTH_CACHE", os.path.join(tempfile.gettempdir(), f"me_synth_cache_{uid}"))   # per user, created 0700
    total = sum(int(np.prod(shp)) for shp in schema.values())
    path = None
    if cache != "0" and total >= 1 << 20:
        # the key covers the schema IN ITS ORDER (the flat image is sliced in insertion order), the seed, the salt and the text of this whole module
        # (synth_tensor and whatever it calls); sha1, not a 32-bit CRC
        key = hashlib.sha1(repr((list(schema.items()), seed, salt)).encode() + inspect.getsource(inspect.getmodule(synth_tensor)).encode()).hexdigest()[:20]
        path = os.path.join(cache, f"w_{key}_{total}.npy")
        if os.path.exists(path):
            try:
                st = os.stat(path)
                if hasattr(os, "getuid") and st.st_uid != uid:      # someone else's file in a shared tmp: never trusted
                    raise PermissionError(path)
                # copy-on-write map: the pages are the file's page-cache pages, shared by every process of the machine that maps the same weights (the
                # gloo test ranks, `bench.py --gpus N`) until someone writes to them -- 6.7 GB once instead of once per process
                flat = np.asarray(np.load(path, mmap_mode="c"))
                if flat.shape == (total,) and flat.dtype == np.float32:
                    out, o = OrderedDict(), 0
                    for k, shp in schema.items():
                        n = int(np.prod(shp))
                        out[k] = flat[o:o + n].reshape(shp)
                        o += n
                    # spot check against the generator: the first and the last tensor of the image must be what synth_tensor makes now
                    ends = [next(iter(schema)), next(reversed(schema))] if isinstance(schema, OrderedDict) or hasattr(schema, "__reversed__") else [next(iter(schema))]
                    if all(np.array_equal(out[k], synth_tensor(salt + k, schema[k], seed)) for k in ends):
                        return out
            except Exception:   # a torn, stale or foreign file: regenerate
                pass
    out = OrderedDict((k, synth_tensor(salt + k, shp, seed)) for k, shp in schema.items())
    if path is not None:
        try:
            os.makedirs(cache, mode=0o700, exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as fh:
                np.save(fh, np.concatenate([v.reshape(-1) for v in out.values()]))
            os.replace(tmp, path)   # atomic: concurrent ranks either see the whole file or none
        except OSError:
            pass
    return out


# ---------------------------------------------------------------------------------------------
# deterministic synthetic inputs: latents ~ N(0,1), text embeddings ~ 0.3 N(0,1),
# ControlNet residuals ~ 0.3 N(0,1), binary person-like masks.  Returned as CPU fp32 torch tensors.
# ---------------------------------------------------------------------------------------------
def synth_normal(name: str, shape: Shape, seed: int = 33, scale: float = 1.0) -> np.ndarray:
    key = (zlib.crc32(("input:" + name).encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF
    rng = np.random.Generator(np.random.Philox(key=key))
    return (rng.standard_normal(int(np.prod(shape)), dtype=np.float32) * scale).reshape(shape)


def synth_masks(f: int, H: int, W: int) -> np.ndarray:
    """[1, f, 1, H, W] binary foreground masks: an ellipse drifting across the frame (the reference's
    `man.mask` PNGs are binary 0/255 person masks, data/dataset.py)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.zeros((1, f, 1, H, W), dtype=np.float32)
    for i in range(f):
        cx = W * (0.35 + 0.3 * i / max(f - 1, 1))
        cy = H * (0.5 + 0.08 * np.sin(i * 0.7))
        out[0, i, 0] = (((xx - cx) / (0.18 * W)) ** 2 + ((yy - cy) / (0.38 * H)) ** 2 <= 1.0).astype(np.float32)
    return out


def make_case_inputs(kind: str, B: int, f: int, h: int, w: int, seed: int = 33, t: int = 981):
    """Inputs of one UNet forward.  kind 'single': sample/ehs only; 'two': + 12 ControlNet-shaped
    residuals [2,C,f,h',w'], mid residual [4,1280,f,h/8,w/8] (zero rows for the recon branch) and
    source masks [1,f,1,8h,8w]."""
    import torch

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    d = {"t": t, "sample": T(synth_normal(f"{kind}.sample", (B, 4, f, h, w), seed)),
         "ehs": T(synth_normal(f"{kind}.ehs", (B, 77, CROSS), seed, 0.3))}
    if kind == "two":
        sizes = [h, h, h, h // 2, h // 2, h // 2, h // 4, h // 4, h // 4, h // 8, h // 8, h // 8]
        wsz = [w, w, w, w // 2, w // 2, w // 2, w // 4, w // 4, w // 4, w // 8, w // 8, w // 8]
        d["down_res"] = [T(synth_normal(f"{kind}.down{i}", (2, c, f, sizes[i], wsz[i]), seed, 0.3)) for i, c in enumerate(ADAPTER_CH)]
        m = synth_normal(f"{kind}.mid", (2, 1280, f, h // 8, w // 8), seed, 0.3)
        mid = np.zeros((4, 1280, f, h // 8, w // 8), dtype=np.float32)
        mid[1], mid[3] = m[0], m[1]
        d["mid_res"] = T(mid)
        d["source_masks"] = T(synth_masks(f, 8 * h, 8 * w))
    return d


def bench_inputs(f: int, h: int, w: int, seed: int = 33):
    """Inputs of the benchmarked two-branch step (bench.py; SURVEY.md 8d "synthetic inputs"): latents [2,4,f,h,w] = [recon, edit],
    50 unconditional embeddings, the two prompts' embeddings, the target skeleton video in [0, 1] and binary source masks.
    CPU fp32 torch tensors; the same generator feeds bench.py, the config-3 golden (oracle/make_golden.py --only-config3) and its
    GPU test, so all three see bit-identical data."""
    import torch

    T = torch.from_numpy
    return dict(latents=T(synth_normal("bench.latents", (2, 4, f, h, w), seed)),
                uncond=[T(synth_normal(f"bench.uncond{i}", (1, 77, 768), seed, 0.3)) for i in range(50)],
                cond=T(synth_normal("bench.cond", (2, 77, 768), seed, 0.3)),
                skeleton=T(np.clip(synth_normal("bench.skel", (1, f, 3, 8 * h, 8 * w), seed, 0.5) + 0.5, 0, 1).astype(np.float32)),
                masks=T(synth_masks(f, 8 * h, 8 * w)))
