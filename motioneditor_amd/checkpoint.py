"""Weight I/O: read real checkpoints into the reference key schema that
``weights.Packed`` consumes.

  * SD-1.5 ``unet/diffusion_pytorch_model.{safetensors,bin}`` (2-D UNet) -> inflated 3-D UNet state dict: the
    reference builds its model from config and loads the 2-D weights non-strictly
    (models/unet_2d_condition.py:548-796), so every module it ADDS keeps its constructor init:
    TemporalConv zeros (resnet_2d.py:15-16), attn_temp.to_out.0.weight zeros (attention_2d.py:462),
    adapter block1/2 zeros (controlnet_adapter.py:418-419), attn_self_temp.to_out.0.weight zeros (:493),
    LayerNorm 1/0, every other new Linear torch's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
  * accelerate ``checkpoint-N/`` (stage-1 tuned UNet) and ``controlnet_adapter_checkpoint-N.pth``
    (inference.py:237-240) are merged on top.
  * ``lllyasviel/sd-controlnet-openpose`` for the ControlNet.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from . import synth


def load_file(path) -> Dict[str, torch.Tensor]:
    """One weight file -> {name: CPU tensor}.  .safetensors, or a torch pickle (.bin/.pth/.pt) holding a flat state dict."""
    path = Path(path)
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file as _lf
        return dict(_lf(str(path), device="cpu"))
    sd = torch.load(str(path), map_location="cpu", weights_only=True)
    if isinstance(sd, Mapping) and "state_dict" in sd and isinstance(sd["state_dict"], Mapping):
        sd = sd["state_dict"]
    if not isinstance(sd, Mapping) or not all(torch.is_tensor(v) for v in sd.values()):
        raise ValueError(f"{path} does not hold a flat tensor state dict")
    return dict(sd)


def find_weights(model_dir, subfolder: Optional[str] = None) -> Path:
    d = Path(model_dir) / subfolder if subfolder else Path(model_dir)
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin", "model.safetensors", "pytorch_model.bin"):
        if (d / name).exists():
            return d / name
    raise FileNotFoundError(f"no diffusion_pytorch_model.{{safetensors,bin}} / model.safetensors / pytorch_model.bin under {d}")


_ZERO_INIT = (".temp_conv1.", ".temp_conv2.", ".block1.", ".block2.")
_ZERO_WEIGHT = ("attn_temp.to_out.0.weight", "attn_self_temp.to_out.0.weight")


def inflate(sd: Mapping[str, torch.Tensor], schema: Mapping[str, Tuple[int, ...]], seed: int = 0):
    """Complete `sd` to `schema`.  Returns (state dict, created keys).  Shape mismatches raise."""
    out: Dict[str, torch.Tensor] = {}
    created = []
    g = np.random.default_rng(seed)
    for k, shape in schema.items():
        if k in sd:
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != expected {tuple(shape)}")
            out[k] = sd[k]
            continue
        created.append(k)
        if any(z in k for z in _ZERO_INIT) or k.endswith(_ZERO_WEIGHT):
            t = np.zeros(shape, dtype=np.float32)
        elif len(shape) == 1 and ("norm" in k) and k.endswith(".weight"):
            t = np.ones(shape, dtype=np.float32)
        elif len(shape) == 1 and ("norm" in k):
            t = np.zeros(shape, dtype=np.float32)
        else:   # nn.Linear / nn.Conv default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
            wshape = schema.get(k[:-len("bias")] + "weight", shape) if k.endswith(".bias") else shape
            fan_in = int(np.prod(wshape[1:])) if len(wshape) > 1 else int(wshape[0])
            b = 1.0 / max(fan_in, 1) ** 0.5
            t = g.uniform(-b, b, size=shape).astype(np.float32)
        out[k] = torch.from_numpy(t)
    return out, created


def load_unet_state_dict(pretrained_model_path, subfolder: Optional[str] = "unet", resume_from_checkpoint: Optional[str] = None,
                         adapter_weight_path: Optional[str] = None, seed: int = 0):
    """The weight-loading part of the reference's inference.py:152-156,237-240 as one call."""
    sd = load_file(find_weights(pretrained_model_path, subfolder))
    if resume_from_checkpoint:
        sd.update(load_file(find_weights(resume_from_checkpoint)))
    if adapter_weight_path:
        sd.update({"controlnet_adapter." + k: v for k, v in load_file(adapter_weight_path).items()})
    unexpected = [k for k in sd if k not in synth.unet_schema()]
    full, created = inflate({k: v for k, v in sd.items() if k not in unexpected}, synth.unet_schema(), seed)
    return full, {"created": created, "unexpected": unexpected}


def load_controlnet_state_dict(controlnet_path, subfolder: Optional[str] = None):
    sd = load_file(find_weights(controlnet_path, subfolder))
    schema = synth.controlnet_schema()
    missing = [k for k in schema if k not in sd]
    if missing:
        raise KeyError(f"ControlNet checkpoint lacks {len(missing)} keys, e.g. {missing[:3]}")
    return {k: sd[k] for k in schema}
