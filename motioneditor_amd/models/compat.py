"""The nn.Module / diffusers ModelMixin surface that the reference's harness touches on its models before the denoising loop
(inference.py:152-248): `requires_grad_`, `enable_xformers_memory_efficient_attention`, `enable_gradient_checkpointing`, `.to(device,
dtype)`, `eval()`, `named_modules()`, `named_parameters()`, `unet.controlnet_adapter.load_state_dict(...)`.  The classes here are not
nn.Modules -- their weights live packed in `weights.Packed` and every forward is a libmotioned launch graph -- so most of these are
state-free acknowledgements; the ones that carry meaning (`to` another device, `load_state_dict`, the parameter iterators) act on
the weight store."""
from __future__ import annotations

from typing import Iterator, Mapping, Tuple

import torch

from ..weights import Packed, _t


class ModuleShims:
    """Mixin for classes with a `P: weights.Packed` and a `device`."""

    training = False

    # -- flags of the reference harness that have no effect on a launch graph without autograd state
    def requires_grad_(self, requires_grad: bool = True):        # inference.py:159-162: freeze everything; nothing here tracks gradients
        return self

    def eval(self):                                              # inference.py:248
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def enable_xformers_memory_efficient_attention(self, *a, **k):   # inference.py:164-168: attention is always the fused me_attn kernel
        return self

    def enable_gradient_checkpointing(self):                     # inference.py:170-171 (training-time memory knob; the tape recomputes GEGLU only)
        return self

    # -- device / dtype
    @property
    def dtype(self):
        return torch.float16

    @dtype.setter
    def dtype(self, v):      # the constructors assign it
        pass

    def to(self, *args, **kwargs):
        """`.to(device)`, `.to(dtype)`, `.to(device, dtype=...)` (inference.py:215-217).  The arithmetic is fp16 storage / fp32 accumulation
        whatever dtype is asked for (float16 and float32 are accepted, anything else raises); a different device re-homes the weight store."""
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        if dtype is not None and dtype not in (torch.float16, torch.float32):
            raise NotImplementedError(f"{type(self).__name__}.to(dtype={dtype}): the kernels compute in fp16 storage / fp32 accumulation")
        if device is not None and torch.device(device) != self.device:
            self.P = Packed(self.P.state, device, prefix=self.P.prefix, dtype=self.P.dtype)
            self.device = torch.device(device)
        return self

    def half(self):
        return self

    def float(self):
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda" if device is None else device))

    # -- parameters, by the reference's names and in the reference's layouts (host tensors)
    def state_dict(self) -> "dict[str, torch.Tensor]":
        return {k[len(self.P.prefix):]: _t(v) for k, v in self.P.state.items() if k.startswith(self.P.prefix)}

    def named_parameters(self, prefix: str = "", recurse: bool = True) -> Iterator[Tuple[str, torch.Tensor]]:
        for k, v in self.state_dict().items():
            yield (prefix + "." if prefix else "") + k, v

    def parameters(self, recurse: bool = True) -> Iterator[torch.Tensor]:
        for _, v in self.named_parameters():
            yield v

    def named_modules(self, memo=None, prefix: str = "") -> Iterator[Tuple[str, object]]:
        """Module paths implied by the parameter names (every dotted prefix once, parents first), as nn.Module.named_modules lists them."""
        seen = {""}
        yield prefix, self
        for k in self.state_dict():
            parts = k.split(".")[:-1]
            for i in range(1, len(parts) + 1):
                name = ".".join(parts[:i])
                if name not in seen:
                    seen.add(name)
                    yield (prefix + "." if prefix else "") + name, _SubModule(self, name + ".")

    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = True):
        return _load(self, "", state_dict, strict)


class _SubModule:
    """A dotted sub-tree of a model's parameters (`unet.controlnet_adapter`): enough of nn.Module for the harness -- `load_state_dict`
    (inference.py:237-240 loads the stage-2 adapter checkpoint this way), `state_dict`, `named_parameters`."""

    def __init__(self, root, prefix: str):
        self._root, self._prefix = root, prefix

    def state_dict(self):
        n = len(self._prefix)
        return {k[n:]: v for k, v in self._root.state_dict().items() if k.startswith(self._prefix)}

    def named_parameters(self, prefix: str = "", recurse: bool = True):
        for k, v in self.state_dict().items():
            yield (prefix + "." if prefix else "") + k, v

    def parameters(self, recurse: bool = True):
        for _, v in self.named_parameters():
            yield v

    def load_state_dict(self, state_dict, strict: bool = True):
        return _load(self._root, self._prefix, state_dict, strict)

    def requires_grad_(self, requires_grad: bool = True):
        return self


def _load(model, prefix: str, state_dict, strict: bool):
    """nn.Module.load_state_dict semantics on the weight store: shapes must match; strict = every parameter under `prefix` present and
    nothing else.  Packed tensors built from a replaced parameter are dropped and re-packed at their next use."""
    P = model.P
    have = {k[len(P.prefix) + len(prefix):] for k in P.state if k.startswith(P.prefix + prefix)}
    missing = sorted(have - set(state_dict))
    unexpected = sorted(set(state_dict) - have)
    if strict and (missing or unexpected):
        raise RuntimeError(f"load_state_dict: missing {missing[:3]}{'...' if len(missing) > 3 else ''}, unexpected {unexpected[:3]}{'...' if len(unexpected) > 3 else ''}")
    P.make_private()
    for k, v in state_dict.items():
        if k not in have:
            continue
        v = _t(v)
        old = _t(P.state[P.prefix + prefix + k])
        if tuple(v.shape) != tuple(old.shape):
            raise RuntimeError(f"load_state_dict: size mismatch for {prefix + k}: {tuple(v.shape)} vs {tuple(old.shape)}")
        P.update(prefix + k, v)
    return type("_IncompatibleKeys", (), {"missing_keys": missing, "unexpected_keys": unexpected})()
