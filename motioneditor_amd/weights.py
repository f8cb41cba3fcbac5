"""Weight packer: reference state-dict (reference key schema, ``SURVEY.md §8b``) -> device-resident
fp16 tensors in the layouts the kernels consume.

  Linear  [N, K]            -> [N, 1, K]
  Conv2d  [Co, Ci, 3, 3]    -> [Co, 9, Ci]   (tap = ky*3 + kx, channels-last K runs)
  Conv2d  [Co, Ci, 1, 1]    -> [Co, 1, Ci]
  Conv1d  [Co, Ci, k]       -> [Co, k, Ci]   (TemporalConv, k in {1, 3})
  GEGLU   [8C, C] (+bias)   -> rows interleaved in blocks of 16 value rows / 16 gate rows
  q|k|v (or k|v)            -> one fused [3C, 1, K] projection
  all ResnetBlock2D.time_emb_proj -> one [sum(Cout), 1, 1280] projection

Accepts numpy arrays or torch tensors (any float dtype) as values -- e.g. a real
``diffusion_pytorch_model.safetensors`` state dict, an accelerate checkpoint, or
``synth.synth_state_dict`` output.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Mapping, Optional

import numpy as np
import torch


def _t(v) -> torch.Tensor:
    if isinstance(v, np.ndarray):
        return torch.from_numpy(v)
    return v.detach().cpu()


class Packed:
    """name -> device fp16 tensor, packed lazily from the host state dict."""

    def __init__(self, state: Mapping[str, object], device, prefix: str = "", dtype=torch.float16):
        self.state = state
        self.dtype = dtype  # fp16 on the GPU; tests/emu_ops.py checks the launch graphs in fp32 on CPU
        self.device = torch.device(device)
        self.prefix = prefix
        self.cache: Dict[str, torch.Tensor] = {}

    def make_private(self) -> None:
        """Give this store its own (mutable) name -> tensor mapping before the first parameter is replaced: the caller's state dict -- often
        shared by several models -- must not change under it."""
        if not getattr(self, "_private", False):
            self.state = dict(self.state)
            self._private = True

    def has(self, name: str) -> bool:
        return (self.prefix + name) in self.state

    def raw(self, name: str) -> torch.Tensor:
        return _t(self.state[self.prefix + name]).float()

    def _put(self, key: str, t: torch.Tensor) -> torch.Tensor:
        d = t.to(self.dtype).contiguous().to(self.device)
        self.cache[key] = d
        return d

    @staticmethod
    def _as_taps(w: torch.Tensor) -> torch.Tensor:
        if w.dim() == 2:      # Linear
            return w[:, None, :]
        if w.dim() == 4:      # Conv2d [Co, Ci, kh, kw] -> [Co, kh*kw, Ci]
            co, ci, kh, kw = w.shape
            return w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
        if w.dim() == 3:      # Conv1d [Co, Ci, k] -> [Co, k, Ci]
            return w.permute(0, 2, 1)
        raise ValueError(f"unsupported weight rank {w.dim()}")

    @staticmethod
    def _geglu_perm(n_out: int) -> torch.Tensor:
        """packed row p = q*32 + r  <-  value row q*16 + r (r < 16) | gate row n_out + q*16 + (r-16)."""
        q = torch.arange(n_out // 16)
        val = (q[:, None] * 16 + torch.arange(16)[None]).reshape(-1, 16)
        gate = val + n_out
        return torch.cat([val, gate], dim=1).reshape(-1)

    def packed_f32(self, key: str) -> torch.Tensor:
        """The packed tensor of cache key `kind:name|name...` in fp32 on the host, built from the state dict (what _put then casts):
        also the fp32 master copy of a trained parameter in the layout its kernels read."""
        kind, _, names = key.partition(":")
        ns = names.split("|")
        if kind == "vec":
            return self.raw(names).reshape(-1)
        if kind == "mat":
            return self._as_taps(self.raw(names)).contiguous()
        if kind == "fused":
            return torch.cat([self._as_taps(self.raw(n)) for n in ns], dim=0).contiguous()
        if kind == "fvec":
            return torch.cat([self.raw(n).reshape(-1) for n in ns])
        if kind == "geglu":
            w = self.raw(names)
            return w[self._geglu_perm(w.shape[0] // 2)][:, None, :].contiguous()
        if kind == "gegluv":
            b = self.raw(names)
            return b[self._geglu_perm(b.shape[0] // 2)].contiguous()
        raise KeyError(key)

    def _get(self, key: str) -> torch.Tensor:
        hit = self.cache.get(key)
        return hit if hit is not None else self._put(key, self.packed_f32(key))

    def vec(self, name: str) -> torch.Tensor:
        """bias / norm parameter as fp16 [n]."""
        return self._get("vec:" + name)

    def mat(self, name: str) -> torch.Tensor:
        return self._get("mat:" + name)

    def mat32(self, name: str) -> torch.Tensor:
        """fp32 [N, taps, K] (conv_small reads its tiny weights through the scalar cache)."""
        k = "mat32:" + name
        if k not in self.cache:
            self.cache[k] = self._as_taps(self.raw(name)).contiguous().to(self.device)
        return self.cache[k]

    def vec32(self, name: str) -> torch.Tensor:
        k = "vec32:" + name
        if k not in self.cache:
            self.cache[k] = self.raw(name).reshape(-1).contiguous().to(self.device)
        return self.cache[k]

    def fused(self, names: Iterable[str]) -> torch.Tensor:
        """Row-concatenation of several projections that share an input (q|k|v, k|v)."""
        return self._get("fused:" + "|".join(names))

    def fused_vec(self, names: Iterable[str]) -> torch.Tensor:
        return self._get("fvec:" + "|".join(names))

    def geglu_mat(self, name: str) -> torch.Tensor:
        return self._get("geglu:" + name)

    def geglu_vec(self, name: str) -> torch.Tensor:
        return self._get("gegluv:" + name)

    def ln_fold(self, norm: str, names: Iterable[str], geglu: bool = False, bias: Optional[str] = None):
        """LayerNorm `norm` folded into the projection(s) `names` that consume it (me_gemm_args.ln_stats, ABI 9):
            LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (W beta + b),   W' = W diag(gamma).
        Returns (W' in the packed layout and dtype of the projection, colsum(W') fp32 [N] -- taken over the ROUNDED W', so that the identity holds
        exactly for the numbers the kernel multiplies --, W beta + b fp32 [N]).  geglu: the value / gate interleaved packing of ff.net.0.proj."""
        names = list(names)
        key = ("lnwg:" if geglu else "lnw:") + "|".join([norm + ".weight", norm + ".bias", *names, *([bias] if bias else [])])
        hit = self.cache.get(key)
        if hit is None:
            gamma, beta = self.raw(norm + ".weight").reshape(-1), self.raw(norm + ".bias").reshape(-1)
            if geglu:
                w = self.packed_f32("geglu:" + names[0])
                b = self.packed_f32("gegluv:" + bias) if bias else None
            else:
                w = torch.cat([self._as_taps(self.raw(n)) for n in names], dim=0)
                b = self.raw(bias).reshape(-1) if bias else None
            if w.shape[1] != 1:
                raise ValueError("ln_fold: dense projections only")
            wq = (w * gamma[None, None, :]).to(self.dtype).contiguous()
            colsum = wq.float().sum(dim=(1, 2))
            cvec = w[:, 0, :] @ beta
            if b is not None:
                cvec = cvec + b
            hit = (wq.to(self.device), colsum.contiguous().to(self.device), cvec.float().contiguous().to(self.device))
            self.cache[key] = hit   # type: ignore[assignment]
        return hit

    def is_zero(self, *names: str) -> bool:
        """True when every named tensor is exactly zero (the reference zero-initialises TemporalConv
        and never trains it, resnet_2d.py:15-16: its branch can be skipped for real checkpoints)."""
        k = "zero:" + "|".join(names)
        if k not in self.cache:
            self.cache[k] = all(bool((self.raw(n) == 0).all()) for n in names)  # type: ignore[assignment]
        return bool(self.cache[k])

    # ---- training support (adapter training step, train_adaptor.py:364-368): packed parameter <-> reference parameter ----
    def trainable_ids(self, name_prefix: str) -> Dict[int, str]:
        """id(packed tensor) -> cache key, for every packed tensor built from parameters whose reference name starts with name_prefix."""
        out = {}
        for key, t in self.cache.items():
            kind, _, names = key.partition(":")
            if isinstance(t, torch.Tensor) and kind in ("mat", "vec", "fused", "fvec", "geglu", "gegluv") and all(n.startswith(name_prefix) for n in names.split("|")):
                out[id(t)] = key
        return out

    def update(self, name: str, value: torch.Tensor) -> None:
        """Replace a parameter (reference name, reference layout); every packed tensor built from it is dropped and re-packed on its
        next use (and its cached transpose forgotten).  The store takes a private copy of the mapping first (make_private)."""
        from . import ops
        self.make_private()
        self.state[self.prefix + name] = value.detach().cpu().clone()   # type: ignore[index]
        stale = [k for k in self.cache if name in k.partition(":")[2].split("|")]
        if self.device.type == "cuda":
            ops.invalidate_transposed([self.cache[k] for k in stale if isinstance(self.cache[k], torch.Tensor) and self.cache[k].dim() == 3])
        for key in stale:
            del self.cache[key]

    def rehome(self, key: str, storage: torch.Tensor) -> torch.Tensor:
        """Move packed tensor `key` into `storage` (a flat slice of a caller-owned bucket of the same dtype and size): the cache hands out the
        bucket view from now on, so that one kernel can refresh every trained tensor from its fp32 master (util.AdapterTrainer)."""
        t = self._get(key)
        if storage.dtype != t.dtype or storage.numel() != t.numel() or not storage.is_contiguous():
            raise ValueError("rehome: storage must be a contiguous slice of the packed tensor's dtype and size")
        storage.copy_(t.reshape(-1))
        v = storage.view(t.shape)
        self.cache[key] = v
        return v

    def unpack_grad(self, key: str, g: torch.Tensor) -> Dict[str, torch.Tensor]:
        """A tensor in the PACKED layout of `key` (a gradient, or an fp32 master copy) -> {reference parameter name: the same values in the
        reference's own layout}: the inverse of packed_f32."""
        kind, _, names = key.partition(":")
        names_l = names.split("|")
        g = g.float().cpu()

        def untap(n: str, t: torch.Tensor) -> torch.Tensor:   # [N, taps, K] -> the layout of raw(n)
            shp = self.raw(n).shape
            if len(shp) == 2:
                return t[:, 0, :]
            if len(shp) == 4:
                return t.reshape(shp[0], shp[2], shp[3], shp[1]).permute(0, 3, 1, 2)
            return t.permute(0, 2, 1)

        if kind == "mat":
            return {names: untap(names, g)}
        if kind == "vec":
            return {names: g.reshape(self.raw(names).shape)}
        if kind in ("fused", "fvec"):
            out, r0 = {}, 0
            for n in names_l:
                rows = self.raw(n).shape[0]
                out[n] = untap(n, g[r0:r0 + rows]) if kind == "fused" else g[r0:r0 + rows]
                r0 += rows
            return out
        if kind in ("geglu", "gegluv"):
            n_out = g.shape[0] // 2
            inv = torch.empty(2 * n_out, dtype=torch.long)
            inv[self._geglu_perm(n_out)] = torch.arange(2 * n_out)
            return {names: (g[inv][:, 0, :] if kind == "geglu" else g[inv])}
        raise KeyError(key)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.cache.values() if isinstance(t, torch.Tensor))
