"""Reverse-mode tape over the C-ABI operators: the host side of the input-gradient (activation) backward that the
null-text optimisation needs (reference: p2p/null_text_optimization.py:133-166 differentiates the guided prev_step loss
w.r.t. the unconditional text embedding through the whole UNet with torch autograd).

The forward launch graph (models/graph.py) stays as it is.  While a tape is recording, every operator call that has a
backward rule is logged with the tensors it read and wrote; `backward()` then walks the log in reverse and calls the
operator's `*_bwd` / `*_dx` primitive of the same backend.  Gradients live in one fp32 buffer per ALLOCATION (a fused
q|k|v tensor, a skip-concat buffer, ...): a view's gradient is the same view of its allocation's buffer, so column slices
and `out=` targets accumulate where they belong without any graph surgery.

The primitives (`ops.gemm_dx`, `ops.geglu_bwd`, `ops.attention_bwd`, `ops.temporal_attention_bwd`, `ops.groupnorm_bwd`,
`ops.layernorm_bwd`, and for trained parameters `ops.gemm_dw`, `ops.colsum_grad`, `ops.layernorm_bwd_params`, `ops.relu_bwd`) are the
kernel-level contract of the backward pass; every accumulation into a gradient buffer goes through `ops.grad_acc`.  On the GPU each is a
HIP kernel of libmotioned (csrc/bwd.hip, attn_bwd.hip, train.hip) or me_gemm itself on transposed weights; tests/emu_ops.py states them on
the CPU (each as the vector-Jacobian product of its forward emulation) and pins this module, through `util.null_optimization` and
`util.adapter_training_grads`, against the reference's own optimisation (tests/golden/null_text.npz, adapter_train.npz).  Still raising
(not differentiated): edited / masked attention segments, shared query items, sharded row orders, the temporal editor's kv_map, the
pad-(0,1,0,1) convolution and 3x3-convolution weights.

Restrictions (asserted): single assignment -- no allocation region is written twice while recording (true for the
single-branch UNet forward; the in-place motion / ControlNet residual adds of the two-branch step are not differentiated).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch


class Tape:
    def __init__(self, backend=None):
        self.backend = backend
        self.entries: List[Tuple[Sequence[torch.Tensor], Callable, Sequence[torch.Tensor]]] = []   # (outputs, backward rule, tensors the call read)
        self.keep: List[torch.Tensor] = []   # every tensor a rule needs stays alive (and un-recycled) until the tape dies

    def record(self, outs: Sequence[torch.Tensor], rule: Callable, *saved: Optional[torch.Tensor]) -> None:
        self.entries.append((tuple(outs), rule, tuple(t for t in saved if t is not None)))
        self.keep.extend(t for t in saved if t is not None)
        self.keep.extend(outs)


_POISON = bool(__import__("os").environ.get("ME_GRAD_POISON"))   # tests: never-zeroed gradient buffers start as NaN, so a read before the first store shows


def _base(t: torch.Tensor) -> torch.Tensor:
    """The allocation `t` lives in.  Operator outputs are fresh contiguous 2-D allocations (or `out=` views of one) on the GPU;
    the CPU emulation sometimes returns a view of a permuted temporary -- such a tensor counts as its own allocation."""
    b = t._base
    return b if (b is not None and b.is_contiguous()) else t


class Grads:
    """fp32 gradient buffers, one per allocation; `view(t)` is the part of it that `t` covers.  Every accumulation is the backend's
    `grad_acc` (a HIP kernel on the GPU)."""

    def __init__(self, backend, trainable: Optional[Dict[int, str]] = None, param_buffers: Optional[Dict[str, torch.Tensor]] = None):
        self.B = backend
        self.buf: Dict[int, torch.Tensor] = {}
        self.fresh: set = set()                       # ids of allocations whose buffer exists but holds nothing yet (first-touch store pending)
        self.keep: List[torch.Tensor] = []
        self.trainable = trainable or {}              # id(packed parameter tensor) -> its key in weights.Packed.cache
        self.params: Dict[str, torch.Tensor] = param_buffers if param_buffers is not None else {}   # key -> fp32 gradient in the packed layout

    def wants(self, t: Optional[torch.Tensor]) -> bool:
        return t is not None and id(t) in self.trainable

    def param(self, t: torch.Tensor) -> torch.Tensor:
        """The fp32 gradient buffer (packed layout) of trainable tensor `t`: a zeroed tensor at first use, or the caller's (a view of the
        trainer's flat gradient bucket)."""
        k = self.trainable[id(t)]
        g = self.params.get(k)
        if g is None:
            g = self.params[k] = torch.zeros(t.shape, dtype=torch.float32, device=t.device)
        return g

    def has(self, t: torch.Tensor) -> bool:
        return id(_base(t)) in self.buf

    @staticmethod
    def _whole(t: torch.Tensor, b: torch.Tensor) -> bool:
        return b is t or (t.shape == b.shape and t.stride() == b.stride() and t.storage_offset() == b.storage_offset())

    def view(self, t: torch.Tensor, first_store: bool = False) -> torch.Tensor:
        """The part of its allocation's gradient buffer that `t` covers.  Buffers are zeroed at their first touch -- unless the caller says the
        touch STORES the whole buffer (`first_store`, and `t` is the whole allocation): then the buffer is allocated uninitialised and marked
        fresh until `take_fresh` hands the store permission out; any other touch of a fresh buffer zeroes it first."""
        b = _base(t)
        g = self.buf.get(id(b))
        whole = self._whole(t, b)
        if g is None:
            if first_store and whole:
                g = torch.full(b.shape, float("nan"), dtype=torch.float32, device=b.device) if _POISON else torch.empty(b.shape, dtype=torch.float32, device=b.device)
                self.fresh.add(id(b))
            else:
                g = torch.zeros(b.shape, dtype=torch.float32, device=b.device)
            self.buf[id(b)] = g
            self.keep.append(b)          # keeps id(b) unique for the life of the buffers
        elif id(b) in self.fresh and not (first_store and whole):
            g.zero_()
            self.fresh.discard(id(b))
        if b is t:
            return g
        return g.as_strided(t.shape, t.stride(), t.storage_offset() - b.storage_offset())

    def take_fresh(self, t: torch.Tensor) -> bool:
        """True exactly once for a buffer `view(t, first_store=True)` left uninitialised: the caller now stores ALL of it."""
        i = id(_base(t))
        if i in self.fresh:
            self.fresh.discard(i)
            return True
        return False

    def add(self, t: Optional[torch.Tensor], g: torch.Tensor, alpha: float = 1.0) -> None:
        if t is None:
            return
        covers = (g.dim() == 2 and t.dim() == 2 and g.shape[0] >= t.shape[0] and g.shape[1] >= t.shape[1]) or (g.dim() != 2 and g.numel() == t.numel())
        v = self.view(t, first_store=covers)
        store = covers and self.take_fresh(t)
        if g.dim() == 2 and v.dim() == 2:
            self.B.grad_acc(v[:g.shape[0], :g.shape[1]], g, alpha, None, store)
        else:
            self.B.grad_acc(v, g.reshape(v.shape), alpha, None, store)


# ---------------------------------------------------------------------------------------------------------------------
# operator rules: (backend, grads) -> None, closed over what the forward call saw
# ---------------------------------------------------------------------------------------------------------------------
def _rule_gemm(B, x, w, out, kw):
    if kw.get("res_rows") or kw.get("res2_rows"):
        # a residual shared by several batch entries (read modulo its row count) would need its gradient SUMMED over the repeats: no rule does that yet
        raise NotImplementedError("autodiff: gemm with a shared residual (res_rows / res2_rows) has no backward rule; record with share=False")

    def rule(G: Grads):
        N, taps, K = w.shape
        dy = G.view(out)
        M = out.shape[0]
        act = kw.get("act", 0)
        for r in (kw.get("res"), kw.get("res2")):           # epilogue residual adds come AFTER the activation: straight through
            if r is not None:
                G.add(r[:M, :dy.shape[1]], dy)
        if act == 1:                                         # ReLU epilogue (adapter block1): needs the output WITHOUT the residuals
            if kw.get("res") is not None or kw.get("res2") is not None:
                raise NotImplementedError("autodiff: ReLU epilogue together with residual terms")
            dy = B.relu_bwd(dy, out)
        elif act:
            raise NotImplementedError("autodiff: gemm with a SiLU epilogue lies on no differentiated path")
        alpha = kw.get("alpha", 1.0)
        if kw.get("geglu"):
            # y = value * gelu(gate) of the biased pre-activation: recompute it (one GEMM) instead of stashing [M, N] per layer
            pre = B.gemm(x, w, M=kw.get("M"), bias=kw.get("bias"))
            dpre = B.geglu_bwd(pre, dy)
        else:
            dpre = dy
        if G.wants(w):                                       # trainable weight: dW[n, tap, k] = sum_m dpre[m, n] * gather(x)[m, tap, k]
            B.gemm_dw(dpre, x, dst=G.param(w), taps=taps, K=K, M=M, alpha=alpha, conv=kw.get("conv"), tconv=kw.get("tconv"))
        if G.wants(kw.get("bias")):
            B.colsum_grad(dpre[:M], dst=G.param(kw["bias"]))
        xv = x[:, :K]
        dst = G.view(xv, first_store=True)
        B.gemm_dx(dpre, w, dst=dst, M=M, alpha=alpha, conv=kw.get("conv"), tconv=kw.get("tconv"), store=G.take_fresh(xv))
    return rule


def _rule_attention(B, q, k, v, out, lse, kw):
    def rule(G: Grads):
        B.attention_bwd(q, k, v, out, G.view(out), dq=G.view(q), dk=G.view(k), dv=G.view(v), lse=lse, **kw)
    return rule


def _rule_tattn(B, q, k, v, out, kw):
    def rule(G: Grads):
        dq, dk, dv = B.temporal_attention_bwd(q, k, v, out, G.view(out), **kw)
        G.add(q, dq)
        G.add(k, dk)
        G.add(v, dv)
    return rule


def _rule_groupnorm(B, x, gamma, beta, out, kw):
    def rule(G: Grads):
        if kw.get("reduce") is not None:
            raise NotImplementedError("autodiff: frame-sharded GroupNorm is not differentiated")
        G.add(x, B.groupnorm_bwd(x, gamma, beta, G.view(out), rows_per_group=kw["rows_per_group"], eps=kw["eps"], silu=kw["silu"], groups=kw.get("groups", 32)))
    return rule


def _rule_layernorm(B, x, gamma, beta, out, eps):
    def rule(G: Grads):
        dy = G.view(out)
        if G.wants(gamma) or G.wants(beta):
            B.layernorm_bwd_params(x, dy, dgamma=G.param(gamma) if G.wants(gamma) else None, dbeta=G.param(beta) if G.wants(beta) else None, eps=eps)
        G.add(x, B.layernorm_bwd(x, gamma, dy, eps=eps))
    return rule


# operators without a backward rule that may run while a tape records: they read nothing the loss is differentiated through (the timestep
# embedding, conv_in on the input latents, layout conversions, allocation) or produce gradient-free side data
_GRADIENT_FREE = {"empty", "timestep_embed", "conv_small", "nchw5_to_rows", "rows_to_nchw5", "rows_to_nchw", "nchw_to_rows", "cfg_ddim", "gaussian_sample",
                  "grad_acc", "gemm_dx", "gemm_dw", "geglu_bwd", "attention_bwd", "temporal_attention_bwd", "groupnorm_bwd", "layernorm_bwd", "layernorm_bwd_params",
                  "colsum_grad", "relu_bwd", "sumsq_absmax", "adamw", "cast_f16", "mse_seed", "invalidate_transposed", "PROFILE", "STEP_PARAMS", "F16"}


class Recorder:
    """Stands in for the `ops` module of models/graph.py while a tape records: same functions, same results, plus the log."""

    recording = True   # models/graph.py keeps single assignment while this is its `ops` (out-of-place residual adds)
    NATIVE = False     # ... and keeps its torch-side data movement (text-embedding cast, clone): the library copies / casts that a recorded plan needs have no backward rule

    def __init__(self, backend, tape: Tape):
        self._b, self._t = backend, tape
        self._written: Dict[int, List[Tuple[int, int, int, int]]] = {}

    def __getattr__(self, name):
        # a compute operator without a rule on a differentiated path would cut the gradient silently: only the gradient-free ones pass
        if name.startswith("_") or name in _GRADIENT_FREE:
            return getattr(self._b, name)
        raise NotImplementedError(f"autodiff: operator `{name}` has no backward rule and is not known to be gradient-free; it may not run while a tape records")

    # -- single-assignment check: (row0, row1, col0, col1) boxes written per allocation must not overlap
    def _mark(self, t: torch.Tensor) -> None:
        b = _base(t)
        off = t.storage_offset() - b.storage_offset()
        ld = t.stride(0) if t.dim() == 2 else t.shape[-1]
        r0, c0 = (off // ld, off % ld) if ld else (0, 0)
        box = (r0, r0 + t.shape[0], c0, c0 + (t.shape[1] if t.dim() == 2 else ld))
        for o in self._written.setdefault(id(b), []):
            if box[0] < o[1] and o[0] < box[1] and box[2] < o[3] and o[2] < box[3]:
                raise RuntimeError("autodiff: a tensor region is written twice while recording (the tape assumes single assignment)")
        self._written[id(b)].append(box)
        self._t.keep.append(b)

    @staticmethod
    def _own(out, kw):
        """An output that is not an `out=` target must be an allocation of its own (the CPU emulation may hand back a view)."""
        return out if (kw.get("out") is not None or out._base is None) else out.contiguous().clone()

    def gemm(self, x, w, **kw):
        out = self._own(self._b.gemm(x, w, **kw), kw)
        self._mark(out)
        self._t.record([out], _rule_gemm(self._b, x, w, out, kw), x, w, kw.get("res"), kw.get("res2"), kw.get("bias"))
        return out

    def attention(self, q, k, v, **kw):
        lse = torch.empty((kw["n_items"] * kw["nq"], kw["heads"]), dtype=torch.float32, device=q.device)   # stashed for the fused backward
        out = self._own(self._b.attention(q, k, v, lse=lse, **kw), kw)
        self._mark(out)
        kw = {a: b for a, b in kw.items() if a != "out"}
        self._t.record([out], _rule_attention(self._b, q, k, v, out, lse, kw), q, k, v, lse, kw.get("mask"))
        return out

    def temporal_attention(self, q, k, v, **kw):
        out = self._own(self._b.temporal_attention(q, k, v, **kw), kw)
        self._mark(out)
        self._t.record([out], _rule_tattn(self._b, q, k, v, out, kw), q, k, v)
        return out

    def groupnorm(self, x, gamma, beta, **kw):
        out = self._own(self._b.groupnorm(x, gamma, beta, **kw), kw)
        self._mark(out)
        self._t.record([out], _rule_groupnorm(self._b, x, gamma, beta, out, {a: b for a, b in kw.items() if a != "out"}), x, gamma, beta)
        return out

    def layernorm(self, x, gamma, beta, eps=1e-5):
        out = self._own(self._b.layernorm(x, gamma, beta, eps), {})
        self._mark(out)
        self._t.record([out], _rule_layernorm(self._b, x, gamma, beta, out, eps), x, gamma, beta)
        return out

    def copy_rows(self, y, x):
        out = self._b.copy_rows(y, x)
        self._mark(y)
        self._t.record([y], lambda G: G.add(x, G.view(y)), x)
        return out

    def axpy_rows(self, y, x, a_, alpha=1.0):
        if _base(y) is _base(x):
            raise RuntimeError("autodiff: in-place axpy_rows is not differentiated")
        out = self._b.axpy_rows(y, x, a_, alpha)
        self._mark(y)
        self._t.record([y], lambda G: (G.add(x, G.view(y)), G.add(a_, G.view(y), alpha)), x, a_)
        return out


class record:
    """`with autodiff.record(graph) as tape:` -- the module's `ops` is a Recorder for the duration."""

    def __init__(self, module):
        self.m = module

    def __enter__(self) -> Tape:
        self.tape = Tape(self.m.ops)
        self.saved = self.m.ops
        self.m.ops = Recorder(self.saved, self.tape)
        return self.tape

    def __exit__(self, *exc):
        self.m.ops = self.saved
        return False


def backward(tape: Tape, seeds: Sequence[Tuple[torch.Tensor, torch.Tensor]], trainable: Optional[Dict[int, str]] = None, backend=None,
             param_buffers: Optional[Dict[str, torch.Tensor]] = None, seed_scale: float = 1.0, wrt: Optional[Sequence[torch.Tensor]] = None) -> Grads:
    """seeds: (tensor the forward produced, gradient of the loss w.r.t. it), entered as seed_scale * gradient (the loss scale).  Returns
    the gradient store; `G.view(t)` of any tensor the forward read is its gradient (zeros if nothing depended on it).  trainable:
    id(packed parameter tensor) -> key (weights.Packed.trainable_ids); their gradients, in the packed layout, end up in `G.params[key]`
    (accumulated into `param_buffers[key]` when the caller provides the buffers).  backend: the ops module the tape recorded on.

    wrt: the input tensors whose gradients the caller will read.  With `wrt` and / or `trainable` given, the walk is PRUNED to the calls that
    lie downstream of one of them: the null-text optimisation differentiates w.r.t. the text rows only, which first enter at the first
    cross-attention -- conv_in's successor resnet and the first block's self-attention need no backward at all; the adapter's parameters
    feed the up path only -- the UNet's down path and mid block need none.  (Without both, every call with a gradient is walked.)"""
    G = Grads(backend if backend is not None else tape.backend, trainable, param_buffers)
    relevant = None
    if wrt is not None or trainable:
        reach = {id(_base(t)) for t in (wrt or ())}
        tr = trainable or {}
        relevant = []
        for outs, _, ins in tape.entries:
            r = any(id(_base(t)) in reach or id(t) in tr for t in ins)
            relevant.append(r)
            if r:
                reach.update(id(_base(o)) for o in outs)
    for t, g in seeds:
        G.add(t, g, seed_scale)
    for i in range(len(tape.entries) - 1, -1, -1):
        outs, rule, _ = tape.entries[i]
        if (relevant is None or relevant[i]) and any(G.has(o) for o in outs):
            rule(G)
    return G
