"""Host logic of the backward pass (motioneditor_amd/autodiff.py, weights.Packed.unpack_grad) on its own, without a model."""
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import emu_ops  # noqa: E402
from motioneditor_amd import autodiff  # noqa: E402
from motioneditor_amd.weights import Packed  # noqa: E402


class _Mod:   # stands for models/graph.py: a module whose `ops` the recorder replaces
    ops = emu_ops


def test_views_and_out_targets_accumulate_in_their_allocation_and_match_autograd():
    """q|k|v-style column views of one fused GEMM output, a residual epilogue, an `out=` target inside a concat buffer: the tape's
    gradients equal torch autograd's on the same little graph."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(12, 16, generator=g)
    w1 = torch.randn(24, 1, 16, generator=g) * 0.3
    w2 = torch.randn(16, 1, 8, generator=g) * 0.3
    gm, bt = torch.ones(16), torch.zeros(16)
    with autodiff.record(_Mod) as tape:
        ops = _Mod.ops
        fused = ops.gemm(x, w1)                                  # [12, 24]
        a, b, c = fused[:, :8], fused[:, 8:16], fused[:, 16:]
        cat = torch.empty(12, 32)
        ops.gemm(a, w2, out=cat[:, :16], res=x)                  # writes the left half of the concat buffer
        ops.copy_rows(cat[:, 16:], ops.layernorm(ops.gemm(b, w2), gm, bt))
        y = ops.gemm(cat, torch.cat([w1[:8], w1[:8]], dim=2), res=c)   # [12, 8] reads the whole buffer
    seed = torch.randn(12, 8, generator=g)
    G = autodiff.backward(tape, [(y, seed)])
    # the same graph under torch autograd
    xa = x.clone().requires_grad_(True)
    f = xa @ w1[:, 0].t()
    left = f[:, :8] @ w2[:, 0].t() + xa
    right = torch.nn.functional.layer_norm(f[:, 8:16] @ w2[:, 0].t(), (16,), gm, bt)
    ya = torch.cat([left, right], 1) @ torch.cat([w1[:8], w1[:8]], dim=2)[:, 0].t() + f[:, 16:]
    ya.backward(seed)
    assert torch.allclose(G.view(x), xa.grad, atol=1e-5, rtol=1e-4)


def test_backward_walk_is_pruned_to_what_lies_downstream_of_the_requested_input():
    """`wrt=`: calls that do not depend on the requested input are not walked (the null-text optimisation asks for the text rows only, which
    first enter at the first cross-attention), the gradient of the requested input is unchanged, and never-zeroed (NaN-poisoned in the tests)
    first-touch buffers never leak into it."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(12, 16, generator=g)          # "latents": the prefix below depends on x only
    txt = torch.randn(12, 8, generator=g)         # "text rows": enter at the second GEMM pair
    w1 = torch.randn(16, 1, 16, generator=g) * 0.3
    wt = torch.randn(16, 1, 8, generator=g) * 0.3
    w2 = torch.randn(8, 1, 16, generator=g) * 0.3
    gm, bt = torch.ones(16), torch.zeros(16)
    called = []
    real = {n: getattr(emu_ops, n) for n in ("gemm_dx", "layernorm_bwd")}

    class Spy:   # the backend the tape calls: counts the backward primitives
        def __getattr__(self, name):
            f = getattr(emu_ops, name)
            if name in real:
                def wrapped(*a, **k):
                    called.append(name)
                    return f(*a, **k)
                return wrapped
            return f

    class M:
        ops = Spy()

    with autodiff.record(M) as tape:
        ops = M.ops
        h = ops.layernorm(ops.gemm(x, w1), gm, bt)            # prefix: two calls that never see the text
        t = ops.gemm(txt, wt)                                  # the text enters
        y = ops.gemm(ops.layernorm(ops.gemm(h, w1, res=t), gm, bt), w2)
    seed = torch.randn(12, 8, generator=g)
    G_all = autodiff.backward(tape, [(y, seed)])
    n_all = len(called)
    del called[:]
    G = autodiff.backward(tape, [(y, seed)], wrt=[txt])
    assert torch.equal(G.view(txt), G_all.view(txt)) and torch.isfinite(G.view(txt)).all()
    assert len(called) < n_all and called.count("layernorm_bwd") == 1     # the prefix LayerNorm (and its GEMM) are not differentiated
    ta = txt.clone().requires_grad_(True)
    ha = torch.nn.functional.layer_norm(x @ w1[:, 0].t(), (16,), gm, bt)
    ya = torch.nn.functional.layer_norm(ha @ w1[:, 0].t() + ta @ wt[:, 0].t(), (16,), gm, bt) @ w2[:, 0].t()
    ya.backward(seed)
    assert torch.allclose(G.view(txt), ta.grad, atol=1e-5, rtol=1e-4)


def test_writing_a_region_twice_while_recording_is_refused():
    x = torch.randn(4, 8)
    w = torch.randn(8, 1, 8)
    buf = torch.empty(4, 16)
    with autodiff.record(_Mod):
        _Mod.ops.gemm(x, w, out=buf[:, :8])
        _Mod.ops.gemm(x, w, out=buf[:, 8:])            # disjoint columns of one allocation: fine
        with pytest.raises(RuntimeError, match="written twice"):
            _Mod.ops.gemm(x, w, out=buf[:, 4:12])
    assert _Mod.ops is emu_ops                           # the module's ops are restored


@pytest.mark.parametrize("kind", ["mat_linear", "mat_conv1d", "mat_conv2d", "fused", "geglu", "gegluv", "vec"])
def test_unpack_grad_is_the_adjoint_of_the_packing(kind):
    """<G, pack(W)> == <unpack(G), W> for random G, W: the gradient w.r.t. a packed tensor mapped back to the reference's parameters."""
    g = torch.Generator().manual_seed(3)
    R = lambda *s: torch.randn(*s, generator=g)   # noqa: E731
    state = {"a.weight": R(12, 8), "b.weight": R(6, 8), "c1.weight": R(5, 7, 3), "c2.weight": R(4, 6, 3, 3), "ff.weight": R(64, 8), "ff.bias": R(64), "n.bias": R(9)}
    P = Packed(state, "cpu", dtype=torch.float32)
    packed, key = {"mat_linear": (P.mat("a.weight"), "mat:a.weight"), "mat_conv1d": (P.mat("c1.weight"), "mat:c1.weight"),
                   "mat_conv2d": (P.mat("c2.weight"), "mat:c2.weight"), "fused": (P.fused(["a.weight", "b.weight"]), "fused:a.weight|b.weight"),
                   "geglu": (P.geglu_mat("ff.weight"), "geglu:ff.weight"), "gegluv": (P.geglu_vec("ff.bias"), "gegluv:ff.bias"),
                   "vec": (P.vec("n.bias"), "vec:n.bias")}[kind]
    assert P.trainable_ids("")[id(packed)] == key
    G = R(*packed.shape)
    back = P.unpack_grad(key, G)
    lhs = float((G * packed).sum())
    rhs = sum(float((back[n] * state[n]).sum()) for n in back)
    assert set(back) == set(key.partition(":")[2].split("|")) and all(back[n].shape == state[n].shape for n in back)
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_packed_update_repacks_on_next_use():
    state = {"a.weight": torch.zeros(4, 8), "b.weight": torch.ones(2, 8)}
    P = Packed(state, "cpu", dtype=torch.float32)
    f0 = P.fused(["a.weight", "b.weight"])
    P.update("a.weight", torch.full((4, 8), 2.0))
    f1 = P.fused(["a.weight", "b.weight"])
    assert f1 is not f0 and float(f1[:4].min()) == 2.0 and float(f1[4:].max()) == 1.0
