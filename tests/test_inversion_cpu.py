"""DDIM inversion (forward half; reference util.py:77-130 as inference.py:289-293 calls it):
oracle vs the reference's golden vectors (tests/golden/inversion.npz, written by oracle/make_golden.py from the
reference's own UNet with normal_infer=True and its in-tree next_step), and the product's host logic
(motioneditor_amd/util.py + the normal_infer launch graph) on the emulated ABI vs the same goldens."""
import numpy as np
import pytest
import torch

import emu_ops
from conftest import GOLD, max_rel
from motioneditor_amd import synth, util
from motioneditor_amd.models import graph
from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
from motioneditor_amd.schedulers import DDIMScheduler
from oracle import ref_cpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "inversion.npz")


def test_oracle_next_step_matches_reference_vectors(gold):
    d = ref_cpu.DDIM()
    x, eps = T(gold["ns_x"]), T(gold["ns_eps"])
    for t in (1, 21, 501, 981):
        want = T(gold[f"next_{t}"])
        assert max_rel(d.next_step(eps, t, x), want) < 1e-5
        ca, cb = d.next_coeffs(t)
        assert max_rel(ca * x + cb * eps, want) < 1e-4


def test_product_next_step_matches_reference_vectors(gold):
    s = DDIMScheduler()
    s.set_timesteps(50)
    x, eps = T(gold["ns_x"]), T(gold["ns_eps"])
    for t in (1, 21, 501, 981):
        assert max_rel(util.next_step(eps, t, x, s), T(gold[f"next_{t}"])) < 1e-4
    assert list(s.timesteps)[-3:] == [41, 21, 1]   # the inversion walks these first (util.py:119)


def test_oracle_normal_infer_unet_and_loop_match_reference(gold, unet_sd_torch):
    c = synth.make_case_inputs("inversion", B=1, f=8, h=16, w=16)
    with torch.no_grad():
        out = ref_cpu.unet_forward(unet_sd_torch, c["sample"], 1, c["ehs"], normal_infer=True)
        assert max_rel(out, T(gold["unet_normal_infer_t1"])) < 2e-5
        lats = ref_cpu.ddim_loop(unet_sd_torch, ref_cpu.DDIM(), c["sample"], 3, c["ehs"], normal_infer=True)
    for i in range(3):
        assert max_rel(lats[i + 1], T(gold[f"loop_latent_{i + 1}"])) < 2e-5


@pytest.fixture()
def emu(monkeypatch):
    monkeypatch.setattr(graph, "ops", emu_ops)
    import motioneditor_amd.models.unet_2d_condition as u
    monkeypatch.setattr(u, "ops", emu_ops)
    monkeypatch.setattr(util, "ops", emu_ops)
    return emu_ops


def test_product_graph_normal_infer_and_ddim_loop_match_reference(emu, gold, unet_sd_np):
    """The launch graph with normal_infer=True (attn1 = per-frame self-attention segments) and util.ddim_loop on the
    emulated ABI."""
    c = synth.make_case_inputs("inversion", B=1, f=8, h=16, w=16)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    out = unet(c["sample"], 1, c["ehs"], normal_infer=True).sample
    assert max_rel(out, T(gold["unet_normal_infer_t1"])) < 2e-4
    # normal_infer differs from the sparse-causal forward (otherwise the flag would be untested)
    assert max_rel(unet(c["sample"], 1, c["ehs"]).sample, T(gold["unet_normal_infer_t1"])) > 1e-2

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet = unet
    s = DDIMScheduler()
    s.set_timesteps(50)
    lats = util.ddim_inversion(pipe, s, c["sample"], 3, normal_infer=True, text_embeddings=c["ehs"])
    assert len(lats) == 4
    for i in range(3):
        assert max_rel(lats[i + 1], T(gold[f"loop_latent_{i + 1}"])) < 2e-4


def test_normal_infer_rejects_editors_and_controlnet_residuals(emu, unet_sd_np):
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    c = synth.make_case_inputs("two", B=4, f=8, h=16, w=16)
    with pytest.raises(NotImplementedError):
        unet(c["sample"], 1, c["ehs"], normal_infer=True, down_block_additional_residuals=c["down_res"], mid_block_additional_residual=c["mid_res"])


def test_oracle_null_text_optimization_matches_the_reference_class_as_written():
    """SURVEY.md 8f rank 1, the OTHER half: tests/golden/null_text.npz holds the unconditional embeddings that the reference's own
    MyNullInversion.null_optimization (p2p/null_text_optimization.py:133-166, compiled from the reference file as written, with
    NUM_DDIM_STEPS = 2) produces around the reference UNet, and the first inner step's gradient.  The oracle restatement must
    land on them: the gradient to 1e-3, the embeddings wherever the gradient is above rounding level (Adam's first update is
    lr * sign(g)).  The HIP path for this row (activation backward of the UNet) is not built yet; this pins its oracle."""
    g = np.load(GOLD / "null_text.npz")
    sd = {k: T(v) for k, v in synth.synth_state_dict(synth.unet_schema()).items()}
    lat = [t for t in T(g["latents"])]
    grads = []
    out = ref_cpu.null_optimization(sd, ref_cpu.DDIM(), lat, T(g["context"]), 2, 1e-5, num_steps=2, grads=grads)
    g0 = T(g["grad0"])
    assert float((grads[0] - g0).norm() / g0.norm()) < 1e-3
    big = g0.abs() > 1e-3 * g0.abs().max()
    for mine, ref in zip(out, T(g["uncond_out"])):
        assert float(((mine - ref).abs() * big).max()) < 2e-3
    assert float((out[0] - T(g["context"])[:1]).abs().max()) > 5e-3      # the embedding did move


def test_oracle_adapter_training_gradients_match_the_reference_unet():
    """the arithmetic of one adapter training step (train_adaptor.py:364-368 -- UNet forward with the
    ControlNet residuals on one clip, mse against the noise, backward into controlnet_adapter.*) as the REFERENCE UNet computes
    it under autograd (tests/golden/adapter_train.npz); the oracle must reproduce the loss and every parameter's gradient.
    The HIP path for this row is not built; this pins its oracle."""
    g = np.load(GOLD / "adapter_train.npz")
    sd = {k: T(v) for k, v in synth.synth_state_dict(synth.unet_schema()).items()}
    names = [str(n) for n in g["names"]]
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    F32 = lambda k: T(g[k].astype(np.float32))   # inputs are stored as fp16 (they are fp16-representable)  # noqa: E731
    down = [F32(f"down{i}") for i in range(12)]
    pred = ref_cpu.unet_forward(sd, F32("noisy"), int(g["t"]), F32("ehs"), down, F32("mid"))
    loss = torch.nn.functional.mse_loss(pred, F32("noise"))
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * float(g["loss"])
    grads = torch.autograd.grad(loss, [sd[k] for k in names])
    norms = np.array([float(x.norm()) for x in grads])
    assert np.allclose(norms, g["grad_norms"], rtol=1e-3, atol=1e-9)
    for i, k in enumerate(str(n) for n in g["full_names"]):
        want = T(g[f"full_{i}"])
        got = grads[names.index(k)]
        assert float((got - want).norm() / want.norm().clamp_min(1e-30)) < 1e-3


def test_null_optimization_on_the_emulated_abi_matches_the_reference_golden(monkeypatch, unet_sd_np):
    """The product's null-text optimisation -- util.null_optimization: the ordinary launch graph recorded on a tape
    (motioneditor_amd/autodiff.py), the backward primitives of the emulated ABI, Adam on the embedding -- against the
    reference class as written (tests/golden/null_text.npz): first gradient and the optimised embeddings of 2 steps x 2 inner."""
    import motioneditor_amd.models.unet_2d_condition as u
    for m in (graph, u, util):
        monkeypatch.setattr(m, "ops", emu_ops)
    g = np.load(GOLD / "null_text.npz")
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    sched = DDIMScheduler()
    sched.set_timesteps(50)

    class Pipe:
        pass
    pipe = Pipe()
    pipe.unet = unet
    grads = []
    out = util.null_optimization(pipe, sched, [t for t in T(g["latents"])], T(g["context"]), 2, 1e-5, num_ddim_steps=2, grads=grads)
    g0 = T(g["grad0"])
    assert float((grads[0] - g0).norm() / g0.norm()) < 1e-3
    big = g0.abs() > 1e-3 * g0.abs().max()
    for mine, ref in zip(out, T(g["uncond_out"])):
        assert float(((mine - ref).abs() * big).max()) < 2e-3


def test_adapter_training_grads_on_the_emulated_abi_match_the_reference_unet(monkeypatch, unet_sd_np):
    """util.adapter_training_grads -- the launch graph on the autodiff tape, parameter gradients mapped back from the packed
    layouts (fused q|k|v, GEGLU-interleaved rows, tap-major Conv1d) to the reference's parameter names -- against what the
    reference UNet's autograd leaves in .grad (tests/golden/adapter_train.npz): loss, the gradient norm of all 372 adapter
    parameters, two tensors in full."""
    import motioneditor_amd.models.unet_2d_condition as u
    for m in (graph, u, util):
        monkeypatch.setattr(m, "ops", emu_ops)
    g = np.load(GOLD / "adapter_train.npz")
    F32 = lambda k: T(g[k].astype(np.float32))   # noqa: E731
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    loss, grads = util.adapter_training_grads(unet, F32("noisy"), int(g["t"]), F32("ehs"), [F32(f"down{i}") for i in range(12)], F32("mid"), F32("noise"))
    assert abs(loss - float(g["loss"])) < 1e-4 * float(g["loss"])
    names = [str(n) for n in g["names"]]
    assert set(names) == set(grads), (set(names) ^ set(grads))
    norms = np.array([float(grads[k].norm()) for k in names])
    assert np.allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-9), float(np.abs(norms / np.maximum(g["grad_norms"], 1e-30) - 1).max())
    for i, k in enumerate(str(n) for n in g["full_names"]):
        want = T(g[f"full_{i}"])
        assert grads[k].shape == want.shape
        assert float((grads[k] - want).norm() / want.norm().clamp_min(1e-30)) < 1e-3
