"""CPU checks of the HOST logic (launch graphs, weight packing, key-segment tables, editors): the
product graph code runs with tests/emu_ops.py (fp32 torch emulation of the C ABI) patched in place of
the HIP ops, and must reproduce the reference's golden outputs.  No GPU, no product fallback."""
import numpy as np
import pytest
import torch

import emu_ops
from conftest import GOLD, max_rel
from motioneditor_amd import synth
from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                           regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
from motioneditor_amd.models import graph
from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel


@pytest.fixture()
def emu(monkeypatch):
    monkeypatch.setattr(graph, "ops", emu_ops)
    import motioneditor_amd.models.unet_2d_condition as u
    monkeypatch.setattr(u, "ops", emu_ops)
    return emu_ops


def test_unet_single_branch_matches_reference_golden(emu, unet_sd_np):
    g = np.load(GOLD / "unet_single.npz")
    c = synth.make_case_inputs("single", B=2, f=8, h=16, w=16)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    out = unet(c["sample"], c["t"], c["ehs"]).sample
    assert out.shape == (2, 4, 8, 16, 16)
    assert max_rel(out, torch.from_numpy(g["out"])) < 2e-4


def controlnet_trunk_case():
    """Inputs of tests/golden/controlnet_trunk.npz (oracle/make_golden.py --only-controlnet) and its expected residuals."""
    g = np.load(GOLD / "controlnet_trunk.npz")
    n, h, t = int(g["n"]), int(g["latent"]), int(g["t"])
    T = torch.from_numpy
    sample = T(synth.synth_normal("cn_trunk.sample", (n, 4, h, h), 33))
    ehs = T(synth.synth_normal("cn_trunk.ehs", (n, 77, 768), 33, 0.3))
    cond = T(np.clip(synth.synth_normal("cn_trunk.cond", (n, 3, 8 * h, 8 * h), 33, 0.5) + 0.5, 0, 1).astype(np.float32))
    want = [T(g[f"down{i}"]) for i in range(12)] + [T(g["mid"])]
    return sample, t, ehs, cond, want


def controlnet_trunk_errors(down, mid, want, metric):
    """Residuals [(n), C, h', w'] against the fixture (levels with h' >= 8 are stored with every second pixel row / column)."""
    errs = []
    for got, w in zip(list(down) + [mid], want):
        got = got.float().cpu()
        if got.shape[-1] != w.shape[-1]:
            got = got[:, :, ::2, ::2]
        errs.append(metric(got, w))
    return errs


def test_controlnet_trunk_matches_reference_blocks(emu, monkeypatch, cn_sd_np, cn_sd_torch):
    """R16: the ControlNet's trunk (conv_in, time embedding, down blocks, mid block) against the REFERENCE's own 2-D-degenerate blocks
    (tests/golden/controlnet_trunk.npz: the reference UNet2DConditionModel at f = 1 with the temporal parts zeroed, loaded with the
    ControlNet's weights) -- first the oracle, then the product graph on the emulated C ABI."""
    import motioneditor_amd.models.controlnet as cm
    from motioneditor_amd.models.controlnet import ControlNetModel
    from oracle import ref_cpu
    monkeypatch.setattr(cm, "ops", emu_ops)
    sample, t, ehs, cond, want = controlnet_trunk_case()
    with torch.no_grad():
        down, mid = ref_cpu.controlnet_forward(cn_sd_torch, sample, t, ehs, cond)
    assert max(controlnet_trunk_errors(down, mid, want, max_rel)) < 1e-4
    cn = ControlNetModel(cn_sd_np, device="cpu", dtype=torch.float32)
    down, mid = cn(sample, t, ehs, cond)
    assert max(controlnet_trunk_errors(down, mid, want, max_rel)) < 2e-4


@pytest.mark.parametrize("tag,step", [("inactive", 0), ("active", 4)])
def test_unet_two_branch_with_editors_matches_reference_golden(emu, unet_sd_np, tag, step):
    g = np.load(GOLD / f"unet_two_{tag}.npz")
    c = synth.make_case_inputs("two", B=4, f=16, h=16, w=16)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)

    class Holder:
        pass

    pipe = Holder()
    pipe.unet = unet
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=c["source_masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed)
    assert ted.num_att_layers == 16 and sed.num_att_layers == 32
    ted.cur_step = sed.cur_step = step
    taps = {}
    out = unet(c["sample"], c["t"], c["ehs"], down_block_additional_residuals=c["down_res"], mid_block_additional_residual=c["mid_res"], taps=taps).sample
    # counters advanced exactly one step, like the reference's (fully_control_utils.py:38-46)
    assert (ted.cur_step, ted.cur_att_layer, sed.cur_step, sed.cur_att_layer) == (step + 1, 0, step + 1, 0)
    assert max_rel(out, torch.from_numpy(g["out"])) < 2e-4
    # per-stage checksums for bisecting (abs-mean of each skip / motion residual)
    for i, s in enumerate(taps["skips"]):
        assert abs(float(s.abs().mean()) - g["skip_stats"][i, 1]) < 1e-3 * g["skip_stats"][i, 1]


def test_zero_temporal_conv_is_skipped_and_equals_running_it(emu, unet_sd_np, monkeypatch):
    """Real checkpoints keep TemporalConv at its zero initialisation (resnet_2d.py:15-16): the launch graph skips those
    GEMMs; the oracle executes them with zero weights.  Same output, and not a single tconv launch."""
    from oracle import ref_cpu
    sd0 = {k: (np.zeros_like(v) if ".temp_conv" in k and not k.startswith("controlnet_adapter.") else v) for k, v in unet_sd_np.items()}
    c = synth.make_case_inputs("single", B=2, f=8, h=16, w=16)
    calls = {"tconv": 0}
    real = emu_ops.gemm

    def counting(x, w, **kw):
        if kw.get("tconv") is not None:
            calls["tconv"] += 1
        return real(x, w, **kw)

    monkeypatch.setattr(emu_ops, "gemm", counting)
    out = UNet2DConditionModel(sd0, device="cpu", dtype=torch.float32)(c["sample"], c["t"], c["ehs"]).sample
    assert calls["tconv"] == 0
    with torch.no_grad():
        want = ref_cpu.unet_forward({k: torch.from_numpy(v) for k, v in sd0.items()}, c["sample"], c["t"], c["ehs"])
    assert max_rel(out, want) < 2e-4



@pytest.mark.parametrize("kind", ["single", "two"])
def test_skips_produced_in_place_equal_the_copied_form(emu, monkeypatch, unet_sd_np, kind):
    """unet_forward writes the down path's skips straight into the up path's concat buffers (graph.INPLACE_SKIPS): the same launches on the same values --
    bitwise the output of the form that copies every skip into its concat -- with 11 fewer row copies: every skip but skip 0 (conv_small has no strided
    output; under the shared classifier-free-guidance prefix of a pipeline step that one is in place too).  Two-branch: the last skip's motion-updated
    clone lands in its slot -- on this emulation the copied form's clone is torch's, not a copy_rows call, hence 10 counted."""
    c = synth.make_case_inputs(kind, B=4 if kind == "two" else 2, f=8, h=8, w=8)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    kw = dict(down_block_additional_residuals=c["down_res"], mid_block_additional_residual=c["mid_res"]) if kind == "two" else {}
    real = emu_ops.copy_rows
    outs, copies = {}, {}
    for flag in (False, True):
        monkeypatch.setattr(graph, "INPLACE_SKIPS", flag)
        n = [0]

        def counting(y, x, n=n):
            n[0] += 1
            return real(y, x)

        monkeypatch.setattr(emu_ops, "copy_rows", counting)
        outs[flag] = unet(c["sample"], c["t"], c["ehs"], **kw).sample
        copies[flag] = n[0]
    assert torch.equal(outs[True], outs[False])
    assert copies[False] - copies[True] == (10 if kind == "two" else 11), copies
