"""CPU check of one full two-branch denoising step (ControlNet -> UNet3D + adapter + editors -> CFG ->
DDIM) built by the product pipeline code on the emulated C ABI, against the oracle restatement
(oracle/ref_cpu.denoise_step, pipeline_motion_editor.py:603-648)."""
import numpy as np
import pytest
import torch

import emu_ops
from conftest import max_rel
from motioneditor_amd import schedulers, synth
from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                           regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
from motioneditor_amd.models import graph
from motioneditor_amd.models.controlnet import ControlNetModel
from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
from motioneditor_amd.pipelines import MotionEditorPipeline
from oracle import ref_cpu


def step_inputs(f=8, h=8, w=8, seed=33):
    T = torch.from_numpy
    return dict(latents=T(synth.synth_normal("step.latents", (2, 4, f, h, w), seed)),
                uncond=T(synth.synth_normal("step.uncond", (1, 77, 768), seed, 0.3)),
                cond=T(synth.synth_normal("step.cond", (2, 77, 768), seed, 0.3)),
                skeleton=T(np.clip(synth.synth_normal("step.skel", (1, f, 3, 8 * h, 8 * w), seed, 0.5) + 0.5, 0, 1)),
                masks=T(synth.synth_masks(f, 8 * h, 8 * w)))


@pytest.mark.parametrize("step", [0, 4])
def test_denoise_step_matches_oracle(monkeypatch, unet_sd_np, cn_sd_np, unet_sd_torch, cn_sd_torch, step):
    import motioneditor_amd.models.unet_2d_condition as u
    import motioneditor_amd.pipelines.pipeline_motion_editor as pm
    for m in (graph, u, pm, schedulers):
        monkeypatch.setattr(m, "ops", emu_ops)
    x = step_inputs()
    f = x["latents"].shape[2]
    # ---- oracle ----
    ddim = ref_cpu.DDIM()
    sp = ref_cpu.SpatialEditor(x["masks"])
    tp = ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5)
    # ---- product pipeline on the emulated ABI ----
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    cn = ControlNetModel(cn_sd_np, device="cpu", dtype=torch.float32)
    pipe = MotionEditorPipeline(unet=unet, controlnet=cn)
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=x["masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed)
    ted.cur_step = sed.cur_step = step
    pipe.scheduler.set_timesteps(50)
    assert pipe.scheduler.timesteps == ddim.timesteps
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]])
    assert pipe.dedup_cfg_prefix      # the classifier-free-guidance prefix is computed once by default
    got = pipe.denoise_step(x["latents"], t, emb, images, 7.5)
    assert max_rel(got, want) < 2e-4
    assert (sed.cur_step, ted.cur_step) == (step + 1, step + 1)
    # the duplicated prefix executed as the reference does: the same step (on the GPU bitwise, test_model_gpu.py; here up to BLAS blocking)
    pipe.dedup_cfg_prefix = False
    ted.cur_step = sed.cur_step = step
    got2 = pipe.denoise_step(x["latents"], t, emb, images, 7.5)
    assert max_rel(got2, got) < 1e-5
    # an editor that edits from layer 0 on keeps the full batch in the first block too
    pipe.dedup_cfg_prefix = True
    sed0 = FullySelfAttentionControlMask(start_step=4, start_layer=0, source_masks=x["masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed0)
    sed0.cur_step = ted.cur_step = step
    assert sed0.edits_next_self_attention() == (step >= 4)
    pipe.denoise_step(x["latents"], t, emb, images, 7.5)
