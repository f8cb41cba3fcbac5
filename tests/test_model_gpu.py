"""Whole-path parity on a real MI355X, through the product classes (ctypes -> libmotioned):
  * UNet3D (+ adapter + both editors) against the golden outputs captured from the REFERENCE's own
    model code (tests/golden/unet_*.npz, oracle/make_golden.py);
  * one full two-branch denoising step (ControlNet -> UNet -> CFG -> DDIM) against the CPU oracle
    (oracle/ref_cpu.py) on the same seeded inputs, at a size the oracle finishes in seconds;
  * size-independent properties at larger sizes (frame-chunk locality of the adapter, batch-row
    independence of the recon branch, determinism).
Stated tolerance (fp16 storage / fp32 accumulate vs the fp32 reference): rel-L2 <= 1e-2 on the UNet
noise prediction, <= 5e-3 on the updated latents of one step.  Measured values are appended to
gpurun_out/parity.jsonl."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, rel_l2

pytestmark = pytest.mark.gpu

UNET_TOL = 1e-2
STEP_TOL = 5e-3
# The CFG-amplified noise prediction (uncond + 7.5 (cond - uncond)): SURVEY 8c asked for this bound to be calibrated against an fp16 eager run of
# the restatement.  tools/calibrate_fp16.py (profiles/r03_fp16_calibration.txt): the oracle itself in fp16 sits at 0.99 - 1.0e-2 from its own fp32 run
# (8 f x 8 x 8, 24 f x 16 x 16 and 8 f x 32 x 32 latents alike; updated latents 8.2 - 8.4e-4); the HIP path (fp32 accumulation / statistics)
# measures 9.4e-3 / 5 - 6e-4 -- the bound is 1.5 x the fp16-eager level instead of the former, uncalibrated 2e-2.
NOISE_PRED_TOL = 1.5e-2


def record(name, value):
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    with open(out / "parity.jsonl", "a") as fh:
        fh.write(json.dumps({"case": name, "rel_l2": value}) + "\n")


@pytest.fixture(scope="module")
def unet(unet_sd_np):
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    return UNet2DConditionModel(unet_sd_np, device="cuda")


@pytest.fixture(scope="module")
def controlnet(cn_sd_np):
    from motioneditor_amd.models.controlnet import ControlNetModel
    return ControlNetModel(cn_sd_np, device="cuda")


def editors(unet, masks):
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)

    class H:
        pass

    h = H()
    h.unet = unet
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(h, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=masks)
    regiter_fully_attention_editor_diffusers(h, sed)
    return sed, ted


def test_unet_single_branch_vs_reference_golden(unet):
    from motioneditor_amd import synth
    unet.spatial_editor = unet.temporal_editor = None
    g = np.load(GOLD / "unet_single.npz")
    c = synth.make_case_inputs("single", B=2, f=8, h=16, w=16)
    out = unet(c["sample"].cuda(), c["t"], c["ehs"].cuda()).sample
    e = rel_l2(out, torch.from_numpy(g["out"]))
    record("unet_single", e)
    assert e <= UNET_TOL, e


def test_unet_single_branch_32x32_vs_reference_golden(unet):
    """The reference UNet's own output at 32x32 latents (level 0: 1024 tokens = 4 query blocks x 2 x 16 key tiles per attn1
    launch, level 3: 4x4), oracle/make_golden.py case A32."""
    from motioneditor_amd import synth
    unet.spatial_editor = unet.temporal_editor = None
    g = np.load(GOLD / "unet_single_32.npz")
    c = synth.make_case_inputs("single32", B=2, f=8, h=32, w=32)
    out = unet(c["sample"].cuda(), c["t"], c["ehs"].cuda()).sample
    e = rel_l2(out, torch.from_numpy(g["out"].astype(np.float32)))
    record("unet_single_32", e)
    assert e <= UNET_TOL, e


@pytest.mark.parametrize("tag,step", [("inactive", 0), ("active", 4)])
def test_unet_two_branch_editors_vs_reference_golden(unet, tag, step):
    from motioneditor_amd import synth
    g = np.load(GOLD / f"unet_two_{tag}.npz")
    c = synth.make_case_inputs("two", B=4, f=16, h=16, w=16)
    sed, ted = editors(unet, c["source_masks"])
    sed.cur_step = ted.cur_step = step
    taps = {}
    out = unet(c["sample"].cuda(), c["t"], c["ehs"].cuda(), down_block_additional_residuals=[d.cuda() for d in c["down_res"]],
               mid_block_additional_residual=c["mid_res"].cuda(), taps=taps).sample
    unet.spatial_editor = unet.temporal_editor = None
    assert (sed.cur_step, sed.cur_att_layer, ted.cur_step, ted.cur_att_layer) == (step + 1, 0, step + 1, 0)
    for i, s in enumerate(taps["skips"]):   # stage checksums localise a failure
        assert abs(float(s.float().abs().mean()) - g["skip_stats"][i, 1]) < 2e-2 * g["skip_stats"][i, 1], f"skip {i}"
    for i, s in enumerate(taps["motion"]):   # adapter outputs [(2 f N), C] (edit rows only); the golden statistics are over [0, m0, 0, m1]
        assert abs(0.5 * float(s.float().abs().mean()) - g["motion_stats"][i, 1]) < 3e-2 * g["motion_stats"][i, 1], f"motion {i}"
    e = rel_l2(out, torch.from_numpy(g["out"]))
    record(f"unet_two_{tag}", e)
    assert e <= UNET_TOL, e
    if tag == "active":  # the edit must actually change the edit rows, and only those
        gi = torch.from_numpy(np.load(GOLD / "unet_two_inactive.npz")["out"])
        d = (out.cpu() - gi).abs().mean(dim=(1, 2, 3, 4))
        assert d[1] > 10 * d[0] and d[3] > 10 * d[2]


@pytest.mark.parametrize("hw,f", [(64, 8), (96, 8), (64, 24)])
def test_unet_two_branch_editors_active_64x64_vs_reference_golden(unet, hw, f):
    """The two-branch UNet with BOTH editors active at a production token count -- batch 4, 8 frames x 64 x 64 latents: 4096 queries, the edit rows against
    20480 materialised keys in the reference (fully_control.py:381-413), [src prev (dual) | src cur (dual) | own cur] segments here -- against
    tests/golden/unet_two_active_64.npz, the output of the REFERENCE's own UNet + editors (oracle/make_golden.py --only-two64, oracle == reference
    asserted there).  What the 16 x 16 golden pins in kind, this pins at the geometry of the benchmark's level-0 launches; hw = 96: the same at the level-0
    geometry of BASELINE configs[4] (9216 queries, 46080 materialised keys; --only-two96); (64, 24): the UNet shape of the BENCHMARK itself -- batch 4,
    24 frames x 64 x 64 latents (BASELINE configs[2]) -- where the reference's hard-coded num_frames = 8 (fully_control.py:377) indexes the masks by head
    (--only-two64-f24; every second frame of the output is stored)."""
    from motioneditor_amd import synth
    tag = f"unet_two_active_{hw}" + ("" if f == 8 else f"_f{f}")
    g = np.load(GOLD / f"{tag}.npz")
    c = synth.make_case_inputs("two", B=4, f=f, h=hw, w=hw)
    sed, ted = editors(unet, c["source_masks"])
    step = int(g["step"])
    sed.cur_step = ted.cur_step = step
    taps = {}
    out = unet(c["sample"].cuda(), c["t"], c["ehs"].cuda(), down_block_additional_residuals=[d.cuda() for d in c["down_res"]],
               mid_block_additional_residual=c["mid_res"].cuda(), taps=taps).sample
    unet.spatial_editor = unet.temporal_editor = None
    assert (sed.cur_step, sed.cur_att_layer, ted.cur_step, ted.cur_att_layer) == (step + 1, 0, step + 1, 0)
    for i, s in enumerate(taps["skips"]):
        assert abs(float(s.float().abs().mean()) - g["skip_stats"][i, 1]) < 2e-2 * g["skip_stats"][i, 1], f"skip {i}"
    for i, s in enumerate(taps["motion"]):   # adapter outputs (edit rows only); the golden statistics are over [0, m0, 0, m1]
        assert abs(0.5 * float(s.float().abs().mean()) - g["motion_stats"][i, 1]) < 3e-2 * g["motion_stats"][i, 1], f"motion {i}"
    e = rel_l2(out[:, :, ::(1 if f == 8 else 2), ::2, ::2], torch.from_numpy(g["out_sub"]))
    record(tag, e)
    assert torch.isfinite(out).all() and e <= UNET_TOL, e


@pytest.mark.parametrize("step", [0, 4])
def test_denoise_step_vs_cpu_oracle(unet, controlnet, unet_sd_torch, cn_sd_torch, step):
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs()
    f = x["latents"].shape[2]
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    otaps = {}
    want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5, taps=otaps)

    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    sed, ted = editors(unet, x["masks"])
    sed.cur_step = ted.cur_step = step
    pipe.scheduler.set_timesteps(50)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    taps = {}
    got = pipe.denoise_step(x["latents"].cuda(), t, emb, images.cuda(), 7.5, taps=taps)
    unet.spatial_editor = unet.temporal_editor = None
    # ControlNet residuals (rows -> reference layout)
    for i, (d, od) in enumerate(zip(taps["cn_down"], otaps["cn_down"])):
        od_rows = od.permute(0, 2, 3, 4, 1).reshape(-1, od.shape[1])
        if d.shape[0] * 2 == od_rows.shape[0]:   # pipeline.dedup_controlnet: the two reference entries are identical (even f)
            half = od_rows.shape[0] // 2
            assert rel_l2(od_rows[half:], od_rows[:half]) < 1e-6
            od_rows = od_rows[:half]
        ei = rel_l2(d, od_rows)
        record(f"step{step}_cn_down{i}", ei)
        assert ei <= 2e-2, (i, ei)
    eps = taps["eps_rows"].float().cpu().reshape(4, f, 64, 4).permute(0, 3, 1, 2).reshape(4, 4, f, 8, 8)
    eu, ec = eps[:2], eps[2:]
    e_np = rel_l2(eu + 7.5 * (ec - eu), otaps["noise_pred"])
    record(f"step{step}_noise_pred", e_np)
    e = rel_l2(got, want)
    record(f"step{step}_latents", e)
    assert e_np <= NOISE_PRED_TOL, e_np   # CFG amplifies (cond - uncond) by 7.5
    assert e <= STEP_TOL, e


def test_denoise_step_soft_masks_vs_cpu_oracle(unet, controlnet, unet_sd_torch, cn_sd_torch):
    """Non-binary source masks (mask / 255 of an anti-aliased PNG: fully_control.py:366-368 accepts any value in [0, 1]): the spatial editor then takes the
    mask-reading general-dual segments (ME_SEG_DUAL_PREV / _CUR, attn_kernel<GD>), which the DEFAULT graph feeds with head-major Q | K | V panels
    (round-5 advisor finding: that combination used to return ME_EINVAL and no model-level test had a soft mask).  One edited step vs the CPU oracle,
    eager and through the step-level C entry point."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs()
    f, step = x["latents"].shape[2], 4
    g = torch.Generator().manual_seed(17)
    soft = (0.15 + 0.7 * x["masks"].float() + 0.1 * torch.rand(x["masks"].shape, generator=g)).clamp(0, 1)
    assert not bool(((soft == 0) | (soft == 1)).all())
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(soft), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5)
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    pipe.scheduler.set_timesteps(50)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    outs = []
    for fn in (pipe.denoise_step, pipe.denoise_step_planned):
        sed, ted = editors(unet, soft)
        assert not sed.binary_masks
        sed.cur_step = ted.cur_step = step
        outs.append(fn(x["latents"].cuda(), t, emb, images.cuda(), 7.5).clone())
    unet.spatial_editor = unet.temporal_editor = None
    e = rel_l2(outs[0], want)
    record("step4_soft_masks_latents", e)
    assert e <= STEP_TOL, e
    assert torch.equal(outs[0], outs[1]), rel_l2(outs[1], outs[0])
    # ... and the soft masks must be what was computed: the binary-mask step is another result
    sed, ted = editors(unet, x["masks"])
    sed.cur_step = ted.cur_step = step
    hard = pipe.denoise_step(x["latents"].cuda(), t, emb, images.cuda(), 7.5)
    unet.spatial_editor = unet.temporal_editor = None
    d = rel_l2(hard, outs[0])
    record("step4_soft_vs_hard_masks", d)
    assert d > 1e-5, d


def test_denoise_step_24_frames_16x16_vs_cpu_oracle(unet, controlnet, unet_sd_torch, cn_sd_torch):
    """The benchmark's frame count at 16x16 latents: three adapter chunks of 8 frames, tattn_kernel<24> inside the graph,
    level 3 = 2x2 pixels; one full two-branch step with both editors active vs the CPU oracle."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs(f=24, h=16, w=16)
    f, step = 24, 4
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 128, 128)
    want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5)
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    sed, ted = editors(unet, x["masks"])
    sed.cur_step = ted.cur_step = step
    pipe.scheduler.set_timesteps(50)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    got = pipe.denoise_step(x["latents"].cuda(), t, emb, images.cuda(), 7.5)
    unet.spatial_editor = unet.temporal_editor = None
    e = rel_l2(got, want)
    record("step4_f24_16x16_latents", e)
    assert e <= STEP_TOL, e


def test_cfg_prefix_sharing_is_bitwise_exact(unet, controlnet):
    """pipeline.dedup_cfg_prefix: conv_in, the first resnet and the first transformer block up to its text cross-attention run once for the
    unconditional / conditional copies of the latents (graph.unet_forward, cfg_dup) -- shared queries in me_attn, shared residual rows in
    the GEMM epilogues.  The step must equal the one that executes the duplicated prefix BIT FOR BIT, with the editors inactive and active,
    at 8 and at 24 frames -- and at 8 frames x 64 x 64 latents, where the full batch takes the LDS-halo convolution kernel and the half batch
    alone would not (me_gemm_args.sel_rows keeps the choice, and with it the summation order, that of the full batch)."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    for f, hw, step in ((8, 8, 0), (8, 8, 4), (24, 16, 4), (8, 64, 4)):
        x = step_inputs(f=f, h=hw, w=hw)
        images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * hw, 8 * hw).cuda()
        emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
        pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
        pipe.scheduler.set_timesteps(50)
        outs = []
        for share in (True, False):
            sed, ted = editors(unet, x["masks"])
            sed.cur_step = ted.cur_step = step
            pipe.dedup_cfg_prefix = share
            outs.append(pipe.denoise_step(x["latents"].cuda(), pipe.scheduler.timesteps[step], emb, images, 7.5).clone())
            assert (sed.cur_step, sed.cur_att_layer) == (step + 1, 0)
        unet.spatial_editor = unet.temporal_editor = None
        assert torch.equal(outs[0], outs[1]), (f, hw, step, rel_l2(outs[0], outs[1]))


def test_graph_replay_six_steps_equals_eager(unet, controlnet):
    """denoise_step_graphed (one captured hipGraph per editor gating, device-resident step scalars) over steps 0..5 --
    across the editors' start at step 4, with a different unconditional embedding and timestep every step -- must
    reproduce the eager loop bit for bit (same kernels, same order, deterministic reductions)."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    x = step_inputs()
    f = x["latents"].shape[2]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64).cuda()
    g = torch.Generator().manual_seed(5)
    uncs = [x["uncond"] + 0.05 * i * torch.randn(x["uncond"].shape, generator=g) for i in range(6)]
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    pipe.scheduler.set_timesteps(50)
    outs = {}
    for mode in ("eager", "graph"):
        sed, ted = editors(unet, x["masks"])
        lat = x["latents"].cuda()
        for i in range(6):
            emb = torch.cat([uncs[i].expand(2, 77, 768), x["cond"]]).cuda()
            fn = pipe.denoise_step if mode == "eager" else pipe.denoise_step_graphed
            lat = fn(lat, pipe.scheduler.timesteps[i], emb, images, 7.5)
            assert sed.cur_step == ted.cur_step == i + 1 and sed.cur_att_layer == ted.cur_att_layer == 0
        outs[mode] = lat.clone()
    unet.spatial_editor = unet.temporal_editor = None
    assert len(pipe._graphs) == 2                      # editors inactive / active
    e = rel_l2(outs["graph"], outs["eager"])
    record("graph_replay_vs_eager_six_steps", e)
    assert e == 0.0, e


def test_planned_step_six_steps_equals_eager(unet, controlnet):
    """denoise_step_planned (csrc/plan.hip: the step's launch list recorded once per editor gating, re-issued by ONE me_denoise_step call on the two live
    streams) over steps 0..5 -- across the editors' start at step 4, with a different unconditional embedding, timestep and latent every step -- must
    reproduce the eager loop bit for bit.  Fresh inputs at every replay: a torch kernel hiding inside the recorded step (executed in the recording
    pass, missing from the replays) would leave stale data behind and show up here."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    x = step_inputs()
    f = x["latents"].shape[2]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64).cuda()
    g = torch.Generator().manual_seed(5)
    uncs = [x["uncond"] + 0.05 * i * torch.randn(x["uncond"].shape, generator=g) for i in range(6)]
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    pipe.scheduler.set_timesteps(50)
    outs = {}
    for mode in ("eager", "plan"):
        sed, ted = editors(unet, x["masks"])
        lat = x["latents"].cuda()
        for i in range(6):
            emb = torch.cat([uncs[i].expand(2, 77, 768), x["cond"]]).cuda()
            fn = pipe.denoise_step if mode == "eager" else pipe.denoise_step_planned
            lat = fn(lat, pipe.scheduler.timesteps[i], emb, images, 7.5)
            assert sed.cur_step == ted.cur_step == i + 1 and sed.cur_att_layer == ted.cur_att_layer == 0
        outs[mode] = lat.clone()
    unet.spatial_editor = unet.temporal_editor = None
    assert len(pipe._plans) == 2                      # editors inactive / active
    for st in pipe._plans.values():
        info = st["plan"].stats()
        assert info["streams"] == 2 and info["launches"] > 500 and info["event_records"] >= info["event_waits"] >= 3 and info["replays"] >= 2, info
        kinds = [n[0] for n in st["plan"].nodes()]
        assert kinds.count(0) == info["launches"] and kinds[0] in (0, 1)
    e = rel_l2(outs["plan"], outs["eager"])
    record("planned_step_vs_eager_six_steps", e)
    assert e == 0.0, e
    # single-branch (no ControlNet, no editors: one stream, no events) and a second replay with new inputs
    pipe1 = MotionEditorPipeline(unet=unet)
    pipe1.scheduler.set_timesteps(50)
    lat = x["latents"][:1].cuda()
    emb = torch.cat([uncs[1][:1], x["cond"][:1]]).cuda()
    for i in (7, 9):
        a = pipe1.denoise_step(lat, pipe1.scheduler.timesteps[i], emb, None, 7.5)
        b = pipe1.denoise_step_planned(lat, pipe1.scheduler.timesteps[i], emb, None, 7.5)
        assert torch.equal(a, b), (i, rel_l2(a, b))
        lat = a
    (st,) = pipe1._plans.values()
    assert st["plan"].stats()["streams"] == 1 and st["plan"].stats()["event_records"] == 0
    # __call__'s loop on the planned executor (3 steps from given latents, no VAE): the eager loop's latents, bit for bit
    res = {}
    for ex in ("eager", "plan"):
        pipe2 = MotionEditorPipeline(unet=unet, controlnet=controlnet)
        pipe2.step_executor = ex
        res[ex] = pipe2(["a source prompt", "a target prompt"], video_length=f, height=64, width=64, num_inference_steps=3, guidance_scale=7.5,
                        latents=x["latents"].cuda(), output_type="latent", text_embeddings=x["cond"].cuda(), negative_text_embeddings=x["uncond"].cuda(),
                        skeleton=x["skeleton"].cuda()).images
    assert torch.equal(res["eager"], res["plan"])


def test_planned_step_failure_rewinds_editors_and_plan_cache_is_bounded(unet, controlnet):
    """(1) A step that raises during the warm-up / recording pass of denoise_step_planned must leave the editors' (cur_step, cur_att_layer) where the
    caller had them and no half-built plan behind, so that the caller's eager retry computes THIS step (round-4 advisor finding).  (2) The plan table is
    an LRU of `max_cached_steps` entries: re-uploaded conditioning tensors do not pin one memory pool each; `release_plans()` empties it."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    x = step_inputs()
    f = x["latents"].shape[2]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64).cuda()
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    pipe.scheduler.set_timesteps(50)
    sed, ted = editors(unet, x["masks"])
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    lat = x["latents"].cuda()
    t = pipe.scheduler.timesteps[0]
    ref = pipe.denoise_step(lat, t, emb, images, 7.5)
    assert sed.cur_step == 1
    sed.cur_step = ted.cur_step = 0
    real, calls = pipe.denoise_step, []

    def flaky(*a, **k):
        calls.append(1)
        if len(calls) == 2:         # the RECORDING pass: advance the counters by a partial step, then fail
            sed.cur_att_layer, ted.cur_att_layer = 7, 3
            sed.cur_step += 1
            raise RuntimeError("injected")
        return real(*a, **k)

    pipe.denoise_step = flaky
    try:
        with pytest.raises(RuntimeError, match="injected"):
            pipe.denoise_step_planned(lat, t, emb, images, 7.5)
    finally:
        pipe.denoise_step = real
    assert (sed.cur_step, sed.cur_att_layer, ted.cur_step, ted.cur_att_layer) == (0, 0, 0, 0) and not pipe._plans
    assert torch.equal(pipe.denoise_step(lat, t, emb, images, 7.5), ref)      # the eager retry is the same step
    # LRU: three conditioning tensors through a table of two
    pipe.max_cached_steps = 2
    for k in range(3):
        sed.cur_step = ted.cur_step = 0
        out = pipe.denoise_step_planned(lat, t, emb, images.clone(), 7.5)
        assert torch.equal(out, ref), k
        assert len(pipe._plans) == min(k + 1, 2)
    pipe.release_plans()
    assert not pipe._plans and not pipe._graphs
    unet.spatial_editor = unet.temporal_editor = None


def test_high_gain_weights_step_vs_cpu_oracle():
    """A weight set that drives the activations to SD-like magnitudes (|x| ~ 1e2 - 1e3 after conv_in, large per-channel
    means in front of the GroupNorms): GroupNorm variance (fp64 statistics) and the fp16 range of the residual stream,
    one two-branch step vs the CPU oracle run on the same weights."""
    from motioneditor_amd import synth
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    usd = dict(synth.synth_state_dict(synth.unet_schema()))
    csd = dict(synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet."))
    for sd in (usd, csd):
        sd["conv_in.weight"] = sd["conv_in.weight"] * 60.0
        sd["conv_in.bias"] = sd["conv_in.bias"] + 80.0 * np.sign(np.arange(sd["conv_in.bias"].shape[0]) % 3 - 0.5).astype(np.float32)
        for k in list(sd):
            if k.endswith(".conv_shortcut.bias") or k.endswith("proj_out.bias"):
                sd[k] = sd[k] + 20.0
    x = step_inputs()
    f, step = x["latents"].shape[2], 4
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    to = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}  # noqa: E731
    otaps = {}
    want = ref_cpu.denoise_step(to(usd), to(csd), ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5, taps=otaps)
    u, c = UNet2DConditionModel(usd, device="cuda"), ControlNetModel(csd, device="cuda")
    pipe = MotionEditorPipeline(unet=u, controlnet=c)
    sed, ted = editors(u, x["masks"])
    sed.cur_step = ted.cur_step = step
    pipe.scheduler.set_timesteps(50)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    taps = {}
    got = pipe.denoise_step(x["latents"].cuda(), t, emb, images.cuda(), 7.5, taps=taps)
    amax = max(float(s.float().abs().max()) for s in taps["skips"])
    record("high_gain_skip_absmax", amax)
    assert amax > 100.0, amax                          # the case really is high-gain
    e = rel_l2(got, want)
    record("high_gain_step_latents", e)
    assert torch.isfinite(got).all() and e <= STEP_TOL, e


def test_six_consecutive_steps_cross_the_editor_start(unet, controlnet, unet_sd_torch, cn_sd_torch):
    """Steps 0..5 of a run with the editors counting on their own (start_step = 4): the un-edited -> edited
    transition, the per-step uncond embedding and the DDIM recursion, product loop vs oracle loop.
    Tolerance: SURVEY 8c allows 2e-2 on the latents after 10 steps."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs()
    f = x["latents"].shape[2]
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    g = torch.Generator().manual_seed(5)
    uncs = [x["uncond"] + 0.05 * i * torch.randn(x["uncond"].shape, generator=g) for i in range(6)]

    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    sed, ted = editors(unet, x["masks"])
    pipe.scheduler.set_timesteps(50)
    want, got = x["latents"], x["latents"].cuda()
    errs = []
    for i in range(6):
        t = ddim.timesteps[i]
        assert sp.cur_step == sed.cur_step == i and tp.cur_step == ted.cur_step == i
        want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, want, t, uncs[i], x["cond"], images, sp, tp, 7.5)
        emb = torch.cat([uncs[i].expand(2, 77, 768), x["cond"]]).cuda()
        got = pipe.denoise_step(got, t, emb, images.cuda(), 7.5)
        errs.append(rel_l2(got, want))
    unet.spatial_editor = unet.temporal_editor = None
    record("six_steps_latents", errs[-1])
    assert errs[-1] <= 2e-2, errs


def test_ddim_inversion_vs_reference_golden(unet):
    """SURVEY 8f rank 1 (forward half): normal_infer UNet and three inversion steps (util.ddim_inversion) on the HIP
    path vs tests/golden/inversion.npz (reference UNet with normal_infer=True + the reference's next_step)."""
    from motioneditor_amd import synth, util
    from motioneditor_amd.schedulers import DDIMScheduler
    g = np.load(GOLD / "inversion.npz")
    c = synth.make_case_inputs("inversion", B=1, f=8, h=16, w=16)
    out = unet(c["sample"], 1, c["ehs"], normal_infer=True).sample.float().cpu()
    e = rel_l2(out, torch.from_numpy(g["unet_normal_infer_t1"]))
    record("unet_normal_infer", e)
    assert e <= UNET_TOL, e

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet = unet
    s = DDIMScheduler()
    s.set_timesteps(50)
    lats = util.ddim_inversion(pipe, s, c["sample"], 3, normal_infer=True, text_embeddings=c["ehs"])
    e3 = rel_l2(lats[-1].float().cpu(), torch.from_numpy(g["loop_latent_3"]))
    record("inversion_3_steps_latents", e3)
    assert e3 <= STEP_TOL, e3


def test_vae_decode_vs_cpu_oracle():
    """SURVEY 8f rank 2 (parity unpinned: diffusers AutoencoderKL is not in the reference tree): decoder on the HIP
    path vs oracle/ref_cpu.py::vae_decode, two 16x16 latent frames -> 128x128 images."""
    from motioneditor_amd import synth
    from motioneditor_amd.models.vae import AutoencoderKL
    from oracle import ref_cpu
    sd_np = synth.synth_state_dict(synth.vae_decoder_schema(), salt="vae.")
    z = torch.from_numpy(synth.synth_normal("vae.z", (2, 4, 16, 16), 33))
    with torch.no_grad():
        want = ref_cpu.vae_decode({k: torch.from_numpy(v) for k, v in sd_np.items()}, z)
    got = AutoencoderKL(sd_np, device="cuda").decode(z.cuda()).sample.float().cpu()
    e = rel_l2(got, want)
    record("vae_decode", e)
    assert got.shape == (2, 3, 128, 128) and e <= 5e-3, e


def test_vae_encode_vs_cpu_oracle():
    """SURVEY 8f rank 2, encode half (parity unpinned): two 128x128 frames -> 16x16 latents, HIP path vs oracle/ref_cpu.py::vae_encode_sample
    with the same noise (asymmetric (0,1,0,1) padding of the stride-2 convolutions, mid attention, DiagonalGaussian sample, x 0.18215)."""
    from motioneditor_amd import synth
    from motioneditor_amd.models.vae import AutoencoderKL
    from oracle import ref_cpu
    sd_np = synth.synth_state_dict(synth.vae_encoder_schema(), salt="vae.")
    x = torch.from_numpy(synth.synth_normal("vae.x", (2, 3, 128, 128), 33)).clamp(-1, 1)
    noise = torch.from_numpy(synth.synth_normal("vae.noise", (2, 4, 16, 16), 33))
    with torch.no_grad():
        want = ref_cpu.vae_encode_sample({k: torch.from_numpy(v) for k, v in sd_np.items()}, x, noise) * 0.18215
    got = AutoencoderKL(sd_np, device="cuda").encode(x.cuda()).latent_dist.sample(noise=noise.cuda(), scale=0.18215).cpu()
    e = rel_l2(got, want)
    record("vae_encode", e)
    assert got.shape == (2, 4, 16, 16) and e <= 5e-3, e


def test_harness_sequence_call_on_gpu_vs_oracle(unet, controlnet, unet_sd_torch, cn_sd_torch):
    """P0 (inference.py:255-326) through examples/run_edit.py on the HIP path: VAE encode -> 2 DDIM-inversion steps (normal_infer)
    -> repeat(2) -> both editors registered -> MotionEditorPipeline.__call__ (2 steps, uncond_embeddings=None so the negative
    embedding is prepended, skeleton = cat([0, tgt, 0, tgt])) -> VAE decode; against the same sequence composed from the oracle."""
    import sys
    sys.path.insert(0, str(ROOT / "examples"))
    import run_edit
    from motioneditor_amd import synth
    from motioneditor_amd.models.vae import AutoencoderKL
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    f, H = 8, 64
    x = run_edit.harness_inputs(f, H, H)
    vsd = dict(synth.synth_state_dict(synth.vae_decoder_schema(), salt="vae."))
    vsd.update(synth.synth_state_dict(synth.vae_encoder_schema(), salt="vae."))
    vt = {k: torch.from_numpy(v) for k, v in vsd.items()}
    steps = inv_steps = 2
    # ---- oracle ----
    with torch.no_grad():
        lat = ref_cpu.vae_encode_sample(vt, x["pixel_values"].reshape(f, 3, H, H), x["encode_noise"])
        lat = lat.reshape(1, f, 4, H // 8, H // 8).permute(0, 2, 1, 3, 4) * 0.18215
        d_inv = ref_cpu.DDIM()
        d_inv.set_timesteps(inv_steps)
        inv = ref_cpu.ddim_loop(unet_sd_torch, d_inv, lat, inv_steps, x["negative_text_embeddings"], normal_infer=True)[-1].repeat(2, 1, 1, 1, 1)
        ddim = ref_cpu.DDIM()
        ddim.set_timesteps(steps)
        sp, tp = ref_cpu.SpatialEditor(x["source_masks"]), ref_cpu.TemporalEditor()
        images = torch.cat([x["target_skeleton"]] * 2).reshape(2 * f, 3, H, H)
        want = inv
        for t in ddim.timesteps:
            want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, want, t, x["negative_text_embeddings"], x["text_embeddings"], images, sp, tp, 7.5)
        video = ref_cpu.decode_latents(vt, want)
    # ---- product ----
    pipe = MotionEditorPipeline(vae=AutoencoderKL(vsd, device="cuda"), unet=unet, controlnet=controlnet)
    xs = {k: v.cuda() for k, v in x.items()}
    s_inv, s_gen, inv_lat = run_edit.run(pipe, xs, steps=steps, inv_steps=inv_steps)
    unet.spatial_editor = unet.temporal_editor = None
    e_inv = rel_l2(inv_lat, inv)
    got = torch.cat([s_inv, s_gen])
    e = float((got.float().cpu() - video).abs().mean() / video.abs().mean())
    record("harness_inversion_latents", e_inv)
    record("harness_video_mean_abs_rel", e)
    assert got.shape == (2, 3, f, H, H) and e_inv <= STEP_TOL and e <= 1e-2, (e_inv, e)


def test_from_pretrained_roundtrip_on_synthetic_safetensors(tmp_path, unet_sd_np, cn_sd_np):
    """SURVEY 8f rank 3 on the GPU: a checkpoint directory as inference.py:152-156,237-240 reads it -- an SD-1.5 style 2-D UNet
    (`unet/diffusion_pytorch_model.safetensors`, no temporal / adapter keys), the accelerate `pytorch_model.bin` of stage 1 with the
    temporal modules, the adapter `.pth`, a ControlNet directory -- written by this test from the synthetic weights, loaded with the
    reference's from_pretrained signature, must reproduce the forward of the model built directly from the full state dict."""
    from safetensors.torch import save_file
    from motioneditor_amd import synth
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    full = {k: torch.from_numpy(v) for k, v in unet_sd_np.items()}
    base = {k: v for k, v in full.items() if "temp" not in k and not k.startswith("controlnet_adapter.")}
    temporal = {k: v for k, v in full.items() if "temp" in k and not k.startswith("controlnet_adapter.")}
    adapter = {k[len("controlnet_adapter."):]: v for k, v in full.items() if k.startswith("controlnet_adapter.")}
    (tmp_path / "sd15" / "unet").mkdir(parents=True)
    save_file(base, str(tmp_path / "sd15" / "unet" / "diffusion_pytorch_model.safetensors"))
    (tmp_path / "ckpt").mkdir()
    torch.save({**base, **temporal}, tmp_path / "ckpt" / "pytorch_model.bin")
    torch.save(adapter, tmp_path / "adapter.pth")
    (tmp_path / "cn").mkdir()
    save_file({k: torch.from_numpy(v) for k, v in cn_sd_np.items()}, str(tmp_path / "cn" / "diffusion_pytorch_model.safetensors"))
    u = UNet2DConditionModel.from_pretrained(str(tmp_path / "sd15"), subfolder="unet", resume_from_checkpoint=str(tmp_path / "ckpt"),
                                             adapter_weight_path=str(tmp_path / "adapter.pth"), device="cuda")
    c = synth.make_case_inputs("two", B=4, f=8, h=16, w=16)
    kw = dict(down_block_additional_residuals=[d.cuda() for d in c["down_res"]], mid_block_additional_residual=c["mid_res"].cuda())
    ref = UNet2DConditionModel(unet_sd_np, device="cuda")
    got = u(c["sample"].cuda(), c["t"], c["ehs"].cuda(), **kw).sample
    want = ref(c["sample"].cuda(), c["t"], c["ehs"].cuda(), **kw).sample
    assert torch.equal(got, want), rel_l2(got, want)
    cn = ControlNetModel.from_pretrained(str(tmp_path / "cn"), device="cuda")
    lat = torch.from_numpy(synth.synth_normal("rt.lat", (2, 4, 16, 16), 33)).cuda()
    ehs = torch.from_numpy(synth.synth_normal("rt.ehs", (2, 77, 768), 33, 0.3)).cuda()
    img = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(3)).cuda()
    d1, m1 = cn(lat, 981, ehs, img)
    d2, m2 = ControlNetModel(cn_sd_np, device="cuda")(lat, 981, ehs, img)
    assert torch.equal(m1, m2) and all(torch.equal(a, b) for a, b in zip(d1, d2))


def test_properties_at_larger_size(unet):
    """Size-independent checks on a bigger clip (B=4, f=16, 32x32 latents): (1) determinism; (2) the
    reconstruction rows do not depend on the editing rows' inputs (K/V injection is one-way);
    (3) the adapter's chunk-of-8 locality: perturbing ControlNet residual frame 9 leaves motion
    residual frames 0-7 untouched except through the causal temporal attention, which only looks back."""
    from motioneditor_amd import synth
    c = synth.make_case_inputs("two", B=4, f=16, h=32, w=32)
    sed, ted = editors(unet, c["source_masks"])

    def run(sample, down):
        sed.reset(); ted.reset()
        sed.cur_step = ted.cur_step = 4
        taps = {}
        out = unet(sample.cuda(), 981, c["ehs"].cuda(), down_block_additional_residuals=[d.cuda() for d in down],
                   mid_block_additional_residual=c["mid_res"].cuda(), taps=taps).sample
        return out, taps

    o1, _ = run(c["sample"], c["down_res"])
    o2, _ = run(c["sample"], c["down_res"])
    # run-to-run: every reduction has a fixed order (GroupNorm statistics: per-chunk partials added in index order) -> bitwise equal
    noise = rel_l2(o1, o2)
    record("run_to_run", noise)
    assert noise == 0.0, noise
    s2 = c["sample"].clone()
    s2[1] += 0.5                                          # perturb the uncond EDIT row only
    o3, _ = run(s2, c["down_res"])
    assert rel_l2(o3[0], o1[0]) < 3e-3 and rel_l2(o3[2], o1[2]) < 3e-3   # recon rows unchanged (to the noise floor)
    assert rel_l2(o3[1], o1[1]) > 3e-2                                    # edit row changed
    d2 = [d.clone() for d in c["down_res"]]
    d2[0][:, :, 9] += 1.0
    _, t4 = run(c["sample"], d2)
    _, t1 = run(c["sample"], c["down_res"])
    m1 = t1["motion"][0].float().reshape(2, 16, 32 * 32, 320)
    m4 = t4["motion"][0].float().reshape(2, 16, 32 * 32, 320)
    assert rel_l2(m4[:, :8], m1[:, :8]) < 3e-3            # frames 0-7: other chunk + causal -> untouched
    assert rel_l2(m4[:, 9:], m1[:, 9:]) > 3e-2            # frame 9 itself, later frames of its chunk, and causal look-back
    unet.spatial_editor = unet.temporal_editor = None


def test_frame_sharded_path_on_one_rank_equals_plain_step(unet, controlnet):
    """World-size-1 RCCL process group: exercises the frame-sharded code path (K|V exchange buffers, sharded tconv / temporal
    attention arguments, GroupNorm split with its all-reduce) through the real kernels; must equal the ordinary step to the noise floor.
    Then the same sharded step with its exchanges issued by RCCL directly, eager (== the process-group run, bitwise) and CAPTURED into a
    hipGraph (denoise_step_graphed(shard=...), RCCL calls as graph nodes) replayed over two steps: bitwise equal to the eager sharded steps.
    Multi-rank correctness of the exchanges is covered by tests/test_frame_shard_cpu.py (gloo, world 2-8)."""
    import torch.distributed as dist
    from motioneditor_amd import parallel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x = step_inputs(f=16)
        f = 16
        pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
        sed, ted = editors(unet, x["masks"])
        pipe.scheduler.set_timesteps(50)
        t = pipe.scheduler.timesteps[4]
        images = x["skeleton"].reshape(f, 3, 64, 64).cuda()
        emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
        sed.cur_step = ted.cur_step = 4
        want = pipe.denoise_step(x["latents"].cuda(), t, emb, torch.cat([images] * 2), 7.5)
        sed.reset(); ted.reset()
        sed.cur_step = ted.cur_step = 4
        shard = parallel.FrameShard(f)
        got = pipe.denoise_step_frame_sharded(x["latents"].cuda(), t, emb, images, 7.5, shard)
        e = rel_l2(got, want)
        record("frame_shard_world1", e)
        assert e < 3e-3, e
        # The same exchanges issued by RCCL directly (motioneditor_amd/rccl.py; no process-group watchdog, so capturable): eager must equal the
        # process-group run bit for bit, and the step CAPTURED into a hipGraph (RCCL calls as graph nodes) must equal the eager steps bit for
        # bit over two consecutive steps with different timesteps / embeddings (editors active in both).
        shard_r = parallel.FrameShard(f, comm="rccl")
        sed.reset(); ted.reset()
        sed.cur_step = ted.cur_step = 4
        got_r = pipe.denoise_step_frame_sharded(x["latents"].cuda(), t, emb, images, 7.5, shard_r)
        assert torch.equal(got_r, got), rel_l2(got_r, got)
        outs = {}
        for mode in ("eager", "graph"):
            sed.reset(); ted.reset()
            sed.cur_step = ted.cur_step = 4
            lat = x["latents"].cuda()
            for i in (4, 5):
                emb_i = torch.cat([(x["uncond"] * (1.0 + 0.1 * i)).expand(2, 77, 768), x["cond"]]).cuda()
                ti = pipe.scheduler.timesteps[i]
                if mode == "eager":
                    lat = pipe.denoise_step_frame_sharded(lat, ti, emb_i, images, 7.5, shard_r)
                else:
                    lat = pipe.denoise_step_graphed(lat, ti, emb_i, images, 7.5, shard=shard_r)
            outs[mode] = lat.clone()
            assert sed.cur_step == ted.cur_step == 6
        eg = rel_l2(outs["graph"], outs["eager"])
        record("frame_shard_graph_vs_eager", eg)
        assert eg == 0.0, eg
    finally:
        unet.spatial_editor = unet.temporal_editor = None
        dist.destroy_process_group()


def test_controlnet_trunk_vs_reference_blocks(controlnet):
    """R16 on the GPU: the drop-in ControlNetModel (diffusers call signature, pipeline_motion_editor.py:618-625) against
    tests/golden/controlnet_trunk.npz -- the residuals of the REFERENCE's own 2-D-degenerate SD-1.5 encoder blocks loaded with the ControlNet's
    trunk weights (oracle/make_golden.py --only-controlnet); only the conditioning-embedding convolutions and the 1x1 zero-convolutions of the
    expected values come from the restatement."""
    from test_graph_cpu import controlnet_trunk_case, controlnet_trunk_errors
    sample, t, ehs, cond, want = controlnet_trunk_case()
    down, mid = controlnet(sample.cuda(), t, ehs.cuda(), cond.cuda())
    errs = controlnet_trunk_errors(down, mid, want, rel_l2)
    record("controlnet_trunk_vs_reference_max", max(errs))
    assert max(errs) <= 5e-3, errs


def _rows_to_5d(rows, B, f, h):
    """[(B f h h), C] token-major rows -> [B, C, f, h, h] (the reference layout of the oracle's taps)."""
    assert rows.shape[0] == B * f * h * h, (rows.shape, B, f, h)
    return rows.float().reshape(B, f, h, h, rows.shape[1]).permute(0, 4, 1, 2, 3)


def _sub5(t, sf, sp):
    return t[:, ::8, ::sf, ::sp, ::sp].cpu()


def _step_vs_golden(tag):
    """One full denoising step on bench.py's inputs and weights against an oracle-generated fixture (oracle/make_golden.py step_golden):
    strided sub-samples of the updated latents and of the guided noise prediction (rel-L2), rel-L2 on sub-samples of two skips / one motion
    residual / one ControlNet residual where the fixture has them, and the abs-mean of every stage."""
    from motioneditor_amd import synth
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    g = np.load(GOLD / f"{tag}.npz")
    f, h, step = int(g["frames"]), int(g["latent"]), int(g["step"])
    single = "single_branch" in g.files and int(g["single_branch"]) == 1
    x = synth.bench_inputs(f, h, h)
    u = UNet2DConditionModel(synth.synth_state_dict(synth.unet_schema()), device="cuda")
    if single:
        pipe = MotionEditorPipeline(unet=u, controlnet=None)
        lat, emb, images = x["latents"][:1].cuda(), torch.cat([x["uncond"][step], x["cond"][:1]]).cuda(), None
    else:
        c = ControlNetModel(synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet."), device="cuda")
        pipe = MotionEditorPipeline(unet=u, controlnet=c)
        sed, ted = editors(u, x["masks"])
        sed.cur_step = ted.cur_step = step
        lat = x["latents"].cuda()
        images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * h).cuda()
        emb = torch.cat([x["uncond"][step].expand(2, 77, 768), x["cond"]]).cuda()
    pipe.scheduler.set_timesteps(50)
    t = pipe.scheduler.timesteps[step]
    assert int(t) == int(g["t"])
    nb = lat.shape[0]
    taps = {}
    got = pipe.denoise_step(lat, t, emb, images, 7.5, taps=taps)
    am = lambda t_: float(t_.float().abs().mean())   # noqa: E731
    for i, s_ in enumerate(taps["skips"]):
        assert abs(am(s_) - g["skip_stats"][i, 1]) < 2e-2 * g["skip_stats"][i, 1], f"skip {i}"
    assert abs(am(taps["mid"]) - g["mid_stats"][1]) < 2e-2 * g["mid_stats"][1]
    if not single:
        for i, d in enumerate(taps["cn_down"]):           # one ControlNet entry (the reference's two are identical)
            assert abs(am(d) - g["cn_down_stats"][i, 1]) < 2e-2 * g["cn_down_stats"][i, 1], f"ControlNet residual {i}"
        assert abs(0.5 * am(taps["cn_mid"]) - g["cn_mid_stats"][1]) < 2e-2 * g["cn_mid_stats"][1]        # golden: [0, m, 0, m]
        for i, m in enumerate(taps["motion"]):            # edit rows only here; the golden statistics are over [0, m0, 0, m1]
            assert abs(0.5 * am(m) - g["motion_stats"][i, 1]) < 3e-2 * g["motion_stats"][i, 1], f"motion residual {i}"
    N = h * h
    eps = taps["eps_rows"].float().reshape(2 * nb, f, N, 4).permute(0, 3, 1, 2).reshape(2 * nb, 4, f, h, h)
    if "np_stride" in g.files:
        sf, sp_np, sp_lat = int(g["np_stride"][0]), int(g["np_stride"][1]), int(g["lat_stride"])
    else:
        sf, sp_np, sp_lat = 2, 4, 2
    npred = (eps[:nb] + 7.5 * (eps[nb:] - eps[:nb]))[:, :, ::sf, ::sp_np, ::sp_np].cpu()
    e_np = rel_l2(npred, torch.from_numpy(g["noise_pred_sub"]))
    e = rel_l2(got[:, :, :, ::sp_lat, ::sp_lat].cpu(), torch.from_numpy(g["latents_sub"]))
    record(f"{tag}_noise_pred", e_np)
    record(f"{tag}_latents", e)
    if "skip1_sub" in g.files:   # rel-L2 on sub-samples of stage tensors: a mis-scaled single block cannot hide behind an abs-mean
        B = 2 * nb
        hs = [h, h, h, h // 2, h // 2, h // 2, h // 4, h // 4, h // 4, h // 8, h // 8, h // 8]
        for name, idx, sp in (("skip1_sub", 1, 4), ("skip7_sub", 7, 2)):
            es = rel_l2(_sub5(_rows_to_5d(taps["skips"][idx], B, f, hs[idx]), sf, sp), torch.from_numpy(g[name]).float())
            record(f"{tag}_{name}", es)
            assert es <= 5e-3, (name, es)
        if not single:
            em = rel_l2(_sub5(_rows_to_5d(taps["motion"][4], 2, f, hs[4]), sf, 2), torch.from_numpy(g["motion4_sub"]).float())
            ec = rel_l2(_sub5(_rows_to_5d(taps["cn_down"][6], 1, f, hs[6]), sf, 2), torch.from_numpy(g["cn_down6_sub"]).float())
            record(f"{tag}_motion4_sub", em)
            record(f"{tag}_cn_down6_sub", ec)
            assert em <= 1e-2 and ec <= 5e-3, (em, ec)
    u.spatial_editor = u.temporal_editor = None
    assert torch.isfinite(got).all() and e_np <= NOISE_PRED_TOL and e <= STEP_TOL, (e_np, e)


def test_full_size_properties_configs4(unet, controlnet):
    """BASELINE configs[4] AS WRITTEN on one GPU -- 48 frames x 768^2 (96 x 96 latents), batch 4, two-branch + ControlNet + adapter, both editors --
    the 918-TFLOP workload no reference can be computed for in a test.  Size-independent properties of the whole step (pipeline_motion_editor.py:597-654
    at video_length = 48): (1) finite; (2) run-to-run bitwise; (3) the planned executor (me_denoise_step) reproduces the eager step bit for bit at this
    size; (4) K/V injection is one-way -- switching the editors on changes the EDIT row and leaves the reconstruction row where it was (to the fp16 noise
    floor: the edited launches batch their items differently); (5) classifier-free guidance + DDIM are affine in the guidance scale: the step at
    g = 4.75 is the midpoint of the steps at g = 2 and g = 7.5 (fp32 latents: to rounding).  The 96 x 96 geometry at 8 frames and the 48-frame count
    at 16 x 16 latents have oracle goldens of their own (test_step_96x96_latents_vs_golden, test_denoise_step_baseline_config0_and_48_frames_vs_cpu_oracle)."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    f, h = 48, 96
    x = step_inputs(f=f, h=h, w=h)
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    pipe.scheduler.set_timesteps(50)
    sed, ted = editors(unet, x["masks"])
    lat = x["latents"].cuda()
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * h).cuda()
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    t = pipe.scheduler.timesteps[4]

    def run(step, g=7.5, fn=None):
        sed.reset(); ted.reset()
        sed.cur_step = ted.cur_step = step
        return (fn or pipe.denoise_step)(lat, t, emb, images, g)

    a1 = run(4)
    a2 = run(4)
    assert torch.isfinite(a1).all() and float(a1.abs().max()) < 50.0
    assert torch.equal(a1, a2)                                             # (2)
    p1 = run(4, fn=pipe.denoise_step_planned)
    p2 = run(4, fn=pipe.denoise_step_planned)                              # the second call replays the recorded launch list
    assert torch.equal(p1, a1) and torch.equal(p2, a1)                     # (3)
    pipe.release_plans()
    off = run(0)                                                           # editors registered but before their start step: plain attention everywhere
    e_recon, e_edit = rel_l2(a1[0], off[0]), rel_l2(a1[1], off[1])
    record("configs4_full_size_recon_row_editors_on_vs_off", e_recon)
    record("configs4_full_size_edit_row_editors_on_vs_off", e_edit)
    assert e_recon < 3e-3 and e_edit > 5 * e_recon and e_edit > 1e-3, (e_recon, e_edit)     # (4)
    lo, mid = run(4, g=2.0), run(4, g=4.75)
    e_aff = float(((lo.double() + a1.double()) / 2 - mid.double()).abs().max() / mid.double().abs().max())
    record("configs4_full_size_guidance_affinity", e_aff)
    assert e_aff < 1e-5, e_aff                                             # (5)
    unet.spatial_editor = unet.temporal_editor = None


def test_step_config3_full_size_vs_golden():
    """BASELINE configs[2] at FULL size -- the benchmarked workload itself (24 frames x 64x64 latents, batch 4, ControlNet + adapter, both editors
    active, bench.py's inputs and weights) -- against tests/golden/step_config3.npz, which oracle/make_golden.py --only-config3 generated in the
    build container (one oracle step, ~12 min on 8 cores)."""
    _step_vs_golden("step_config3")


def test_step_single_branch_config2_vs_golden():
    """BASELINE configs[1]: 8 frames x 512^2 (64x64 latents), ONE clip through the single-branch UNet3D (no ControlNet, no adapter input, no
    editors), classifier-free guidance + DDIM -- against tests/golden/step_single.npz (oracle/make_golden.py --only-single)."""
    _step_vs_golden("step_single")


def test_step_single_branch_96x96_vs_reference_golden():
    """The 96 x 96-latent geometry of BASELINE configs[4] -- 9216 tokens, 18432 [prev | cur] keys per query, 18 query blocks per (item, head) -- single-branch
    UNet3D at 8 frames, classifier-free guidance + DDIM, against tests/golden/step_single96.npz: noise prediction and latents from the REFERENCE's own UNet
    (oracle/make_golden.py --only-single96: the reference model through the chunked xformers stand-in, oracle == reference asserted there)."""
    _step_vs_golden("step_single96")


def test_step_96x96_latents_vs_golden():
    """The spatial geometry of BASELINE configs[4] (768^2 images: 96x96 latents -- 9216 tokens / 18 query blocks of 512 / 72 + 72 (+ 72) stages of
    keys per item at level 0, 48x48 at level 1 (dh = 80, 2304 keys), 24x24, 12x12) on one GPU at 8 frames, two-branch + ControlNet + adapter, both
    editors active -- against tests/golden/step_geom96.npz (oracle/make_golden.py --only-geom96)."""
    _step_vs_golden("step_geom96")


@pytest.mark.parametrize("f,hw", [(8, 32), (48, 16)])
def test_denoise_step_baseline_config0_and_48_frames_vs_cpu_oracle(unet, controlnet, unet_sd_torch, cn_sd_torch, f, hw):
    """(8, 32): BASELINE configs[0]'s shape -- 8 frames x 256^2 -- one full two-branch step, editors active, vs the CPU oracle run on the box.
    (48, 16): BASELINE configs[4]'s frame count through the whole graph (six adapter chunks, 48-frame temporal attention in every block)."""
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from oracle import ref_cpu
    from test_step_cpu import step_inputs
    x = step_inputs(f=f, h=hw, w=hw)
    step = 4
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = step
    t = ddim.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * hw, 8 * hw)
    with torch.no_grad():
        want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, x["latents"], t, x["uncond"], x["cond"], images, sp, tp, 7.5)
    pipe = MotionEditorPipeline(unet=unet, controlnet=controlnet)
    sed, ted = editors(unet, x["masks"])
    sed.cur_step = ted.cur_step = step
    pipe.scheduler.set_timesteps(50)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
    got = pipe.denoise_step(x["latents"].cuda(), t, emb, images.cuda(), 7.5)
    unet.spatial_editor = unet.temporal_editor = None
    e = rel_l2(got, want)
    record(f"step4_f{f}_{hw}x{hw}_latents", e)
    assert e <= STEP_TOL, e


def test_null_text_optimization_on_the_gpu_vs_reference_golden(unet_sd_np):
    """util.null_optimization end to end on the GPU: the forward launch graph on a tape, the six backward primitives (five kernels /
    kernel compositions + the matrix-materialising attention backward), Adam on the embedding -- against the reference class as
    written (tests/golden/null_text.npz; 2 DDIM steps x 2 inner steps).  fp16 activations and fp16 inter-layer gradients (loss-scaled)
    vs the reference's fp32: the first gradient to 3e-2 rel-L2 (measured 1.7e-3), the first optimised embedding to 3e-3 where the gradient is significant (measured 6e-5)."""
    from conftest import GOLD
    from motioneditor_amd import util
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.schedulers import DDIMScheduler
    g = np.load(GOLD / "null_text.npz")
    T = torch.from_numpy
    unet = UNet2DConditionModel(unet_sd_np, device="cuda")
    sched = DDIMScheduler()
    sched.set_timesteps(50)

    class Pipe:
        pass
    pipe = Pipe()
    pipe.unet = unet
    grads = []
    out = util.null_optimization(pipe, sched, [t for t in T(g["latents"])], T(g["context"]), 2, 1e-5, num_ddim_steps=2, grads=grads)
    g0 = T(g["grad0"])
    rel = float((grads[0].float().cpu() - g0).norm() / g0.norm())
    big = g0.abs() > 5e-2 * g0.abs().max()
    ref = T(g["uncond_out"])
    d0 = float(((out[0].float().cpu() - ref[0]).abs() * big).max())
    # later steps: Adam's update is ~ lr * sign(g), so an element whose gradient sits at rounding level may land 2 lr away; what
    # must hold is that all but a handful of elements agree
    d1 = (out[1].float().cpu() - ref[1]).abs()
    frac = float((d1 < 2e-3).float().mean())
    print("null-text on GPU: grad rel-L2", rel, "step-1 embedding diff (significant elements)", d0, "step-2 elements within 2e-3:", frac)
    assert rel < 3e-2 and d0 < 3e-3 and frac > 0.99, (rel, d0, frac)   # measured 1.7e-3, 6e-5, 0.9963


def test_adapter_training_gradients_on_the_gpu_vs_reference_golden(unet_sd_np):
    """util.adapter_training_grads on the GPU: the launch graph on the tape, all backward and parameter-gradient primitives, the packed
    gradients mapped back to the reference's parameter names -- against what the reference UNet's autograd leaves in .grad
    (tests/golden/adapter_train.npz).  fp16 activations and loss-scaled fp16 inter-layer gradients vs the reference's fp32."""
    from conftest import GOLD
    from motioneditor_amd import util
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    g = np.load(GOLD / "adapter_train.npz")
    T = torch.from_numpy
    F32 = lambda k: T(g[k].astype(np.float32))   # noqa: E731
    unet = UNet2DConditionModel(unet_sd_np, device="cuda")
    loss, grads = util.adapter_training_grads(unet, F32("noisy"), int(g["t"]), F32("ehs"), [F32(f"down{i}") for i in range(12)], F32("mid"), F32("noise"))
    names = [str(n) for n in g["names"]]
    assert set(names) == set(grads)
    norms = np.array([float(grads[k].norm()) for k in names])
    live = g["grad_norms"] > 0                   # six 1x1-pixel attn_pose q / k projections get exactly zero gradient (one key: softmax = 1)
    assert float(norms[~live].max(initial=0.0)) < 1e-6
    rel = np.abs(norms[live] / g["grad_norms"][live] - 1)
    tot = float(np.sqrt((norms ** 2).sum()) / np.sqrt((g["grad_norms"] ** 2).sum()))
    fulls = []
    for i, k in enumerate(str(n) for n in g["full_names"]):
        want = T(g[f"full_{i}"])
        fulls.append(float((grads[k].float().cpu() - want).norm() / want.norm().clamp_min(1e-30)))
    print("adapter training on GPU: loss", loss, "vs", float(g["loss"]), " total grad norm ratio", tot, " median / max per-parameter norm error", float(np.median(rel)), float(rel.max()),
          " full tensors rel-L2", fulls)
    assert abs(loss - float(g["loss"])) < 5e-3 * float(g["loss"]) and abs(tot - 1) < 2e-2 and float(np.median(rel)) < 2e-2 and max(fulls) < 5e-2, (loss, tot, fulls)


def test_adapter_trainer_step_on_the_gpu_vs_oracle_autograd_adamw(unet_sd_np):
    """util.AdapterTrainer.step entirely on the device, on a world-1 RCCL group (the gradient bucket and the loss go through dist.all_reduce on
    device tensors): tape gradients accumulated into one flat fp32 bucket, global-norm clip from the device reduction, me_adamw on the packed
    fp32 masters, packed fp16 weights refreshed in place -- against the oracle's UNet under torch autograd + clip_grad_norm_ +
    torch.optim.AdamW on the same clip (train_adaptor.py:364-385).  Then a second step: the forward must read the UPDATED weights (the
    transposed-weight cache of the backward is invalidated), i.e. its loss equals the oracle's loss at the updated parameters."""
    import torch.distributed as dist
    from conftest import GOLD
    from motioneditor_amd import ops, util
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from oracle import ref_cpu
    g = np.load(GOLD / "adapter_train.npz")
    T = torch.from_numpy
    F32 = lambda k: T(g[k].astype(np.float32))   # noqa: E731
    clip = dict(noisy=F32("noisy"), t=int(g["t"]), ehs=F32("ehs"), down=[F32(f"down{i}") for i in range(12)], mid=F32("mid"), noise=F32("noise"))
    lr = 1e-3                                            # a visible step (the reference's 3e-5 moves fp32 weights by 1e-5)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        unet = UNet2DConditionModel(unet_sd_np, device="cuda")
        tr = util.AdapterTrainer(unet, lr=lr)
        loss = tr.step(clip["noisy"], clip["t"], clip["ehs"], clip["down"], clip["mid"], clip["noise"])
        cache1 = len(ops._wT_cache)     # transposes of the FROZEN weights the backward walked through; the trained ones were dropped again
        # oracle: the same step with torch autograd
        sd = {k: T(v) for k, v in unet_sd_np.items()}
        names = tr.names
        params = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
        opt = torch.optim.AdamW(list(params.values()), lr=lr, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        sd2 = dict(sd)
        sd2.update(params)
        l0 = torch.nn.functional.mse_loss(ref_cpu.unet_forward(sd2, clip["noisy"], clip["t"], clip["ehs"], clip["down"], clip["mid"]), clip["noise"])
        for k, gr in zip(names, torch.autograd.grad(l0, [params[k] for k in names])):
            params[k].grad = gr
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        got = tr.export_state_dict()
        num = sum(float((got[k] - params[k].detach()).pow(2).sum()) for k in names)
        den = sum(float((params[k].detach() - sd[k]).pow(2).sum()) for k in names)
        upd = (num / den) ** 0.5
        # AdamW's FIRST update is lr * g / (|g| + eps) ~ lr * sign(g): an element whose gradient is smaller than the fp16 path's error on it
        # (~1e-3 of the tensor's typical gradient) may land 2 lr away from the autograd step, whatever the implementation -- so the whole update
        # is bounded loosely, and tightly where the oracle's gradient is significant (> 1 % of its tensor's rms)
        num_s = den_s = 0.0
        zero_worst = 0.0
        for k in names:
            gk, dref, dgot = params[k].grad, params[k].detach() - sd[k], got[k] - sd[k]
            rms = float(gk.pow(2).mean().sqrt())
            if rms == 0.0:     # exactly zero gradient under autograd (single-key softmax at the 1x1-pixel level): weight decay only
                zero_worst = max(zero_worst, float((dgot - dref).abs().max()) / lr)
                continue
            sig = gk.abs() > 1e-2 * rms
            num_s += float(((dgot - dref) * sig).pow(2).sum())
            den_s += float((dref * sig).pow(2).sum())
        upd_s = (num_s / den_s) ** 0.5
        record("adapter_trainer_update_rel_l2", upd)
        record("adapter_trainer_update_rel_l2_significant", upd_s)
        record("adapter_trainer_zero_grad_tensors_worst_over_lr", zero_worst)
        print("adapter trainer on GPU: loss", loss, "vs", float(l0), " update rel-L2", upd, " on significant gradients", upd_s, " zero-gradient tensors: worst |delta| / lr", zero_worst)
        # measured on MI355X: loss 1.30982 vs 1.30988, whole update 2.8e-2, on significant gradients 1.0e-2, zero-gradient tensors exactly the weight-decay step
        assert abs(loss - float(l0)) < 5e-3 * float(l0) and upd < 5e-2 and upd_s < 2e-2 and zero_worst < 0.05, (loss, float(l0), upd, upd_s, zero_worst)
        # second step: forward on the updated weights
        with torch.no_grad():
            sd3 = dict(sd)
            sd3.update({k: v.detach() for k, v in params.items()})
            l1 = float(torch.nn.functional.mse_loss(ref_cpu.unet_forward(sd3, clip["noisy"], clip["t"], clip["ehs"], clip["down"], clip["mid"]), clip["noise"]))
        loss2 = tr.step(clip["noisy"], clip["t"], clip["ehs"], clip["down"], clip["mid"], clip["noise"])
        assert abs(loss2 - l1) < 5e-3 * l1 and abs(l1 - float(l0)) > 1e-5, (loss2, l1, float(l0))
        tr.step(clip["noisy"], clip["t"], clip["ehs"], clip["down"], clip["mid"], clip["noise"])
        assert len(ops._wT_cache) <= cache1, "transposed-weight cache grows with the training steps"
    finally:
        dist.destroy_process_group()


def test_training_example_sequence_on_the_gpu_matches_the_oracle_loss(unet_sd_np, cn_sd_np):
    """examples/train_adapter.py::step on the GPU -- VAE encode, add_noise, ControlNet, UNet + adapter on the tape, backward, clip, AdamW -- reports
    the oracle's loss for the same tensors and moves the adapter."""
    import sys
    sys.path.insert(0, str(ROOT / "examples"))
    import train_adapter as ex
    from motioneditor_amd import synth, util
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.models.vae import AutoencoderKL
    from oracle import ref_cpu
    vsd = synth.synth_state_dict(synth.vae_encoder_schema(), 33, salt="vae.")
    vae, unet, cn = AutoencoderKL(vsd, device="cuda"), UNet2DConditionModel(unet_sd_np, device="cuda"), ControlNetModel(cn_sd_np, device="cuda")
    tr = util.AdapterTrainer(unet, lr=1e-3)
    f, H, t = 8, 64, 401
    b = ex.training_batch(f, H, H)
    before = tr.export_state_dict()["controlnet_adapter.body.0.block2.weight"].clone()
    loss = ex.step(tr, vae, cn, b, t)
    T = torch.from_numpy
    usd, csd = {k: T(x) for k, x in unet_sd_np.items()}, {k: T(x) for k, x in cn_sd_np.items()}
    with torch.no_grad():
        lat = ref_cpu.vae_encode_sample({k: T(x) for k, x in vsd.items()}, b["pixel_values"].reshape(f, 3, H, H), b["encode_noise"])
        lat = lat.reshape(1, f, 4, 8, 8).permute(0, 2, 1, 3, 4) * 0.18215
        a = float(ex.alphas_cumprod()[t])
        noisy = a ** 0.5 * lat + (1 - a) ** 0.5 * b["noise"]
        down, mid = ref_cpu.controlnet_forward(csd, noisy.permute(0, 2, 1, 3, 4).reshape(f, 4, 8, 8), t, b["ehs"].repeat(f, 1, 1), b["skeleton"].reshape(f, 3, H, H))
        down = [d.reshape(1, f, *d.shape[1:]).permute(0, 2, 1, 3, 4) for d in down]
        mid = mid.reshape(1, f, *mid.shape[1:]).permute(0, 2, 1, 3, 4)
        want = float(torch.nn.functional.mse_loss(ref_cpu.unet_forward(usd, noisy, t, b["ehs"], down, mid), b["noise"]))
    record("train_example_loss_rel", abs(loss - want) / want)
    assert abs(loss - want) < 1e-2 * want, (loss, want)
    assert float((tr.export_state_dict()["controlnet_adapter.body.0.block2.weight"] - before).abs().max()) > 1e-4
