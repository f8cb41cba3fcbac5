"""End-to-end `MotionEditorPipeline.__call__` on the emulated ABI (reference pipeline_motion_editor.py:505-667 as
inference.py:305-323 drives it): two prompts, target skeleton, per-step null-text embeddings, both editors with their
own counters, two DDIM steps, VAE decode to a video tensor -- against the oracle loop."""
import torch

import emu_ops
from motioneditor_amd import schedulers, synth
from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                           regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
from motioneditor_amd.models import graph, vae
from motioneditor_amd.models.controlnet import ControlNetModel
from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
from motioneditor_amd.pipelines import MotionEditorPipeline
from oracle import ref_cpu
from test_step_cpu import step_inputs


def test_call_two_steps_with_vae_decode_matches_oracle_loop(monkeypatch, unet_sd_np, cn_sd_np, unet_sd_torch, cn_sd_torch):
    import motioneditor_amd.models.unet_2d_condition as u
    import motioneditor_amd.pipelines.pipeline_motion_editor as pm
    for m in (graph, u, pm, schedulers, vae):
        monkeypatch.setattr(m, "ops", emu_ops)
    x = step_inputs()
    f = x["latents"].shape[2]
    vae_sd_np = synth.synth_state_dict(synth.vae_decoder_schema(), salt="vae.")
    g = torch.Generator().manual_seed(9)
    uncs = [x["uncond"] + 0.05 * torch.randn(x["uncond"].shape, generator=g) for _ in range(2)]
    lat0 = x["latents"] * 0.5

    # ---- oracle: two steps (the second with the editors active), then decode ----
    ddim = ref_cpu.DDIM()
    ddim.set_timesteps(2)
    sp, tp = ref_cpu.SpatialEditor(x["masks"], start_step=1), ref_cpu.TemporalEditor(start_step=1)
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    want = lat0
    with torch.no_grad():
        for i, t in enumerate(ddim.timesteps):
            want = ref_cpu.denoise_step(unet_sd_torch, cn_sd_torch, ddim, want, t, uncs[i], x["cond"], images, sp, tp, 7.5)
        video = ref_cpu.decode_latents({k: torch.from_numpy(v) for k, v in vae_sd_np.items()}, want)

    # ---- product ----
    pipe = MotionEditorPipeline(vae=vae.AutoencoderKL(vae_sd_np, device="cpu", dtype=torch.float32),
                                unet=UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32),
                                controlnet=ControlNetModel(cn_sd_np, device="cpu", dtype=torch.float32))
    ted = TemporalSelfAttentionControl(start_step=1, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=1, start_layer=10, source_masks=x["masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed)
    seen = []
    out = pipe(["a source prompt", "a target prompt"], video_length=f, height=64, width=64, num_inference_steps=2, guidance_scale=7.5,
               latents=lat0, uncond_embeddings=uncs, skeleton=x["skeleton"], source_masks=x["masks"], text_embeddings=x["cond"],
               output_type="tensor", callback=lambda i, t, l: seen.append((i, int(t))))
    assert seen == [(0, 501), (1, 1)]
    assert (sed.cur_step, ted.cur_step) == (2, 2)
    got = out.images
    assert got.shape == (2, 3, f, 64, 64) and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    assert float((got - video).abs().max()) < 2e-3
    # latents path as well
    sed.reset(), ted.reset()
    lat = pipe(["a", "b"], video_length=f, height=64, width=64, num_inference_steps=2, latents=lat0, uncond_embeddings=uncs,
               skeleton=x["skeleton"], text_embeddings=x["cond"], output_type="latent").images
    assert float((lat - want).abs().max() / want.abs().mean()) < 2e-4


def test_step_cache_is_an_lru_that_closes_what_it_evicts():
    """MotionEditorPipeline._plans / _graphs (round-4 advisor finding): every entry pins one step's activations, so the tables are LRU-bounded by
    `max_cached_steps`, an evicted entry's plan is closed, a hit refreshes its recency, and release_plans() empties both."""
    class Unet:
        device = torch.device("cpu")
        spatial_editor = temporal_editor = None
    closed = []

    class Plan:
        def __init__(self, n):
            self.n = n

        def close(self):
            closed.append(self.n)

    pipe = MotionEditorPipeline(unet=Unet())
    pipe.max_cached_steps = 2
    for n in range(3):
        pipe._cache_put(pipe._plans, ("k", n), dict(plan=Plan(n), lat=torch.zeros(1)))
        if n == 1:
            assert pipe._cache_get(pipe._plans, ("k", 0)) is not None      # 0 becomes the most recent: 1 is evicted next
    assert closed == [1] and list(pipe._plans) == [("k", 0), ("k", 2)]
    pipe._cache_put(pipe._graphs, "g", dict(graph=object()))
    pipe.release_plans()
    assert sorted(closed) == [0, 1, 2] and not pipe._plans and not pipe._graphs
