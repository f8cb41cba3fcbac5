"""Multi-process (world_size 2, gloo, CPU) check of the CFG-parallel denoising step: rank 0 runs the unconditional
(recon, edit) pair, rank 1 the conditional pair, ONE all-gather of the noise prediction, and both ranks must end with
the latents of the single-process step.  Runs the product pipeline / graph / editor code on tests/emu_ops.py."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, port, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    import emu_ops
    import motioneditor_amd.models.unet_2d_condition as u
    import motioneditor_amd.pipelines.pipeline_motion_editor as pm
    from motioneditor_amd import schedulers, synth
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
    from motioneditor_amd.models import graph
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    for m in (graph, u, pm, schedulers):
        m.ops = emu_ops
    x = step_inputs()
    f = x["latents"].shape[2]
    unet = UNet2DConditionModel(synth.synth_state_dict(synth.unet_schema()), device="cpu", dtype=torch.float32)
    cn = ControlNetModel(synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet."), device="cpu", dtype=torch.float32)
    pipe = MotionEditorPipeline(unet=unet, controlnet=cn)
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=x["masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed)
    pipe.scheduler.set_timesteps(50)
    step = 4
    t = pipe.scheduler.timesteps[step]
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 64, 64)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]])
    ted.cur_step = sed.cur_step = step
    got = pipe.denoise_step_cfg_parallel(x["latents"], t, emb, images, 7.5)
    assert (sed.cur_step, ted.cur_step, sed.cur_att_layer, ted.cur_att_layer) == (step + 1, step + 1, 0, 0)
    gathered = [torch.empty_like(got) for _ in range(2)]
    dist.all_gather(gathered, got)
    assert torch.equal(gathered[0], gathered[1]), "ranks disagree on the updated latents"
    if rank == 0:   # single-process reference of the same step
        ted.reset(); sed.reset()
        ted.cur_step = sed.cur_step = step
        want = pipe.denoise_step(x["latents"], t, emb, images, 7.5)
        err = float((got - want).abs().max() / want.abs().mean())
        torch.save({"err": err}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_parallel_two_ranks_equal_single_process(tmp_path):
    out = tmp_path / "r.pt"
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(out)), nprocs=2, join=True)
    err = torch.load(out)["err"]
    assert err < 1e-4, err
