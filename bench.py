"""bench.py -- denoise-steps/sec of MotionEditor's two-branch DDIM step on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]                       (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one iteration of the reference loop body (pipeline_motion_editor.py:603-648): ControlNet on
the two edit rows, one batch-4 UNet3D forward with the content-aware motion adapter and both attention
editors ACTIVE (steps >= 4, i.e. 46 of the 50 steps of a run), classifier-free guidance, DDIM update.
Workload = BASELINE.json configs[2]: 24 frames x 512^2 (64x64 latents), synthetic inputs, seeded random
weights of the SD-1.5 / ControlNet-openpose / adapter architectures (no checkpoints exist offline).

Multi-GPU (round 1): one process per GPU.  Even N (default `--parallel cfg`): GPUs pair up and split ONE clip along
the classifier-free-guidance axis (rank 2i: unconditional (recon, edit) pair, rank 2i+1: conditional pair); the only
data-path exchange is one RCCL all-gather of the 4-channel noise prediction per step; N/2 clips run side by side.
`--parallel replicas`: every rank denoises its own clip, no collective.  value = clips*steps / max-over-ranks time.
Frame sharding with the RCCL temporal-K/V all-gather is the next row.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel family, HIP-event timed on the launch
stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
REF_TFLOP_PER_STEP = {"c3": 145.7}  # BASELINE.md §2, reference semantics (2*MAC)


def build_inputs(f, h, w, seed=33):
    from motioneditor_amd import synth
    T = torch.from_numpy
    return dict(latents=T(synth.synth_normal("bench.latents", (2, 4, f, h, w), seed)),
                uncond=[T(synth.synth_normal(f"bench.uncond{i}", (1, 77, 768), seed, 0.3)) for i in range(50)],
                cond=T(synth.synth_normal("bench.cond", (2, 77, 768), seed, 0.3)),
                skeleton=T(np.clip(synth.synth_normal("bench.skel", (1, f, 3, 8 * h, 8 * w), seed, 0.5) + 0.5, 0, 1).astype(np.float32)),
                masks=T(synth.synth_masks(f, 8 * h, 8 * w)))


def make_pipeline(device, usd, csd, masks):
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    pipe = MotionEditorPipeline(unet=UNet2DConditionModel(usd, device), controlnet=ControlNetModel(csd, device))
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=masks)
    regiter_fully_attention_editor_diffusers(pipe, sed)
    pipe.scheduler.set_timesteps(50)
    return pipe, sed, ted


def cpu_baseline(usd, csd, budget_s=25.0):
    """CPU oracle (oracle/ref_cpu.py, fp32 torch) on a bounded sample: ONE full two-branch step with editors
    active at 8 frames x 64^2 (8x8 latents), torch's default thread count; scaled to the bench workload by the
    reference-semantics FLOP ratio."""
    from oracle import ref_cpu
    from motioneditor_amd import synth
    cores = torch.get_num_threads()   # torch's default (physical cores); forcing every SMT thread made it 20x slower
    f, h, w = 8, 8, 8
    x = build_inputs(f, h, w)
    to = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}  # noqa: E731
    u, c = to(usd), to(csd)
    ddim = ref_cpu.DDIM()
    sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
    sp.cur_step = tp.cur_step = 4
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * w)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_cpu.denoise_step(u, c, ddim, x["latents"], ddim.timesteps[4], x["uncond"][4], x["cond"], images, sp, tp, 7.5)
    dt = time.perf_counter() - t0
    return dt, cores, f, h, w


def step_tflop(f, h, w):
    """Reference-semantics TFLOP of one two-branch step, scaled from BASELINE.md's config-3 figure by token count
    (exact at config 3; attention terms scale super-linearly in h*w, so other sizes are approximate)."""
    return REF_TFLOP_PER_STEP["c3"] * (f * h * w) / (24 * 64 * 64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--latent", type=int, default=64, help="latent height = width (image size / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--parallel", choices=["cfg", "replicas", "frames"], default="cfg",
                    help="N > 1: CFG-parallel GPU pairs (even N), independent replicas, or ONE clip with its frames sharded over all N ranks")
    ap.add_argument("--shapes", action="store_true", help="print the GEMM shapes with the largest time share to stderr")
    ap.add_argument("--editors", choices=["active", "inactive"], default="active",
                    help="secondary measurement: 'inactive' times the un-edited step (steps 0-3 of a 50-step run); the headline metric is 'active'")
    ap.add_argument("--inversion", action="store_true",
                    help="secondary measurement (SURVEY 8f rank 1): DDIM-inversion steps/s -- single-branch UNet with normal_infer + next_step, B = 1")
    ap.add_argument("--vae-decode", action="store_true",
                    help="secondary measurement (SURVEY 8f rank 2): frames/s of the VAE decoder (64x64 latents -> 512x512), no UNet involved")
    ap.add_argument("--no-overlap", action="store_true", help="A/B: run ControlNet on the main stream instead of beside the UNet's down path")
    ap.add_argument("--zero-tconv", action="store_true",
                    help="secondary measurement: UNet TemporalConv weights exactly zero, as in real checkpoints (resnet_2d.py:15-16) -> the launch is skipped")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from motioneditor_amd import capi, ops, synth
    capi.lib()  # no HIP library -> hard failure (no fallback path exists)
    if args.vae_decode:
        if dist_on:
            raise SystemExit("--vae-decode is a single-GPU secondary measurement")
        from motioneditor_amd.models.vae import AutoencoderKL
        vae = AutoencoderKL.from_synthetic(device)
        z = torch.from_numpy(synth.synth_normal("bench.vae", (args.frames, 4, args.latent, args.latent), 33)).to(device)
        for _ in range(args.warmup):
            vae.decode(z)
        torch.cuda.synchronize()
        ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = vae.decode(z).sample
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof, ops.PROFILE = ops.PROFILE, None
        fam = {}
        for name, fl, by, e0, e1, detail in prof:
            d = fam.setdefault(name, [0.0, 0.0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += fl
        assert torch.isfinite(img).all()
        print(json.dumps({"metric": "vae-decode frames/sec (SD-1.5 AutoencoderKL decoder)", "value": round(args.frames * args.steps / dt, 3), "unit": "frames/s",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
                          "dtype": "f16", "data": "synthetic", "config": {"workload": f"{args.frames} latent frames {args.latent}x{args.latent} -> {8 * args.latent}x{8 * args.latent} images"},
                          "kernel_families": {k: {"ms_per_step": round(v[0] / args.steps * 1e3, 2), "tflops": round(v[1] / v[0] / 1e12, 1) if v[0] else 0} for k, v in sorted(fam.items())}}))
        return
    usd = synth.synth_state_dict(synth.unet_schema())
    csd = synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")
    if args.zero_tconv:
        usd = {k: (np.zeros_like(v) if ".temp_conv" in k and not k.startswith("controlnet_adapter.") else v) for k, v in usd.items()}
    f, h, w = args.frames, args.latent, args.latent
    cfg_par = dist_on and args.parallel == "cfg" and world % 2 == 0
    group = None
    if cfg_par:
        pairs = [dist.new_group([2 * i, 2 * i + 1]) for i in range(world // 2)]   # every rank creates every group
        group = pairs[rank // 2]
    frame_par = dist_on and args.parallel == "frames"
    shard = None
    if frame_par:
        from motioneditor_amd import parallel
        shard = parallel.FrameShard(f)               # f / world frames per rank
    clip = 0 if frame_par else (rank // 2 if cfg_par else rank)
    n_clips = 1 if frame_par else (world // 2 if cfg_par else world)
    x = build_inputs(f, h, w, seed=33 + clip)   # one clip per rank (replicas), per GPU pair (CFG-parallel) or for all ranks (frames)
    pipe, sed, ted = make_pipeline(device, usd, csd, x["masks"])
    pipe.overlap_controlnet = pipe.overlap_adapter = not args.no_overlap
    images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * w).to(device)
    lat = x["latents"].to(device)
    if frame_par:   # this rank's frames only
        lo, hi = shard.frame0, shard.frame0 + shard.f_loc
        images = images[lo:hi].contiguous()
        lat = lat[:, :, lo:hi].contiguous()
    cond = x["cond"].to(device)
    unc = [u.to(device) for u in x["uncond"]]
    ts = pipe.scheduler.timesteps

    if args.inversion:
        from motioneditor_amd import util
        pipe.unet.spatial_editor = pipe.unet.temporal_editor = None
        lat = lat[:1].contiguous()

    def run_step(i, lat):
        if args.inversion:   # one body of util.ddim_loop (reference util.py:118-123)
            return util.ddim_loop(pipe, pipe.scheduler, lat, 1, normal_infer=True, text_embeddings=cond[:1])[-1]
        if args.editors == "inactive":
            sed.cur_step = ted.cur_step = 0      # the editors count steps themselves: hold them before start_step
        emb = torch.cat([unc[i].expand(2, 77, 768), cond])
        if frame_par:
            return pipe.denoise_step_frame_sharded(lat, ts[i], emb, images, 7.5, shard)
        if cfg_par:
            return pipe.denoise_step_cfg_parallel(lat, ts[i], emb, images, 7.5, group=group)
        return pipe.denoise_step(lat, ts[i], emb, images, 7.5)

    sed.cur_step = ted.cur_step = 4 if args.editors == "active" else 0   # active: the steady-state step (46 of 50)
    i0 = 4
    for k in range(args.warmup):
        lat = run_step(i0 + k, lat)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        lat = run_step(i0 + args.warmup + k, lat)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # Roofline pass: the SAME steps once more, with a HIP event pair around every launch and on a single stream.  It is
    # not folded into the timed region because (a) ~1100 event pairs per step cost ~6 % of the step and (b) the timed
    # region overlaps two streams (ControlNet + adapter beside the UNet), which stretches every kernel's own duration
    # by whatever shares the GPU with it.  profiles/*kernel_stats* is the rocprofv3 trace of `--no-overlap`.
    prof = None
    if not args.no_profile:
        ov = (getattr(pipe, "overlap_controlnet", False), getattr(pipe, "overlap_adapter", False))
        pipe.overlap_controlnet = pipe.overlap_adapter = False
        if rank == 0:
            ops.PROFILE = []
        lat2 = lat
        for k in range(args.steps):
            lat2 = run_step(i0 + args.warmup + k, lat2)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        pipe.overlap_controlnet, pipe.overlap_adapter = ov
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(lat).all(), "non-finite latents"

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = n_clips * args.steps / dt
        out = {"metric": ("ddim-inversion steps/sec (single-branch UNet3D, normal_infer)" if args.inversion else
                          f"denoise-steps/sec, {f}f x {8 * h}^2 two-branch UNet3D+ControlNet(+adapter+K/V injection)"), "value": round(value, 4),
               "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "strong" if (frame_par or (cfg_par and world == 2)) else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[2]: case-1 shape, {f} frames x {8*h}x{8*w}, two-branch + ControlNet + adapter + K/V injection "
                                      f"(editors {args.editors}), 1 DDIM step = 1 unit; seeded random SD-1.5-architecture weights",
                          "frames": f, "latent_hw": [h, w], "batch": 4, "guidance": 7.5, "editors": args.editors, "zero_temporal_conv": bool(args.zero_tconv), "controlnet_dedup": bool(pipe.dedup_controlnet and f % 2 == 0), "controlnet_side_stream": bool(pipe.overlap_controlnet and world == 1),
                          "parallelism": "single GPU" if world == 1 else (f"cfg2 x dp{world // 2}: each GPU pair splits one clip along the CFG axis "
                                                                          f"(one RCCL all-gather of the noise prediction per step), {world // 2} clip(s) side by side"
                                                                          if cfg_par else (f"frames{world}: one clip, {f // world} frames per GPU; RCCL all-gather of K|V (attn1, adapter, temporal attention), "
                                                                                           f"TemporalConv halos, GroupNorm-statistic all-reduce" if frame_par
                                                                                           else f"dp{world}: one independent clip per GPU, no data-path collective"))},
               "step_tflop_reference_semantics": round(step_tflop(f, h, w), 2),
               "achieved_tflops_whole_job": round(step_tflop(f, h, w) * n_clips * args.steps / dt, 1)}
        if prof:
            fam = {}
            shapes = {}
            for name, fl, by, e0, e1, detail in prof:
                d = fam.setdefault(name, [0.0, 0.0, 0.0, 0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += fl
                d[2] += by
                d[3] += 1
                if detail:
                    sd = shapes.setdefault(detail, [0.0, 0.0, 0])
                    sd[0] += e0.elapsed_time(e1) * 1e-3
                    sd[1] += fl
                    sd[2] += 1
            if args.shapes:
                for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:60]:
                    print(f"[shape] {k:44s} {v[0] / args.steps * 1e3:8.2f} ms/step {v[2] // args.steps:4d} launches {v[1] / v[0] / 1e12:7.1f} TF/s", file=sys.stderr)
            tot = sum(v[0] for v in fam.values())
            dom = max(fam, key=lambda k: fam[k][0])
            tsec, fl, by, n = fam[dom]
            traffic = None   # HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_summary.py), same workload only
            pmc = ROOT / "profiles" / "pmc_traffic.json"
            if pmc.exists() and (f, h, w) == (24, 64, 64):
                traffic = json.loads(pmc.read_text()).get(dom, {}).get("hbm_bytes_per_launch")
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(fl / tsec / 1e12, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(fl / tsec / 1e12 / PEAK_MFMA_TFLOPS, 4), "traffic": None if traffic is None else round(traffic),
                               "algorithmic_bytes_per_launch": round(by / n),
                               "launches_per_step": n // args.steps, "avg_launch_ms": round(tsec / n * 1e3, 4),
                               "share_of_gpu_time": round(tsec / tot, 3)}
            out["roofline"]["measured_in"] = ("event-instrumented single-stream pass of the same steps inside this run, right after the timed region "
                                              "(the timed region is un-instrumented and overlaps two streams)")
            out["kernel_families"] = {k: {"ms_per_step": round(v[0] / args.steps * 1e3, 2), "tflops": round(v[1] / v[0] / 1e12, 1) if v[0] else 0,
                                          "gbs": round(v[2] / v[0] / 1e9, 1) if v[0] else 0, "launches_per_step": v[3] // args.steps} for k, v in sorted(fam.items())}
        if world == 1 and not args.no_cpu_baseline:
            cdt, cores, cf, ch, cw = cpu_baseline(usd, csd)
            scale = step_tflop(f, h, w) / step_tflop(cf, ch, cw)
            out["cpu_baseline"] = {"value": round(1.0 / (cdt * scale), 6), "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": f"oracle/ref_cpu.py (fp32 torch CPU restatement of the reference step) timed on ONE full two-branch step at {cf} frames x "
                                             f"{8*ch}x{8*cw} ({cdt:.1f} s on {cores} threads), scaled to the bench workload by the token ratio x{scale:.0f}",
                                   "sample_seconds": round(cdt, 2)}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
