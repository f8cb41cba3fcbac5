"""The frame-sharded step with REAL kernels and REAL (non-degenerate) exchanges on a one-GPU box: several ranks share cuda:0 -- RCCL refuses two
ranks on one device, so the exchanges are staged through host memory and a gloo group (parallel.HostStagedExchange, a verification adapter) -- and each
runs `denoise_step_frame_sharded` on its own frames through libmotioned: remote one-frame K|V halos for attn1, the frame<->pixel all-to-all of temporal
attention over R > 1 parts (me_copy_blocks + me_tattn with q_parts = kv_parts = R), the adapter's two-frame halos across a straddling 8-frame chunk,
TemporalConv halos with the interior / boundary row-range launches, GroupNorm statistics summed over ranks.  The gathered latents must be those of the
single-process step on the same GPU (the CPU tests prove the same pattern on the emulated ABI; the world-1 RCCL test proves the real kernels on
degenerate exchanges; this one closes the gap between them)."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, f, out_path, hybrid, temporal, adapter, hw=8, bench_inputs=False, overlap=False):
    import time
    t00 = time.time()

    def lap(what):   # ME_TEST_TIMES=<file>: where a multi-rank case spends its wall time (rank 0)
        log = os.environ.get("ME_TEST_TIMES")
        if log and rank == 0:
            with open(log, "a") as fh:
                fh.write(f"world {world} f {f} hw {hw} overlap {overlap}: {what} at {time.time() - t00:.1f} s\n")
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # the ranks pack 1.2 G parameters each on the host (fp32 -> packed fp16) at their first step: with torch's default of one thread per core in EVERY rank
    # four ranks took 360 s over it (8 x 8 and 64 x 64 latents alike: profiles/r06_shard_test_times.txt), two ranks 60 s
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // (2 * world))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from motioneditor_amd import parallel, synth
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    # bench_inputs: bench.py's own inputs (synth.bench_inputs) -- what tests/golden/step_config3.npz was generated on
    x = synth.bench_inputs(f, hw, hw) if bench_inputs else step_inputs(f=f, h=hw, w=hw)
    lap("imports + process group + inputs")
    unet = UNet2DConditionModel(synth.synth_state_dict(synth.unet_schema()), device="cuda")
    cn = ControlNetModel(synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet."), device="cuda")
    lap("models built")
    pipe = MotionEditorPipeline(unet=unet, controlnet=cn)
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10)
    regiter_temporal_attention_editor_diffusers(pipe, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=x["masks"])
    regiter_fully_attention_editor_diffusers(pipe, sed)
    pipe.scheduler.set_timesteps(50)
    step = 4
    t = pipe.scheduler.timesteps[step]
    H = x["skeleton"].shape[-1]
    images = x["skeleton"].reshape(f, 3, H, H).cuda()
    unc = x["uncond"][step] if bench_inputs else x["uncond"]      # (bench_inputs holds one unconditional embedding per step: null-text inversion)
    emb = torch.cat([unc.expand(2, 77, 768), x["cond"]]).cuda()
    lat = x["latents"].cuda()
    cfg_x = shard_group = side_group = None
    if hybrid:   # rank = shard * 2 + cfg half (bench.py's layout): CFG pairs {0,1},{2,3}; frame-shard groups {0,2},{1,3}
        ns = world // 2
        for s_ in range(ns):
            g = dist.new_group([2 * s_, 2 * s_ + 1])
            if rank // 2 == s_:
                cfg_x = parallel.exchange(g, "staged")
        for k in range(2):
            g = dist.new_group([2 * s_ + k for s_ in range(ns)])
            g2 = dist.new_group([2 * s_ + k for s_ in range(ns)]) if overlap else None     # the adapter's own communicator (FrameShard.side_shard)
            if rank % 2 == k:
                shard_group, side_group = g, g2
    elif overlap:
        side_group = dist.new_group(list(range(world)))
    parallel.reset_stats()
    shard = parallel.FrameShard(f, shard_group, temporal=temporal, adapter=adapter, comm="staged", side_group=side_group)
    lo, hi = shard.frame0, shard.frame0 + shard.f_loc
    ted.cur_step = sed.cur_step = step
    lap("groups + shard")
    got = pipe.denoise_step_frame_sharded(lat[:, :, lo:hi].contiguous(), t, emb, images[lo:hi].contiguous(), 7.5, shard, cfg_group=cfg_x)
    torch.cuda.synchronize()
    lap("sharded step")
    if overlap:   # the same step with ControlNet + adapter on the side stream (the adapter's exchanges on the second group): the same launches, bit for bit
        assert shard.side_shard is not None and not pipe.shard_overlap
        pipe.shard_overlap = True
        ted.cur_step = sed.cur_step = step
        ted.cur_att_layer = sed.cur_att_layer = 0
        got2 = pipe.denoise_step_frame_sharded(lat[:, :, lo:hi].contiguous(), t, emb, images[lo:hi].contiguous(), 7.5, shard, cfg_group=cfg_x)
        torch.cuda.synchronize()
        lap("overlapped sharded step")
        pipe.shard_overlap = False
        assert pipe._side_stream is not None
        assert torch.equal(got2, got), float((got2.double() - got.double()).norm() / got.double().norm())
        for v in parallel.STATS.values():     # two steps were counted
            v[0] //= 2
            v[1] //= 2
    assert (sed.cur_step, ted.cur_step, sed.cur_att_layer, ted.cur_att_layer) == (step + 1, step + 1, 0, 0)
    st = parallel.stats_summary()
    assert st["all_reduce(groupnorm stats)"]["calls_per_step"] == 45, st
    if shard.world > 1:
        assert st.get("p2p(TemporalConv halo)", {"calls_per_step": 0})["calls_per_step"] > 0, st
        if temporal == "a2a":   # 8 / 8 / 8 / 4 temporal attentions at the four levels (16 UNet blocks + 12 adapter blocks); a level whose pixel count the ranks do not divide all-gathers
            want_a2a = sum(n for n, lv in ((8, 0), (8, 1), (8, 2), (4, 3)) if ((hw >> lv) ** 2) % shard.world == 0)
            assert st["all_to_all(temporal in)"]["calls_per_step"] == st["all_to_all(temporal out)"]["calls_per_step"] == want_a2a, st
    parts = [torch.empty(got.shape, dtype=got.dtype) for _ in range(world)]
    dist.all_gather(parts, got.cpu())
    if hybrid:
        assert torch.equal(parts[rank], parts[rank ^ 1])     # both members of a CFG pair hold the same frames and must agree exactly
        parts = parts[0::2]
    full = torch.cat(parts, dim=2)
    if rank == 0:   # the single-process step of the same inputs on the same GPU
        ted.reset(); sed.reset()
        ted.cur_step = sed.cur_step = step
        want = pipe.denoise_step(lat, t, emb, torch.cat([images] * 2), 7.5).cpu()
        lap("plain step")
        err = float((full.double() - want.double()).norm() / want.double().norm())
        torch.save({"err": err, "stats": st, "latents": full if bench_inputs else None}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, world, f, hybrid=False, temporal="a2a", adapter="halo", hw=8, bench_inputs=False, overlap=False):
    from motioneditor_amd import synth
    synth.synth_state_dict(synth.unet_schema())                               # fill the per-machine weight cache once: the ranks map it instead of
    synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")     # generating 1.7 G parameters each
    out = tmp_path / "r.pt"
    port = 29100 + (os.getpid() % 2000) + world * 7 + f + (3 if hybrid else 0) + hw
    mp.spawn(_worker, args=(world, port, f, str(out), hybrid, temporal, adapter, hw, bench_inputs, overlap), nprocs=world, join=True)
    return torch.load(out)


def test_two_ranks_on_one_gpu_frame_sharded_step_equals_the_plain_step(tmp_path):
    """24 frames over 2 ranks (the 8-frame chunk [8, 16) straddles the boundary at frame 12).  Not bitwise: a rank's launches are half as tall (other
    tile shapes on some layers) and the GroupNorm statistics are summed in another order -- the yardstick is the per-kernel fp16 tolerance."""
    r = _run(tmp_path, 2, 24)
    from test_model_gpu import record
    record("frame_shard_two_ranks_one_gpu", r["err"])
    assert r["err"] < 2e-3, r


def test_hybrid_cfg_x_frames_four_ranks_on_one_gpu(tmp_path):
    """The 8-GPU default layout at half size (CFG pair x 2 frame shards = 4 ranks on the one GPU): every frame-shard exchange at batch 2, the pair's
    all-gather of the noise prediction, K|V all-gathers instead of halos (`--shard-exchange gather`, the flavour BASELINE configs[3] names)."""
    r = _run(tmp_path, 4, 16, hybrid=True, temporal="gather", adapter="gather")
    from test_model_gpu import record
    record("frame_shard_hybrid_four_ranks_one_gpu", r["err"])
    assert r["err"] < 2e-3, r


def test_four_frame_shards_on_one_gpu_the_layout_of_baseline_configs3(tmp_path):
    """BASELINE configs[3] as written -- 24 frames sharded 6 per rank over 4 ranks -- at 8 x 8 latents: the frame<->pixel all-to-all over 4 parts, the
    adapter's chunk-first / previous frames fetched from other ranks for every rank but the first (ranges start at frames 6, 12, 18 of chunks that start
    at 0, 8, 16), TemporalConv halos on both sides of the two middle ranks."""
    r = _run(tmp_path, 4, 24)
    from test_model_gpu import record
    record("frame_shard_four_ranks_one_gpu_config3_layout", r["err"])
    assert r["err"] < 2e-3, r


def test_four_frame_shards_at_configs3_size(tmp_path):
    """BASELINE configs[3] AS WRITTEN, at its own size: 24 frames x 512^2 (64 x 64 latents), batch 4, two-branch + ControlNet + adapter + both editors,
    the frames sharded 6 per rank over 4 ranks (here: 4 processes sharing the one GPU, exchanges staged through the host) -- the gathered latents against
    tests/golden/step_config3.npz (the oracle's step at this very workload, oracle/make_golden.py --only-config3) at the step tolerance, and against
    the plain step on the same GPU.  Level-0 launches of a rank are 6 frames x 4096 pixels x batch 4 = 98304 rows: the real tile shapes, the 4-part
    frame<->pixel all-to-all at 1024 pixels per part, one-frame K | V halos of 4096 keys."""
    import numpy as np
    from conftest import GOLD
    from test_model_gpu import STEP_TOL, record
    from conftest import rel_l2
    r = _run(tmp_path, 4, 24, hw=64, bench_inputs=True)
    g = np.load(GOLD / "step_config3.npz")
    assert int(g["frames"]) == 24 and int(g["latent"]) == 64
    sp_lat = int(g["lat_stride"]) if "lat_stride" in g.files else 2
    e = rel_l2(r["latents"][:, :, :, ::sp_lat, ::sp_lat], torch.from_numpy(g["latents_sub"]))
    record("frame_shard_four_ranks_configs3_size_vs_plain", r["err"])
    record("frame_shard_four_ranks_configs3_size_vs_golden", e)
    assert r["err"] < 2e-3, r["err"]
    assert e <= STEP_TOL, e


def test_eight_frame_shards_the_layout_of_baseline_configs4(tmp_path):
    """BASELINE configs[4]'s layout -- 48 frames sharded 6 per rank over 8 ranks (8 processes on the one GPU) -- at 32 x 32 latents: the world-8 frame<->pixel
    all-to-all (128 pixels per part at level 0, 2 at level 3), the adapter's chunk halos on all 7 rank boundaries (ranges start at frames 6, 12, ..., 42 of
    chunks that start at 0, 8, ..., 40), TemporalConv halos on both sides of the six middle ranks, GroupNorm statistics summed over 8 ranks -- against the plain
    step on the same GPU (the 48-frame count against the oracle: test_denoise_step_baseline_config0_and_48_frames_vs_cpu_oracle)."""
    from test_model_gpu import record
    r = _run(tmp_path, 8, 48, hw=32)
    record("frame_shard_eight_ranks_one_gpu_configs4_layout", r["err"])
    assert r["err"] < 2e-3, r


@pytest.mark.parametrize("hybrid", [False, True])
def test_sharded_step_with_controlnet_and_adapter_on_the_side_stream(tmp_path, hybrid):
    """`--shard-overlap` (round 6): inside the frame-sharded step the ControlNet and the content-aware adapter run on the side stream beside the UNet's down path
    and mid block -- as in the single-process step -- with the adapter's chunk halos, TemporalConv halos and frame<->pixel all-to-alls on a SECOND process
    group (parallel.FrameShard.side_shard), so that the two streams' exchanges do not queue behind each other.  4 ranks on the one GPU (frames4, and the
    CFG pair x frames2 hybrid): the overlapped step must equal the serialised sharded step bit for bit (same launches, other streams) and the plain step to
    the usual tolerance."""
    r = _run(tmp_path, 4, 16 if hybrid else 24, hybrid=hybrid, overlap=True)
    from test_model_gpu import record
    record("frame_shard_overlap_" + ("hybrid" if hybrid else "frames4"), r["err"])
    assert r["err"] < 2e-3, r


def test_eight_frame_shards_at_configs4_size(tmp_path):
    """BASELINE configs[4] AS WRITTEN, at its own size and in its own layout: 48 frames x 768^2 (96 x 96 latents), batch 4, two-branch + ControlNet + adapter + both
    editors, the frames sharded 6 per rank over 8 ranks (8 processes sharing the one GPU, exchanges staged through the host) -- against the plain step of the same
    inputs on the same GPU (the 918-TFLOP workload has no oracle fixture: the unsharded step's own checks are test_full_size_properties_configs4, the 96 x 96
    geometry and the 48-frame count have oracle goldens of their own).  A rank's level-0 launches are 6 frames x 9216 pixels x batch 4 = 221184 rows; the world-8
    frame<->pixel all-to-all moves 1152 pixels per part."""
    from test_model_gpu import record
    r = _run(tmp_path, 8, 48, hw=96)
    record("frame_shard_eight_ranks_configs4_size", r["err"])
    assert r["err"] < 2e-3, r
