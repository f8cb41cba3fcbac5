"""Multi-process (gloo, CPU) check of the FRAME-SHARDED denoising step (SURVEY.md 8e / BASELINE config 4): the clip's frames
are split over the ranks; K|V all-gathers (attn1, adapter sparse-causal, temporal attention), one-frame halos (TemporalConv)
and GroupNorm-statistic all-reduces must reproduce the single-process step.  f = 16 over 2 ranks (8-frame adapter chunks
aligned with the shards) and f = 24 over 2 ranks (the chunk [8, 16) straddles the rank boundary).  Product pipeline / graph /
editor code on tests/emu_ops.py."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, f, out_path, hybrid=False, temporal="a2a", adapter="halo", comm="torch"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_ops
    import motioneditor_amd.models.unet_2d_condition as u
    import motioneditor_amd.pipelines.pipeline_motion_editor as pm
    from motioneditor_amd import parallel, schedulers, synth
    from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                               regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
    from motioneditor_amd.models import graph
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from test_step_cpu import step_inputs
    for m in (graph, u, pm, schedulers):
        m.ops = emu_ops
    x = step_inputs(f=f, h=8, w=8)
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
    H = x["skeleton"].shape[-1]
    images = x["skeleton"].reshape(f, 3, H, H)
    emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]])
    cfg_group = shard_group = None
    if hybrid:   # rank = shard * 2 + cfg half (bench.py's layout): CFG pairs {0,1},{2,3}; frame-shard groups {0,2},{1,3}
        ns = world // 2
        for s_ in range(ns):
            g = dist.new_group([2 * s_, 2 * s_ + 1])
            if rank // 2 == s_:
                cfg_group = g
        for k in range(2):
            g = dist.new_group([2 * s_ + k for s_ in range(ns)])
            if rank % 2 == k:
                shard_group = g
    parallel.reset_stats()
    shard = parallel.FrameShard(f, shard_group, temporal=temporal, adapter=adapter, comm=comm)
    if comm == "staged" and cfg_group is not None:
        cfg_group = parallel.exchange(cfg_group, "staged")
    lo, hi = shard.frame0, shard.frame0 + shard.f_loc
    ted.cur_step = sed.cur_step = step
    got = pipe.denoise_step_frame_sharded(x["latents"][:, :, lo:hi].contiguous(), t, emb, images[lo:hi].contiguous(), 7.5, shard, cfg_group=cfg_group)
    assert (sed.cur_step, ted.cur_step, sed.cur_att_layer, ted.cur_att_layer) == (step + 1, step + 1, 0, 0)
    st = parallel.stats_summary()
    # exchange budget of one step: 16 attn1 halos, 45 GroupNorm all-reduces, 12 K|V all-gathers (adapter
    # sparse-causal attention) and 28 temporal attentions: frame<->pixel all-to-all pairs, or K|V all-gathers where the
    # pixel count of the level does not divide over the shards (the 1x1 level of these 8x8 latents) / under temporal="gather"
    a2a = st.get("all_to_all(temporal in)", {"calls_per_step": 0})["calls_per_step"]
    assert a2a == st.get("all_to_all(temporal out)", {"calls_per_step": 0})["calls_per_step"] == (24 if temporal == "a2a" else 0), st
    halo = 12 if adapter == "halo" else 0   # the adapter's 12 sparse-causal attentions fetch <= 2 halo frames instead of the all-gather
    assert st["all_gather(K|V rows)"]["calls_per_step"] + a2a + halo == 40 and st["all_reduce(groupnorm stats)"]["calls_per_step"] == 45, st
    if adapter == "halo":
        sends = sum(1 for r in range(shard.world) if r != shard.rank and any(g is not None and g // shard.f_loc == shard.rank for g in shard.chunk_view(8).needs(r)))
        assert st.get("p2p(adapter K|V halo)", {"calls_per_step": 0})["calls_per_step"] == 12 * sends, st
    parts = [torch.empty_like(got) for _ in range(world)]
    dist.all_gather(parts, got)
    if hybrid:   # both members of a CFG pair hold the same frames and must agree exactly
        assert torch.equal(parts[rank], parts[rank ^ 1])
        parts = parts[0::2]
    full = torch.cat(parts, dim=2)
    if rank == 0:   # single-process reference of the same step
        ted.reset(); sed.reset()
        ted.cur_step = sed.cur_step = step
        want = pipe.denoise_step(x["latents"], t, emb, torch.cat([images] * 2), 7.5)
        torch.save({"err": float((full - want).abs().max() / want.abs().mean())}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("f,temporal", [(16, "a2a"), (24, "a2a"), (24, "gather")])
def test_frame_sharded_step_equals_single_process(tmp_path, f, temporal):
    out = tmp_path / "r.pt"
    port = 29700 + (os.getpid() % 2000) + f + (3 if temporal == "gather" else 0)
    mp.spawn(_worker, args=(2, port, f, str(out), False, temporal, "halo" if temporal == "a2a" else "gather"), nprocs=2, join=True)
    err = torch.load(out)["err"]
    assert err < 1e-4, err


def test_host_staged_exchange_adapter_carries_the_sharded_step(tmp_path):
    """parallel.HostStagedExchange (the verification adapter behind tests/test_frame_shard_gpu.py: several ranks on ONE GPU, exchanges staged through host
    memory and gloo) on the emulated ABI, hybrid layout: halos into strided row blocks, all-to-all, all-gathers, statistics all-reduce, the CFG pair's
    all-gather -- the result must be the single-process step, as with the plain adapter."""
    out = tmp_path / "r.pt"
    port = 29700 + (os.getpid() % 2000) + 91
    mp.spawn(_worker, args=(4, port, 16, str(out), True, "a2a", "halo", "staged"), nprocs=4, join=True)
    err = torch.load(out)["err"]
    assert err < 1e-4, err


def test_four_frame_shards_equal_single_process(tmp_path):
    """4 frame shards of 6 frames: the frame<->pixel all-to-all runs over 4 ranks, and the adapter's chunk-first / previous frames
    come from other ranks for every rank but the first (ranges start at frames 6, 12, 18 of chunks that start at 0, 8, 16)."""
    out = tmp_path / "r.pt"
    port = 29700 + (os.getpid() % 2000) + 55
    mp.spawn(_worker, args=(4, port, 24, str(out), False, "a2a", "halo"), nprocs=4, join=True)
    err = torch.load(out)["err"]
    assert err < 1e-4, err


def test_hybrid_cfg_x_frame_sharded_step_equals_single_process(tmp_path):
    """4 ranks = CFG pair x 2 frame shards (the 8-GPU layout at half size): every frame-shard exchange runs at batch 2 and the
    pair trades its noise predictions once; the result must be the single-process step."""
    out = tmp_path / "r.pt"
    port = 29700 + (os.getpid() % 2000) + 77
    mp.spawn(_worker, args=(4, port, 16, str(out), True), nprocs=4, join=True)
    err = torch.load(out)["err"]
    assert err < 1e-4, err


def test_prev_frame_halo_view_indexing_and_segment_tables():
    """attn1 under frame sharding reads [halo | local] K|V, not the all-gather: item map and the key-segment tables."""
    from types import SimpleNamespace

    import torch

    from motioneditor_amd import segments
    from motioneditor_amd.parallel import PrevFrameHalo

    B, f_loc = 2, 4
    for rank in (0, 1, 2):
        sh = SimpleNamespace(world=3, rank=rank, f_loc=f_loc, f_total=12, frame0=rank * f_loc, group=None, _ranks=[0, 1, 2])
        hv = PrevFrameHalo(sh)
        assert hv.item(B, 1, rank * f_loc + 2) == B + 1 * f_loc + 2
        if rank > 0:
            assert hv.item(B, 1, rank * f_loc - 1) == 1          # the halo item of batch row 1
        for bad in (rank * f_loc + f_loc, rank * f_loc - 2):
            if bad >= 0:
                try:
                    hv.item(B, 0, bad)
                    raise AssertionError("out-of-view frame accepted")
                except IndexError:
                    pass
        si, sm = segments.prev_cur(B, f_loc, torch.device("cpu"), hv)
        si = si.reshape(B, f_loc, 2)
        for b in range(B):
            for i in range(f_loc):
                cur = B + b * f_loc + i
                if rank == 0 and i == 0:      # global frame 0 would attend itself twice: collapsed to one segment
                    assert si[b, i].tolist() == [cur, -1]
                    continue
                prev = b if i == 0 else cur - 1
                assert si[b, i].tolist() == [prev, cur], (rank, b, i, si[b, i].tolist())
    # the table cache distinguishes the two layouts of one rank
    g = SimpleNamespace(world=3, rank=1, f_loc=f_loc, f_total=12, frame0=f_loc, layout="gather", item=lambda B_, b, gg: (gg // f_loc) * (B_ * f_loc) + b * f_loc + gg % f_loc)
    a, _ = segments.prev_cur(B, f_loc, torch.device("cpu"), g)
    h, _ = segments.prev_cur(B, f_loc, torch.device("cpu"), PrevFrameHalo(SimpleNamespace(world=3, rank=1, f_loc=f_loc, f_total=12, frame0=f_loc, group=None, _ranks=[0, 1, 2])))
    assert a.tolist() != h.tolist()


def _chunk_halo_worker(rank, world, port, f, chunk):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_ops
    from motioneditor_amd import parallel, segments
    B, npix, cols = 2, 3, 8
    sh = parallel.FrameShard(f, None).chunk_view(chunk)

    def rows_of(b, g):   # the value pattern of K|V rows (b, global frame g)
        return (torch.arange(npix * cols, dtype=torch.float32).reshape(npix, cols) + 1000 * g + 100000 * b).to(torch.float16)

    ext, loc = sh.kv_buffer(B * sh.f_loc * npix, cols, B, npix, torch.zeros(1, dtype=torch.float16))
    for b in range(B):
        for i in range(sh.f_loc):
            loc[(b * sh.f_loc + i) * npix:(b * sh.f_loc + i + 1) * npix] = rows_of(b, sh.frame0 + i)
    kv = sh.complete_kv(ext, B, npix, emu_ops.copy_rows)
    item, mode = segments.first_prev_chunked(B, sh.f_loc, chunk, torch.device("cpu"), sh)
    item = item.reshape(B * sh.f_loc, -1)
    for b in range(B):
        for i in range(sh.f_loc):
            g = sh.frame0 + i
            c0 = g - g % chunk
            want = [c0] if g % chunk <= 1 else [c0, g - 1]
            got = [int(v) for v in item[b * sh.f_loc + i] if v >= 0]
            assert len(got) == len(want), (rank, g, got, want)
            for it, gw in zip(got, want):
                assert torch.equal(kv[it * npix:(it + 1) * npix], rows_of(b, gw)), (rank, b, g, gw)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,f,chunk", [(8, 24, 8), (4, 24, 8), (8, 16, 8), (3, 24, 8)])
def test_chunk_halo_fetches_first_and_previous_frames_from_any_rank(world, f, chunk):
    """f_loc = 3 puts the chunk's first frame up to two ranks back and the previous frame on the neighbour: every key item the
    adapter's [first | previous] table names must hold that frame's rows after the point-to-point exchange."""
    port = 29700 + (os.getpid() % 2000) + 100 + world + f
    mp.spawn(_chunk_halo_worker, args=(world, port, f, chunk), nprocs=world, join=True)
