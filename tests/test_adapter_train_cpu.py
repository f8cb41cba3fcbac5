"""Adapter training step (train_adaptor.py:364-385) on the emulated ABI, two data-parallel ranks on gloo:
util.AdapterTrainer (tape gradients accumulated into one flat bucket -> one all-reduce -> clip_grad_norm -> AdamW on packed fp32 masters -> packed weights refreshed in place) against the same
step taken with torch autograd through the oracle on both clips in one process."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _clip(rank, f=8, h=8):
    g = torch.Generator().manual_seed(1000 + rank)
    r16 = lambda x: x.half().float()   # noqa: E731
    from motioneditor_amd import synth
    sizes = [h, h, h, h // 2, h // 2, h // 2, h // 4, h // 4, h // 4, h // 8, h // 8, h // 8]
    return dict(noisy=r16(torch.randn(1, 4, f, h, h, generator=g)), noise=r16(torch.randn(1, 4, f, h, h, generator=g)),
                ehs=r16(torch.randn(1, 77, 768, generator=g) * 0.3),
                down=[r16(torch.randn(1, c, f, sizes[i], sizes[i], generator=g) * 0.3) for i, c in enumerate(synth.ADAPTER_CH)],
                mid=r16(torch.randn(1, 1280, f, h // 8, h // 8, generator=g) * 0.3), t=501 - 100 * rank)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_ops
    import motioneditor_amd.models.unet_2d_condition as u
    from motioneditor_amd import synth, util
    from motioneditor_amd.models import graph
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    for m in (graph, u, util):
        m.ops = emu_ops
    sd_np = synth.synth_state_dict(synth.unet_schema())
    unet = UNet2DConditionModel(sd_np, device="cpu", dtype=torch.float32)
    tr = util.AdapterTrainer(unet, lr=1e-3)            # a visible step (the reference's 3e-5 moves fp32 weights by 1e-5)
    c = _clip(rank)
    loss = tr.step(c["noisy"], c["t"], c["ehs"], c["down"], c["mid"], c["noise"])
    if rank == 0:
        from oracle import ref_cpu
        sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        names = tr.names
        params = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
        opt = torch.optim.AdamW(list(params.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        tot, losses = None, []
        for r in range(world):
            cr = _clip(r)
            sd2 = dict(sd)
            sd2.update(params)
            l = torch.nn.functional.mse_loss(ref_cpu.unet_forward(sd2, cr["noisy"], cr["t"], cr["ehs"], cr["down"], cr["mid"]), cr["noise"])
            gr = torch.autograd.grad(l, [params[k] for k in names])
            losses.append(float(l))
            tot = [g / world for g in gr] if tot is None else [a + g / world for a, g in zip(tot, gr)]
        for k, g in zip(names, tot):
            params[k].grad = g
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        got = tr.export_state_dict()       # the trained parameters under the reference's names, in the reference's layouts
        assert sorted(got) == names and all(got[k].shape == sd[k].shape for k in names)
        worst = max(float((got[k] - params[k].detach()).abs().max() / (params[k].detach() - sd[k]).abs().max().clamp_min(1e-12)) for k in names)
        moved = max(float((params[k].detach() - sd[k]).abs().max()) for k in names)
        num = sum(float((got[k] - params[k].detach()).pow(2).sum()) for k in names)
        den = sum(float((params[k].detach() - sd[k]).pow(2).sum()) for k in names)
        torch.save({"worst": worst, "moved": moved, "update_rel_l2": (num / den) ** 0.5, "loss": loss, "want_loss": sum(losses) / world,
                    "repacked": float((unet.P.mat(names_w(names)) .float().reshape(-1)[:8] - params[names_w(names)].detach().reshape(-1)[:8]).abs().max())}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def names_w(names):
    return next(n for n in names if n.endswith("attn_pose.to_q.weight"))


def test_two_rank_adapter_training_step_equals_the_single_process_autograd_step(tmp_path):
    out = tmp_path / "r.pt"
    port = 29700 + (os.getpid() % 2000) + 321
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    r = torch.load(out)
    assert abs(r["loss"] - r["want_loss"]) < 1e-4 * r["want_loss"]
    assert r["moved"] > 5e-4                 # AdamW moved the parameters ...
    # ... to where the autograd step puts them.  Adam's first update is lr * g / (|g| + eps): elements whose gradient sits at eps level
    # depend on its last digits, so the whole update is compared in L2 and the single worst element only loosely
    assert r["update_rel_l2"] < 1e-2 and r["worst"] < 0.3, r
    assert r["repacked"] < 1e-6              # and the weight store hands out the updated values


def test_training_sequence_of_the_example_matches_the_oracle_loss(monkeypatch, unet_sd_np, cn_sd_np):
    """examples/train_adapter.py::step -- VAE encode, add_noise, ControlNet, UNet + adapter, loss, AdamW -- on the emulated ABI:
    the loss it reports is the oracle's for the same tensors, and the adapter's weights moved."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "examples"))
    import emu_ops
    import motioneditor_amd.models.controlnet as c
    import motioneditor_amd.models.unet_2d_condition as u
    import motioneditor_amd.models.vae as v
    import train_adapter as ex
    from motioneditor_amd import synth, util
    from motioneditor_amd.models import graph
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.models.vae import AutoencoderKL
    from oracle import ref_cpu
    for m in (graph, u, c, v, util):
        monkeypatch.setattr(m, "ops", emu_ops)
    vsd = synth.synth_state_dict(synth.vae_encoder_schema(), 33, salt="vae.")
    vae = AutoencoderKL(vsd, device="cpu", dtype=torch.float32)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    cn = ControlNetModel(cn_sd_np, device="cpu", dtype=torch.float32)
    tr = util.AdapterTrainer(unet, lr=1e-3)
    f, H = 8, 64
    b = ex.training_batch(f, H, H)
    t = 401
    before = tr.export_state_dict()["controlnet_adapter.body.0.block2.weight"].clone()
    loss = ex.step(tr, vae, cn, b, t)
    # oracle: the same sequence in torch
    T = torch.from_numpy
    usd = {k: T(x) for k, x in unet_sd_np.items()}
    csd = {k: T(x) for k, x in cn_sd_np.items()}
    lat = ref_cpu.vae_encode_sample({k: T(x) for k, x in vsd.items()}, b["pixel_values"].reshape(f, 3, H, H), b["encode_noise"])
    lat = lat.reshape(1, f, 4, 8, 8).permute(0, 2, 1, 3, 4) * 0.18215
    a = float(ex.alphas_cumprod()[t])
    noisy = a ** 0.5 * lat + (1 - a) ** 0.5 * b["noise"]
    down, mid = ref_cpu.controlnet_forward(csd, noisy.permute(0, 2, 1, 3, 4).reshape(f, 4, 8, 8), t, b["ehs"].repeat(f, 1, 1), b["skeleton"].reshape(f, 3, H, H))
    down = [d.reshape(1, f, *d.shape[1:]).permute(0, 2, 1, 3, 4) for d in down]
    mid = mid.reshape(1, f, *mid.shape[1:]).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        want = float(torch.nn.functional.mse_loss(ref_cpu.unet_forward(usd, noisy, t, b["ehs"], down, mid), b["noise"]))
    assert abs(loss - want) < 1e-3 * want, (loss, want)
    assert float((tr.export_state_dict()["controlnet_adapter.body.0.block2.weight"] - before).abs().max()) > 1e-4
