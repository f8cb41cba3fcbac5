"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own model code.

CONTAINER-ONLY (needs /root/reference, which never travels to the GPU box).  Test infrastructure,
not product code.  What it does:

  1. imports ``motion_editor.models`` / ``motion_editor.attn_control`` from /root/reference through
     the stub packages in oracle/shim (no diffusers / xformers in this image);
  2. builds the reference ``UNet2DConditionModel`` (SD-1.5 widths) with the deterministic synthetic
     weights of ``motioneditor_amd.synth`` (zero-init tensors re-randomised);
  3. runs single-branch and two-branch forwards (both attention editors registered, inactive and
     active steps) on seeded inputs, and checks that ``oracle/ref_cpu.py`` reproduces every output
     to fp32 round-off -- this is what PINS the restatement;
  4. stores the reference outputs (+ per-stage checksums for bisecting) as small .npz fixtures,
     the state-dict key schema as unet_keys.txt, and DDIM vectors from the reference's in-tree
     ``prev_step`` (p2p/null_text_optimization.py:26-36).

  5. (--only-inversion / full run) DDIM inversion: the reference UNet with ``normal_infer=True`` and the reference's
     in-tree ``next_step`` (util.py:77-87) walked over three inversion steps -> inversion.npz.

Usage:  python oracle/make_golden.py [--skip-two-branch | --only-inversion | --only-null-text | --only-adapter-train | --only-config3 | --only-geom96 | --only-single | --only-single96 | --only-two64 | --only-two96 | --only-two64-f24 | --only-controlnet]
"""
from __future__ import annotations

import ast
import contextlib
import io
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "oracle" / "shim"))
sys.path.insert(1, str(REF))
sys.path.insert(2, str(ROOT))

from motioneditor_amd import synth  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from motioneditor_amd.synth import make_case_inputs  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def stats(t: torch.Tensor):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).sum().sqrt().item()], dtype=np.float64)


def relerr(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a - b).abs().max() / b.abs().mean().clamp_min(1e-12)).item()


def build_reference_unet(sd_np):
    from motion_editor.models.unet_2d_condition import UNet2DConditionModel
    m = quiet(UNet2DConditionModel, sample_size=64, cross_attention_dim=768, attention_head_dim=8, use_sc_attn=True, use_st_attn=False)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    assert not missing and not unexpected
    return m.eval()


def reference_ddim_vectors():
    """Execute the reference's own prev_step (p2p/null_text_optimization.py:26-36) without importing the
    module (it loads a CLIP tokenizer at import time)."""
    src = (REF / "motion_editor/p2p/null_text_optimization.py").read_text()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "prev_step":
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"Union": __import__("typing").Union, "torch": torch, "np": np}
    exec(compile(mod, "ref_prev_step", "exec"), ns)
    d = ref_cpu.DDIM()

    class Sched:
        class config:
            num_train_timesteps = 1000
        num_inference_steps = 50
        alphas_cumprod = d.alphas_cumprod
        final_alpha_cumprod = d.alphas_cumprod[0]

    class Self:
        scheduler = Sched

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 8, 8, generator=g)
    e = torch.randn(2, 4, 8, 8, generator=g)
    out = {}
    for t in (981, 501, 21, 1):
        ref = ns["prev_step"](Self, e, t, x)
        mine = d.step(e, t, x)
        assert relerr(mine, ref) < 1e-6, (t, relerr(mine, ref))
        ca, cb = d.coeffs(t)
        assert relerr(ca * x + cb * e, ref) < 1e-5
        out[f"prev_{t}"] = ref.numpy()
    out["x"], out["eps"] = x.numpy(), e.numpy()
    out["timesteps"] = np.array(d.timesteps, dtype=np.int64)
    assert d.timesteps[0] == 981 and d.timesteps[-1] == 1 and len(d.timesteps) == 50
    np.savez_compressed(GOLD / "ddim.npz", **out)
    print("ddim.npz written; oracle DDIM == reference prev_step")


def _ref_function(path: Path, name: str):
    """Compile ONE function of a reference file without importing the module (its imports need packages this image lacks)."""
    tree = ast.parse(path.read_text())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"Union": __import__("typing").Union, "torch": torch, "np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), f"ref_{name}", "exec"), ns)
    return ns[name]


def prepare_image_golden():
    """MotionEditorPipeline.prepare_image (pipeline_motion_editor.py:418-459) executed as written -- PIL and tensor inputs, one
    image broadcast over the batch, classifier-free-guidance doubling -- on seeded uint8 images; the images travel with the
    expected tensors (tests/golden/prepare_image.npz)."""
    import PIL.Image
    lanczos = PIL.Image.Resampling.LANCZOS if hasattr(PIL.Image, "Resampling") else PIL.Image.LANCZOS
    tree = ast.parse((REF / "motion_editor/pipelines/pipeline_motion_editor.py").read_text())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "prepare_image")
    ns = {"torch": torch, "np": np, "PIL": PIL, "PIL_INTERPOLATION": {"lanczos": lanczos}}     # diffusers.utils.PIL_INTERPOLATION["lanczos"]
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "ref_prepare_image", "exec"), ns)
    prepare_image = ns["prepare_image"]
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (40, 56), dtype=np.uint8)            # a grey ("L") image: .convert("RGB") replicates it
    ims = [PIL.Image.fromarray(a), PIL.Image.fromarray(b)]
    out = {"img_a": a, "img_b": b}
    out["pil_list_cfg"] = prepare_image(None, ims, 32, 24, 2, 1, "cpu", torch.float32, True).numpy()
    out["pil_one_b3"] = prepare_image(None, ims[0], 32, 24, 3, 1, "cpu", torch.float32, False).numpy()
    t = torch.from_numpy(rng.random((2, 3, 24, 32), dtype=np.float32))
    out["tensor_in"] = t.numpy()
    out["tensor_cfg"] = prepare_image(None, t, 32, 24, 2, 1, "cpu", torch.float32, True).numpy()
    out["tensor_list"] = prepare_image(None, [t[:1], t[1:]], 32, 24, 2, 1, "cpu", torch.float32, False).numpy()
    np.savez_compressed(GOLD / "prepare_image.npz", **out)
    print("prepare_image.npz written:", {k: v.shape for k, v in out.items()})


def null_text_golden(unet, sd):
    """MyNullInversion.null_optimization (p2p/null_text_optimization.py:133-166) executed AS WRITTEN -- the class body is compiled
    from the reference file with only the methods the optimisation touches, NUM_DDIM_STEPS set to 2 (the file's global is 50) --
    around the reference UNet with the synthetic weights: the optimised unconditional embeddings of two DDIM steps x two inner
    Adam steps, plus the first inner step's gradient (computed here with the reference UNet and the reference prev_step).
    The oracle's restatement is checked against both before the fixture is written."""
    from types import SimpleNamespace
    src = (REF / "motion_editor/p2p/null_text_optimization.py").read_text()
    cls = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ClassDef) and n.name == "MyNullInversion")
    keep = {"prev_step", "get_noise_pred_single", "get_noise_pred", "null_optimization", "scheduler"}
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in keep]

    class _Bar:
        def __init__(self, *a, **k): pass
        def update(self, *a): pass
        def close(self): pass

    ns = {"Union": __import__("typing").Union, "torch": torch, "np": np, "nnf": torch.nn.functional, "Adam": torch.optim.Adam,
          "tqdm": _Bar, "NUM_DDIM_STEPS": 2}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "ref_null_inversion", "exec"), ns)
    d = ref_cpu.DDIM()

    class Sched:
        class config:
            num_train_timesteps = 1000
        num_inference_steps = 50
        alphas_cumprod = d.alphas_cumprod
        final_alpha_cumprod = d.alphas_cumprod[0]
        timesteps = torch.tensor(d.timesteps)

    g = torch.Generator().manual_seed(123)
    f, h, w = 4, 8, 8
    latents = [torch.randn(1, 4, f, h, w, generator=g) for _ in range(3)]          # stands for the inversion trajectory x_0 .. x_2
    context = torch.randn(2, 77, 768, generator=g) * 0.3                              # [uncond, cond]
    obj = ns["MyNullInversion"].__new__(ns["MyNullInversion"])
    obj.model = SimpleNamespace(unet=lambda x, t, encoder_hidden_states=None, normal_infer=False: {"sample": quiet(unet, x, t, encoder_hidden_states, normal_infer=normal_infer).sample},
                                scheduler=Sched)
    obj.context = context
    t0 = time.time()
    ref_list = obj.null_optimization(latents, 2, 1e-5)
    print(f"reference null_optimization (2 steps x 2 inner) {time.time()-t0:.1f}s")
    # first inner step's gradient, with the reference pieces
    unc = context[:1].clone().requires_grad_(True)
    t = Sched.timesteps[0]
    with torch.no_grad():
        eps_c = obj.get_noise_pred_single(latents[-1], t, context[1:])
    eps_u = obj.get_noise_pred_single(latents[-1], t, unc)
    loss = torch.nn.functional.mse_loss(obj.prev_step(eps_u + 7.5 * (eps_c - eps_u), t, latents[-1]), latents[1])
    loss.backward()
    grads = []
    mine = ref_cpu.null_optimization(sd, d, latents, context, 2, 1e-5, num_steps=2, grads=grads)
    eg = relerr(grads[0], unc.grad)
    # Adam's first update is lr * sign(g): elements whose gradient is at rounding level may flip; compare where |g| is not tiny
    big = unc.grad.abs() > 1e-3 * unc.grad.abs().max()
    ee = [float(((m - r).abs() * big).max()) for m, r in zip(mine, ref_list)]
    print("null-text oracle vs reference: grad rel err", eg, " max |emb diff| on significant elements", ee, " first loss", float(loss))
    assert eg < 1e-3 and max(ee) < 2e-3, (eg, ee)
    np.savez_compressed(GOLD / "null_text.npz", latents=torch.stack(latents).numpy(), context=context.numpy(),
                        uncond_out=torch.stack(ref_list).numpy(), grad0=unc.grad.numpy(), loss0=float(loss), oracle_grad_relerr=eg)
    print("null_text.npz written")


def adapter_train_golden(unet, sd):
    """The arithmetic of one adapter training step (train_adaptor.py:364-368): model_pred = unet(noisy_latents, t, ehs,
    down_block_additional_residuals, mid_block_additional_residual).sample on ONE clip (batch 1: the adapter sees every row,
    unet_2d_condition.py:482-485), loss = mse(model_pred, noise), backward.  The script itself is one accelerate loop and
    cannot be executed piecewise, so the fixture holds what its lines 364-368 compute with the REFERENCE UNet: the loss and
    the gradient of every controlnet_adapter parameter (norms of all, two tensors in full).  Checked against the oracle
    (ref_cpu.unet_forward under autograd) before it is written."""
    B, f, h, w = 1, 8, 8, 8
    g = torch.Generator().manual_seed(77)
    r16 = lambda x: x.half().float()   # fp16-representable inputs: the fixture stores them as fp16 without loss  # noqa: E731
    noisy = r16(torch.randn(B, 4, f, h, w, generator=g))
    noise = r16(torch.randn(B, 4, f, h, w, generator=g))
    ehs = r16(torch.randn(B, 77, 768, generator=g) * 0.3)
    sizes = [h, h, h, h // 2, h // 2, h // 2, h // 4, h // 4, h // 4, h // 8, h // 8, h // 8]
    down = [r16(torch.randn(B, c, f, sizes[i], sizes[i], generator=g) * 0.3) for i, c in enumerate(synth.ADAPTER_CH)]
    mid = r16(torch.randn(B, 1280, f, h // 8, w // 8, generator=g) * 0.3)
    t = 501
    names = [k for k in sd if k.startswith("controlnet_adapter.")]
    unet.zero_grad(set_to_none=True)
    for p_ in unet.parameters():
        p_.requires_grad_(False)
    ref_params = dict(unet.named_parameters())
    for k in names:
        ref_params[k].requires_grad_(True)
    t0 = time.time()
    pred = quiet(unet, noisy, torch.tensor(t), ehs, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    loss = torch.nn.functional.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    print(f"reference adapter training forward+backward {time.time()-t0:.1f}s, loss {float(loss):.6f}")
    ref_g = {k: ref_params[k].grad.detach().clone() for k in names}
    # oracle
    sd2 = dict(sd)
    for k in names:
        sd2[k] = sd[k].clone().requires_grad_(True)
    pred2 = ref_cpu.unet_forward(sd2, noisy, t, ehs, down, mid)
    loss2 = torch.nn.functional.mse_loss(pred2, noise)
    gr = torch.autograd.grad(loss2, [sd2[k] for k in names])
    tot = (sum(float((a - ref_g[k]).pow(2).sum()) for a, k in zip(gr, names)) / sum(float(ref_g[k].pow(2).sum()) for k in names)) ** 0.5
    print("adapter-training oracle vs reference: loss", float(loss2), "vs", float(loss), " gradient rel err (all adapter parameters)", tot)
    assert abs(float(loss2) - float(loss)) < 1e-4 * float(loss) and tot < 1e-3, (float(loss2), float(loss), tot)
    full = ["controlnet_adapter.body.0.block2.weight", "controlnet_adapter.body.11.attn_self_temp.to_out.0.bias"]
    full = [k for k in full if k in ref_g] or names[:2]
    h16 = lambda x: x.numpy().astype(np.float16)  # noqa: E731
    np.savez_compressed(GOLD / "adapter_train.npz", noisy=h16(noisy), noise=h16(noise), ehs=h16(ehs), mid=h16(mid), t=t,
                        **{f"down{i}": h16(d_) for i, d_ in enumerate(down)}, loss=float(loss), names=np.array(names),
                        grad_norms=np.array([float(ref_g[k].norm()) for k in names], dtype=np.float64),
                        **{"full_" + str(i): ref_g[k].numpy() for i, k in enumerate(full)}, full_names=np.array(full), oracle_grad_relerr=tot)
    for p_ in unet.parameters():
        p_.requires_grad_(True)
    print("adapter_train.npz written:", len(names), "adapter parameters")


def inversion_goldens(unet, sd):
    """DDIM inversion (util.py:111-124 as inference.py:289-293 calls it): normal_infer UNet forward + next_step."""
    next_step = _ref_function(REF / "motion_editor/util.py", "next_step")
    d = ref_cpu.DDIM()

    class Sched:
        class config:
            num_train_timesteps = 1000
        num_inference_steps = 50
        alphas_cumprod = d.alphas_cumprod
        final_alpha_cumprod = d.alphas_cumprod[0]
        timesteps = d.timesteps

    ci = make_case_inputs("inversion", B=1, f=8, h=16, w=16)
    cond = ci["ehs"]
    out = {}
    with torch.no_grad():
        ref = quiet(unet, ci["sample"], torch.tensor(1), cond, normal_infer=True).sample
        mine = ref_cpu.unet_forward(sd, ci["sample"], 1, cond, normal_infer=True)
        e = relerr(mine, ref)
        print("normal_infer oracle vs reference rel err", e)
        assert e < 2e-4, e
        out["unet_normal_infer_t1"] = ref.numpy().astype(np.float32)
        # next_step vectors
        g = torch.Generator().manual_seed(7)
        x, eps = torch.randn(1, 4, 8, 8, 8, generator=g), torch.randn(1, 4, 8, 8, 8, generator=g)
        for t in (1, 21, 501, 981):
            r = next_step(eps, t, x, Sched)
            assert relerr(d.next_step(eps, t, x), r) < 1e-6
            ca, cb = d.next_coeffs(t)
            assert relerr(ca * x + cb * eps, r) < 1e-5
            out[f"next_{t}"] = r.numpy()
        out["ns_x"], out["ns_eps"] = x.numpy(), eps.numpy()
        # three inversion steps with the reference's own pieces (ddim_loop body, util.py:118-123)
        lat = ci["sample"]
        for i in range(3):
            t = Sched.timesteps[len(Sched.timesteps) - i - 1]
            noise = quiet(unet, lat, torch.tensor(t), cond, normal_infer=True).sample
            lat = next_step(noise, t, lat, Sched)
            out[f"loop_latent_{i + 1}"] = lat.numpy().astype(np.float32)
        mine = ref_cpu.ddim_loop(sd, d, ci["sample"], 3, cond, normal_infer=True)
        e2 = relerr(mine[-1], lat)
        print("3-step inversion oracle vs reference rel err", e2)
        assert e2 < 2e-4, e2
    out["oracle_relerr"] = np.array([e, e2])
    np.savez_compressed(GOLD / "inversion.npz", **out)
    print("inversion.npz written")


def _sub(t: torch.Tensor, sf: int, sp: int) -> np.ndarray:
    """Strided sub-sample of a [B, C, f, h, w] tensor (every 8th channel, every sf-th frame, every sp-th pixel row / column), fp16."""
    return t[:, ::8, ::sf, ::sp, ::sp].numpy().astype(np.float16)   # (every 8th channel)


def step_golden(tag: str, f: int, h: int, step: int = 4, single_branch: bool = False):
    """One full denoising step through the ORACLE at a size the reference's own modules cannot hold in the container (their materialised
    5N-key scores alone are 64 GB at config 3), on bench.py's inputs and weights.  The oracle is pinned against the reference at 16x16 /
    32x32 (cases above); these fixtures pin the HIP path to the oracle at the benchmarked geometries:
      step_config3   BASELINE configs[2]: 24 frames x 64x64 latents, two-branch + ControlNet + adapter, both editors ACTIVE (step 4)
      step_geom96    the spatial geometry of BASELINE configs[4] (96x96 latents: 9216 tokens at level 0, 48x48 / 24x24 / 12x12 below) at 8 frames
      step_single    BASELINE configs[1]: 8 frames x 64x64 latents, ONE clip, single-branch UNet3D (no ControlNet, no adapter input, no editors);
                     round 5: the REFERENCE UNet itself runs this one (chunked xformers stand-in) -- oracle == reference asserted, the fixture's noise
                     prediction / latents come from the reference's eps
    Stored: strided sub-samples of the updated latents and of the guided noise prediction, rel-L2-comparable sub-samples of two skips and one
    motion residual, and per-stage statistics (12 skips, 12 motion residuals, mid, 12 + 1 ControlNet residuals).  Needs no reference import."""
    torch.set_num_threads(os.cpu_count() or 8)
    x = synth.bench_inputs(f, h, h)
    usd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.unet_schema()).items()}
    ddim = ref_cpu.DDIM()
    t = ddim.timesteps[step]
    taps = {}
    t0 = time.time()
    if single_branch:
        with torch.no_grad():
            want = ref_cpu.denoise_step(usd, None, ddim, x["latents"][:1], t, x["uncond"][step], x["cond"][:1], None, None, None, 7.5, taps=taps)
    else:
        csd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.").items()}
        sp, tp = ref_cpu.SpatialEditor(x["masks"]), ref_cpu.TemporalEditor()
        sp.cur_step = tp.cur_step = step
        images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * h, 8 * h)
        with torch.no_grad():
            want = ref_cpu.denoise_step(usd, csd, ddim, x["latents"], t, x["uncond"][step], x["cond"], images, sp, tp, 7.5, taps=taps)
    dt = time.time() - t0
    print(f"oracle step {tag} ({f} f x {h}x{h}, {'single-branch' if single_branch else 'two-branch'}): {dt:.0f} s on {torch.get_num_threads()} threads")
    assert torch.isfinite(want).all()
    ref_err = None
    if single_branch and "--no-reference" not in sys.argv:
        # The single-branch step fits the reference's OWN UNet in the container once the test-only xformers stand-in chunks its score matrix
        # (oracle/shim/xformers/ops.py, exact): run models/unet_2d_condition.py:363-546 on the same cat([latents] * 2) / timestep / [uncond, cond]
        # at the production token count (N = 4096, [prev | cur] keys 8192), assert the oracle's eps equals it, and write the fixture's noise
        # prediction and updated latents FROM THE REFERENCE's eps (CFG :643-645; DDIM through the oracle's step, itself pinned by the reference's
        # in-tree prev_step vectors) -- step_single.npz is then reference-generated where it matters.
        unet = build_reference_unet(synth.synth_state_dict(synth.unet_schema()))
        if 2 * f * 8 * (h * h) * (2 * h * h) * 4 > 20e9:
            # the [prev | cur] score tensor of level 0 in one piece (87 GB at 96 x 96 latents) does not fit the container: run the reference the way inference.py does --
            # `unet.enable_xformers_memory_efficient_attention()` (inference.py:187), i.e. BasicTransformerBlock.set_use_memory_efficient_attention_xformers
            # (attention_2d.py:465-491) -- by setting the two flags that method sets (the method itself insists on CUDA); xformers.ops is the exact, chunked stand-in
            n_on = 0
            for m in unet.modules():
                if hasattr(m, "set_use_memory_efficient_attention_xformers") and hasattr(m, "attn1"):
                    m.attn1._use_memory_efficient_attention_xformers = True
                    if getattr(m, "attn2", None) is not None:
                        m.attn2._use_memory_efficient_attention_xformers = True
                    n_on += 1
            print(f"reference UNet: xformers path enabled on {n_on} transformer blocks (attention_2d.py:488-490)")
        xin = torch.cat([x["latents"][:1]] * 2)
        emb = torch.cat([x["uncond"][step].expand(1, 77, 768), x["cond"][:1]])
        t1 = time.time()
        with torch.no_grad():
            eps_ref = quiet(unet, xin, torch.tensor(t), emb).sample
        del unet
        eu, ec = eps_ref.chunk(2)
        np_ref = eu + 7.5 * (ec - eu)
        ref_err = relerr(taps["noise_pred"], np_ref)
        print(f"reference UNet at {f} f x {h}x{h} (B = 2): {time.time() - t1:.0f} s; oracle guided noise prediction vs reference: max-abs / mean-abs {ref_err:.2e}")
        assert ref_err < 2e-4, ref_err
        taps["noise_pred"] = np_ref
        want = ddim.step(np_ref, t, x["latents"][:1])
    sp_lat, sp_np = (2, 4) if h >= 64 else (1, 2)
    sf = 2 if f > 8 else 1
    out = dict(frames=f, latent=h, step=step, t=t, oracle_seconds=dt, single_branch=int(single_branch),
               latents_sub=want[:, :, :, ::sp_lat, ::sp_lat].numpy().astype(np.float32), latents_stats=stats(want), lat_stride=sp_lat,
               noise_pred_sub=taps["noise_pred"][:, :, ::sf, ::sp_np, ::sp_np].numpy().astype(np.float32), noise_pred_stats=stats(taps["noise_pred"]),
               np_stride=np.array([sf, sp_np]),
               skip_stats=np.stack([stats(s) for s in taps["skips"]]), mid_stats=stats(taps["mid"]),
               # level-0 skip after the first transformer block, and a level-2 skip: rel-L2 on a strided sub-sample catches a mis-scaled block
               skip1_sub=_sub(taps["skips"][1], sf, 4), skip7_sub=_sub(taps["skips"][7], sf, 2))
    if ref_err is not None:
        out.update(reference_generated=1, oracle_vs_reference=ref_err)
    if not single_branch:
        cn = taps["cn_down"]
        # the two ControlNet batch entries of the reference are the same computation (even frame count): the product computes one
        assert all(float((d[0] - d[1]).abs().max()) == 0.0 for d in cn)
        out.update(motion_stats=np.stack([stats(s) for s in taps["motion"]]), cn_down_stats=np.stack([stats(d) for d in cn]), cn_mid_stats=stats(taps["cn_mid"]),
                   motion4_sub=_sub(taps["motion"][4][[1, 3]], sf, 2), cn_down6_sub=_sub(cn[6][:1], sf, 2))
    np.savez_compressed(GOLD / f"{tag}.npz", **out)
    print(f"{tag}.npz written")


def two_branch_64_golden(unet, sd, Spatial, reg_spatial, Temporal, reg_temporal, h=64, f=8):
    """The two-branch UNet forward with BOTH reference editors ACTIVE at a production token count: batch 4 = [u.rec, u.edit, c.rec, c.edit], 8 frames x 64x64
    latents (N = 4096 queries; the edit rows attend 5 N = 20480 materialised keys, fully_control.py:381-413), adapter fed with ControlNet-shaped residuals.
    The reference's own code runs it in the container because every self-attention of a model with registered editors goes through
    xformers.ops.memory_efficient_attention (control_utils.py / fully_control_utils.py AttentionBase.forward, fully_control.py:418) -- here the exact,
    chunked stand-in of oracle/shim.  (attn_batch hard-codes num_frames = 8, fully_control.py:377: the shape has 8 frames.)  Asserts oracle == reference and
    writes tests/golden/unet_two_active_64.npz from the REFERENCE output (every other latent row / column)."""
    cb = make_case_inputs("two", B=4, f=f, h=h, w=h)      # (f = 24, h = 64: the UNet shape of the benchmark itself, BASELINE configs[2]; h = 96: the level-0 geometry of BASELINE configs[4], 9216 queries / 46080 materialised keys)

    class Holder:
        pass

    holder = Holder()
    holder.unet = unet
    ted = quiet(Temporal, start_step=4, start_layer=10)
    quiet(reg_temporal, holder, ted)
    sed = quiet(Spatial, start_step=4, start_layer=10, source_masks=cb["source_masks"])
    quiet(reg_spatial, holder, sed)
    step = 4
    ted.reset(); sed.reset()
    ted.cur_step = sed.cur_step = step
    my_sp, my_tp = ref_cpu.SpatialEditor(cb["source_masks"]), ref_cpu.TemporalEditor()
    my_sp.cur_step = my_tp.cur_step = step
    with torch.no_grad():
        t0 = time.time()
        ref = quiet(unet, cb["sample"], torch.tensor(cb["t"]), cb["ehs"], down_block_additional_residuals=cb["down_res"],
                    mid_block_additional_residual=cb["mid_res"]).sample
        print(f"reference two-branch forward, editors active, {f} f x {h}x{h} (B = 4): {time.time() - t0:.0f} s")
        t0 = time.time()
        taps = {}
        mine = ref_cpu.unet_forward(sd, cb["sample"], cb["t"], cb["ehs"], cb["down_res"], cb["mid_res"], my_sp, my_tp, taps=taps)
        print(f"oracle: {time.time() - t0:.0f} s")
    e = relerr(mine, ref)
    print(f"two-branch (active) {h}x{h} oracle vs reference rel err", e)
    assert e < 2e-4, e
    assert (ted.cur_step, ted.cur_att_layer, sed.cur_step, sed.cur_att_layer) == (step + 1, 0, step + 1, 0)
    tag = f"unet_two_active_{h}" + ("" if f == 8 else f"_f{f}")
    np.savez_compressed(GOLD / f"{tag}.npz", out_sub=ref[:, :, ::(1 if f == 8 else 2), ::2, ::2].numpy().astype(np.float32), out_stats=stats(ref), step=step, latent=h, frames=f,
                        skip_stats=np.stack([stats(s_) for s_ in taps["skips"]]), motion_stats=np.stack([stats(s_) for s_ in taps["motion"]]),
                        mid_stats=stats(taps["mid"]), oracle_relerr=e)
    print(f"{tag}.npz written")


def controlnet_trunk_golden():
    """R16 (diffusers ControlNetModel, source not in the reference tree): pin the TRUNK against the reference's own blocks.
    conv_in, the time embedding, down_blocks.* and mid_block.* of the ControlNet share the SD-1.5 key schema with the reference's
    UNet2DConditionModel, and that model run at f = 1 with temp_conv* and attn_temp.to_out zeroed IS the 2-D SD-1.5 encoder:
    GroupNorm over one frame (resnet_2d.py:199-249), TemporalConv zero -> identity (resnet_2d.py:15-16, 205-206), [frame 0 | frame 0]
    keys = plain self-attention (attention_2d.py:732-740), a one-frame temporal attention times a zero out-projection = 0
    (attention_2d.py:534-545).  So: load the reference UNet with the ControlNet's trunk weights, add the oracle's conditioning embedding to
    conv_in's output with a forward hook, capture the 12 down-block residuals + the mid-block output with hooks, and compare with the
    oracle's tensors in front of the zero-convolutions.  The fixture holds the zero-convolved REFERENCE tensors (a 1x1 convolution applied
    here to the reference's outputs).  What stays self-pinned: the 8 conditioning-embedding convolutions and the 13 1x1 zero-convolutions."""
    csd_np = synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")
    usd_np = dict(synth.synth_state_dict(synth.unet_schema()))
    trunk = [k for k in csd_np if k in usd_np]
    assert {k.split(".")[0] for k in trunk} == {"conv_in", "time_embedding", "down_blocks", "mid_block"}, {k.split(".")[0] for k in trunk}
    assert not [k for k in csd_np if k not in usd_np and not k.startswith(("controlnet_cond_embedding.", "controlnet_down_blocks.", "controlnet_mid_block."))]
    for k in trunk:
        assert usd_np[k].shape == csd_np[k].shape, k
        usd_np[k] = csd_np[k]
    nz = 0
    for k in usd_np:
        if "temp_conv" in k or "attn_temp.to_out" in k:
            usd_np[k] = np.zeros_like(usd_np[k])
            nz += 1
    unet = build_reference_unet(usd_np)
    csd = {k: torch.from_numpy(v) for k, v in csd_np.items()}
    n, h, t = 2, 16, 501
    T = torch.from_numpy
    sample = T(synth.synth_normal("cn_trunk.sample", (n, 4, h, h), 33))
    ehs = T(synth.synth_normal("cn_trunk.ehs", (n, 77, 768), 33, 0.3))
    cond = T(np.clip(synth.synth_normal("cn_trunk.cond", (n, 3, 8 * h, 8 * h), 33, 0.5) + 0.5, 0, 1).astype(np.float32))
    taps = {}
    with torch.no_grad():
        down_o, mid_o = ref_cpu.controlnet_forward(csd, sample, t, ehs, cond, taps=taps)
    cemb = taps["cond_emb"]
    got = {"res": [], "mid": None}

    def conv_in_hook(_m, _i, o):
        o = o + cemb[:, :, None]
        got["res"].append(o)
        return o

    hooks = [unet.conv_in.register_forward_hook(conv_in_hook)]
    for blk in unet.down_blocks:
        hooks.append(blk.register_forward_hook(lambda _m, _i, o: got["res"].extend(o[1])))
    hooks.append(unet.mid_block.register_forward_hook(lambda _m, _i, o: got.__setitem__("mid", o)))
    with torch.no_grad():
        quiet(unet, sample[:, :, None], torch.tensor(t), ehs)
    for hk in hooks:
        hk.remove()
    assert len(got["res"]) == 12 and got["mid"] is not None
    errs = [relerr(a, b[:, :, 0]) for a, b in zip(taps["outs"], got["res"])] + [relerr(taps["mid"], got["mid"][:, :, 0])]
    print(f"ControlNet trunk: oracle vs the reference's 2-D blocks ({len(trunk)} shared tensors, {nz} temporal tensors zeroed): max rel err {max(errs):.2e}")
    assert max(errs) < 1e-4, errs
    zc = lambda name, x: torch.nn.functional.conv2d(x, csd[name + ".weight"], csd[name + ".bias"])   # noqa: E731
    with torch.no_grad():
        down_r = [zc(f"controlnet_down_blocks.{i}", r[:, :, 0]) for i, r in enumerate(got["res"])]
        mid_r = zc("controlnet_mid_block", got["mid"][:, :, 0])
    e2 = max(relerr(a, b) for a, b in zip(down_o + [mid_o], down_r + [mid_r]))
    assert e2 < 1e-4, e2
    out = {f"down{i}": (d[:, :, ::2, ::2] if d.shape[-1] >= 8 else d).numpy().astype(np.float32) for i, d in enumerate(down_r)}
    np.savez_compressed(GOLD / "controlnet_trunk.npz", n=n, latent=h, t=t, mid=mid_r.numpy().astype(np.float32), oracle_relerr=max(max(errs), e2),
                        trunk_tensors=len(trunk), **out)
    print("controlnet_trunk.npz written; oracle ControlNet trunk == reference SD-1.5 2-D encoder blocks")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    GOLD.mkdir(parents=True, exist_ok=True)
    if "--only-config3" in sys.argv:
        step_golden("step_config3", 24, 64)
        return
    if "--only-geom96" in sys.argv:
        step_golden("step_geom96", 8, 96)
        return
    if "--only-single96" in sys.argv:   # the 96 x 96-latent geometry of BASELINE configs[4] (9216 tokens, 18432 [prev | cur] keys) through the reference UNet itself
        step_golden("step_single96", 8, 96, single_branch=True)
        return
    if "--only-single" in sys.argv:
        step_golden("step_single", 8, 64, single_branch=True)
        return
    if "--only-controlnet" in sys.argv:
        controlnet_trunk_golden()
        return
    if "--only-prepare-image" in sys.argv:
        prepare_image_golden()
        return
    reference_ddim_vectors()
    prepare_image_golden()

    t0 = time.time()
    schema = synth.unet_schema()
    sd_np = synth.synth_state_dict(schema)
    print(f"weights: {sum(v.size for v in sd_np.values())/1e6:.1f} M params in {time.time()-t0:.1f}s")
    unet = build_reference_unet(sd_np)
    ref_sd = unet.state_dict()
    with open(GOLD / "unet_keys.txt", "w") as fh:
        for k, v in ref_sd.items():
            fh.write(f"{k} {' '.join(str(int(s)) for s in v.shape)}\n")
    assert {k: tuple(v.shape) for k, v in ref_sd.items()} == dict(schema), "schema mismatch vs reference"
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}

    if "--only-inversion" in sys.argv:
        inversion_goldens(unet, sd)
        return
    if "--only-null-text" in sys.argv:
        null_text_golden(unet, sd)
        return
    if "--only-adapter-train" in sys.argv:
        adapter_train_golden(unet, sd)
        return

    from motion_editor.attn_control.fully_control import FullySelfAttentionControlMask
    from motion_editor.attn_control.fully_control_utils import regiter_fully_attention_editor_diffusers
    from motion_editor.attn_control.temporal_control import TemporalSelfAttentionControl
    from motion_editor.attn_control.temporal_control_utils import regiter_temporal_attention_editor_diffusers

    if "--only-two64" in sys.argv or "--only-two96" in sys.argv or "--only-two64-f24" in sys.argv:
        two_branch_64_golden(unet, sd, FullySelfAttentionControlMask, regiter_fully_attention_editor_diffusers, TemporalSelfAttentionControl,
                             regiter_temporal_attention_editor_diffusers, h=96 if "--only-two96" in sys.argv else 64, f=24 if "--only-two64-f24" in sys.argv else 8)
        return

    # ---- case A: single branch, no editors (config-2 shape family, small) ----
    ca = make_case_inputs("single", B=2, f=8, h=16, w=16)
    with torch.no_grad():
        t0 = time.time()
        ref = quiet(unet, ca["sample"], torch.tensor(ca["t"]), ca["ehs"]).sample
        print(f"reference single-branch forward {time.time()-t0:.1f}s")
        taps = {}
        mine = ref_cpu.unet_forward(sd, ca["sample"], ca["t"], ca["ehs"], taps=taps)
    e = relerr(mine, ref)
    print("single-branch oracle vs reference rel err", e)
    assert e < 2e-4, e
    np.savez_compressed(GOLD / "unet_single.npz", out=ref.numpy().astype(np.float32), out_stats=stats(ref),
                        skip_stats=np.stack([stats(s) for s in taps["skips"]]), mid_stats=stats(taps["mid"]), oracle_relerr=e)

    # ---- case A32: the same at 32x32 latents (level 0 = 1024 tokens, level 3 = 4x4): a golden whose attention launches have
    # several query blocks and 16+ key tiles per segment ----
    if "--skip-32" not in sys.argv:
        c32 = make_case_inputs("single32", B=2, f=8, h=32, w=32)
        with torch.no_grad():
            t0 = time.time()
            ref = quiet(unet, c32["sample"], torch.tensor(c32["t"]), c32["ehs"]).sample
            print(f"reference single-branch 32x32 forward {time.time()-t0:.1f}s")
            mine = ref_cpu.unet_forward(sd, c32["sample"], c32["t"], c32["ehs"])
        e = relerr(mine, ref)
        print("single-branch 32x32 oracle vs reference rel err", e)
        assert e < 2e-4, e
        np.savez_compressed(GOLD / "unet_single_32.npz", out=ref.numpy().astype(np.float16), out_stats=stats(ref), oracle_relerr=e)
    if "--only-32" in sys.argv:
        return

    if "--skip-two-branch" in sys.argv:
        return

    # ---- case B: two branch (B=4), f=16 (exposes the adapter's chunk-of-8 quirk), both editors ----
    cb = make_case_inputs("two", B=4, f=16, h=16, w=16)

    class Holder:
        pass

    holder = Holder()
    holder.unet = unet
    ted = quiet(TemporalSelfAttentionControl, start_step=4, start_layer=10)
    quiet(regiter_temporal_attention_editor_diffusers, holder, ted)
    sed = quiet(FullySelfAttentionControlMask, start_step=4, start_layer=10, source_masks=cb["source_masks"])
    quiet(regiter_fully_attention_editor_diffusers, holder, sed)
    assert ted.num_att_layers == 16 and sed.num_att_layers == 32

    for tag, step in (("inactive", 0), ("active", 4)):
        ted.reset(); sed.reset()
        ted.cur_step = sed.cur_step = step
        my_sp = ref_cpu.SpatialEditor(cb["source_masks"])
        my_tp = ref_cpu.TemporalEditor()
        my_sp.cur_step = my_tp.cur_step = step
        with torch.no_grad():
            t0 = time.time()
            ref = quiet(unet, cb["sample"], torch.tensor(cb["t"]), cb["ehs"], down_block_additional_residuals=cb["down_res"],
                        mid_block_additional_residual=cb["mid_res"]).sample
            print(f"reference two-branch forward ({tag}) {time.time()-t0:.1f}s")
            taps = {}
            mine = ref_cpu.unet_forward(sd, cb["sample"], cb["t"], cb["ehs"], cb["down_res"], cb["mid_res"], my_sp, my_tp, taps=taps)
        e = relerr(mine, ref)
        print(f"two-branch ({tag}) oracle vs reference rel err", e)
        assert e < 2e-4, e
        assert (ted.cur_step, ted.cur_att_layer, sed.cur_step, sed.cur_att_layer) == (my_tp.cur_step, my_tp.cur_att_layer, my_sp.cur_step, my_sp.cur_att_layer) == (step + 1, 0, step + 1, 0)
        np.savez_compressed(GOLD / f"unet_two_{tag}.npz", out=ref.numpy().astype(np.float32), out_stats=stats(ref),
                            skip_stats=np.stack([stats(s) for s in taps["skips"]]), motion_stats=np.stack([stats(s) for s in taps["motion"]]),
                            mid_stats=stats(taps["mid"]), oracle_relerr=e)
    # editors were registered on `unet` above: the inversion forward runs on a fresh, un-patched model
    inversion_goldens(build_reference_unet(sd_np), sd)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
