"""P0 of SURVEY 8a: the harness call sequence (inference.py:296-323).  The synthetic inputs of examples/run_edit.py are pinned by
tests/golden/harness_inputs.npz (shapes + checksums), the tensor conventions by direct assertions."""
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "examples"))


def test_harness_inputs_match_the_fixture():
    import run_edit
    g = np.load(GOLD / "harness_inputs.npz")
    x = run_edit.harness_inputs(8, 64, 64)
    assert set(k.split(".")[0] for k in g.files) == set(x)
    for k, v in x.items():
        a = v.numpy().astype(np.float64)
        assert tuple(g[k + ".shape"]) == a.shape, k
        np.testing.assert_allclose([a.sum(), np.abs(a).sum(), a.min(), a.max()], g[k + ".stats"], rtol=1e-9, atol=1e-9, err_msg=k)
    assert float(x["pixel_values"].abs().max()) <= 1.0 and set(np.unique(x["source_masks"].numpy())) <= {0.0, 1.0}
    assert 0.0 <= float(x["target_skeleton"].min()) and float(x["target_skeleton"].max()) <= 1.0


def test_skeleton_convention_only_the_last_entry_is_used():
    """inference.py:300-302 builds cat([0, target, 0, target]); the pipeline takes skeleton[-1] (pipeline_motion_editor.py:556),
    repeats it for the batch and doubles it for classifier-free guidance -> [2 f, 3, H, W]."""
    from motioneditor_amd.pipelines import MotionEditorPipeline

    class U:   # the constructor only reads .device
        device = torch.device("cpu")

    pipe = MotionEditorPipeline(unet=U())
    tgt = torch.rand(1, 4, 3, 16, 16)
    skeleton = torch.cat([torch.zeros_like(tgt), tgt, torch.zeros_like(tgt), tgt])
    img = pipe.prepare_image(torch.unsqueeze(skeleton[-1], 0), 16, 16, 1, 1, "cpu", torch.float32, True)
    assert img.shape == (2, 4, 3, 16, 16) and torch.equal(img[0], tgt[0]) and torch.equal(img[1], tgt[0])


def test_foreign_scheduler_config_is_adopted_and_bad_ones_rejected():
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from motioneditor_amd.schedulers import DDIMScheduler

    class U:
        device = torch.device("cpu")

    class Frozen(dict):   # diffusers' FrozenDict refuses item assignment
        def __setitem__(self, k, v):
            raise RuntimeError("frozen")
        __getattr__ = dict.__getitem__

    class Foreign:
        config = Frozen(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True,
                        set_alpha_to_one=False, steps_offset=1)

    p = MotionEditorPipeline(unet=U(), scheduler=Foreign())
    assert isinstance(p.scheduler, DDIMScheduler) and p.scheduler.config.clip_sample is False
    p.scheduler.set_timesteps(50)
    assert p.scheduler.timesteps[0] == 981 and p.scheduler.timesteps[-1] == 1
    try:
        MotionEditorPipeline(unet=U(), scheduler=object())
        raise AssertionError("scheduler without a config accepted")
    except TypeError:
        pass


def test_prepare_image_accepts_pil_and_tensor_lists_like_the_reference():
    """pipeline_motion_editor.py:418-459: PIL -> RGB -> Lanczos resize -> [0,1] NCHW; lists of tensors are concatenated;
    one image is repeated over the batch; classifier-free guidance doubles the batch."""
    import numpy as np
    import PIL.Image
    from motioneditor_amd.pipelines.pipeline_motion_editor import MotionEditorPipeline
    pipe = MotionEditorPipeline.__new__(MotionEditorPipeline)
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (24, 20, 3), dtype=np.uint8)
    ims = [PIL.Image.fromarray(a), PIL.Image.fromarray(a[::-1].copy()).convert("L")]
    out = pipe.prepare_image(ims, 16, 8, 2, 1, "cpu", torch.float32, True)
    assert out.shape == (4, 3, 8, 16) and out.dtype == torch.float32 and 0.0 <= float(out.min()) and float(out.max()) <= 1.0
    want0 = np.asarray(ims[0].convert("RGB").resize((16, 8), resample=PIL.Image.Resampling.LANCZOS)).astype(np.float32) / 255.0
    assert torch.equal(out[0], torch.from_numpy(want0.transpose(2, 0, 1))) and torch.equal(out[:2], out[2:])
    assert torch.equal(out[1, 0], out[1, 1]) and torch.equal(out[1, 1], out[1, 2])       # the "L" image became grey RGB
    one = pipe.prepare_image(ims[0], 16, 8, 3, 1, "cpu", torch.float16, False)
    assert one.shape == (3, 3, 8, 16) and one.dtype == torch.float16 and torch.equal(one[0], one[2])
    ts = [torch.rand(1, 3, 8, 16), torch.rand(1, 3, 8, 16)]
    cat = pipe.prepare_image(ts, 16, 8, 2, 1, "cpu", torch.float32, False)
    assert torch.equal(cat, torch.cat(ts))


def test_prepare_image_matches_the_reference_function_golden():
    """tests/golden/prepare_image.npz = outputs of the reference's own prepare_image (oracle/make_golden.py executes the
    function as written) on seeded uint8 images, which travel in the fixture."""
    import numpy as np
    import PIL.Image
    from conftest import GOLD
    from motioneditor_amd.pipelines.pipeline_motion_editor import MotionEditorPipeline
    g = np.load(GOLD / "prepare_image.npz")
    pipe = MotionEditorPipeline.__new__(MotionEditorPipeline)
    ims = [PIL.Image.fromarray(g["img_a"]), PIL.Image.fromarray(g["img_b"])]
    T = torch.from_numpy
    assert torch.equal(pipe.prepare_image(ims, 32, 24, 2, 1, "cpu", torch.float32, True), T(g["pil_list_cfg"]))
    assert torch.equal(pipe.prepare_image(ims[0], 32, 24, 3, 1, "cpu", torch.float32, False), T(g["pil_one_b3"]))
    t = T(g["tensor_in"])
    assert torch.equal(pipe.prepare_image(t, 32, 24, 2, 1, "cpu", torch.float32, True), T(g["tensor_cfg"]))
    assert torch.equal(pipe.prepare_image([t[:1], t[1:]], 32, 24, 2, 1, "cpu", torch.float32, False), T(g["tensor_list"]))


def test_reference_harness_setup_calls_run_on_the_drop_in_classes(tmp_path, unet_sd_np, cn_sd_np):
    """The calls inference.py:152-248 makes on its models between loading them and the first batch -- requires_grad_(False),
    enable_xformers_memory_efficient_attention, the pipeline constructor with DDIMScheduler.from_pretrained(..., subfolder="scheduler"),
    enable_vae_slicing, .to(device, dtype=...), unet.controlnet_adapter.load_state_dict(adapter checkpoint), named_modules /
    named_parameters, eval() -- replayed on motioneditor_amd's classes (weights from state dicts; from_pretrained itself is covered by
    test_checkpoint_cpu.py and the GPU round trip)."""
    import json
    from motioneditor_amd import synth
    from motioneditor_amd.models.controlnet import ControlNetModel
    from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
    from motioneditor_amd.models.vae import AutoencoderKL
    from motioneditor_amd.pipelines import MotionEditorPipeline
    from motioneditor_amd.schedulers import DDIMScheduler
    (tmp_path / "scheduler").mkdir()
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(dict(   # SD-1.5's scheduler_config.json
        _class_name="PNDMScheduler", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
        set_alpha_to_one=False, steps_offset=1, skip_prk_steps=True, trained_betas=None)))
    vae = AutoencoderKL(synth.synth_state_dict(synth.vae_decoder_schema(), salt="vae."), device="cpu", dtype=torch.float32)
    unet = UNet2DConditionModel(unet_sd_np, device="cpu", dtype=torch.float32)
    controlnet = ControlNetModel(cn_sd_np, device="cpu", dtype=torch.float32)
    for m in (vae, unet, controlnet):
        assert m.requires_grad_(False) is m                                     # :159-162
    unet.enable_xformers_memory_efficient_attention()                           # :166
    unet.enable_gradient_checkpointing()                                        # :171
    pipe = MotionEditorPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler"),
                                safety_checker=None, feature_extractor=None, controlnet=controlnet)          # :187-196
    pipe.enable_vae_slicing()                                                   # :197
    inv = DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")   # :198-199
    inv.set_timesteps(50)
    assert inv.timesteps[0] == 981 and inv.timesteps[-1] == 1 and pipe.scheduler.config.steps_offset == 1
    vae.to("cpu", dtype=torch.float32)                                          # :215-217
    controlnet.to(torch.device("cpu"), dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        controlnet.to("cpu", dtype=torch.bfloat16)
    # :237-240 -- the stage-2 adapter checkpoint: a state dict without the "controlnet_adapter." prefix
    g = torch.Generator().manual_seed(4)
    adapter = {k[len("controlnet_adapter."):]: torch.randn(v.shape, generator=g) for k, v in unet_sd_np.items() if k.startswith("controlnet_adapter.")}
    torch.save(adapter, tmp_path / "adapter.pth")
    packed_before = unet.P.mat("controlnet_adapter.body.0.block2.weight")
    unet.controlnet_adapter.load_state_dict(torch.load(tmp_path / "adapter.pth"))
    k0 = "body.0.block2.weight"
    assert torch.equal(unet.P.raw("controlnet_adapter." + k0), adapter[k0])
    assert unet.P.mat("controlnet_adapter." + k0) is not packed_before and torch.equal(unet.P.mat("controlnet_adapter." + k0)[:, 0, :], adapter[k0][:, :, 0])   # re-packed
    with pytest.raises(RuntimeError):
        unet.controlnet_adapter.load_state_dict({k0: adapter[k0]})               # strict: the rest is missing
    with pytest.raises(RuntimeError):
        unet.controlnet_adapter.load_state_dict({**adapter, k0: torch.zeros(3)})   # size mismatch
    mods = [n for n, _ in pipe.unet.named_modules()]                            # :242-246
    params = [n for n, _ in pipe.unet.named_parameters()]
    assert mods[0] == "" and "controlnet_adapter.body.11.attn_self_temp" in mods and "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q" in mods
    assert params == list(synth.unet_schema()) and len(set(mods)) == len(mods)
    assert unet.eval() is unet and unet.training is False                       # :248
    assert unet.dtype == torch.float16 and unet.device == torch.device("cpu")   # :268 source_masks.to(device=unet.device, dtype=unet.dtype)
    # a foreign config that lacks the SD-1.5 keys means diffusers' defaults (linear betas): refused, not silently replaced
    (tmp_path / "bare").mkdir()
    (tmp_path / "bare" / "scheduler_config.json").write_text(json.dumps(dict(num_train_timesteps=1000)))
    with pytest.raises(NotImplementedError):
        DDIMScheduler.from_pretrained(str(tmp_path / "bare"))
    for bad in (dict(timestep_spacing="trailing"), dict(thresholding=True), dict(rescale_betas_zero_snr=True), dict(trained_betas=[0.1])):
        with pytest.raises(NotImplementedError):
            DDIMScheduler.from_config(dict(beta_schedule="scaled_linear", **bad))
    d = DDIMScheduler.from_config(dict(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012))
    assert d.config.steps_offset == 0 and d.config.set_alpha_to_one is True     # diffusers' defaults for absent keys
