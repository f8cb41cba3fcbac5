import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from motioneditor_amd import synth, ops
from motioneditor_amd.models.unet_2d_condition import UNet2DConditionModel
from motioneditor_amd.models.controlnet import ControlNetModel
from motioneditor_amd.pipelines import MotionEditorPipeline
from motioneditor_amd.attn_control import (FullySelfAttentionControlMask, TemporalSelfAttentionControl,
                                           regiter_fully_attention_editor_diffusers, regiter_temporal_attention_editor_diffusers)
from test_step_cpu import step_inputs
unet = UNet2DConditionModel(synth.synth_state_dict(synth.unet_schema()), device="cuda")
cn = ControlNetModel(synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet."), device="cuda")
f, hw, step = 24, 16, 4
x = step_inputs(f=f, h=hw, w=hw)
images = torch.cat([x["skeleton"]] * 2).reshape(2 * f, 3, 8 * hw, 8 * hw).cuda()
emb = torch.cat([x["uncond"].expand(2, 77, 768), x["cond"]]).cuda()
pipe = MotionEditorPipeline(unet=unet, controlnet=cn)
pipe.scheduler.set_timesteps(50)
outs = {}
for share in (True, True, False, False):
    class H: pass
    h = H(); h.unet = unet
    ted = TemporalSelfAttentionControl(start_step=4, start_layer=10); regiter_temporal_attention_editor_diffusers(h, ted)
    sed = FullySelfAttentionControlMask(start_step=4, start_layer=10, source_masks=x["masks"]); regiter_fully_attention_editor_diffusers(h, sed)
    sed.cur_step = ted.cur_step = step
    pipe.dedup_cfg_prefix = share
    taps = {}
    o = pipe.denoise_step(x["latents"].cuda(), pipe.scheduler.timesteps[step], emb, images, 7.5, taps=taps if "--taps" in sys.argv else None).clone()
    outs.setdefault(share, []).append((o, taps))
r = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print("share run-to-run", r(outs[True][0][0], outs[True][1][0]), " noshare run-to-run", r(outs[False][0][0], outs[False][1][0]), " share vs noshare", r(outs[True][0][0], outs[False][0][0]))
if "--taps" in sys.argv:
    for i, (a, b) in enumerate(zip(outs[True][0][1]["skips"], outs[False][0][1]["skips"])):
        print("skip", i, r(a, b))
