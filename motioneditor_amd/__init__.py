"""motioneditor_amd -- MI355X-native (gfx950) implementation of MotionEditor's two-branch DDIM denoising
step behind the reference's own Python interface (motion_editor.pipelines / .attn_control / .models).
All arithmetic runs in ``libmotioned.so`` (hand-written HIP, see ``csrc/`` and ``include/motioned.h``);
there is no CPU or PyTorch fallback."""
__version__ = "0.1.0"
