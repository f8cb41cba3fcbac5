"""Weight I/O: safetensors / torch-pickle readers and the 2-D -> 3-D inflation rules (reference
models/unet_2d_condition.py:548-796; resnet_2d.py:15-16; attention_2d.py:462; controlnet_adapter.py:418-419,493)."""
import numpy as np
import pytest
import torch

from motioneditor_amd import checkpoint, synth


def test_load_file_roundtrip_safetensors_and_pickle(tmp_path):
    from safetensors.torch import save_file
    sd = {"a.weight": torch.randn(4, 3), "a.bias": torch.randn(4)}
    save_file(sd, str(tmp_path / "diffusion_pytorch_model.safetensors"))
    torch.save(sd, tmp_path / "adapter.pth")
    for got in (checkpoint.load_file(checkpoint.find_weights(tmp_path)), checkpoint.load_file(tmp_path / "adapter.pth")):
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError):
        checkpoint.find_weights(tmp_path / "nope")


def test_inflate_follows_the_reference_constructor_inits():
    schema = {k: v for k, v in synth.unet_schema().items() if k.startswith(("down_blocks.0.resnets.0.", "down_blocks.0.attentions.0.", "controlnet_adapter.body.0."))}
    sd2d = {k: torch.full(v, 0.5) for k, v in schema.items()
            if "temp" not in k and "controlnet_adapter" not in k}     # what an SD-1.5 checkpoint holds
    full, created = checkpoint.inflate(sd2d, schema)
    assert set(full) == set(schema) and all(tuple(full[k].shape) == schema[k] for k in schema)
    assert all(torch.equal(full[k], sd2d[k]) for k in sd2d)
    assert all(("temp" in k) or k.startswith("controlnet_adapter") for k in created)
    z = lambda k: float(full[k].abs().max()) == 0.0   # noqa: E731
    assert z("down_blocks.0.resnets.0.temp_conv1.weight") and z("down_blocks.0.resnets.0.temp_conv2.bias")
    assert z("down_blocks.0.attentions.0.transformer_blocks.0.attn_temp.to_out.0.weight")
    assert z("controlnet_adapter.body.0.block1.weight") and z("controlnet_adapter.body.0.attn_self_temp.to_out.0.weight")
    assert float(full["down_blocks.0.attentions.0.transformer_blocks.0.norm_temp.weight"].min()) == 1.0
    q = full["down_blocks.0.attentions.0.transformer_blocks.0.attn_temp.to_q.weight"]
    assert 0 < float(q.abs().max()) <= 1 / 320 ** 0.5 + 1e-6
    with pytest.raises(ValueError):
        checkpoint.inflate({"down_blocks.0.resnets.0.conv1.weight": torch.zeros(3, 3)}, schema)
