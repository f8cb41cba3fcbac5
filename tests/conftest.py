import os
import sys

os.environ.setdefault("ME_GRAD_POISON", "1")   # never-zeroed gradient buffers of the autodiff tape start as NaN: a read before the first store fails a test
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
GOLD = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def unet_sd_np():
    from motioneditor_amd import synth
    return synth.synth_state_dict(synth.unet_schema())


@pytest.fixture(scope="session")
def cn_sd_np():
    from motioneditor_amd import synth
    return synth.synth_state_dict(synth.controlnet_schema(), salt="controlnet.")


@pytest.fixture(scope="session")
def unet_sd_torch(unet_sd_np):
    return {k: torch.from_numpy(v) for k, v in unet_sd_np.items()}


@pytest.fixture(scope="session")
def cn_sd_torch(cn_sd_np):
    return {k: torch.from_numpy(v) for k, v in cn_sd_np.items()}


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().mean().clamp_min(1e-30))
