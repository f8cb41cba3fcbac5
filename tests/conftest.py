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


@pytest.fixture(autouse=True)
def _give_memory_back():
    """The fp32 emulation and the oracle allocate and free gigabytes per test; glibc keeps freed heap pages (its mmap threshold adapts upwards to the
    sizes it sees), so this process would sit on ~14 GB by the time the world-4 / world-8 gloo cases need room for their ranks.  Collect and trim after
    every test."""
    yield
    import ctypes
    import gc
    gc.collect()
    try:
        ctypes.CDLL("libc.so.6").malloc_trim(0)
    except OSError:
        pass


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().mean().clamp_min(1e-30))


# ME_TEST_RSS_LOG=<file>: one line per test with the resident set of the pytest process after it (the multi-process gloo cases need room for world x
# ~9 GB of fp32 weights beside this process; a test that leaves gigabytes behind in a module global or a cache shows up here)
def pytest_runtest_teardown(item, nextitem):
    log = os.environ.get("ME_TEST_RSS_LOG")
    if not log:
        return
    rss_kb = 0
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith("VmRSS:"):
                rss_kb = int(line.split()[1])
    with open(log, "a") as fh:
        fh.write(f"{rss_kb / 1048576:7.2f} GB  {item.nodeid}\n")
