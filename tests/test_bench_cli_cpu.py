"""bench.py launch contract: `python bench.py --gpus N` with no WORLD_SIZE in the environment must start N ranks by itself
(torch.distributed.run) and print ONE JSON line with n_gpus == N and the per-step exchange budget.  Runs the product host
code on the CPU emulation of the C ABI (gloo) at a tiny shape."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_bench_gpus_2_spawns_two_ranks_and_reports_them():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--emulate", "--frames", "8", "--latent", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "strong"
    assert d["config"]["parallel_mode"] == "cfg"
    assert d["comm"]["total"]["calls_per_step"] == 1      # one all-gather of the noise prediction per step
    # ... and how long the rank's stream was held by it (two extra untimed steps with an event / clock pair around every exchange, parallel.TIMING)
    assert d["comm"]["total"]["stream_held_ms_per_step"] > 0 and "stream_held_note" in d["comm"]
    assert d["comm"]["all_gather(noise prediction, CFG pair)"]["stream_held_ms_per_step"] == d["comm"]["total"]["stream_held_ms_per_step"]
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert k in d


def test_mode_resolution():
    sys.path.insert(0, str(ROOT))
    import bench
    A = type("A", (), {})
    a = A()
    a.parallel = "auto"
    assert bench.resolve_mode(a, 1) == ("single", 1, 1, 1)
    assert bench.resolve_mode(a, 2) == ("cfg", 2, 1, 1)
    assert bench.resolve_mode(a, 4) == ("cfg-frames", 2, 2, 1)
    assert bench.resolve_mode(a, 8) == ("cfg-frames", 2, 4, 1)
    a.parallel = "frames"
    assert bench.resolve_mode(a, 8) == ("frames", 1, 8, 1)
    a.parallel = "replicas"
    assert bench.resolve_mode(a, 4) == ("replicas", 1, 1, 4)


def test_cpu_baseline_full_mode_prints_one_clocked_line_without_a_gpu():
    """`bench.py --cpu-baseline full`: host-only, ONE oracle step at the requested workload, one JSON line (what is committed as
    profiles/cpu_baseline_full.json at the bench workload) -- here at 8 frames x 8x8 latents."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline", "full", "--frames", "8", "--latent", "8"], capture_output=True, text=True, timeout=900,
                       cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])["cpu_baseline_full"]
    assert d["extrapolated"] is False and d["seconds"] > 0 and d["kind"] == "port" and d["workload"].startswith("8 frames x 64x64 ")
    # the committed full-config figure is the bench workload's and is what the default run reports as cpu_baseline.value
    full = json.loads((ROOT / "profiles" / "cpu_baseline_full.json").read_text())["cpu_baseline_full"]
    assert full["workload"].startswith("24 frames x 512x512 ") and full["extrapolated"] is False and full["seconds"] > 60


def test_synthetic_weight_cache_returns_the_generated_tensors(tmp_path, monkeypatch):
    import numpy as np
    sys.path.insert(0, str(ROOT))
    from motioneditor_amd import synth
    schema = {"a.weight": (1024, 640), "b.bias": (640,), "c.weight": (640, 3, 320)}
    monkeypatch.setenv("ME_SYNTH_CACHE", "0")
    ref = synth.synth_state_dict(schema, seed=5, salt="t.")
    monkeypatch.setenv("ME_SYNTH_CACHE", str(tmp_path))
    first = synth.synth_state_dict(schema, seed=5, salt="t.")
    assert len(list(tmp_path.iterdir())) == 1
    again = synth.synth_state_dict(schema, seed=5, salt="t.")
    for k in schema:
        assert np.array_equal(ref[k], first[k]) and np.array_equal(ref[k], again[k]) and again[k].shape == tuple(schema[k])
    other = synth.synth_state_dict(schema, seed=6, salt="t.")
    assert not np.array_equal(other["a.weight"], ref["a.weight"]) and len(list(tmp_path.iterdir())) == 2


def test_planned_or_eager_falls_back_only_while_nothing_was_replayed():
    """bench.planned_or_eager: a failure while RECORDING the step's launch plan (nothing replayed yet) switches the run to the eager executor for good and
    is reported; a failure of a plan that exists propagates."""
    import bench

    class Pipe:
        def __init__(self, fail):
            self._plans, self.fail, self.calls = {}, fail, []

        def denoise_step_planned(self, lat, t, emb, images, g):
            self.calls.append("plan")
            if self.fail:
                raise RuntimeError("no MemPool here")
            self._plans["k"] = object()
            return lat + 1

        def denoise_step(self, lat, t, emb, images, g):
            self.calls.append("eager")
            return lat + 1

    st = {"on": True, "error": None}
    p = Pipe(fail=True)
    assert bench.planned_or_eager(p, st, 1, 0, None, None) == 2 and p.calls == ["plan", "eager"]
    assert st == {"on": False, "error": "RuntimeError: no MemPool here"}
    assert bench.planned_or_eager(p, st, 2, 0, None, None) == 3 and p.calls == ["plan", "eager", "eager"]      # stays eager
    st = {"on": True, "error": None}
    p = Pipe(fail=False)
    assert bench.planned_or_eager(p, st, 1, 0, None, None) == 2 and st["on"] and p.calls == ["plan"]
    p.fail = True                                                                                                  # a recorded plan that fails to replay
    import pytest as _pt
    with _pt.raises(RuntimeError):
        bench.planned_or_eager(p, st, 1, 0, None, None)
