"""Build libmotioned.so (hipcc, gfx950 only) in-tree.  Invoked by ``__graft_entry__.build()``;
also ``python -m motioneditor_amd.build``.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "csrc" / "_obj"
LIB = PKG / "libmotioned.so"
SOURCES = ["capi.hip", "gemm.hip", "attn.hip", "tattn.hip", "norm.hip", "eltwise.hip", "bwd.hip", "attn_bwd.hip", "train.hip", "plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
# attn.hip: without nnan, every fmaxf on an MFMA result gets a canonicalising v_max_f32 in front of it (21 extra VALU per
# 64-key tile in a kernel whose VALU time adds to its MFMA time); the kernel never produces or tests NaN / Inf.
EXTRA_FLAGS = {"attn.hip": ["-ffinite-math-only"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    cc = hipcc()
    headers = [CSRC / "me_common.h", PKG.parent / "include" / "motioned.h"]

    def one(src: str) -> Path:
        s, o = CSRC / src, OBJ / (Path(src).stem + ".o")
        if force or _stale(o, [s, *headers, Path(__file__)]):
            cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", str(s), "-o", str(o)]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
