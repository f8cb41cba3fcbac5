#!/bin/bash
# GPU-box A/B of the GroupNorm statistics pass (loads in flight per thread), then the suite pieces that depend on it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/gn_ab.txt
: > $out
for v in old u4 "" u16 u8a8 old ""; do
  lib=$R/motioneditor_amd/libmotioned.so
  [ -n "$v" ] && lib=$R/tools/_bin/libmotioned_gn_$v.so
  echo "== ${v:-u8 (tree)}" >> $out
  ME_LIB=$lib timeout 200 python tools/kbench.py gn 2>&1 | grep -i "groupnorm" >> $out
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "groupnorm or norm or planned or full_size_properties or unet_single or config0" > gpurun_out/gn_pytest.log 2>&1
echo "pytest exit $?" >> $out
tail -3 gpurun_out/gn_pytest.log >> $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gn_smoke.log 2>&1
echo "smoke exit $?" >> $out
tail -2 gpurun_out/gn_smoke.log >> $out
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/gn_bench.json
python - <<'PY' >> $out
import json
d = json.load(open("gpurun_out/gn_bench.json"))
print("bench ms/step", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], "plan", d["config"].get("launch_plan_replay"), {k: v for k, v in d.get("kernel_families", {}).items()} if isinstance(d.get("kernel_families"), dict) else "")
PY
cat $out
