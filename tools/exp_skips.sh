#!/bin/bash
# GPU-box check of the in-place skips (graph.INPLACE_SKIPS): model-level tests, then the bench with and without (ME_INPLACE_SKIPS=0), same session.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/skips_ab.txt
: > $out
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/skips_pytest.log 2>&1
echo "pytest exit $?" >> $out
tail -2 gpurun_out/skips_pytest.log >> $out
for v in 0 1 0 1; do
  ME_INPLACE_SKIPS=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/skips_bench_tmp.json
  python - $v <<'PY' >> $out
import json, sys
try:
    d = json.load(open("gpurun_out/skips_bench_tmp.json"))
    fam = d.get("kernel_families", {})
    print(f"ME_INPLACE_SKIPS={sys.argv[1]} ms/step {d['ms_per_step']:8.2f} host {d['host_enqueue_ms_per_step']:5.2f} launches {d.get('launch_plan', {}).get('launches')} gemm {fam.get('gemm', {}).get('ms_per_step')} attn40 {fam.get('attn_dh40', {}).get('ms_per_step')} gn {fam.get('groupnorm', {}).get('ms_per_step')}")
except Exception as e:
    print("bench FAILED", sys.argv[1], e, open("gpurun_out/skips_bench_tmp.json").read()[-300:])
PY
done
cat $out
