#!/bin/bash
# GPU-box check of the step-level C entry points (csrc/plan.hip): bitwise test, then eager / plan / graph A/B of the bench workload in one session.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "planned or graph_replay_six or frame_sharded_path_on_one_rank" > gpurun_out/plan_pytest.log 2>&1
echo "pytest exit $?" > gpurun_out/plan_summary.txt
tail -5 gpurun_out/plan_pytest.log >> gpurun_out/plan_summary.txt
for mode in "" "--plan" "" "--plan" "--graph"; do
  timeout 400 python bench.py $mode --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/plan_bench_tmp.json
  python - "$mode" <<'PY' >> gpurun_out/plan_summary.txt
import json, sys
try:
    d = json.load(open("gpurun_out/plan_bench_tmp.json"))
    print(f"bench {sys.argv[1] or 'eager':8s} ms/step {d['ms_per_step']:8.2f} host_enqueue_ms {d['host_enqueue_ms_per_step']:6.2f} plan={d['config'].get('launch_plan_replay')} graph={d['config'].get('hip_graph_replay')}")
except Exception as e:
    print("bench", sys.argv[1], "FAILED", e, open("gpurun_out/plan_bench_tmp.json").read()[-400:])
PY
done
for mode in "" "--plan"; do
  timeout 300 python bench.py $mode --frames 8 --latent 32 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/plan_bench_tmp.json
  python - "$mode" <<'PY' >> gpurun_out/plan_summary.txt
import json, sys
try:
    d = json.load(open("gpurun_out/plan_bench_tmp.json"))
    print(f"bench 8f x 32^2 {sys.argv[1] or 'eager':8s} ms/step {d['ms_per_step']:8.2f} host_enqueue_ms {d['host_enqueue_ms_per_step']:6.2f}")
except Exception as e:
    print("bench small", sys.argv[1], "FAILED", e, open("gpurun_out/plan_bench_tmp.json").read()[-400:])
PY
done
cat gpurun_out/plan_summary.txt
