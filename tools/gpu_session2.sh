#!/bin/bash
# Targeted re-validation: bash tools/gpu_session2.sh <tag>
tag=${1:-s2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/${tag}_summary.txt
rm -f gpurun_out/parity.jsonl $S
run() {
  local name=$1 to=$2; shift 2
  local t0=$(date +%s)
  timeout $to "$@" > gpurun_out/${tag}_${name}.log 2>&1
  echo "$name rc $? ($(( $(date +%s) - t0 )) s)" >> $S
  tail -4 gpurun_out/${tag}_${name}.log | cut -c1-400 >> $S
}
PT="python -m pytest -q -p no:cacheprovider -m gpu"
run k_gradacc 300 $PT tests/test_kernels_gpu.py -k "grad_acc"
run m_shard 600 $PT tests/test_model_gpu.py -k "frame_sharded" -s
run m_trainer 900 $PT tests/test_model_gpu.py -k "adapter_trainer or training_example" -s
cp gpurun_out/parity.jsonl gpurun_out/${tag}_parity.jsonl 2>/dev/null
run bench_frames_graph 600 python bench.py --parallel frames --graph --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run bench_frames_eager_rccl 600 python bench.py --parallel frames --comm rccl --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run bench_frames_eager_torch 600 python bench.py --parallel frames --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run bench_single_graph 600 python bench.py --graph --steps 4 --warmup 2 --no-cpu-baseline --no-profile
cat $S
