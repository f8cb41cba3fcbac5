#!/bin/bash
tag=${1:-s3}
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
  tail -4 gpurun_out/${tag}_${name}.log | cut -c1-600 >> $S
}
PT="python -m pytest -q -p no:cacheprovider -m gpu"
run k_bwd 400 $PT tests/test_kernels_gpu.py -k "groupnorm or attention_bwd"
run m_bwd 900 $PT tests/test_model_gpu.py -k "null_text or adapter_training_gradients or adapter_trainer"
run kbench_bwd 300 python tools/kbench.py bwd
run nulltext 400 python bench.py --null-text --steps 3 --warmup 1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_nt -o r -- python $R/bench.py --null-text --steps 1 --warmup 1 > $R/gpurun_out/${tag}_rocprof_nt.log 2>&1 )
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof_nt -name "*.db" | head -1) gpurun_out/${tag}_nulltext_kernel_stats.csv 1 >> $S 2>&1
rm -rf gpurun_out/${tag}_prof_nt
cat $S
