#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k properties > gpurun_out/model2.log 2>&1
echo "model2 exit $?" > gpurun_out/summary.txt
timeout 900 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 1200 python bench.py > gpurun_out/bench_full.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1 )
echo "rocprof exit $?" >> gpurun_out/summary.txt
find gpurun_out/prof_r1 -name "*.csv" | head; 
tail -3 gpurun_out/model2.log; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench_full.log; cat gpurun_out/summary.txt
