#!/bin/bash
# Round-6 evidence on ONE GPU box at HEAD: GPU test suite (with per-test durations), smoke, bench (default flags and the driver's), rocprofv3 kernel stats of the same
# command, PMC traffic passes, per-shape kernel bench incl. the LayerNorm-fold A/B, the fold's step-level A/B, secondary workloads.
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
if [ "$2" != "notests" ]; then
  timeout 1750 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/${tag}_pytest_gpu.log 2>&1
  echo "pytest exit $?" > gpurun_out/${tag}_summary.txt
  cp gpurun_out/parity.jsonl gpurun_out/${tag}_parity.jsonl 2>/dev/null
else
  : > gpurun_out/${tag}_summary.txt
fi
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
echo "build+smoke exit $?" >> gpurun_out/${tag}_summary.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_summary.txt
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_c3.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_c3_driver_flags.json
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o r -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_rocprof.log 2>&1 )
echo "rocprof exit $?" >> gpurun_out/${tag}_summary.txt
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof -name "*.db" | head -1) gpurun_out/${tag}_bench_c3_kernel_stats.csv 3 >> gpurun_out/${tag}_summary.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_pmc_f -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_f.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_pmc_w -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_w.log 2>&1 )
python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*.db" | head -1) $(find gpurun_out/${tag}_pmc_w -name "*.db" | head -1) gpurun_out/${tag}_pmc_hbm.csv gpurun_out/${tag}_pmc_traffic.json 3 >> gpurun_out/${tag}_summary.txt 2>&1
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
timeout 500 python tools/kbench.py gemm attn misc > gpurun_out/${tag}_kbench.txt 2>&1
timeout 200 python tools/kbench.py lnfold > gpurun_out/${tag}_kbench_lnfold.txt 2>&1
# the fold at step level, alternating runs on this box
ab=gpurun_out/${tag}_lnfold_ab.txt
: > $ab
for i in 1 2; do
  for v in 1 0; do
    ME_LN_FOLD=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ME_LN_FOLD=$v', d['ms_per_step'], 'ms/step', {k:v['ms_per_step'] for k,v in d['kernel_families'].items()}, 'launches', d['launch_plan']['launches'])" >> $ab
  done
done
# later in the round: the resident-key cross-attention, dh = 80 at 32 queries per wave, the dispatch thresholds (A/B on this box)
timeout 200 python tools/kbench.py attnkvres > gpurun_out/${tag}_kbench_attnkvres.txt 2>&1
timeout 200 python tools/exp_attn80.py > gpurun_out/${tag}_attn80.txt 2>&1
timeout 600 bash tools/exp_dispatch_ab.sh > gpurun_out/${tag}_dispatch_ab_collection_box.txt 2>&1
# secondary workloads
sec=gpurun_out/${tag}_secondary.jsonl
: > $sec
run() { echo "{\"cmd\": \"bench.py $*\"}" >> $sec; timeout 400 python bench.py "$@" 2>/dev/null | grep '^{' | tail -1 >> $sec; }
run --editors inactive --steps 8 --warmup 3 --no-cpu-baseline --no-profile
run --single-branch --frames 8 --steps 10 --warmup 3 --no-cpu-baseline
run --frames 8 --latent 32 --steps 6 --warmup 2 --no-cpu-baseline --no-profile
run --frames 48 --latent 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile
run --eager --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run --parallel frames --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run --parallel frames --shard-overlap --steps 4 --warmup 2 --no-cpu-baseline --no-profile
cat gpurun_out/${tag}_summary.txt; tail -n 26 gpurun_out/${tag}_pytest_gpu.log 2>/dev/null; tail -c 300 gpurun_out/${tag}_bench_c3.json; cat $ab
