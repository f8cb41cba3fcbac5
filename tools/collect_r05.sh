#!/bin/bash
# Round-5 evidence on ONE GPU box at HEAD (a trimmed tools/collect_profiles.sh: the round's A/B experiments have their own scripts -- exp_fill.sh, exp_attn_order.sh,
# exp_qhm.sh, kbench.py rowepi): GPU test suite, smoke, bench (+ rocprofv3 kernel stats of the same command, PMC traffic passes), per-shape kernel bench, secondary workloads.
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > gpurun_out/${tag}_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/${tag}_summary.txt
cp gpurun_out/parity.jsonl gpurun_out/${tag}_parity.jsonl 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
echo "build+smoke exit $?" >> gpurun_out/${tag}_summary.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_summary.txt
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_c3.json
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o r -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_rocprof.log 2>&1 )
echo "rocprof exit $?" >> gpurun_out/${tag}_summary.txt
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof -name "*.db" | head -1) gpurun_out/${tag}_bench_c3_kernel_stats.csv 3 >> gpurun_out/${tag}_summary.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_pmc_f -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_f.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_pmc_w -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_w.log 2>&1 )
python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*.db" | head -1) $(find gpurun_out/${tag}_pmc_w -name "*.db" | head -1) gpurun_out/${tag}_pmc_hbm.csv gpurun_out/${tag}_pmc_traffic.json 3 >> gpurun_out/${tag}_summary.txt 2>&1
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
timeout 500 python tools/kbench.py gemm attn misc gn bwd > gpurun_out/${tag}_kbench.txt 2>&1
# secondary workloads
sec=gpurun_out/${tag}_secondary.jsonl
: > $sec
run() { echo "{\"cmd\": \"bench.py $*\"}" >> $sec; timeout 400 python bench.py "$@" 2>/dev/null | grep '^{' | tail -1 >> $sec; }
run --editors inactive --steps 8 --warmup 3 --no-cpu-baseline --no-profile
run --single-branch --frames 8 --steps 10 --warmup 3 --no-cpu-baseline
run --frames 8 --latent 32 --steps 6 --warmup 2 --no-cpu-baseline --no-profile
run --frames 48 --latent 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile
run --eager --steps 4 --warmup 2 --no-cpu-baseline --no-profile
run --null-text --steps 3 --warmup 1
# one more A/B that costs a minute: the LDS-halo convolution kernel vs the 8-phase gather kernel on the stride-1 3x3 convolutions (the halo kernel predates the 8-phase schedule)
{ echo "== default (conv3_halo_kernel)"; timeout 120 python tools/kbench.py gemm 2>/dev/null | grep -E "conv3x3|conv [0-9]"; echo "== ME_CONV_HALO=0 (gemm8p gather)"; ME_CONV_HALO=0 timeout 120 python tools/kbench.py gemm 2>/dev/null | grep -E "conv3x3|conv [0-9]"; } > gpurun_out/${tag}_conv_halo_ab.txt 2>&1
cat gpurun_out/${tag}_summary.txt; tail -n 22 gpurun_out/${tag}_pytest_gpu.log; tail -c 400 gpurun_out/${tag}_bench_c3.json; cat gpurun_out/${tag}_conv_halo_ab.txt
