#!/bin/bash
# Re-take the bench line, the rocprofv3 kernel summary and the PMC traffic passes at HEAD (the part of tools/collect_r05.sh that depends on kernel code), plus smoke().
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke exit $?" > gpurun_out/${tag}_summary2.txt
timeout 900 python bench.py > gpurun_out/${tag}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_summary2.txt
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_c3.json
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o r -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_rocprof.log 2>&1 )
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof -name "*.db" | head -1) gpurun_out/${tag}_bench_c3_kernel_stats.csv 3 >> gpurun_out/${tag}_summary2.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_pmc_f -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_f.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_pmc_w -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_w.log 2>&1 )
python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*.db" | head -1) $(find gpurun_out/${tag}_pmc_w -name "*.db" | head -1) gpurun_out/${tag}_pmc_hbm.csv gpurun_out/${tag}_pmc_traffic.json 3 > /dev/null 2>&1
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
timeout 300 python tools/kbench.py gemm attn misc > gpurun_out/${tag}_kbench2.txt 2>&1
cat gpurun_out/${tag}_summary2.txt; tail -c 300 gpurun_out/${tag}_smoke.log; python -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench_c3.json').read()); r=d['roofline']
print(d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['frac_executed'], r['traffic'], r['algorithmic_bytes_per_launch'])
print({k:v['ms_per_step'] for k,v in d['kernel_families'].items()})
t=json.load(open('gpurun_out/${tag}_pmc_traffic.json')); print(t['attn2_kernel<40,2,16,fold>'], t['gemm'])
"
