#!/bin/bash
# Collect the judged evidence on the GPU box: usage  [LEAN=1 KBENCH_EXTRA=] bash tools/collect_profiles.sh <tag>   (e.g. r01_v4; LEAN skips what did not change: 8-phase A/B, head-major A/B, null-text kernel trace)
# Writes gpurun_out/<tag>_*; copy what should be kept into profiles/ afterwards.  The rocprofv3 passes run the eager executor so that the trace holds exactly
# warm-up + timed steps (the default executor records its plan in two extra passes); the kernels are the same.
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${tag}_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/${tag}_summary.txt
cp gpurun_out/parity.jsonl gpurun_out/${tag}_parity.jsonl 2>/dev/null
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
echo "build+smoke exit $?" >> gpurun_out/${tag}_summary.txt
timeout 1200 python bench.py > gpurun_out/${tag}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${tag}_summary.txt
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_c3.json
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o r -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_rocprof.log 2>&1 )
echo "rocprof exit $?" >> gpurun_out/${tag}_summary.txt
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof -name "*.db" | head -1) gpurun_out/${tag}_bench_c3_kernel_stats.csv 3 >> gpurun_out/${tag}_summary.txt 2>&1
( cd /tmp && timeout 1200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_pmc_f -o f -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_f.log 2>&1 )
( cd /tmp && timeout 1200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_pmc_w -o w -- python $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap > $R/gpurun_out/${tag}_pmc_w.log 2>&1 )
python tools/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*.db" | head -1) $(find gpurun_out/${tag}_pmc_w -name "*.db" | head -1) gpurun_out/${tag}_pmc_hbm.csv gpurun_out/${tag}_pmc_traffic.json 3 >> gpurun_out/${tag}_summary.txt 2>&1
timeout 600 python tools/kbench.py gemm attn misc gn ${KBENCH_EXTRA:-bwd} > gpurun_out/${tag}_kbench.txt 2>&1
# A/B builds present under tools/_bin (tools/build_abl.sh): the same attention micro-benchmark under each, same session
for ab in tools/_bin/libmotioned_attn_*.so; do
  [ -f "$ab" ] || continue
  { echo "== $(basename $ab)"; ME_LIB=$R/$ab timeout 200 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids; echo "== tree"; timeout 200 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids; } >> gpurun_out/${tag}_attn_ab.txt
done
[ -z "$LEAN" ] && timeout 300 python tools/kbench.py gemm8p > gpurun_out/${tag}_kbench_8p.txt 2>&1
# secondary measurements: null-text inner iteration, other shapes, the frame-sharded path on one rank (eager / captured)
timeout 400 python bench.py --null-text --steps 3 --warmup 1 > gpurun_out/${tag}_nulltext.log 2>&1; tail -1 gpurun_out/${tag}_nulltext.log > gpurun_out/${tag}_bench_nulltext.json
timeout 300 python bench.py --frames 8 --latent 32 --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_8f_256.json
timeout 300 python bench.py --single-branch --frames 8 --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/${tag}_bench_single_branch_8f_512.json   # BASELINE configs[1]
[ -z "$LEAN" ] && timeout 120 python tools/kbench.py attnhmp > gpurun_out/${tag}_attn_headmajor.txt 2>&1
[ -z "$LEAN" ] && timeout 600 python bench.py --frames 48 --latent 96 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_48f_768.json
[ -z "$LEAN" ] && timeout 300 python bench.py --parallel frames --graph --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | grep '^{' | tail -1 > gpurun_out/${tag}_bench_frames1_graph.json
[ -z "$LEAN" ] && timeout 300 python bench.py --parallel frames --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | grep '^{' | tail -1 > gpurun_out/${tag}_bench_frames1_eager.json
[ -z "$LEAN" ] && timeout 300 python bench.py --no-overlap --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_no_overlap.json
timeout 300 python bench.py --graph --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_graph.json
timeout 300 python bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_eager.json   # the default executor is me_denoise_step (csrc/plan.hip): this is the Python-enqueued step
timeout 300 python bench.py --frames 8 --latent 32 --eager --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/${tag}_bench_8f_256_eager.json
if [ -z "$LEAN" ]; then
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_nt -o r -- python $R/bench.py --null-text --steps 1 --warmup 1 > $R/gpurun_out/${tag}_rocprof_nt.log 2>&1 )
python tools/rocpd_summary.py $(find gpurun_out/${tag}_prof_nt -name "*.db" | head -1) gpurun_out/${tag}_nulltext_kernel_stats.csv 1 >> gpurun_out/${tag}_summary.txt 2>&1
rm -rf gpurun_out/${tag}_prof_nt
fi
# (instruction-rate micro-benchmarks and the attention kernel's SQ counters: tools/ubench.hip, tools/exp_pmc_attn.sh -- unchanged since round 2, profiles/r02_*)
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
cat gpurun_out/${tag}_summary.txt; tail -c 600 gpurun_out/${tag}_bench_c3.json
