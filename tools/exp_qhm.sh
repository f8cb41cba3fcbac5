#!/bin/bash
# Round 5: head-major Q (ABI 8) -- HBM-side traffic of the level-0 attention launches with Q as row slices vs per-head panels (heads-slowest block order),
# and the step with / without it.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/qhm.txt
: > $out
for kind in pc ed; do
  for q in rows qhm; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_out
      ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_out -o p -- python $R/tools/attn_one.py $kind 4 $q > /dev/null 2>&1 )
      echo "== L0 $kind, Q as $q, $c [KB]" >> $out
      python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
    done
  done
done
cd $R
for m in 1 0 1 0; do
  ME_HEAD_MAJOR_Q=$m timeout 200 python bench.py --steps 8 --warmup 3 --cpu-baseline off --no-profile > gpurun_out/bench_qhm_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/bench_qhm_$m.json'));print('ME_HEAD_MAJOR_Q=$m', d['ms_per_step'])" >> $out
done
cat $out
