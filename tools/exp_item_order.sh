#!/bin/bash
# Round 5: processing order of the items inside a head's run for the edited attention launches (recon f, edit f, recon f + 1, ... vs ascending):
# time (alternating, one process each), HBM-side traffic, the step.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/item_order.txt
: > $out
cd $R
for m in 1 0 1 0; do
  echo "== ME_ATTN_ITEM_ORDER=$m: L0 / L1 edited (tools/kbench.py attnorder, heads-slowest column)" >> $out
  ME_ATTN_ITEM_ORDER=$m timeout 200 python tools/kbench.py attnorder 2>/dev/null | grep -E "edited" >> $out
done
for m in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_out
    ( cd /tmp && ME_ATTN_ITEM_ORDER=$m timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_out -o p -- python $R/tools/attn_one.py ed 4 qhm > /dev/null 2>&1 )
    echo "== L0 ed, ME_ATTN_ITEM_ORDER=$m, $c [KB]" >> $out
    python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
  done
done
for m in 1 0 1 0; do
  ME_ATTN_ITEM_ORDER=$m timeout 200 python bench.py --steps 8 --warmup 3 --cpu-baseline off --no-profile > gpurun_out/bench_io_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/bench_io_$m.json'));print('ME_ATTN_ITEM_ORDER=$m', d['ms_per_step'])" >> $out
done
cat $out
