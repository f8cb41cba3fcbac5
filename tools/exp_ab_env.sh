#!/bin/bash
# Step-level A/B of an environment switch on one box, alternating runs:  tools/exp_ab_env.sh VAR [steps] [value]   (VAR=value, default 0, against unset)
cd ${GRAFT_REPO_ROOT:-/root/repo}
v=$1; n=${2:-8}; x=${3:-0}
for i in 1 2; do
  for val in "" $x; do
    env ${val:+$v=$val} python bench.py --steps $n --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v=${val:-unset}', d['ms_per_step'], 'ms/step', {k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
  done
done
