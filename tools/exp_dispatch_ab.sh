#!/bin/bash
# Round 6: step-level A/B of me_gemm's new dispatch thresholds (192-row 8-phase kernel from 192 tiles / 4 K tiles, GEGLU 8-phase from 240 tiles, dense 128-row
# 8-phase from 192 tiles) against the old ones, one box, alternating runs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for old in 0 1; do
    if [ $old = 1 ]; then export ME_GEMM_8P_192=448 ME_GEMM_192_MINK=8 ME_GEMM_GEGLU_MIN=640 ME_GEMM_8P_128=0; else unset ME_GEMM_8P_192 ME_GEMM_192_MINK ME_GEMM_GEGLU_MIN ME_GEMM_8P_128; fi
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('old thresholds' if $old else 'new thresholds', d['ms_per_step'], 'ms/step', {k:v['ms_per_step'] for k,v in d['kernel_families'].items()}, 'launches', d['launch_plan']['launches'])"
  done
done
