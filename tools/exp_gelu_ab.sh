#!/bin/bash
# Round 6: packed-fp32 GELU in the GEGLU epilogues vs the scalar form (tools/build_abl.sh gelu_scalar gemm.hip -DME_GELU_PK=0), alternating on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for lib in "" tools/_bin/libmotioned_gelu_scalar.so; do
    echo "== ${lib:-packed (shipped)}"
    ME_LIB=$lib python tools/kbench.py gemm 2>/dev/null | grep -i "geglu"
  done
done
for i in 1 2; do
  for lib in "" tools/_bin/libmotioned_gelu_scalar.so; do
    ME_LIB=$lib python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-packed}', d['ms_per_step'], 'ms/step', {k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
  done
done
