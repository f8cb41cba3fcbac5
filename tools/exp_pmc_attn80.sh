#!/bin/bash
# Round 6: SQ counters of the level-1 [prev | cur] attention launch (dh = 80) at 32 queries per wave (default) and at 16 (ME_ATTN_80_QT2=0): counter-only passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/r06_attn_dh80_sq_counters.txt
: > $out
for v in 1 0; do
  echo "== ME_ATTN_80_QT2=$v  ($([ $v = 1 ] && echo 'attn2_kernel<80,2,8,fold>: 8 waves x 32 queries, 128-key stages' || echo 'attn2_kernel<80,1,8,fold>: 8 waves x 16 queries, 64-key stages'))" >> $out
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    rm -rf /tmp/pmc_out
    ( cd /tmp && ME_ATTN_80_QT2=$v timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/attn_one80.py > /dev/null 2>&1 )
    python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
  done
done
cat $out
