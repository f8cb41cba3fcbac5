#!/bin/bash
# SQ counters of the L0 [prev|cur] attention launch for a few kernel variants
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2c
for v in "$@"; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    rm -rf /tmp/pmc_out
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/kbench.py attn1 > /dev/null 2>&1 )
    echo "== variant $v" >> $R/gpurun_out/r2c/pmc_attn.txt
    python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $R/gpurun_out/r2c/pmc_attn.txt 2>&1
  done
done
cat $R/gpurun_out/r2c/pmc_attn.txt
