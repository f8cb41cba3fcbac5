#!/bin/bash
# SQ counters of the big-tile GEMM launches (tools/kbench.py gemmabl): one-barrier kernel (ME_GEMM_8P=0) vs the 8-phase kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/pmc_gemm.txt
rm -f $out
for v in 0 1; do
  for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"; do   # (more than ~4 counters per pass leave the DB without its views)
    rm -rf /tmp/pmc_out
    ( cd /tmp && ME_GEMM_8P=$v timeout 200 rocprofv3 --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/kbench.py gemmabl > /dev/null 2>&1 )
    echo "== ME_GEMM_8P=$v" >> $out
    if [ $v = 0 ]; then key="gemm_kernel<256, 320"; else key="gemm8p_kernel<256, 320, false"; fi
    python $R/tools/pmc_sq.py "$key" $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
  done
done
cat $out
