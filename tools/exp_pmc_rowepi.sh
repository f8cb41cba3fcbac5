#!/bin/bash
# Round 5: SQ counters of gemm8p_kernel<256,320,false> with the row-contiguous epilogue (ME_GEMM_ROWEPI=1, default) vs the direct epilogue (=0) over tools/kbench.py gemmabl
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/pmc_rowepi.txt
rm -f $out
for v in 1 0; do
  for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc_out
    ( cd /tmp && ME_GEMM_ROWEPI=$v timeout 200 rocprofv3 --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/kbench.py gemmabl > /dev/null 2>&1 )
    echo "== ME_GEMM_ROWEPI=$v" >> $out
    python $R/tools/pmc_sq.py "gemm8p_kernel<256, 320, false" $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
  done
done
cat $out
