#!/bin/bash
# Round 5: block order of the multi-segment attention launches (heads slowest vs items slowest): time (same process, alternating), HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, per the MI355X guide; 2 x FETCH_SIZE + WRITE_SIZE) and the SQ counters of the dh = 40 kernel as it is now.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
out=$R/gpurun_out/attn_order.txt
mkdir -p $R/gpurun_out
: > $out
( cd $R && timeout 300 python tools/kbench.py attnorder ) >> $out 2>&1
for kind in pc ed; do
  for order in 0 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_out
      ( cd /tmp && ME_ATTN_ORDER=$order timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_out -o p -- python $R/tools/attn_one.py $kind 4 > /dev/null 2>&1 )
      echo "== L0 $kind, ME_ATTN_ORDER=$order (1 = heads slowest), $c [KB]" >> $out
      python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $out 2>&1
    done
  done
done
sq=$R/gpurun_out/attn_dh40_sq_counters.txt
: > $sq
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pmc_out
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/attn_one.py pc 4 > /dev/null 2>&1 )
  python $R/tools/pmc_sq.py attn2_kernel $(find /tmp/pmc_out -name "*.db" | head -1) >> $sq 2>&1
done
cat $out $sq
