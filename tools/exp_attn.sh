#!/bin/bash
# attention variants / ablations on the L0 [prev|cur] launch; prints ms and the effective shader clock (GRBM_GUI_ACTIVE / time is not
# available without rocprof, so only time here)
mkdir -p gpurun_out/r2b
for v in "$@"; do
  echo -n "ME_ATTN_VARIANT=$v  " >> gpurun_out/r2b/attn.txt
  ME_ATTN_VARIANT=$v python tools/kbench.py attn1 2>/dev/null | tail -1 >> gpurun_out/r2b/attn.txt
done
cat gpurun_out/r2b/attn.txt
