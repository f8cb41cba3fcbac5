#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > gpurun_out/t_attn.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_attn.log
tail -3 gpurun_out/t_attn.log
for v in 0 2 0; do
ME_ATTN_VARIANT=$v timeout 300 python tools/kbench.py attn 2>&1 | grep "L0 \|L1 \|L2 "
done
