#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > gpurun_out/t_attn.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_attn.log
timeout 300 python tools/kbench.py attn > gpurun_out/kb_attn2.log 2>&1
tail -5 gpurun_out/t_attn.log; cat gpurun_out/kb_attn2.log
