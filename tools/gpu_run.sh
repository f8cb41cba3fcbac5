#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -3 gpurun_out/t_all.log
ME_GEMM_BIG_MIN=1 ME_CONV_HALO=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t_pers.log 2>&1
tail -2 gpurun_out/t_pers.log
timeout 300 python tools/kbench.py gemmk 2>&1 | tail -11
timeout 300 python tools/kbench.py gemm > gpurun_out/kb_ep.log 2>&1
