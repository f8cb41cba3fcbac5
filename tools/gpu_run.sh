#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
ME_GEMM_PP=1 ME_GEMM_BIG_MIN=1 ME_CONV_HALO=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t_pp.log 2>&1
tail -4 gpurun_out/t_pp.log
ME_GEMM_PP=1 ME_CONV_HALO=0 timeout 300 python tools/kbench.py gemm > gpurun_out/kb_pp.log 2>&1
ME_CONV_HALO=0 timeout 300 python tools/kbench.py gemm > gpurun_out/kb_nopp.log 2>&1
