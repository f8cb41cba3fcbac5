#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "epilogue_term" > gpurun_out/t_epi.log 2>&1
tail -5 gpurun_out/t_epi.log
