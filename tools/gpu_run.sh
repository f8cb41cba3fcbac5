#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -3
timeout 300 python tools/kbench.py gemm 2>&1 | grep "cond"
