#!/bin/bash
cd /root/repo
timeout 300 python tools/kbench.py gemms 2>&1 | tail -11
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k gemm 2>&1 | tail -2
