#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k attention > gpurun_out/k_all.log 2>&1; echo "pytest kernels exit $?"
tail -3 gpurun_out/k_all.log
timeout 600 python tools/kbench.py attn > gpurun_out/kb2.log 2>&1; cat gpurun_out/kb2.log
