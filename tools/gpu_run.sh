#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/all.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/all.log
