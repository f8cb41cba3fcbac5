#!/bin/bash
# round-3 GPU session 4: the 8-phase GEMM kernel -- parity / race screen, then old-vs-new timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "8phase or gemm" > gpurun_out/s4_k_gemm.log 2>&1
echo "pytest gemm exit $?" > gpurun_out/s4_summary.txt
timeout 600 python tools/kbench.py gemm8p > gpurun_out/s4_kbench_8p.log 2>&1
echo "kbench exit $?" >> gpurun_out/s4_summary.txt
cat gpurun_out/s4_summary.txt; tail -5 gpurun_out/s4_k_gemm.log; cat gpurun_out/s4_kbench_8p.log
