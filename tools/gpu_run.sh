#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/k_all.log 2>&1; echo "pytest kernels exit $?"
tail -3 gpurun_out/k_all.log
timeout 600 python tools/kbench.py gemm > gpurun_out/kb2.log 2>&1; cat gpurun_out/kb2.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench3.log 2>&1; tail -1 gpurun_out/bench3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']); [print(k,v) for k,v in d['kernel_families'].items()]"
