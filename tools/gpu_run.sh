#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/all.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/all.log
timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --shapes > gpurun_out/bench3.log 2>&1; grep "^{" gpurun_out/bench3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']); [print(k,v) for k,v in d['kernel_families'].items()]"; grep shape gpurun_out/bench3.log | head -12
