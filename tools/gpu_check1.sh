#!/bin/bash
# first GPU pass: kernel parity, model parity, short bench. Everything logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep "Model name" >> gpurun_out/device.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/summary.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "model exit $?" >> gpurun_out/summary.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
tail -5 gpurun_out/kernels.log; tail -5 gpurun_out/model.log; tail -3 gpurun_out/bench_c3.log; cat gpurun_out/summary.txt
