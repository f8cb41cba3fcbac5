#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -3 gpurun_out/t_all.log
timeout 900 python bench.py --shapes --no-cpu-baseline > gpurun_out/bench_shapes3.log 2>&1
grep "^\[shape\]" gpurun_out/bench_shapes3.log | head -12
tail -1 gpurun_out/bench_shapes3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['steps'], d['warmup']); print({k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
