#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
grep -n "passed\|failed" gpurun_out/t_all.log | tail -2
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
