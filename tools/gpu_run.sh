#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "inversion" 2>&1 | tail -4
grep -i "inver\|normal" gpurun_out/parity.jsonl
timeout 600 python bench.py --inversion --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['metric'], d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
