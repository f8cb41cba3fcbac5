#!/bin/bash
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['roofline']); print({k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
