#!/bin/bash
cd /root/repo
for fl in "" "--no-overlap" "" "--no-overlap"; do
echo "flags: $fl"
timeout 600 python bench.py --no-cpu-baseline $fl 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "denoise_step or six" 2>&1 | tail -2
