#!/bin/bash
cd /root/repo
for v in 1 0 1 0; do
echo "early $v"
ME_EARLY=$v timeout 600 python bench.py --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
