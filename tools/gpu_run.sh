#!/bin/bash
cd /root/repo
for v in 0 1 2 0 1 2; do
echo "overlap $v"
ME_OVERLAP=$v timeout 600 python bench.py --no-cpu-baseline $( [ $v = 0 ] && echo --no-overlap ) 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
