#!/bin/bash
cd /root/repo
for n in 2048 1024 512 256; do
echo "GN blocks $n"
ME_GN_BLOCKS=$n timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('main', d['ms_per_step'], d['kernel_families']['groupnorm'])"
ME_GN_BLOCKS=$n timeout 600 python bench.py --inversion --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('inv ', d['ms_per_step'], d['kernel_families']['groupnorm'])"
done
