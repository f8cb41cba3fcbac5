#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for fl in "" "--editors inactive" "--zero-tconv" "--frames 8" "--frames 16" "--frames 8 --latent 32"; do
echo "flags: $fl"
timeout 600 python bench.py --no-cpu-baseline $fl 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['achieved_tflops_whole_job'], {k:v['ms_per_step'] for k,v in d['kernel_families'].items()})"
done
