#!/bin/bash
cd /root/repo
timeout 900 python bench.py --shapes --no-cpu-baseline 2> gpurun_out/shapes_flags.txt > /dev/null
grep "^\[shape\]" gpurun_out/shapes_flags.txt | head -40
