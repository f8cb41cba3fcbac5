#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "six_consecutive" 2>&1 | tail -5
grep six gpurun_out/parity.jsonl
