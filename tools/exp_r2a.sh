#!/bin/bash
# round-2 experiment batch A: instruction rates, attention / GEMM ablations (diagnostics only)
mkdir -p gpurun_out/r2a
tools/_bin/ubench > gpurun_out/r2a/ubench.txt 2>&1
for v in 0 101 102 104 108 103 112 117 119 1; do
  echo "== ME_ATTN_VARIANT=$v" >> gpurun_out/r2a/attn_abl.txt
  ME_ATTN_VARIANT=$v python tools/kbench.py attn1 2>/dev/null | tail -1 >> gpurun_out/r2a/attn_abl.txt
done
for v in 0 1 2 4 5; do
  echo "== ME_GEMM_ABL=$v" >> gpurun_out/r2a/gemm_abl.txt
  ME_GEMM_ABL=$v python tools/kbench.py gemmabl 2>/dev/null >> gpurun_out/r2a/gemm_abl.txt
done
cat gpurun_out/r2a/ubench.txt gpurun_out/r2a/attn_abl.txt gpurun_out/r2a/gemm_abl.txt
