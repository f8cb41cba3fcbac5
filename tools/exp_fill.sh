#!/bin/bash
# GPU-box run of the two K = 320 tile experiments: the operand fill alone (tools/ubench_fill.hip) and the ring-buffered GEMM model
# (tools/ubench_ring_gemm.hip, self-checking), next to the library's numbers for the same shapes.  ~1 minute.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out tools/_bin
out=gpurun_out/fill_ring.txt
: > $out
for t in ubench_fill ubench_ring_gemm; do
  [ -x tools/_bin/$t ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o tools/_bin/$t tools/$t.hip >> $out 2>&1
  echo "== $t" >> $out
  timeout 120 tools/_bin/$t >> $out 2>&1
  echo "exit $?" >> $out
done
echo "== library (tools/kbench.py gemm): the same shapes through gemm8p_kernel" >> $out
timeout 200 python tools/kbench.py gemm 2>&1 | grep -E "^L0 (qkv|out|ff1|ff2)|^L1 (qkv|out)" >> $out
cat $out
