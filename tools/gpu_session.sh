#!/bin/bash
# One gpurun session: new-kernel tests in separate processes (a device fault kills only its group), the rest of the GPU suite, the bench lines.
#   bash tools/gpu_session.sh <tag> [quick]
tag=${1:-s}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
S=gpurun_out/${tag}_summary.txt
rm -f gpurun_out/parity.jsonl $S
run() {   # name, timeout, command...
  local name=$1 to=$2; shift 2
  local t0=$(date +%s)
  timeout $to "$@" > gpurun_out/${tag}_${name}.log 2>&1
  echo "$name rc $? ($(( $(date +%s) - t0 )) s)" >> $S
  tail -3 gpurun_out/${tag}_${name}.log | cut -c1-300 >> $S
}
PT="python -m pytest -q -p no:cacheprovider -m gpu"
NEWK="attention_bwd or gemm_dw or layernorm_param or grad_acc or sumsq or adamw or cast_kernels or conv_small_wide or gemm_dx or zero_stuffed or shared_residual or shared_query"
run k_attn_bwd 400 $PT tests/test_kernels_gpu.py -k "attention_bwd"
run k_train 400 $PT tests/test_kernels_gpu.py -k "gemm_dw or layernorm_param or grad_acc or sumsq or adamw or cast_kernels or conv_small_wide"
run k_gather 400 $PT tests/test_kernels_gpu.py -k "gemm_dx or zero_stuffed or shared_residual or shared_query"
run k_rest 900 $PT tests/test_kernels_gpu.py -k "not ($NEWK)"
run model 1500 $PT tests/test_model_gpu.py
cp gpurun_out/parity.jsonl gpurun_out/${tag}_parity.jsonl 2>/dev/null
run bench 900 python bench.py
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_c3.json
if [ "$2" != "quick" ]; then
  run nulltext 600 python bench.py --null-text --steps 3 --warmup 1
  run shapes 400 python bench.py --shapes --steps 2 --warmup 1 --no-cpu-baseline
  run kbench 400 python tools/kbench.py gemm attn misc
  run bench_frames_graph 600 python bench.py --parallel frames --graph --steps 4 --warmup 2 --no-cpu-baseline --no-profile
  run bench_frames_eager 600 python bench.py --parallel frames --steps 4 --warmup 2 --no-cpu-baseline --no-profile
fi
cat $S
