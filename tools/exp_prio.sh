#!/bin/bash
# GPU-box A/B: the step's main stream at high HIP priority (the side stream keeps the default), same session.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
out=gpurun_out/prio_ab.txt
: > $out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')" >> $out 2>&1
for v in ${PRIO_SET:-0 -1 0 -1}; do
  timeout 400 python bench.py --${PRIO_WHICH:-main}-priority $v --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 > gpurun_out/prio_tmp.json
  python - "$v" <<'PY' >> $out
import json, sys
try:
    d = json.load(open("gpurun_out/prio_tmp.json"))
    print(f"priority {sys.argv[1]:>2s}: ms/step {d['ms_per_step']:8.2f} host {d['host_enqueue_ms_per_step']:5.2f}")
except Exception as e:
    print("FAILED", sys.argv[1], e, open("gpurun_out/prio_tmp.json").read()[-300:])
PY
done
[ -z "$PRIO_WHICH" ] && timeout 300 python bench.py --main-priority -1 --eager --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('main priority -1, --eager: ms/step', d['ms_per_step'])" >> $out 2>&1
cat $out
