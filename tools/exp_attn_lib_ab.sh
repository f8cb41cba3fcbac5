#!/bin/bash
# A/B of an experiment build of attn.hip (ME_LIB) against the shipped library on the level-0 / level-1 attention launches, alternating processes on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
lib=$1
for i in 1 2 3; do
  for l in "" $lib; do
    echo "== ${l:-shipped}"; ME_LIB=$l python tools/kbench.py attn 2>/dev/null | grep "^L0 prev|cur   \|^L0 edited\|^L0 self\|^L1"
  done
done
