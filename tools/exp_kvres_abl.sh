#!/bin/bash
# Round 6 experiments on the resident-key cross-attention kernel: ablation / occupancy builds (tools/build_abl.sh ... attn.hip -DME_KVRES_ABL=n | -DME_KV40_QT=q -DME_KV40_MINW=w)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" "$@"; do
  echo "== ${lib:-shipped}"; ME_LIB=$lib python tools/kbench.py attnkvres 2>/dev/null | grep "^L"
done
