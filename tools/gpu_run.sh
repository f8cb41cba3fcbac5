#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "build+smoke exit $?"; tail -1 gpurun_out/smoke.log
