#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02b
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02b/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc" | tee gpurun_out/r02b/status.txt; tail -3 gpurun_out/r02b/t_all.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh r02b stats sq fetch write
