#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02d/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc" | tee gpurun_out/r02d/status.txt; tail -15 gpurun_out/r02d/t_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02d/bench.log 2>gpurun_out/r02d/bench.err; echo "bench rc=$?"; head -c 400 gpurun_out/r02d/bench.log; echo
bash tools/profile.sh r02d stats
