#!/bin/bash
# round-5 working call: all GPU tests, then the default bench line.  usage: tools/gpu_r5.sh <tag> [pytest -k expression]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r5}
K=${2:-}
mkdir -p gpurun_out/$T
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu -s -k "$K" > gpurun_out/$T/t_sel.log 2>&1; rc=$?; echo "selected gpu tests rc=$rc"; tail -5 gpurun_out/$T/t_sel.log
else
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -5 gpurun_out/$T/t_all.log
fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; rc=$?; echo "bench rc=$rc"; head -c 400 gpurun_out/$T/bench.json; echo
exit $rc
