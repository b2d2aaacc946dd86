#!/bin/bash
# all GPU tests, then the default bench line.  usage: tools/gpu_tests_bench.sh <tag> [pytest args]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-tb}; shift
mkdir -p gpurun_out/$T
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -s "$@" > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 gpurun_out/$T/t_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; rc=$?; echo "bench rc=$rc"; head -c 400 gpurun_out/$T/bench.json; echo
exit $rc
