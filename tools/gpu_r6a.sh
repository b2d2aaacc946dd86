#!/bin/bash
# round 6, first GPU call: all GPU tests (no -x: every failure is information), then the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r6a}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests -q -m gpu -s > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "passed|failed|FAILED|Error" gpurun_out/$T/t_all.log | tail -30
[ $rc -le 1 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; rc2=$?; echo "bench rc=$rc2"; head -c 600 gpurun_out/$T/bench.json; echo
exit $((rc + rc2))
