#!/bin/bash
# quick GPU check of a kernel change: model + ops parity, then kernel-trace stats of the bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-quick}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/$T/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/$T/t.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh $T stats
grep -o '"value": [0-9.]*' gpurun_out/$T/stats.log | head -1
