#!/bin/bash
# round 6: new multi-rank product entry test, rank start-up record, file-contract bench at 256^3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r6b}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_volume.py tests/test_gpu_model.py::test_mixed_branch_batch_runs_the_network_once -x -q -m gpu -s > gpurun_out/$T/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/$T/t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/rank_startup.py 256 2 gloo > gpurun_out/$T/rank_startup.txt 2> gpurun_out/$T/rank_startup.err; rc=$?; echo "rank_startup rc=$rc"; cat gpurun_out/$T/rank_startup.txt
[ $rc -eq 0 ] || { tail -20 gpurun_out/$T/rank_startup.err; exit $rc; }
timeout -k 10 300 python tools/file_predictor_bench.py 256 > gpurun_out/$T/file_predictor.txt 2> gpurun_out/$T/file_predictor.err; rc=$?; echo "file_predictor rc=$rc"; cat gpurun_out/$T/file_predictor.txt
[ $rc -eq 0 ] || tail -20 gpurun_out/$T/file_predictor.err
exit $rc
