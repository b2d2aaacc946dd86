#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-full}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -3 gpurun_out/$T/t_all.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh $T stats sq fetch write > gpurun_out/$T/profile.log 2>&1; echo "profile rc=$?"
cat gpurun_out/$T/pmc_sq_summary.txt
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/$T/bench.json; echo
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --steps 6 --warmup 1 --map 256 > gpurun_out/$T/bench_2rank_gloo.json 2> gpurun_out/$T/bench_2rank.err; echo "2-rank rehearsal rc=$?"; head -c 300 gpurun_out/$T/bench_2rank_gloo.json; echo
