#!/bin/bash
# everything the round's records need, on one box: all GPU tests, rocprofv3 passes of bench.py (kernel trace + SQ counters + HBM
# traffic), the default bench line, the 2-rank rehearsal of the N>1 path on one GPU (gloo, host-staged: NOT an RCCL measurement),
# the file-based predictor's rate.  usage: tools/gpu_full.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-full}
mkdir -p gpurun_out/$T
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -3 gpurun_out/$T/t_all.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh $T stats sq fetch write > gpurun_out/$T/profile.log 2>&1; echo "profile rc=$?"
cat gpurun_out/$T/pmc_sq_summary.txt
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/$T/bench.json; echo
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --steps 8 --warmup 2 > gpurun_out/$T/bench_2rank_gloo.json 2> gpurun_out/$T/bench_2rank.err; echo "2-rank rehearsal rc=$?"; head -c 300 gpurun_out/$T/bench_2rank_gloo.json; echo
timeout -k 10 300 python tools/file_predictor_bench.py 192 > gpurun_out/$T/file_predictor.txt 2>&1; echo "file predictor rc=$?"; cat gpurun_out/$T/file_predictor.txt | tail -4
timeout -k 10 300 python tools/e2e_bench.py 384 4 > gpurun_out/$T/e2e.txt 2>&1; echo "e2e rc=$?"; tail -3 gpurun_out/$T/e2e.txt
