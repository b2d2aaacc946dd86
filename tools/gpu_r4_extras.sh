#!/bin/bash
# round-4 records beside the tests: the RCCL branch on one rank, the strong-scaling mode (1 rank and the 2-rank gloo rehearsal on
# one GPU), the file-based predictor and the host-to-host streamed rate.  usage: tools/gpu_r4_extras.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r4x}
mkdir -p gpurun_out/$T
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map"
timeout -k 10 300 python bench.py --gpus 1 --backend nccl --force-exchange --steps 16 --warmup 3 $Q > gpurun_out/$T/bench_rccl_1rank.json 2> gpurun_out/$T/bench_rccl_1rank.err; rc=$?; echo "rccl 1-rank rc=$rc"; head -c 600 gpurun_out/$T/bench_rccl_1rank.json; echo; [ $rc -eq 0 ] || tail -5 gpurun_out/$T/bench_rccl_1rank.err
timeout -k 10 300 python bench.py --gpus 1 --strong --map 256 > gpurun_out/$T/bench_strong_1rank_256.json 2> gpurun_out/$T/bench_strong_1rank_256.err; rc=$?; echo "strong 1-rank 256 rc=$rc"; head -c 900 gpurun_out/$T/bench_strong_1rank_256.json; echo; [ $rc -eq 0 ] || tail -5 gpurun_out/$T/bench_strong_1rank_256.err
timeout -k 10 300 python bench.py --gpus 1 --strong --map 256 --backend nccl --force-exchange > gpurun_out/$T/bench_strong_rccl_1rank_256.json 2> gpurun_out/$T/bench_strong_rccl_1rank_256.err; rc=$?; echo "strong rccl 1-rank 256 rc=$rc"; head -c 900 gpurun_out/$T/bench_strong_rccl_1rank_256.json; echo; [ $rc -eq 0 ] || tail -5 gpurun_out/$T/bench_strong_rccl_1rank_256.err
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --strong --map 256 > gpurun_out/$T/bench_strong_2rank_gloo_256.json 2> gpurun_out/$T/bench_strong_2rank_gloo_256.err; rc=$?; echo "strong 2-rank gloo rc=$rc"; head -c 900 gpurun_out/$T/bench_strong_2rank_gloo_256.json; echo; [ $rc -eq 0 ] || tail -5 gpurun_out/$T/bench_strong_2rank_gloo_256.err
timeout -k 10 300 python tools/file_predictor_bench.py 256 > gpurun_out/$T/file_predictor.txt 2>&1; echo "file predictor rc=$?"; tail -4 gpurun_out/$T/file_predictor.txt
timeout -k 10 300 python tools/e2e_bench.py 384 4 > gpurun_out/$T/e2e.txt 2>&1; echo "e2e rc=$?"; tail -3 gpurun_out/$T/e2e.txt
