#!/bin/bash
# round-2 GPU check: new config tests, whole GPU suite, bench on the 512^3 metric config, kernel-trace profile
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q -m gpu -s > gpurun_out/r02/t_new.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/r02/status.txt
tail -5 gpurun_out/r02/t_new.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_default.log 2> gpurun_out/r02/bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r02/status.txt
tail -c 3000 gpurun_out/r02/bench_default.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r02/prof -o bench -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-tiling > gpurun_out/r02/prof_bench.log 2>&1; echo "prof rc=$?" | tee -a gpurun_out/r02/status.txt
find gpurun_out/r02/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} python tools/prof_summary.py {} 64 24 > gpurun_out/r02/kernel_stats.txt 2>&1
cat gpurun_out/r02/kernel_stats.txt
find gpurun_out/r02/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +1M -delete
