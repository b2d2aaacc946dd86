#!/bin/bash
# copy the outputs of tools/gpu_r5_full.sh <tag> from gpurun_out/<tag>/ into the tracked round-5 records under profiles/
# usage: tools/copy_records.sh <tag>
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1
cp $T/bench.json profiles/r05_bench_default.json
cp $T/strong_256.json profiles/r05_bench_strong_1rank_256.json
cp $T/strong_512.json profiles/r05_bench_strong_1rank_512.json
cp $T/strong_256_rccl.json profiles/r05_bench_strong_rccl_allgather_1rank_256.json
cp $T/strong_256_rccl_root.json profiles/r05_bench_strong_rccl_gather_1rank_256.json
grep -v amdgpu.ids $T/e2e.txt > profiles/r05_e2e_streamed.txt
grep -v amdgpu.ids $T/file_predictor.txt > profiles/r05_file_predictor.txt
cp $T/kernel_stats.txt profiles/r05_kernel_stats.txt
cp $T/stats/r_kernel_stats.csv profiles/r05_kernel_stats.csv
cp $T/pmc_sq_summary.txt profiles/r05_pmc_sq_summary.txt
cp $T/pmc_traffic.json profiles/r05_pmc_traffic.json
cp $T/parity_margins.txt profiles/r05_parity_margins.txt
tail -1 $T/t_all.log
cat $T/source_hash.txt
