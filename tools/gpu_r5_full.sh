#!/bin/bash
# round-5 records on one box: all GPU tests, rocprofv3 passes of bench.py (kernel trace + SQ counters + HBM traffic, stamped with the
# library source hash), the default bench line, the parity margins table, the getData + nnPred chain, strong-mode runs.
# usage: tools/gpu_r5_full.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r05}
mkdir -p gpurun_out/$T
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -3 gpurun_out/$T/t_all.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile.sh $T stats sq fetch write > gpurun_out/$T/profile.log 2>&1; echo "profile rc=$?"
cat gpurun_out/$T/pmc_sq_summary.txt | head -24
mkdir -p profiles_new && cp gpurun_out/$T/pmc_traffic.json profiles/r05_pmc_traffic.json 2>/dev/null; cp gpurun_out/$T/pmc_sq_summary.txt profiles/r05_pmc_sq_summary.txt 2>/dev/null; rmdir profiles_new
timeout -k 10 400 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/$T/bench.json; echo
timeout -k 10 600 python tools/parity_margins.py 0,1,3 > gpurun_out/$T/parity_margins.txt 2> gpurun_out/$T/parity_margins.err; echo "margins rc=$?"
timeout -k 10 600 python tools/file_predictor_bench.py 256 > gpurun_out/$T/file_predictor.txt 2>&1; echo "file predictor rc=$?"; tail -7 gpurun_out/$T/file_predictor.txt
for n in 256 512; do
  timeout -k 10 300 python bench.py --strong --map $n --grid 48 --pad 8 > gpurun_out/$T/strong_$n.json 2> gpurun_out/$T/strong_$n.err; echo "strong $n rc=$?"; head -c 120 gpurun_out/$T/strong_$n.json; echo
done
timeout -k 10 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange > gpurun_out/$T/strong_256_rccl.json 2> gpurun_out/$T/strong_256_rccl.err; echo "strong rccl rc=$?"; head -c 120 gpurun_out/$T/strong_256_rccl.json; echo
timeout -k 10 300 python bench.py --strong --map 256 --grid 48 --pad 8 --backend nccl --force-exchange --gather-to-root > gpurun_out/$T/strong_256_rccl_root.json 2> gpurun_out/$T/strong_256_rccl_root.err; echo "strong rccl root rc=$?"; head -c 120 gpurun_out/$T/strong_256_rccl_root.json; echo
timeout -k 10 300 python tools/e2e_bench.py 384 4 > gpurun_out/$T/e2e.txt 2>&1; echo "e2e rc=$?"; tail -3 gpurun_out/$T/e2e.txt
