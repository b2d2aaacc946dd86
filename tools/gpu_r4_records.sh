#!/bin/bash
# round-4 records: rocprofv3 passes of bench.py (kernel trace + SQ counters + HBM traffic), then the default bench line.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r4rec}
mkdir -p gpurun_out/$T
bash tools/profile.sh $T stats sq fetch write > gpurun_out/$T/profile.log 2>&1; echo "profile rc=$?"
cat gpurun_out/$T/pmc_sq_summary.txt | head -12
cp gpurun_out/$T/pmc_traffic.json profiles/r04_pmc_traffic.json 2>/dev/null     # bench.py reads the newest committed traffic file
timeout -k 10 500 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/$T/bench.json; echo
