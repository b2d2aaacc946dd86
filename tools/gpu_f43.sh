#!/bin/bash
# F(4,3) bring-up: its op tests first (short leash), then the whole GPU suite, then the bench A/B (MICA_F43=0 | 1) on the same box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-f43}
mkdir -p gpurun_out/$T
timeout -k 10 240 python -m pytest tests/test_gpu_ops.py -x -q -s -k "f43" > gpurun_out/$T/t_ops.log 2>&1; rc=$?; echo "f43 op tests rc=$rc"; tail -15 gpurun_out/$T/t_ops.log
[ $rc -eq 0 ] || exit $rc
if [ "$2" != "noall" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > gpurun_out/$T/t_all.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 gpurun_out/$T/t_all.log
[ $rc -eq 0 ] || exit $rc
fi
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map"
MICA_F43=0 timeout -k 10 300 python bench.py $Q > gpurun_out/$T/bench_f23.json 2> gpurun_out/$T/bench_f23.err; echo "bench F23 rc=$?"; head -c 200 gpurun_out/$T/bench_f23.json; echo
MICA_F43=1 timeout -k 10 300 python bench.py $Q > gpurun_out/$T/bench_f43.json 2> gpurun_out/$T/bench_f43.err; echo "bench F43 rc=$?"; head -c 200 gpurun_out/$T/bench_f43.json; echo
MICA_F43=0 timeout -k 10 300 python bench.py $Q > gpurun_out/$T/bench_f23_b.json 2> gpurun_out/$T/bench_f23_b.err; echo "bench F23 (again) rc=$?"; head -c 200 gpurun_out/$T/bench_f23_b.json; echo
