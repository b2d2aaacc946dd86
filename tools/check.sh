#!/bin/bash
# Working call after a change (one gpurun call): selected GPU tests, then a short bench line.
# usage: tools/check.sh <tag> [pytest selection, default: the op and model tests] ; CHECK_BENCH=0 skips the bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-chk}; shift
SEL=${*:-tests/test_gpu_ops.py tests/test_gpu_model.py}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest $SEL -x -q -m gpu -s > gpurun_out/$T/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/$T/t.log
[ $rc -eq 0 ] || exit $rc
[ "${CHECK_BENCH:-1}" = 1 ] || exit 0
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; rc=$?
python - <<PY
import json
d = json.load(open("gpurun_out/$T/bench.json")); r = d["roofline"]; h = d["hbm_conv3d"]
print("value %.2f sub-grids/s (%.2f ms/step); conv43<128> %.3f ms x%d (frac %.3f); conv43<64> %.3f ms x%d; wino16 %.3f ms x%d; depthwise %.1f GB/s (frac %.3f)" % (
    d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_per_batch"], r["frac"], r["conv_wino43_64"]["avg_launch_ms"], r["conv_wino43_64"]["launches_per_batch"],
    r["conv_wino16"]["avg_launch_ms"], r["conv_wino16"]["launches_per_batch"], h["achieved"], h["frac"]))
PY
exit $rc
