#!/bin/bash
# op tests of the F(4,3) kernel, then A/B of conv variants on the default bench.  usage: tools/gpu_r5k2.sh <tag> "<variants>"
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r5k2}
VARS=${2:-"1 3 1 3"}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f43 or conv3d" > gpurun_out/$T/t_ops.log 2>&1; rc=$?; echo "op tests rc=$rc"; tail -2 gpurun_out/$T/t_ops.log
[ $rc -eq 0 ] || exit $rc
for v in $VARS; do
  MICA_F43=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/$T/bench_v$v.json 2> gpurun_out/$T/bench_v$v.err; rc=$?
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_v$v.json"))
r=d["roofline"]
print("variant $v: %.2f sub-grids/s  %.2f ms/step; F(4,3) launches %d x %.3f ms, F(2,3) launches %d x %.3f ms" % (d["value"], d["ms_per_step"], r["launches_per_batch"], r["avg_launch_ms"], r["conv_wino16"]["launches_per_batch"], r["conv_wino16"]["avg_launch_ms"]))
PY
done
