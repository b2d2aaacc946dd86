#!/bin/bash
# round 5: blocked raw layout around the depthwise conv: model + op tests, then A/B MICA_RAW_CBLK = 0 | 32 on the default bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r5d}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/$T/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/$T/t.log
[ $rc -eq 0 ] || exit $rc
for v in 0 32 0 32; do
  MICA_RAW_CBLK=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/$T/bench_cb$v.json 2> gpurun_out/$T/bench_cb$v.err; rc=$?
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_cb$v.json"))
r=d["roofline"]; h=d["hbm_conv3d"]
print("raw cblk $v: %.2f sub-grids/s %.2f ms/step; depthwise %.1f GB/s (frac %.3f, %.3f ms); conv43<128> %.3f ms x%d; wino16 %.3f ms x%d; 3x3x3 total %.2f ms" % (d["value"], d["ms_per_step"], h["achieved"], h["frac"], h["avg_launch_ms"], r["avg_launch_ms"], r["launches_per_batch"], r["conv_wino16"]["avg_launch_ms"], r["conv_wino16"]["launches_per_batch"], r["all_3x3x3_convs"]["ms_per_batch"]))
PY
done
