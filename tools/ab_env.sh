#!/bin/bash
# A/B of an environment switch of the library on the default bench, alternating on one box (MICA_F43 = conv variant, MICA_RAW_CBLK,
# MICA_STEM_MFMA, MICA_TRUNK_PER_RUN ...; two BUILDS are compared with tools/exp/bench_libs.sh).
# usage: tools/ab_env.sh <tag> <VAR> "<v1 v2 v1 v2>" [bench args]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:?tag}; VAR=${2:?variable}; VALS=${3:?values}; shift 3
mkdir -p gpurun_out/$T
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map --af-coverage 0 "$@" > gpurun_out/$T/bench_$v.json 2> gpurun_out/$T/bench_$v.err || { tail -3 gpurun_out/$T/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/$T/bench_$v.json")); r = d["roofline"]; h = d["hbm_conv3d"]
print("$VAR=$v: %.2f sub-grids/s %.2f ms/step; conv43<128> %.3f ms x%d; wino16 %.3f ms x%d; 3x3x3 total %.2f ms; depthwise %.1f GB/s" % (
    d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_per_batch"], r["conv_wino16"]["avg_launch_ms"], r["conv_wino16"]["launches_per_batch"], r["all_3x3x3_convs"]["ms_per_batch"], h["achieved"]))
PY
done
