#!/bin/bash
# round 5 power experiment: do the matrix cores draw less power when the lo halves of the split operands carry fewer mantissa bits?
# tools/exp/libmica_lomaskN.so = the library with -DMICA_EXP_LOMASK=N (N low bits of every lo half cleared, activations and weights).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/lomask
for lib in mica_amd/lib/libmica_hip.so tools/exp/libmica_lomask4.so tools/exp/libmica_lomask7.so mica_amd/lib/libmica_hip.so tools/exp/libmica_lomask4.so tools/exp/libmica_lomask7.so; do
  MICA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/lomask/b.json 2> gpurun_out/lomask/b.err
  python - <<PY
import json
d=json.load(open("gpurun_out/lomask/b.json")); r=d["roofline"]
print("$lib: %.2f sub-grids/s; conv43<128> %.3f ms x%d; wino16 %.3f ms x%d; all 3x3x3 %.2f ms" % (d["value"], r["avg_launch_ms"], r["launches_per_batch"], r["conv_wino16"]["avg_launch_ms"], r["conv_wino16"]["launches_per_batch"], r["all_3x3x3_convs"]["ms_per_batch"]))
PY
done
