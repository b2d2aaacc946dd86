#!/bin/bash
# round 5 experiment: non-temporal stores for the depthwise conv's output (tools/exp/libmica_dwnt.so = the library with -DMICA_DW_NT)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/dwnt
for lib in mica_amd/lib/libmica_hip.so tools/exp/libmica_dwnt.so mica_amd/lib/libmica_hip.so tools/exp/libmica_dwnt.so; do
  MICA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/dwnt/b.json 2> gpurun_out/dwnt/b.err
  python - <<PY
import json
d=json.load(open("gpurun_out/dwnt/b.json")); h=d["hbm_conv3d"]
print("$lib: %.2f sub-grids/s; depthwise %.1f GB/s (frac %.3f, %.3f ms)" % (d["value"], h["achieved"], h["frac"], h["avg_launch_ms"]))
PY
done
