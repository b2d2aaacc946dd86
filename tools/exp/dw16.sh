#!/bin/bash
# round 5 experiment: the depthwise conv's 32-channel variant with 16 waves per workgroup (<2, 8, 8>: two outputs per thread, 128 VGPRs,
# four waves per SIMD) against the shipped 8 waves of four outputs (<4, 8, 4>: 218 VGPRs, two waves per SIMD).
# `make -C mica_amd/csrc exp_dw16` builds tools/exp/libmica_dw16.so.  Measured: 0.559-0.567 against 0.568-0.574 of 8 TB/s - more waves do
# not help (LDS reads per output go up by 56 %: 4 window rows per 2 outputs instead of 6 per 4, the 27 weight reads per thread stay).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/dw16
MICA_HIP_LIB=$PWD/tools/exp/libmica_dw16.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "depthwise" > gpurun_out/dw16/tests.log 2>&1; rc=$?; tail -2 gpurun_out/dw16/tests.log
[ $rc -eq 0 ] || exit $rc
for lib in mica_amd/lib/libmica_hip.so tools/exp/libmica_dw16.so mica_amd/lib/libmica_hip.so tools/exp/libmica_dw16.so; do
  MICA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/dw16/b.json 2> gpurun_out/dw16/b.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/dw16/b.json")); h=d["hbm_conv3d"]
print("$lib: %.2f sub-grids/s; depthwise %.1f GB/s (frac %.3f, %.3f ms)" % (d["value"], h["achieved"], h["frac"], h["avg_launch_ms"]))
PY
done
