#!/bin/bash
# bench.py under the three conv variants (MICA_F43 = 1 shipped: encoder.2; 2: plus encoder.1's transition; 0: F(2,3) everywhere), same box, 1 run first and last
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/f43modes
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map"
for m in 1 2 0 2 1; do
  MICA_F43=$m timeout -k 10 300 python bench.py $Q > gpurun_out/f43modes/bench_$m.json 2> gpurun_out/f43modes/bench_$m.err; rc=$?
  echo "MICA_F43=$m rc=$rc $(python -c "import json;j=json.load(open('gpurun_out/f43modes/bench_$m.json'));r=j['roofline'];print(round(j['value'],2), round(j['ms_per_step'],2), 'wino43', r['launches_per_batch'], round(r['avg_launch_ms'],3), 'wino16', r['conv_wino16']['launches_per_batch'], round(r['conv_wino16']['avg_launch_ms'],3))" 2>/dev/null)"
done
