#!/bin/bash
# bench.py with the stem on the matrix cores (default) and on the f32 VALU kernel (MICA_STEM_MFMA=0), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/stem
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map"
for m in 1 0 1 0; do
  MICA_STEM_MFMA=$m timeout -k 10 300 python bench.py $Q > gpurun_out/stem/bench_$m.json 2> gpurun_out/stem/bench_$m.err; rc=$?
  echo "MICA_STEM_MFMA=$m rc=$rc $(python -c "import json;j=json.load(open('gpurun_out/stem/bench_$m.json'));print(round(j['value'],2), round(j['ms_per_step'],2))" 2>/dev/null)"
done
