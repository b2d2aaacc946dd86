#!/bin/bash
# bench.py with alternative builds of the library (MICA_HIP_LIB), same box, alternating with the shipped one.  usage: bench_libs.sh <tag> lib...
cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; shift
mkdir -p gpurun_out/$T
Q="--no-cpu-baseline --no-alt-tiling --no-whole-map"
for L in shipped "$@" shipped; do
  if [ "$L" = shipped ]; then unset MICA_HIP_LIB; else export MICA_HIP_LIB=$PWD/tools/exp/$L; fi
  timeout -k 10 300 python bench.py $Q > gpurun_out/$T/bench_$L.json 2> gpurun_out/$T/bench_$L.err; rc=$?
  echo "$L rc=$rc $(python -c "import json;j=json.load(open('gpurun_out/$T/bench_$L.json'));print(round(j['value'],2), round(j['ms_per_step'],2), round(j['roofline']['avg_launch_ms'],3))" 2>/dev/null)"
done
