#!/bin/bash
# round 5 experiment: the 1x1 conv kernel with its loads re-ordered for the in-order vector-memory queue (fragments two steps ahead, the
# activation fetch behind step 0's request, peeled loop with unconditional loads) against the previous build (tools/exp/libmica_prev.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > gpurun_out/c1/tests.log 2>&1; rc=$?; tail -2 gpurun_out/c1/tests.log
[ $rc -eq 0 ] || exit $rc
for lib in tools/exp/libmica_prev.so mica_amd/lib/libmica_hip.so tools/exp/libmica_prev.so mica_amd/lib/libmica_hip.so; do
  MICA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/c1/b.json 2> gpurun_out/c1/b.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/c1/b.json"))
print("$lib: %.2f sub-grids/s (%.2f ms per step)" % (d["value"], d["ms_per_step"]))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-tiling --no-whole-map > $GRAFT_REPO_ROOT/gpurun_out/c1/prof.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(find gpurun_out/c1/prof -name "*kernel_stats.csv" | head -1) && python tools/prof_summary.py $f 72 40 | grep -i "conv1x1\|total"
