#!/bin/bash
# quick GPU check of a kernel change: ops + model parity, then a short bench line (no CPU baseline / whole map)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-q2}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/$T/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/$T/t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench.json"))
print("value", round(d["value"],2), "conv3 TF", round(d["roofline"]["achieved"],1), "dw GB/s", round(d["hbm_conv3d"]["achieved"],1), "dw frac", round(d["hbm_conv3d"]["frac"],3), "dw ms", round(d["hbm_conv3d"]["avg_launch_ms"],4))
PY
exit $rc
