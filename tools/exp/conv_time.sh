#!/bin/bash
# per-layer conv_wino16 timing through mica_op_conv3d (one tile): kernel-trace averages
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/convt
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3d or fused" 2>&1 | tail -2
for sh in "64 128" "256 128" "128 256" "512 256" "256 512"; do
  set -- $sh
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/convt/$1_$2 -o r -- python3 tools/conv_bench.py $1 $2 3 64 4 > gpurun_out/convt/$1_$2.log 2>&1
  f=$(find gpurun_out/convt/$1_$2 -name "*kernel_stats.csv" | head -1)
  echo "$1->$2: $(grep conv_wino16 $f | awk -F, '{print "calls "$2" avg_us "$4/1000}')"
done
