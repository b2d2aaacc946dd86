#!/bin/bash
# Round-3 review item 4 (wide fused launches for the narrow layers of a dense block): single-layer kernel times (one 64^3 tile, kernel-trace
# averages, mica_op_conv3d) of the layers as they run today and of the launches a fused graph would use instead.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/fuse
for sh in "64 32" "96 32" "128 64" "64 128" "32 32" "64 64" "128 64" "192 64" "256 128" "128 256" "128 128" "192 192" "192 256"; do
  set -- $sh
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fuse/$1_$2 -o r -- python3 tools/conv_bench.py $1 $2 3 64 6 > gpurun_out/fuse/$1_$2.log 2>&1
  f=$(find gpurun_out/fuse/$1_$2 -name "*kernel_stats.csv" | head -1)
  echo "$1->$2: $(grep -E 'conv_wino16' $f | awk -F, '{print $1" avg_us "$4/1000}' | sed 's/.*conv_wino16_kernelILi\([0-9]*\).*avg_us/<\1> avg_us/')"
done
