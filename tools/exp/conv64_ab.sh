#!/bin/bash
# Round 5: the tap-split F(4,3) kernel (conv_wino43_kernel<64>) against conv_wino16_kernel<64> on the late narrow layers' shapes, one 64^3
# tile, kernel-trace averages; cin = 64 ... 256 separates the per-chunk time (slope) from the per-item overhead (intercept).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/c64
for v in 0 1; do
  for cin in 64 128 192 256; do
    VARIANT=$v timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c64/v${v}_$cin -o r -- python3 tools/conv_bench.py $cin 64 3 64 4 > gpurun_out/c64/v${v}_$cin.log 2>&1
    f=$(find gpurun_out/c64/v${v}_$cin -name "*kernel_stats.csv" | head -1)
    echo "variant $v  $cin->64 (chunks $((cin/16))): $(grep -E 'conv_wino(16|43)_kernel' $f | awk -F, '{printf "%s calls %s avg_us %.1f  ", substr($1,1,40), $2, $4/1000}')"
  done
done
find gpurun_out/c64 -name "*.csv" -size +1M -delete
