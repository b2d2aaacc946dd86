#!/bin/bash
# Round 6, review item 4: conv_wino43_kernel<64> with every wave touching the slab lines of chunk k+2 behind chunk k+1's last DMA
# (dev build: make -C mica_amd/csrc exp_warm43) against the shipped kernel, single layers, one 64^3 tile, kernel-trace averages.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/warm43
for rep in 1 2; do
for v in normal ${WARM43_LIST:-WARM}; do
  if [ $v = normal ]; then L=$PWD/mica_amd/lib/libmica_hip.so; else L=$PWD/tools/exp/libmica43_$v.so; fi
  [ -f $L ] || { echo "missing $L"; exit 1; }
  for cin in 64 128 192 256; do
    MICA_HIP_LIB=$L VARIANT=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/warm43/${v}_${cin}_$rep -o r -- python3 tools/conv_bench.py $cin 64 3 64 6 > gpurun_out/warm43/${v}_${cin}_$rep.log 2>&1
    f=$(find gpurun_out/warm43/${v}_${cin}_$rep -name "*kernel_stats.csv" | head -1)
    echo "$v  $cin->64 (chunks $((cin/16))) rep $rep: $(grep -E 'conv_wino43_kernel' $f | awk -F, '{printf "calls %s avg_us %.1f min_us %.1f", $2, $4/1000, $5/1000}')  $(tail -1 gpurun_out/warm43/${v}_${cin}_$rep.log | sed 's/.*out mean/mean/')"
  done
done
done
find gpurun_out/warm43 -name "*.csv" -size +1M -delete
