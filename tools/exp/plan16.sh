#!/bin/bash
# Round 6: slab-DMA placements of the tap-split F(2,3) kernels conv_wino16_kernel<64> / <32> (dev builds: make -C mica_amd/csrc exp_plan16 PLAN16_VARIANTS="1 2 3 4"; TWOAHEAD is always built)
# against the shipped placement, single layers, one 64^3 tile, kernel-trace averages.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/plan16
for rep in 1 2; do
for v in normal ${PLAN16_LIST:-PLAN1 PLAN2 PLAN3 PLAN4}; do
  if [ $v = normal ]; then L=$PWD/mica_amd/lib/libmica_hip.so; else L=$PWD/tools/exp/libmica16_$v.so; fi
  [ -f $L ] || { echo "missing $L"; exit 1; }
  for sh in 64:64 192:64 64:32 128:32; do
    cin=${sh%%:*}; cout=${sh##*:}
    MICA_HIP_LIB=$L VARIANT=0 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/plan16/${v}_${cin}_${cout}_$rep -o r -- python3 tools/conv_bench.py $cin $cout 3 64 6 > gpurun_out/plan16/${v}_${cin}_${cout}_$rep.log 2>&1
    f=$(find gpurun_out/plan16/${v}_${cin}_${cout}_$rep -name "*kernel_stats.csv" | head -1)
    echo "$v  $cin->$cout rep $rep: $(grep -E 'conv_wino16_kernel' $f | awk -F, '{printf "calls %s avg_us %.1f", $2, $4/1000}')"
  done
done
done
find gpurun_out/plan16 -name "*.csv" -size +1M -delete
