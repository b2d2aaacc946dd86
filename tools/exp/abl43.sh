#!/bin/bash
# ablations of conv_wino43 (dev builds: make -C mica_amd/csrc exp_abl43) on single layers with random operands (one 64^3 tile,
# mica_op_conv3d_variant): kernel-trace averages of the conv kernel; F(2,3) kernel on the same layer for reference.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/abl43
for sh in ${ABL43_SHAPES:-512:256 256:512 256:128}; do
  set -- ${sh%%:*} ${sh##*:}
  [ -n "$2" ] && [ "$1" -gt 0 ] && [ "$2" -gt 0 ] || { echo "bad shape $sh"; exit 1; }
  for v in f23 normal $ABL43_LIST; do
    if [ $v = normal ] || [ $v = f23 ]; then L=$PWD/mica_amd/lib/libmica_hip.so; else L=$PWD/tools/exp/libmica43_$v.so; fi
    [ -f $L ] || continue
    export MICA_HIP_LIB=$L
    if [ $v = f23 ]; then export VARIANT=0; else export VARIANT=1; fi
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl43/${v}_$1_$2 -o r -- python3 tools/conv_bench.py $1 $2 3 64 6 > gpurun_out/abl43/${v}_$1_$2.log 2>&1
    f=$(find gpurun_out/abl43/${v}_$1_$2 -name "*kernel_stats.csv" | head -1)
    echo "$1->$2 $v: $(grep -E 'conv_wino16|conv_wino43' $f | awk -F, '{print "calls "$2" avg_us "$4/1000" min_us "$5/1000}')"
  done
done
