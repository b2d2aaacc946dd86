#!/bin/bash
# energy ablations of conv_wino16 (dev builds: make -C mica_amd/csrc exp_abl) on single layers with random operands (one 64^3 tile,
# mica_op_conv3d): the kernel is power-bound, so the time an ablation saves is (roughly) the share of the chip's power that the
# ablated data movement costs.  kernel-trace averages of the conv kernel.  R2W / R2S / R2WS = ROWFRAGS=2 combined with W_FIXED /
# SLAB_FIXED / both (built by hand: hipcc -DMICA_EXP_ROWFRAGS=2 -DMICA_EXP_W_FIXED ... -c kernels_conv.hip, linked like exp_abl).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
for sh in "512 256" "256 512" "128 64" "64 64"; do
  set -- $sh
  for v in normal SLAB_FIXED W_FIXED NOEPI ROWFRAGS=4 ROWFRAGS=2 R2W R2S R2WS; do
    [ -f $PWD/tools/exp/libmica_$v.so ] || [ $v = normal ] || continue
    if [ $v = normal ]; then L=$PWD/mica_amd/lib/libmica_hip.so; else L=$PWD/tools/exp/libmica_$v.so; fi
    export MICA_HIP_LIB=$L
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/${v}_$1_$2 -o r -- python3 tools/conv_bench.py $1 $2 3 64 6 > gpurun_out/abl/${v}_$1_$2.log 2>&1
    f=$(find gpurun_out/abl/${v}_$1_$2 -name "*kernel_stats.csv" | head -1)
    echo "$1->$2 $v: $(grep conv_wino16 $f | awk -F, '{print "calls "$2" avg_us "$4/1000" min_us "$5/1000}')"
  done
done
