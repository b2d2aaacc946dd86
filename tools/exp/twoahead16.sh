#!/bin/bash
# Round 6: conv_wino16_kernel<64> / <32> with the weight fragments requested TWO steps ahead (four register sets; dev build:
# make -C mica_amd/csrc exp_plan16 -> libmica16_TWOAHEAD.so) against the shipped one-step-ahead schedule: op tests on the dev build first
# (results must not change), then single layers, one 64^3 tile, kernel-trace averages, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/two16
MICA_HIP_LIB=$PWD/tools/exp/libmica16_TWOAHEAD.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3d or fused" > gpurun_out/two16/ops.log 2>&1; rc=$?; echo "op tests on the dev build rc=$rc"; tail -2 gpurun_out/two16/ops.log
[ $rc -eq 0 ] || exit $rc
PLAN16_LIST=TWOAHEAD tools/exp/plan16.sh
