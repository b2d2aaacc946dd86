#!/bin/bash
# ablations of stem_mfma_kernel (dev builds: make -C mica_amd/csrc exp_stem; one K-step per kernel size; no output): kernel-trace average of
# the stem kernel in bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/stemabl
for v in shipped NOLOOP NOEPI; do
  if [ $v = shipped ]; then unset MICA_HIP_LIB; else export MICA_HIP_LIB=$PWD/tools/exp/libmica_stem_$v.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stemabl/$v -o r -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/stemabl/$v.log 2>&1 || { echo "failed $v"; exit 1; }
  f=$(find gpurun_out/stemabl/$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep -E 'stem_mfma' $f | awk -F, '{print "calls "$2" avg_us "$4/1000}')"
done
