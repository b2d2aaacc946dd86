#!/bin/bash
# timing experiment: share of the output-transform epilogue in conv_wino16 (normal library vs a build without the passes)
# build the variant first (in the container): make -C mica_amd/csrc exp_noepi ; delete tools/exp/libmica_noepi.so afterwards
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/epi
for v in normal noepi; do
  if [ $v = noepi ]; then export MICA_HIP_LIB=$PWD/tools/exp/libmica_noepi.so; fi
  for sh in "64 64" "192 64" "64 32" "256 128" "512 256"; do
    set -- $sh
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/epi/${v}_$1_$2 -o r -- python3 tools/conv_bench.py $1 $2 3 64 4 > gpurun_out/epi/${v}_$1_$2.log 2>&1
    f=$(find gpurun_out/epi/${v}_$1_$2 -name "*kernel_stats.csv" | head -1)
    echo "$v $1->$2: $(grep conv_wino16 $f | awk -F, '{print "calls "$2" avg_us "$4/1000}')"
  done
done
