#!/bin/bash
# Is the conv kernel limited by power?  Same binary, same launches, random against all-zero operands (MI355X_MICROARCH.md,
# "DVFS give-back" item 1): kernel-trace averages of the conv kernel over 40 back-to-back single-layer calls on one 64^3 tile.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/zero_ab
for sh in ${ZERO_AB_SHAPES:-512:256 256:512}; do
  set -- ${sh%%:*} ${sh##*:}
  [ -n "$2" ] && [ "$1" -gt 0 ] && [ "$2" -gt 0 ] || { echo "bad shape $sh"; exit 1; }
  for v in 0 1; do
    for z in 0 1; do
      export VARIANT=$v ZERO=$z
      d=gpurun_out/zero_ab/v${v}_z${z}_$1_$2
      timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 tools/conv_bench.py $1 $2 3 64 40 > $d.log 2>&1 || { echo "run failed: $d"; exit 1; }
      f=$(find $d -name "*kernel_stats.csv" | head -1)
      echo "$1->$2 $([ $v = 1 ] && echo 'F(4,3)' || echo 'F(2,3)') $([ $z = 1 ] && echo zero || echo random): $(grep -E 'conv_wino16|conv_wino43' $f | awk -F, '{print "calls "$2" avg_us "$4/1000" min_us "$5/1000}')"
    done
  done
done
