#!/bin/bash
# per-phase cycle stamps of the depthwise kernel (dev builds: make -C mica_amd/csrc exp_dwclk): normal, without HBM reads, without taps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/dwclk
for v in dwclk dwnofetch dwnocompute; do
  MICA_HIP_LIB=$PWD/tools/exp/libmica_$v.so timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/dwclk/$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/dwclk/$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dw GB/s', round(d['hbm_conv3d']['achieved']), 'ms', round(d['hbm_conv3d']['avg_launch_ms'],4))")"
  grep "^dw<" gpurun_out/dwclk/$v.log | grep "C=256 blk 77" | sort | uniq | head -4
done
