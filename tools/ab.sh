#!/bin/bash
# A/B two builds of libmica_hip.so in one process-free interleaved run on the same device (development aid).
# usage: tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    v=$(MICA_HIP_LIB=$L timeout -k 10 250 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['roofline']['achieved'],1))")
    echo "round $i $(basename $L): $v"
  done
done
