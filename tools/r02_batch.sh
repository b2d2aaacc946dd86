#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h
for b in 4 8 12 16; do
  timeout -k 10 200 python bench.py --batch $b --steps $((96 / b)) --warmup 2 --no-cpu-baseline --no-alt-tiling > gpurun_out/r02h/b$b.json 2> gpurun_out/r02h/b$b.err || { echo "batch $b failed"; tail -3 gpurun_out/r02h/b$b.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/r02h/b$b.json')); print('batch $b', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'conv TF', round(d['roofline']['achieved'],1))"
done
