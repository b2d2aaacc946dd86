#!/bin/bash
# sample the socket power and clocks while bench.py's timed loop runs (development aid: is the conv kernel at the power cap?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/power
python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-alt-tiling --no-whole-map > gpurun_out/power/bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk" | sed 's/=//g; s/GPU\[0\]//; s/\t//g' | tr '\n' ' '; echo
  sleep 0.7
done > gpurun_out/power/samples.txt
wait $BP
sort gpurun_out/power/samples.txt | uniq -c | sort -k1 -n | tail -12
python -c "import json; d=json.load(open('gpurun_out/power/bench.json')); print('value', d['value'])"
