#!/bin/bash
# Round 6: tools/xcd_share_bench (build: hipcc --offload-arch=gfx950 -O3 tools/xcd_share_bench.hip -o tools/xcd_share_bench) - timing, then
# FETCH_SIZE of the shared and the private mode in counter passes of their own.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/xcd
timeout -k 10 120 ./tools/xcd_share_bench > gpurun_out/xcd/timing.txt 2>&1; echo "timing rc=$?"; cat gpurun_out/xcd/timing.txt
for m in 0 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/xcd/pmc$m -o r -- ./tools/xcd_share_bench $m > gpurun_out/xcd/pmc$m.log 2>&1 || { tail -3 gpurun_out/xcd/pmc$m.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/xcd/pmc$m/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
vals = [per[k] for k in sorted(per, key=int)]
# three launches per buffer size: 32 MB first, then 2048 MB
print("mode %s (%s): FETCH_SIZE per launch, KiB x 2 (gfx950: 64 B counted per 128-B request) -> GB: %s" % ($m, "shared" if $m else "private", ["%.2f" % (v * 2048 / 1e9) for v in vals]))
PY
done
