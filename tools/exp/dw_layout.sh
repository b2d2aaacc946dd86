#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/dwl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dwl/t -o r -- python3 tools/exp/dw_layout.py > gpurun_out/dwl/log.txt 2>&1; echo rc=$?
cat gpurun_out/dwl/log.txt | tail -5
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/dwl/t/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "depthwise" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:44], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, "launches", len(v), "avg us %.1f  min %.1f" % (sum(v) / len(v), min(v)))
PY
find gpurun_out/dwl -name "*.csv" -size +2M -delete
