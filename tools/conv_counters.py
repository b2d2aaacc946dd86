"""Summarise a rocprofv3 --pmc run of tools/conv_bench.py for one conv shape (development aid)."""
import csv, sys, collections, glob
d, cin, cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr)) if "conv_wino" in r["Kernel_Name"]]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    if "conv_wino" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
v = list(agg.values())[-1]
cyc = v["GRBM_GUI_ACTIVE"] / 8
fl = 2 * 27 * cin * cout * 262144
print("%d->%d: %.0f us  %.0f TF  cycles %.3fM clock %.2f GHz  mfma-busy %.2f  wait_any/wave %.2f wait_inst/wave %.2f" % (
    cin, cout, t[-1] * 1e3, fl / t[-1] / 1e9, cyc / 1e6, cyc / t[-1] / 1e6, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
    v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
