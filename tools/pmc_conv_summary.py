"""Per-kernel summary of a rocprofv3 `--kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE` run of bench.py (profiles/rNN_pmc_sq_summary.txt).
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
is cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
usage: python tools/pmc_conv_summary.py <dir with *_kernel_trace.csv and *_counter_collection.csv>"""
import collections, csv, glob, re, sys

d = sys.argv[1]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]


def short(n):
    m = re.search(r"conv_wino16_kernelILi(\d+)E", n)
    if m:
        return "conv_wino16_kernel<%s>" % m.group(1)
    m = re.search(r"conv_wino43_kernelILi(\d+)E", n) or re.search(r"conv_wino43_kernel<(\d+)>", n)
    if m:
        return "conv_wino43_kernel<%s>" % m.group(1)
    m = re.search(r"conv2_kernelILi(\d+)ELi(\d+)E", n)
    if m:
        return "conv2_kernel<%s,%s>" % (m.group(1), m.group(2))
    m = re.search(r"conv1x1_kernelILi(\d+)E", n)
    if m:
        return "conv1x1_kernel<%s>" % m.group(1)
    m = re.search(r"(conv_wino43_kernel|prep_wino43_kernel|stem_mfma_kernel|conv_wino16_kernel<\d+>|conv2_kernel<[\d, ]+>|depthwise_kernel<[\d, ]+>|prep_wino_kernel|prep_kernel|prep_ncdhw_wino_kernel|stem_kernel|"
                  r"stats_finalize_kernel|feat_gate_kernel|head_final_kernel|postprocess_kernel|gather_tiles_kernel|stitch_tiles_kernel)", n)
    return m.group(1) if m else None


dur = {}
for r in csv.DictReader(open(tr)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
agg = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for r in csv.DictReader(open(cc)):
    k = short(r["Kernel_Name"])
    if k:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = k
per = collections.defaultdict(lambda: collections.defaultdict(float))
for did, v in agg.items():
    k = name[did]
    per[k]["n"] += 1
    per[k]["us"] += dur.get(did, 0.0)
    for c, x in v.items():
        per[k][c] += x
print("%-28s %5s %10s %9s %7s %9s %9s %9s %9s %9s %10s" % ("kernel", "calls", "avg us", "clock GHz", "mfma%", "wait_any", "wait_inst", "active", "lds_act%", "bank_cf%", "valu/wave-cyc"))
for k, v in sorted(per.items(), key=lambda kv: -kv[1]["us"]):
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print("%-28s %5d %10.1f %9.2f %7.1f %9.3f %9.3f %9.3f %9.1f %9.2f %10.3f" % (
        k, v["n"], v["us"] / v["n"], cyc / max(v["us"], 1e-9) / 1e3, 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024.0 * cyc, 1.0),
        v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc,
        100.0 * v["SQ_LDS_IDX_ACTIVE"] / max(256.0 * cyc, 1.0), 100.0 * v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1.0),
        v["SQ_INSTS_VALU"] / wc))
print("mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = GRBM_GUI_ACTIVE/8 / wall time (profiled passes clock lower than "
      "un-profiled ones); wait_any / wait_inst / active are shares of SQ_WAVE_CYCLES")
