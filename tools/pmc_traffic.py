"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports 1/2 of the bytes
of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [label] [library_source_hash]"""
import csv, json, sys, collections

def load(path, counter):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per

def short(n):
    import re
    m = re.search(r"conv_wino43_kernelILi(\d+)E", n) or re.search(r"conv_wino43_kernel<(\d+)>", n)
    if m:
        return "conv_wino43_kernel<%s>" % m.group(1)
    for key in ("conv_wino43_kernel", "prep_wino43_kernel", "stem_mfma_kernel", "conv_wino16_kernel", "conv1x1_kernel", "depthwise_kernel", "prep_wino_kernel", "prep_kernel",
                "stats_kernel", "stem_kernel", "gather_tiles_kernel", "stitch_tiles_kernel", "postprocess_kernel", "head_final_kernel",
                "hist_kernel", "finish_kernel"):
        if key in n:
            return key
    return None

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"label": sys.argv[4] if len(sys.argv) > 4 else "", "library_source_hash": sys.argv[5] if len(sys.argv) > 5 else None, "units": "bytes per launch (mean over launches)",
       "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request)", "kernels": {}}
agg = collections.defaultdict(lambda: {"launches": 0, "fetch": 0.0, "write": 0.0})
for name, d in fetch.items():
    k = short(name)
    if k:
        agg[k]["launches"] += len(d)
        agg[k]["fetch"] += sum(d.values()) * 1024 * 2
for name, d in write.items():
    k = short(name)
    if k:
        agg[k]["write"] += sum(d.values()) * 1024
# conv_wino43_kernel<128> runs encoder.2's four 3^3 convs, always in this order within a forward pass: the per-layer means separate the
# weight stream (re-read through every XCD's L2 once per round of items) from the operand slabs ("Round 6")
LAYERS43 = ("encoder.2 conv1 256->128", "encoder.2 conv2 384->128", "encoder.2 conv3 512->256", "encoder.2 transition 256->512")
per_layer = {}
for name in fetch:
    if short(name) == "conv_wino43_kernel<128>":
        ids = sorted(fetch[name], key=int)
        wr = write.get(name, {})
        for pos, lay in enumerate(LAYERS43):
            sel = ids[pos::4]
            if sel and len(ids) % 4 == 0:
                per_layer[lay] = {"launches": len(sel), "hbm_read_bytes": sum(fetch[name][i] for i in sel) * 2048 / len(sel),
                                  "hbm_write_bytes": sum(wr.get(i, 0.0) for i in sel) * 1024 / len(sel)}
for k, v in agg.items():
    n = max(v["launches"], 1)
    out["kernels"][k] = {"launches": v["launches"], "hbm_read_bytes": v["fetch"] / n, "hbm_write_bytes": v["write"] / n,
                         "hbm_bytes": (v["fetch"] + v["write"]) / n}
if per_layer:
    out["kernels"]["conv_wino43_kernel<128>"]["per_layer"] = per_layer
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
