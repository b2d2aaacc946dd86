"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv (development aid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ntiles = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    t = float(r["TotalDurationNs"])
    print("%-58s calls %4s total %8.2f ms %5.1f%% avg %8.1f us" % (r["Name"][:58], r["Calls"], t / 1e6, 100 * t / tot, float(r["AverageNs"]) / 1e3))
print("total ms %.2f  per tile %.3f ms" % (tot / 1e6, tot / 1e6 / ntiles))
