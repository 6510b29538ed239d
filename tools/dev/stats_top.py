"""Top kernels of a rocprofv3 --stats csv: python tools/dev/stats_top.py <dir-or-csv> [n]"""
import csv, glob, os, sys
p = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6)
for r in rows[:n]:
    print("%-72s %7d %9.1f us %5.1f%%" % (r["Name"][:72], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
