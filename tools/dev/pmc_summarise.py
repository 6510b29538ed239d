"""Per-launch averages of rocprofv3 --pmc counter_collection csv files for kernels matching a name."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[2:] if len(v) > 4 else v          # drop the first launches (cold)
    print("PMC", k, sum(v) / len(v), len(v), "total", sum(v), "kernels~" + pat)
