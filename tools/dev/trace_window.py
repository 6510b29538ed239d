"""Print the kernels between two consecutive launches of a marker kernel (name substring) from a rocprofv3 kernel trace:
name, duration, idle gap before it.  usage: trace_window.py <dir> <marker> [which=-2] [width=60]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker, which = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -2
w = int(sys.argv[4]) if len(sys.argv) > 4 else 60
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[which - 1], marks[which]
prev_end, busy, idle = int(rows[a]["End_Timestamp"]), 0, 0
agg = {}
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"at::native::|void |\(anonymous namespace\)::", "", r["Kernel_Name"])[:w]
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    busy += e - s; idle += max(0, s - prev_end); prev_end = max(prev_end, e)
    k = agg.setdefault(n, [0, 0]); k[0] += 1; k[1] += e - s
print(f"-- {b - a} kernels, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / 1e3:9.1f} us {c:5d}  {n}")
