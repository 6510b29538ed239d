import csv, re, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"at::native::|void |\(anonymous namespace\)::", "", r["Kernel_Name"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 70] for r in rows]
half = len(names) // 2
for n in names[half:]:
    print(n)
