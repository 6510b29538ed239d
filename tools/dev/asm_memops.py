"""Static count of global/scratch/ds/s_load instructions per device function (float instantiations)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
names = re.findall(r"^\s*\.type\s+(\S+),@function", txt, re.M)
rows = []
for n in names:
    i = txt.find("\n%s:" % n)
    j = txt.find(".Lfunc_end", i)
    body = txt[i:j]
    c = lambda pat: len(re.findall(pat, body))
    rows.append((n, c(r"\bglobal_load"), c(r"\bscratch_load|\bbuffer_load"), c(r"\bscratch_store|\bbuffer_store"), c(r"\bds_read|\bds_load"), c(r"\bds_write|\bds_store"), c(r"\bs_load|\bs_buffer_load"), c(r"\bs_waitcnt"), c(r"s_swappc")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'gload':>6} {'scrld':>6} {'scrst':>6} {'dsrd':>6} {'dswr':>6} {'sload':>6} {'wait':>6} {'calls':>6}")
for r, d in zip(rows, dem):
    if "double" in d or "Id" in r[0]:
        continue
    print(" ".join(f"{x:6d}" for x in r[1:]), d[:90].replace("DevModel<float> const&", "M").replace("Scratch<float>&", "S").replace("TaskDev const&", "K"))
