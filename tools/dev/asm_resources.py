"""Per-function VGPR / scratch / code size of the device asm (hipcc -S --cuda-device-only)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
# blocks end with "; Function info:" or kernel info comments; find ".type NAME,@function" then metrics after
names = re.findall(r"^\s*\.type\s+(\S+),@function", txt, re.M)
out = []
for n in names:
    i = txt.find("\n%s:" % n)
    j = txt.find(".Lfunc_end", i)
    k = txt.find("\n\t.section", j) if j >= 0 else -1
    blk = txt[j:j + 3000]
    g = lambda key: (re.search(key + r":?\s+(\d+)", blk) or [None, "-1"])[1]
    out.append((int(g("; NumVgprs")), int(g("; ScratchSize")), int(g("; codeLenInByte")), n))
dem = subprocess.run(["c++filt"], input="\n".join(o[3] for o in out), capture_output=True, text=True).stdout.split("\n")
for (v, s, c, n), d in sorted(zip(out, dem)):
    if "Id" in n or "double" in d:
        continue
    print(f"{v:4d} vgpr {s:5d} scratch {c:7d} B  {d[:120]}")
