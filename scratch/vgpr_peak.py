"""highest vector register index touched per basic block of one kernel (where a kernel's register count comes from)
usage: python scratch/vgpr_peak.py LISTING.s KERNEL_SUBSTRING [MIN]"""
import re, sys
s = open(sys.argv[1]).read()
m = [x for x in re.finditer(r"^(_Z\w+):", s, re.M) if sys.argv[2] in x.group(1)][0]
b = s[m.start():s.index(".Lfunc_end", m.start())]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cur = ["entry", 0, 0, ""]
out = []
for l in b.splitlines()[1:]:
    t = l.strip()
    mm = re.match(r"(\.LBB\d+_\d+):\s*(;.*)?", t)
    if mm:
        out.append(cur); cur = [mm.group(1), 0, 0, (mm.group(2) or "")[:70]]; continue
    if not t or t[0] in ";.": continue
    cur[2] += 1
    for r in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", t):
        cur[1] = max(cur[1], int(r.group(1)) if r.group(1) else int(r.group(3)))
out.append(cur)
for x in out:
    if x[1] >= lo: print(x)
