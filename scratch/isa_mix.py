"""Instruction mix of the longest loop of a kernel in a hipcc -save-temps .s file, with the issue-cost model of
scratch/issue_bench.hip (one wave per SIMD, gfx950): fp64 matrix instruction 64 cycles, fp64 vector 4.9, v_accvgpr_* 7.7,
other vector 4.6 -- none of them overlaps another; LDS / memory / scalar instructions issue in the matrix shadow.
usage: python scratch/isa_mix.py FILE.s KERNEL_SUBSTRING"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(name), s, re.M)
a = m.start()
b = s.index(".Lfunc_end", a)
body = s[a:b].split("\n")
labels, loops = {}, []
for i, l in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = i
for i, l in enumerate(body):
    mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if mm and labels.get(mm.group(1), 1 << 30) < i:
        loops.append((labels[mm.group(1)], i))
lo, hi = max(loops, key=lambda t: t[1] - t[0])
cnt = collections.Counter()
for l in body[lo:hi]:
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    cnt[l.split()[0]] += 1
f64 = sum(v for k, v in cnt.items() if re.match(r"v_(fma|fmac|mul|add)_f64", k))
mfma = sum(v for k, v in cnt.items() if k.startswith("v_mfma"))
acc = sum(v for k, v in cnt.items() if k.startswith("v_accvgpr"))
valu = sum(v for k, v in cnt.items() if k.startswith("v_")) - f64 - mfma - acc
other = sum(cnt.values()) - f64 - mfma - acc - valu
model = 64 * mfma + 4.9 * f64 + 7.7 * acc + 4.6 * valu
print(f"{m.group(1)[:60]}: loop of {sum(cnt.values())} instructions")
print(f"  matrix {mfma}  fp64 vector {f64}  accvgpr {acc}  other vector {valu}  LDS/memory/scalar {other}")
print(f"  modelled {model:.0f} cycles = {64 * mfma} + {4.9 * f64:.0f} + {7.7 * acc:.0f} + {4.6 * valu:.0f}")
if len(sys.argv) > 3:
    for k, v in cnt.most_common(30):
        print(f"    {k:32s}{v}")
