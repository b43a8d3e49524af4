"""Basic-block instruction counts and resource figures of one kernel in a hipcc -S listing.
usage: python scratch/kernel_blocks.py LISTING.s KERNEL_SUBSTRING [--blocks]"""
import re, sys

def kernel_body(s, sub):
    for m in re.finditer(r"^(_Z\w+):", s, re.M):
        if sub in m.group(1):
            i = m.start(); j = s.index(".Lfunc_end", i)
            return m.group(1), s[i:j]
    raise SystemExit("no kernel " + sub)

def main():
    s = open(sys.argv[1]).read()
    name, body = kernel_body(s, sys.argv[2])
    blocks = []; blk = ["entry", 0, 0]; blocks.append(blk)
    for l in body.splitlines()[1:]:
        t = l.strip()
        m = re.match(r"(\.LBB\d+_\d+):", t)
        if m:
            blk = [m.group(1), 0, 0]; blocks.append(blk); continue
        if not t or t[0] in ";.": continue
        blk[1] += 1
        if t.startswith("v_"): blk[2] += 1
    md = s[s.index("amdhsa.kernels"):]
    k = md.index(".name:           " + name)
    a = md.rfind("  - .agpr_count", 0, k); b = md.find("  - .agpr_count", k)
    ent = md[a:b if b > 0 else len(md)]
    g = lambda key: re.search(r"\.%s:\s*(\S+)" % key, ent).group(1)
    print(name[:60], "instr", sum(b[1] for b in blocks), "valu", sum(b[2] for b in blocks), "vgpr", g("vgpr_count"), "agpr", g("agpr_count"),
          "sgpr", g("sgpr_count"), "spill", g("vgpr_spill_count"), "sgpr_spill", g("sgpr_spill_count"), "lds", g("group_segment_fixed_size"))
    if "--blocks" in sys.argv:
        for b in blocks: print("  ", *b)

main()
