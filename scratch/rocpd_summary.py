"""Summarise rocprofv3 (rocpd sqlite) outputs: kernel stats csv and per-kernel PMC means.

usage: python scratch/rocpd_summary.py stats <dir> > kernel_stats.csv
       python scratch/rocpd_summary.py pmc <dir> [<dir> ...] > pmc.txt
"""
import glob
import sqlite3
import sys


def dbs(d):
    return sorted(glob.glob(d + "/**/*.db", recursive=True))


def stats(d):
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for f in dbs(d):
        c = sqlite3.connect(f)
        rows = c.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
            "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        for name, n, t, avg, mn, mx in rows:
            print('"%s",%d,%d,%.0f,%.2f,%d,%d' % (name, n, t, avg, 100.0 * t / tot, mn, mx))


def pmc(dirs):
    for d in dirs:
        for f in dbs(d):
            c = sqlite3.connect(f)
            rows = c.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
                "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(scratch_size) "
                "from counters_collection group by kernel_name, counter_name order by 1, 2").fetchall()
            for k, cn, n, avg, mn, mx, vg, ag, sg, lds, scr in rows:
                if not any(t in k for t in ("tensor_", "general", "contact", "read8", "read16", "write8")):
                    continue
                print("%-70s %-14s dispatches %3d mean %.6g min %.6g max %.6g  (vgpr %s agpr %s sgpr %s lds %s scratch %s)"
                      % (k[:70], cn, n, avg, mn, mx, vg, ag, sg, lds, scr))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
