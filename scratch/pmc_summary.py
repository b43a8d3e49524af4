"""Per-kernel sums of a rocprofv3 --pmc run (counter_collection.csv): mean per dispatch of every counter.
usage: python scratch/pmc_summary.py DIR [DIR...]"""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add(row["Dispatch_Id"])
        print("#", f)
        for k in acc:
            n = max(len(cnt[k]), 1)
            print(k, "dispatches", n, " ".join(f"{c}={v / n:.6g}" for c, v in sorted(acc[k].items())))
