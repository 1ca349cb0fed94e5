#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc output directories.
usage: tools/pmc_table.py <dir> [<dir> ...]   (prints kernel, launches, counter = average per launch)"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mfcc::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if k.startswith("__amd") or "at::" in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        print("    %-24s %14.0f  (%d launches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
