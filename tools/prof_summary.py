"""Summarises a tools/prof_pmc.sh output directory: kernel stats and mean counter values per kernel."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob(d + "/kt/*/*_kernel_stats.csv"):
    print(open(f).read().strip())
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (d, sub)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-42s %-24s %.4g  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
