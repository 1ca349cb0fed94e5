#!/usr/bin/env python3
"""Summarise an SQ counter pass (rocprofv3 --pmc ... --output-format csv) for the static-feature
kernel: averages per launch and per frame.  python tools/front_counters.py <dir> <frames>"""
import csv
import glob
import os
import sys
from collections import defaultdict

paths = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"), key=os.path.getmtime)
frames = float(sys.argv[2])
acc, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(paths[-1])):
    if "k_front" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
for k in sorted(acc):
    v = acc[k] / cnt[k]
    print("%-24s %14.4g per launch  %10.2f per frame" % (k, v, v / frames))
