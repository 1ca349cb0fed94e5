#!/usr/bin/env python3
"""Summarise SQ counter passes (rocprofv3 --pmc ... --output-format csv) for one kernel: averages
per launch and per frame.  python tools/front_counters.py <frames> <dir> [<dir> ...] [--kernel substr]
[--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

args = sys.argv[1:]
kernel, out_json = "k_front", None
if "--kernel" in args:
    i = args.index("--kernel")
    kernel = args[i + 1]
    del args[i:i + 2]
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    del args[i:i + 2]
frames = float(args[0])
acc, cnt = defaultdict(float), defaultdict(int)
for d in args[1:]:
    paths = sorted(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(paths[-1])):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
res = {}
for k in sorted(acc):
    v = acc[k] / cnt[k]
    res[k] = {"per_launch": v, "per_frame": v / frames, "launches": cnt[k]}
    print("%-24s %14.4g per launch  %10.2f per frame" % (k, v, v / frames))
if out_json:
    with open(out_json, "w") as f:
        json.dump({"kernel": kernel, "frames_per_launch": frames, "counters": res}, f, indent=1)
