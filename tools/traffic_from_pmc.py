#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes written by tools/collect_profiles.sh into
profiles/traffic_<workload>.json (HBM bytes per k_front launch), which bench.py reports as
roofline.traffic.

Units and corrections: both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as it is.
"""
import csv
import glob
import json
import os
import sys


def per_launch(dirname, counter):
    vals = []
    paths = sorted(glob.glob(dirname + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    for path in paths[-1:]:  # the newest run only: merged result directories keep older files
        for r in csv.DictReader(open(path)):
            if "k_front" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for k_front under %s" % (counter, dirname))
    return sum(vals) / len(vals), len(vals)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_r1"
    workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
    utts = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    precision = sys.argv[4] if len(sys.argv) > 4 else "f32"
    fetch, nf = per_launch(src + "/fetch", "FETCH_SIZE")
    write, nw = per_launch(src + "/write", "WRITE_SIZE")
    out = {
        "workload": workload, "utts": utts, "precision": precision,
        "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
        "launches_averaged": [nf, nw],
        "hbm_read_bytes_per_launch": 2.0 * fetch * 1024.0,
        "hbm_write_bytes_per_launch": write * 1024.0,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE x1",
    }
    with open("profiles/traffic_%s.json" % workload, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
