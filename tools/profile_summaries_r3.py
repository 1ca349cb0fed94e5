#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/collect_profiles_r3.sh into the tracked summaries
  profiles/r03_<w>_<prec>_kernel_stats.csv          per-kernel times of the steady-state run
  profiles/r03_<w>_<prec>_bench_under_rocprof.json  the bench line of that same run
  profiles/r03_<w>_<prec>_traffic.json              HBM bytes per launch of the static-feature kernel
  profiles/r03_<w>_<prec>_counters.json             SQ counters per launch and per frame, busy fractions
usage: tools/profile_summaries_r3.py <workload> <f32|f64> [src dir]
(frames per launch and utterances come from the bench line of the traced run)

Units and corrections: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as it
is.  Busy fractions: a wave-level vector instruction occupies one of a CU's four SIMDs for four
cycles, so VALU busy = SQ_ACTIVE_INST_VALU x 4 / (4 x SQ_BUSY_CU_CYCLES); the LDS serves one
wave-level access pass per cycle per CU, LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

w, prec = sys.argv[1], sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/profiles_r3/%s_%s" % (w, prec)
KERNELS = ("k_lean256", "k_wide1024", "k_pipe64", "k_front")
stem = "profiles/r03_%s_%s_" % (w, prec)


def counters(dirname):
    acc, cnt = defaultdict(float), defaultdict(int)
    paths = sorted(glob.glob(dirname + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    name = None
    for r in csv.DictReader(open(paths[-1])):
        if any(k in r["Kernel_Name"] for k in KERNELS):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    return {k: acc[k] / cnt[k] for k in acc}, name


line = json.loads(open(src + "/bench_under_rocprof.json").readline())
utts = line["config"]["utterances_per_gpu"]
frames = float(line["config"]["frames_per_gpu"])
shutil.copy(src + "/kernel_stats.csv", stem + "kernel_stats.csv")
shutil.copy(src + "/bench_under_rocprof.json", stem + "bench_under_rocprof.json")
fetch, _ = counters(src + "/fetch")
write, _ = counters(src + "/write")
traffic = {
    "workload": w, "utts": utts, "precision": prec,
    "fetch_size_kib_raw": fetch["FETCH_SIZE"], "write_size_kib_raw": write["WRITE_SIZE"],
    "hbm_read_bytes_per_launch": 2.0 * fetch["FETCH_SIZE"] * 1024.0,
    "hbm_write_bytes_per_launch": write["WRITE_SIZE"] * 1024.0,
    "hbm_bytes_per_launch": (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE x1",
}
json.dump(traffic, open(stem + "traffic.json", "w"), indent=1)
c1, kname = counters(src + "/sq")
c2, _ = counters(src + "/sq2")
c1.update(c2)
busy = c1["SQ_BUSY_CU_CYCLES"]
out = {
    "workload": w, "precision": prec, "kernel": kname, "frames_per_launch": frames,
    "per_launch": c1, "per_frame": {k: v / frames for k, v in c1.items()},
    "valu_frac": c1["SQ_ACTIVE_INST_VALU"] * 4.0 / (4.0 * busy),
    "lds_frac": c1["SQ_LDS_IDX_ACTIVE"] / busy,
    "lds_bank_conflict_share": c1["SQ_LDS_BANK_CONFLICT"] / max(c1["SQ_LDS_IDX_ACTIVE"], 1.0),
    "notes": "averages over the launches of one rocprofv3 --pmc run (two passes of 8 SQ counters); fractions as "
             "defined in tools/profile_summaries_r3.py",
}
json.dump(out, open(stem + "counters.json", "w"), indent=1)
print(json.dumps({"w": w, "prec": prec, "kernel": kname, "traffic_bytes": traffic["hbm_bytes_per_launch"],
                  "valu_frac": out["valu_frac"], "lds_frac": out["lds_frac"],
                  "valu_per_frame": out["per_frame"]["SQ_INSTS_VALU"],
                  "lds_cycles_per_frame": out["per_frame"]["SQ_LDS_IDX_ACTIVE"]}))
