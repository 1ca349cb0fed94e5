#!/usr/bin/env python3
"""Prints per-frame SQ counters of the static-feature kernel from tools/gpu_counters.sh passes.
usage: tools/counters_table.py <gpurun_out/tag> <workload> <frames per launch>"""
import csv, glob, os, sys
from collections import defaultdict
tag, w, frames = sys.argv[1], sys.argv[2], float(sys.argv[3])
KERNELS = ("k_lean256", "k_wide1024", "k_front", "k_pipe64")
acc, cnt = defaultdict(float), defaultdict(int)
for d in ("sq_", "sq2_"):
    paths = sorted(glob.glob("%s/%s%s/*/*_counter_collection.csv" % (tag, d, w)), key=os.path.getmtime)
    for r in csv.DictReader(open(paths[-1])):
        if any(k in r["Kernel_Name"] for k in KERNELS):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            vg = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
c = {k: acc[k] / cnt[k] for k in acc}
busy = c["SQ_BUSY_CU_CYCLES"]
print("per frame:", {k: round(v / frames, 1) for k, v in sorted(c.items())})
print("VALU busy %.3f  LDS busy %.3f  conflicts/LDS cycles %.3f  CU-busy cycles per frame per CU %.1f" % (
    c["SQ_ACTIVE_INST_VALU"] * 4 / (4 * busy), c["SQ_LDS_IDX_ACTIVE"] / busy, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), busy / frames))
print("wave cycles: wait_any %.3f  active_any %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
