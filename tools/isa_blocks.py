#!/usr/bin/env python3
"""Per basic block instruction mix of one kernel in an ISA listing (hipcc -S --cuda-device-only).
usage: tools/isa_blocks.py <file.s> <mangled-name-substring> [min_instructions]"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E:", ")")) or (l.startswith("_Z") and key in l and ":" in l))
blocks, name, cur, note = [], "entry", Counter(), ""
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB[0-9_]+):\s*(;.*)?", l)
    if m:
        blocks.append((name, cur, note))
        name, cur, note = m.group(1), Counter(), (m.group(2) or "")
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
    if not m:
        continue
    op = m.group(1)
    cur["total"] += 1
    if op.startswith("v_mfma"):
        cur["mfma"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
        if op.startswith(("v_mov", "v_pk_mov", "v_accvgpr")):
            cur["mov"] += 1
        if op.startswith("v_pk_"):
            cur["pk"] += 1
        if re.match(r"v_(add|sub|mul|fma|fmac|mac)_f32", op):
            cur["f32"] += 1
        if re.match(r"v_.*(u32|i32|u64|b32|b64|lshl|ashr|and_|or_|xor)", op) and not op.startswith(("v_mov", "v_cvt", "v_pk")):
            cur["int"] += 1
        if op.startswith(("v_cmp", "v_cndmask")):
            cur["sel"] += 1
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            cur["lane"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op == "s_waitcnt":
            cur["wait"] += 1
        if op == "s_nop":
            cur["nop"] += 1
    elif op.startswith("ds_"):
        cur["ds"] += 1
    else:
        cur["vmem"] += 1
blocks.append((name, cur, note))
for name, c, note in blocks:
    if c["total"] >= minn:
        print("%-12s %s %s" % (name, " ".join("%s=%d" % (k, c[k]) for k in ("total", "valu", "pk", "f32", "int", "sel", "mov", "lane", "mfma", "salu", "wait", "nop", "ds", "vmem")), note[:60]))
