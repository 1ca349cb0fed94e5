#!/usr/bin/env python3
"""Time one device-resident pass of a preset with option overrides, e.g.
  python tools/time_config.py c3 znormFlag=0 regreOrder=2 --utts 10000 --seconds 10
Prints the whole pass and the static-feature kernel (HIP events) in ms."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch  # noqa: E402
import mfcc_amd as ma  # noqa: E402
from mfcc_amd.presets import CONFIGS  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("preset")
    ap.add_argument("overrides", nargs="*")
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--seconds", type=float, default=10.0)
    ns = ap.parse_args()
    args = [ns.preset] + ns.overrides
    utts, seconds = ns.utts, ns.seconds
    kw = dict(CONFIGS[args[0]])
    for a in args[1:]:
        k, v = a.split("=")
        kw[k] = type(kw.get(k, 0))(float(v)) if k in kw else int(v)
    ch = 2 if kw.get("vecNum", 1) > 1 else 1
    bps = 3 if kw["sampleRate"] == 44100 else 2
    n = int(seconds * kw["sampleRate"])
    one = ma.synth_pcm(n, ch, bps, seed=1, sample_rate=kw["sampleRate"], tone_hz=0)
    stride = (one.size + 15) & ~15
    buf = np.zeros(stride * utts, dtype=np.uint8)
    for u in range(utts):
        buf[u * stride:u * stride + one.size] = one
    ctx = ma.Context(ma.make_config(precision=ma.PREC_F32, **kw), 0)
    plan = ma.Plan(ctx, ch, bps, np.arange(utts, dtype=np.int64) * stride, np.full(utts, n, dtype=np.int64))
    dev = torch.from_numpy(buf).cuda()
    out = plan.alloc_output()
    plan.enable_timing(True)
    for _ in range(3):
        plan.extract(dev, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps, kms = 10, []
    for _ in range(reps):
        plan.extract(dev, out)
        kms.append(plan.last_kernel_ms())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%s %s: pass %.3f ms, static kernel %.3f ms, rest %.3f ms, %d frames" % (
        args[0], " ".join(args[1:]), ms, float(np.mean(kms)), ms - float(np.mean(kms)), plan.total_rows), flush=True)


if __name__ == "__main__":
    main()
