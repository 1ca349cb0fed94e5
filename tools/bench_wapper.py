#!/usr/bin/env python3
"""Latency of the per-file entry point MFCCWapperMem (WAV image in host memory -> feature matrix
in host memory) for the example configuration and the 16 kHz mono one.
usage: tools/bench_wapper.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (loads the HIP runtime first)
import mfcc_amd as ma  # noqa: E402
from mfcc_amd.presets import CONFIGS  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for name, ch, bps, n in (("c1a", 2, 3, 1323001), ("c2", 1, 2, 160000), ("c3", 1, 2, 160000)):
    kw = CONFIGS[name]
    rng = np.random.default_rng(3)
    hi = 1 << (8 * bps - 2)
    x = rng.integers(-hi, hi, size=(n, ch)).astype(np.int64)
    pcm = np.zeros((n, ch, bps), dtype=np.uint8)
    for b in range(bps):
        pcm[:, :, b] = (x >> (8 * b)) & 0xff
    wav = ma.wav_bytes(pcm.reshape(-1), ch, bps, kw["sampleRate"])
    for prec, label in ((ma.PREC_F64, "f64"), (ma.PREC_F32, "f32")):
        cfg = ma.make_config(precision=prec, **kw)
        for _ in range(3):
            feat = ma.wapper_mem(cfg, wav)
        t0 = time.perf_counter()
        for _ in range(calls):
            feat = ma.wapper_mem(cfg, wav)
        dt = (time.perf_counter() - t0) / calls
        print("%s %s: %d samples x %d ch -> %s in %.3f ms per call" % (name, label, n, ch, feat.shape, dt * 1e3), flush=True)
