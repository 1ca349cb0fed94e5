#!/usr/bin/env python3
"""GPU box: one c3 utterance through the whole-utterance-in-LDS, row-block and streaming finishing kernels; where bits differ."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS
from util import run_product_batch
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
zn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kw = dict(CONFIGS[name], znormFlag=zn)
ch, bps = (1, 2) if kw["sampleRate"] == 16000 else (2, 3)
pcm = ma.synth_pcm(3 * kw["sampleRate"] + 77, ch, bps, seed=0x5EED0000 + 31, sample_rate=kw["sampleRate"], tone_hz=0)
cfg = ma.make_config(precision=ma.PREC_F32, **kw)
res = {}
for f in (1, 3, 2):
    ma.tuning(finish=f)
    res[f] = run_product_batch(ma, cfg, [pcm.tobytes()], ch, bps)[0]
ma.tuning()
for f in (3, 2):
    d = res[f] != res[1]
    d &= ~(np.isnan(res[f]) & np.isnan(res[1]))
    print("finish", f, "differs in", int(d.sum()), "of", d.size, "cols", np.unique(np.nonzero(d)[1])[:50], "rows", np.unique(np.nonzero(d)[0])[:20])
    if d.any():
        r, c = np.argwhere(d)[0]
        print("  first", r, c, res[1][r, c], res[f][r, c])
