#!/usr/bin/env python3
"""GPU box: a 600-utterance ragged c3 batch through the row-block and streaming finishing kernels; where bits differ."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS
from util import run_product_batch
kw = dict(CONFIGS["c3"], znormFlag=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
rng = np.random.default_rng(12)
lengths = rng.integers(2 * 16000, 20 * 16000 + 1, size=600)
pcms = [ma.synth_pcm(int(n), 1, 2, seed=0x5EED0000 + 3000 + u, sample_rate=16000, tone_hz=0).tobytes() for u, n in enumerate(lengths)]
cfg = ma.make_config(precision=ma.PREC_F32, **kw)
res = {}
for f in (3, 2):
    ma.tuning(finish=f)
    res[f] = run_product_batch(ma, cfg, pcms, 1, 2)
ma.tuning()
bad = 0
for u in range(600):
    a, b = res[3][u], res[2][u]
    d = (a != b) & ~(np.isnan(a) & np.isnan(b))
    if d.any():
        bad += 1
        if bad <= 5:
            r, c = np.argwhere(d)[0]
            print("utt", u, "rows", a.shape[0], "differs in", int(d.sum()), "first", r, c, a[r, c], b[r, c], "cols", np.unique(np.nonzero(d)[1])[:45], "max abs", float(np.nanmax(np.abs(a - b))))
print("utterances that differ:", bad)
