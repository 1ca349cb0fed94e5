#!/usr/bin/env python3
"""GPU box: rows left unwritten (all zero) by a batch extraction of the c2 preset.
usage: python tools/dev_zero_rows.py [utterances]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")))
import mfcc_amd as ma  # noqa: E402
from mfcc_amd.presets import CONFIGS  # noqa: E402
from util import run_product_batch  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kw = dict(CONFIGS["c2"], preemphasise=0.0)
pcms = [ma.synth_pcm(160000, 1, 2, seed=0x5EED0000 + 500 + u, sample_rate=16000, tone_hz=0) for u in range(U)]
a = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [p.tobytes() for p in pcms], 1, 2)
allrows = np.concatenate(a)
zero = np.nonzero(np.all(allrows == 0, axis=1))[0]
print("rows", allrows.shape[0], "zero rows", len(zero))
tiles = sorted(set((zero // 998 * 250 + (zero % 998) // 4).tolist()))
print("tiles with zero rows", len(tiles), tiles[:60])
