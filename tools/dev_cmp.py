#!/usr/bin/env python3
"""Compare one preset (with overrides) against the oracle on a short utterance and print where the
two differ most, column by column.  python tools/dev_cmp.py c4 [key=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma  # noqa: E402
from mfcc_amd.presets import CONFIGS  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from util import run_product_batch, column_kinds  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

kw = dict(CONFIGS[sys.argv[1]])
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = type(kw.get(k, 0))(float(v))
ch, bps = tp._layout(kw)
n = kw["sampleRate"] + 123
pcm = tp._pcm(n, ch, bps, 7, kw["sampleRate"])
ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], ch, bps)[0]
kinds = column_kinds(kw) * kw.get("vecNum", 1)
print("shape", got.shape, "nan got/ref", int(np.isnan(got).sum()), int(np.isnan(ref).sum()))
bad = np.argwhere(np.isnan(got) != np.isnan(ref))
print("first nan mismatches (row, col):", bad[:12].tolist())
d = np.abs(np.where(np.isfinite(got) & np.isfinite(ref), got - ref, 0.0))
for c in range(min(got.shape[1], len(kinds))):
    if d[:, c].max() > 1e-3 or np.isnan(got[:, c]).any():
        r = int(np.argmax(d[:, c]))
        print("col %d (%s): max diff %.4g at row %d (got %.6g ref %.6g), nan rows %s" % (
            c, kinds[c], d[:, c].max(), r, got[r, c], ref[r, c], np.where(np.isnan(got[:, c]))[0][:8].tolist()))
