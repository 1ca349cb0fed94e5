#!/usr/bin/env python3
"""GPU box: rows of a preset's one-utterance extraction that differ from the oracle.
usage: python tools/dev_rows.py <preset> [samples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")))
import mfcc_amd as ma  # noqa: E402
from mfcc_amd.presets import CONFIGS  # noqa: E402
from oracle import binding as ob  # noqa: E402
from util import run_product_batch  # noqa: E402

name = sys.argv[1]
kw = CONFIGS[name]
ch, bps = (1 if kw["vecNum"] == 1 else 2), (2 if kw["sampleRate"] == 16000 else 3)
n = int(sys.argv[2]) if len(sys.argv) > 2 else kw["sampleRate"] + 123
pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + 7, sample_rate=kw["sampleRate"], tone_hz=440)
ob.build()
ref = ob.extract_pcm_bytes(ob.make_config(**kw), pcm, ch, bps)
got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], ch, bps)[0]
err = np.max(np.abs(got - ref), axis=1)
bad = np.nonzero(err > 1e-3)[0]
print("rows", got.shape[0], "bad rows", len(bad), bad[:40].tolist())
for r in bad[:3]:
    print(r, "got", got[r, :8], "ref", ref[r, :8])
cols = np.nonzero(np.max(np.abs(got - ref), axis=0) > 1e-3)[0]
print("bad columns", cols.tolist())
