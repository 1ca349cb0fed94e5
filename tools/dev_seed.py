#!/usr/bin/env python3
"""Diagnose one seed of tests/test_gpu_parity.py::test_random_configurations: where the largest
log-mel difference sits and how that channel's energy compares with the frame's."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from util import run_product_batch, column_kinds  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
sr = int(rng.choice([8000, 16000, 44100]))
wlen = int(rng.choice([64, 200, 256, 400, 441, 512, 800, 1024, 1764, 2048]))
inc = int(max(1, wlen // int(rng.choice([2, 3, 4])) + int(rng.integers(-1, 2))))
ch = int(rng.choice([1, 2]))
bps = int(rng.choice([2, 2, 3, 4]))
vec = int(rng.integers(1, 5)) if ch == 2 else 1
bank = int(rng.integers(3, 50))
fbank = int(rng.integers(0, 2))
order = int(rng.choice([1, 1, 2, 3, 4]))
plain = bool(rng.integers(0, 2))
kw = dict(sampleRate=sr, lowpassfre=-1, hipassfre=-1, preemphasise=float(rng.choice([0.0, 0.95, 0.97])),
          wlen=wlen, inc=inc, vecNum=vec, fbankFlag=fbank, bankNum=bank, MFCCNum=int(rng.integers(1, bank + 1)),
          MFCC0thFlag=int(rng.integers(0, 2)), energyFlag=int(rng.integers(0, 2)),
          zeroCrossingFlag=0 if plain else int(rng.integers(0, 2)),
          brightFlag=0 if plain else int(rng.integers(0, 2)),
          subBandEFlag=0 if plain else int(rng.choice([0, 0, 3, 8])),
          fftLength=0 if plain else int(rng.choice([0, 0, 5])),
          regreOrder=order, delwin=int(rng.integers(1, 6)) if order > 1 else 0,
          zeroMeanSigFlag=0 if plain else int(rng.integers(0, 2)), znormFlag=int(rng.integers(0, 2)))
if plain:
    ch, bps, kw["vecNum"] = 1, 2, 1
    if kw["inc"] % 2:
        kw["inc"] += 1
lengths = [int(wlen + kw["inc"] * rng.integers(0, 60) + rng.integers(0, kw["inc"])) for _ in range(3)] + [wlen - 1]
pcms = [tp._pcm(n, ch, bps, seed * 10 + i, sr, tone=0).tobytes() for i, n in enumerate(lengths)]
print(kw, ch, bps)
ocfg = oracle.make_config(**kw)
if len(sys.argv) > 2:  # tuning overrides, e.g. noLean=1
    ma.tuning(**{k: int(v) for k, v in (item.split("=") for item in sys.argv[2].split(","))})
refs = [oracle.extract_pcm_bytes(ocfg, np.frombuffer(p, dtype=np.uint8), ch, bps) for p in pcms]
gots = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), pcms, ch, bps, pad=int(rng.integers(0, 7)))
kinds = np.array(column_kinds(kw) * kw["vecNum"])
t = oracle.fbank_tables(wlen, sr, bank, -1, -1)
for u, (g, r) in enumerate(zip(gots, refs)):
    if r.shape[0] == 0:
        continue
    m = np.where(kinds == "mel")[0]
    if m.size == 0:
        continue
    d = np.abs(g[:, m] - r[:, m])
    i, j = np.unravel_index(np.argmax(d), d.shape)
    c = m[j] % len(column_kinds(kw))
    bins = int(np.sum((t["loChan"] == c) | (t["loChan"] == c + 1)))
    print("utt %d: max |d log-mel| %.3g at frame %d column %d (channel %d of vec %d, %d bins): ref %.6f, row max %.3f" % (
        u, d[i, j], i, m[j], c + 1, m[j] // len(column_kinds(kw)), bins, r[i, m[j]], np.max(r[i, m])))
# where the whole matrix differs most (any column kind, delta blocks included)
for u, (g, r) in enumerate(zip(gots, refs)):
    if r.shape[0] == 0:
        continue
    fin = np.isfinite(r)
    d = np.where(fin, np.abs(g - r), 0.0)
    i, j = np.unravel_index(np.argmax(d), d.shape)
    vs = len(kinds)
    rows = np.unique(np.nonzero(d > 1e-3)[0])
    cols = np.unique(np.nonzero(d > 1e-3)[1])
    print("utt %d (%d rows): max |d| %.3g at row %d column %d (block %d, column %d of it); rows off by > 1e-3: %s; columns: %s" % (
        u, r.shape[0], d[i, j], i, j, j // vs, j % vs, rows[:20], cols[:20]))
