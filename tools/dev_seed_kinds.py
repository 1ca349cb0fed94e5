#!/usr/bin/env python3
"""One seed of tests/test_gpu_parity.py::test_random_configurations, both precisions: largest
difference per column kind and block (static, delta, delta-delta), with its place.
usage: tools/dev_seed_kinds.py <seed> [tuning key=value,...]"""
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
pad = int(rng.integers(0, 7))
print(kw, "channels", ch, "bytes", bps, "lengths", lengths, "pad", pad)
if seed % 3 == 2:
    ma.tuning(noUttFinish=1)
if seed % 3 == 1:
    ma.tuning(finish=2, streamRows=(0, 16, 8)[seed % 9 // 3])
if len(sys.argv) > 2:
    ma.tuning(**{k: int(v) for k, v in (item.split("=") for item in sys.argv[2].split(","))})
ocfg = oracle.make_config(**kw)
refs = [oracle.extract_pcm_bytes(ocfg, np.frombuffer(p, dtype=np.uint8), ch, bps) for p in pcms]
kinds = np.array(column_kinds(kw) * kw["vecNum"])
vs = len(kinds)
for prec, name in ((ma.PREC_F32, "f32"), (ma.PREC_F64, "f64")):
    gots = run_product_batch(ma, ma.make_config(precision=prec, **kw), pcms, ch, bps, pad=pad)
    for u, (g, r) in enumerate(zip(gots, refs)):
        if r.shape[0] == 0:
            continue
        for o in range(min(kw["regreOrder"], 3)):
            d = np.abs(g[:, o * vs:(o + 1) * vs] - r[:, o * vs:(o + 1) * vs])
            d = np.where(np.isfinite(d), d, np.inf)
            for kind in sorted(set(kinds)):
                m = np.where(kinds == kind)[0]
                i, j = np.unravel_index(np.argmax(d[:, m]), d[:, m].shape)
                print("%s utt %d (%d rows) block %d %-7s max |diff| %.3g at row %d col %d (got %.6g, ref %.6g)" %
                      (name, u, r.shape[0], o, kind, d[i, m[j]], i, m[j], g[i, o * vs + m[j]], r[i, o * vs + m[j]]))
if os.environ.get("DUMP_ROW"):
    u, i = (int(x) for x in os.environ["DUMP_ROW"].split(","))
    np.set_printoptions(linewidth=200, precision=5, suppress=False)
    print("got", gots[u][i, :vs])
    print("ref", refs[u][i, :vs])
    print("bad cols", np.where(np.abs(gots[u][i, :vs] - refs[u][i, :vs]) > 1e-3 * (1 + np.abs(refs[u][i, :vs])))[0])
    for j in range(max(0, i - 3), i + 1):
        print("row", j, "bad cols", np.where(np.abs(gots[u][j, :vs] - refs[u][j, :vs]) > 1e-3 * (1 + np.abs(refs[u][j, :vs])))[0])
