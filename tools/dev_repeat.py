#!/usr/bin/env python3
"""GPU box: one fuzz seed of test_random_configurations many times over; every run's matrices must equal the first run's bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma
from util import run_product_batch
import test_gpu_parity as tp
seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if len(sys.argv) > 3:
    ma.tuning(**{k: int(v) for k, v in (item.split("=") for item in sys.argv[3].split(","))})
rng = np.random.default_rng(1000 + seed)
sr = int(rng.choice([8000, 16000, 44100])); wlen = int(rng.choice([64, 200, 256, 400, 441, 512, 800, 1024, 1764, 2048]))
inc = int(max(1, wlen // int(rng.choice([2, 3, 4])) + int(rng.integers(-1, 2)))); ch = int(rng.choice([1, 2])); bps = int(rng.choice([2, 2, 3, 4]))
vec = int(rng.integers(1, 5)) if ch == 2 else 1; bank = int(rng.integers(3, 50)); fbank = int(rng.integers(0, 2)); order = int(rng.choice([1, 1, 2, 3, 4]))
plain = bool(rng.integers(0, 2))
kw = dict(sampleRate=sr, lowpassfre=-1, hipassfre=-1, preemphasise=float(rng.choice([0.0, 0.95, 0.97])), wlen=wlen, inc=inc, vecNum=vec, fbankFlag=fbank,
          bankNum=bank, MFCCNum=int(rng.integers(1, bank + 1)), MFCC0thFlag=int(rng.integers(0, 2)), energyFlag=int(rng.integers(0, 2)),
          zeroCrossingFlag=0 if plain else int(rng.integers(0, 2)), brightFlag=0 if plain else int(rng.integers(0, 2)),
          subBandEFlag=0 if plain else int(rng.choice([0, 0, 3, 8])), fftLength=0 if plain else int(rng.choice([0, 0, 5])),
          regreOrder=order, delwin=int(rng.integers(1, 6)) if order > 1 else 0, zeroMeanSigFlag=0 if plain else int(rng.integers(0, 2)), znormFlag=int(rng.integers(0, 2)))
if plain:
    ch, bps, kw["vecNum"] = 1, 2, 1
    if kw["inc"] % 2: kw["inc"] += 1
lengths = [int(wlen + kw["inc"] * rng.integers(0, 60) + rng.integers(0, kw["inc"])) for _ in range(3)] + [wlen - 1]
pcms = [tp._pcm(n, ch, bps, seed * 10 + i, sr, tone=0).tobytes() for i, n in enumerate(lengths)]
pad = int(rng.integers(0, 7))
vs = len(tp.column_kinds(kw)) * kw["vecNum"]


def run_nan(cfg):
    """As util.run_product_batch, but the output matrix starts as NaN: elements no kernel writes stay NaN."""
    import torch
    ctx = ma.Context(cfg, 0)
    offs, total = [], pad
    for p in pcms:
        offs.append(total)
        total += len(p) + pad
    buf = np.zeros(max(total, 16), dtype=np.uint8)
    for o, p in zip(offs, pcms):
        buf[o:o + len(p)] = np.frombuffer(bytes(p), dtype=np.uint8)
    ns = np.array([len(p) // (ch * bps) for p in pcms], dtype=np.int64)
    plan = ma.Plan(ctx, ch, bps, np.array(offs, dtype=np.int64), ns)
    dev = torch.from_numpy(buf).to("cuda:0")
    out = plan.alloc_output()
    out.fill_(float("nan"))
    plan.extract(dev, out)
    torch.cuda.synchronize()
    host = out.double().cpu().numpy()
    res = [host[plan.row_offset(u):plan.row_offset(u) + plan.frames(u)] for u in range(len(ns))]
    nans = [int(np.isnan(r).sum()) for r in res]
    if any(nans):
        print("NaN left in", nans)
    plan.close(); ctx.close()
    return res

bad = 0
for prec in (ma.PREC_F32, ma.PREC_F64):
    cfg = ma.make_config(precision=prec, **kw)
    first = None
    for it in range(reps):
        got = run_nan(cfg)
        if first is None:
            first = got
            continue
        for u, (a, b) in enumerate(zip(first, got)):
            d = (a != b) & ~(np.isnan(a) & np.isnan(b))
            if d.any():
                bad += 1
                r, c = np.argwhere(d)[0]
                if bad <= 3:
                    print("prec %d run %d utt %d (%d rows): %d elements differ, first at row %d column %d (block %d): %g vs %g; rows %s" % (
                        prec, it, u, a.shape[0], int(d.sum()), r, c, c // vs, a[r, c], b[r, c], np.unique(np.nonzero(d)[0])[:12]))
                    for rr in np.unique(np.nonzero(d)[0]):
                        print("   row", rr, "columns", np.nonzero(d[rr])[0] % vs)
print("seed", seed, kw, "runs that differed:", bad)
