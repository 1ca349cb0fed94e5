#!/usr/bin/env python3
"""One hand-made case against the oracle: prints the places that differ.
usage: tools/dev_case.py key=value ...   (config keys; ch, bps, n, prec=f32|f64, pad)"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from util import run_product_batch  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.95, wlen=2048, inc=512, vecNum=1, fbankFlag=0,
          bankNum=39, MFCCNum=39, MFCC0thFlag=0, energyFlag=0, zeroCrossingFlag=1, brightFlag=1, subBandEFlag=8,
          fftLength=5, regreOrder=4, delwin=2, zeroMeanSigFlag=0, znormFlag=0)
extra = dict(ch=1, bps=3, n=2048 + 512 * 26 + 100, prec="f64", pad=0)
for item in sys.argv[1:]:
    k, v = item.split("=")
    if k in extra:
        extra[k] = v if k == "prec" else int(v)
    elif k in kw:
        kw[k] = type(kw[k])(float(v)) if isinstance(kw[k], float) else int(v)
    else:
        ma.tuning(**{k: int(v)})
ch, bps, n = extra["ch"], extra["bps"], extra["n"]
pcm = tp._pcm(n, ch, bps, 5, kw["sampleRate"], tone=0).tobytes()
ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), np.frombuffer(pcm, dtype=np.uint8), ch, bps)
got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F64 if extra["prec"] == "f64" else ma.PREC_F32, **kw), [pcm], ch, bps,
                        pad=extra["pad"])[0]
bad = np.abs(got - ref) > 1e-3 * (1 + np.abs(ref))
print(sys.argv[1:], "rows", ref.shape, "bad elements", int(bad.sum()), "max |diff| %.3g" % np.nanmax(np.abs(np.where(np.isfinite(ref), got - ref, 0))))
for r in np.where(bad.any(axis=1))[0]:
    print("  row", r, "cols", np.where(bad[r])[0].tolist())
