"""Debug aid: rerun one seed of tests/test_gpu_parity.py::test_random_configurations and print where
the product and the oracle differ (GPU box, repo root: python tools/dev_fuzz.py <seed>)."""
import sys
import numpy as np
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mfcc_amd as ma
from oracle import binding as ob
from util import column_kinds, run_product_batch
import test_gpu_parity as T

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
print("ch", ch, "bps", bps, kw, lengths)
pcms = [T._pcm(n, ch, bps, seed * 10 + i, sr, tone=0).tobytes() for i, n in enumerate(lengths)]
ob.build()
ocfg = ob.make_config(**kw)
refs = [ob.extract_pcm_bytes(ocfg, np.frombuffer(p, dtype=np.uint8), ch, bps) for p in pcms]
kinds = column_kinds(kw) * kw["vecNum"]
for f64 in (False, True):
    cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
    pad = int(rng.integers(0, 7))
    gots = run_product_batch(ma, cfg, pcms, ch, bps, pad=pad)
    for u, (g, r) in enumerate(zip(gots, refs)):
        if g.size == 0:
            continue
        d = np.abs(np.where(np.isfinite(r), g - r, 0.0)) / np.maximum(np.abs(np.where(np.isfinite(r), r, 0)), 1.0)
        worst = np.unravel_index(np.argmax(d), d.shape)
        col = worst[1]
        print("f64" if f64 else "f32", "pad", pad, "utt", u, g.shape, "max rel", d.max(), "at", worst,
              "kind", kinds[col % len(kinds)], "block", col // len(kinds), "got", g[worst], "ref", r[worst],
              "nan mismatch", int((np.isnan(g) != np.isnan(r)).sum()))
