import time, numpy as np, sys
import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS
kw = CONFIGS["c3"]
cfg = ma.make_config(precision=ma.PREC_F32, **kw)
pcm = ma.synth_pcm(160000, 1, 2, seed=3, sample_rate=16000, tone_hz=0)
wav = ma.wav_bytes(pcm, 1, 2, 16000)
ma.wapper_mem(cfg, wav)
t0 = time.perf_counter()
for _ in range(200):
    f = ma.wapper_mem(cfg, wav)
dt = (time.perf_counter() - t0) / 200
print("MFCCWapperMem 10 s file: %.3f ms per call, %d x %d" % (dt * 1e3, f.shape[0], f.shape[1]))
