import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mfcc_amd as ma
from oracle import binding as ob
import util
for name in ['c2','c3','c4','c1a','c1b','c5']:
    kw = util.CONFIGS[name]
    sr = kw['sampleRate']; ch = 1 if kw['vecNum']==1 else 2; bps = 2 if sr==16000 else 3
    n = sr//2 + 37
    pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000+7, sample_rate=sr, tone_hz=440)
    ocfg = ob.make_config(**kw)
    ref = ob.extract_pcm_bytes(ocfg, pcm, ch, bps)
    for prec in (1,0):
        cfg = ma.make_config(precision=prec, **kw)
        got = util.run_product_batch(ma, cfg, [pcm.tobytes()], ch, bps)[0]
        fin = np.isfinite(ref)&np.isfinite(got)
        err = np.abs(got-ref)[fin]
        print(name, 'f64' if prec else 'f32', got.shape, ref.shape, 'max abs err', err.max() if err.size else None, 'argmax col', np.unravel_index(np.argmax(np.abs(np.where(fin,got-ref,0))), got.shape), 'nan mismatch', (np.isnan(ref)!=np.isnan(got)).sum())
