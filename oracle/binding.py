"""ctypes binding of the CPU oracle (oracle/mfcc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (mfcc_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libmfcc_oracle.so")


class OrcConfig(C.Structure):
    _fields_ = [
        ("sampleRate", C.c_int),
        ("hipassfre", C.c_double),
        ("lowpassfre", C.c_double),
        ("preemphasise", C.c_double),
        ("wlen", C.c_int),
        ("inc", C.c_int),
        ("vecNum", C.c_int),
        ("fbankFlag", C.c_int),
        ("bankNum", C.c_int),
        ("MFCCNum", C.c_int),
        ("MFCC0thFlag", C.c_int),
        ("energyFlag", C.c_int),
        ("zeroCrossingFlag", C.c_int),
        ("brightFlag", C.c_int),
        ("subBandEFlag", C.c_int),
        ("fftLength", C.c_int),
        ("regreOrder", C.c_int),
        ("delwin", C.c_int),
        ("zeroMeanSigFlag", C.c_int),
        ("znormFlag", C.c_int),
        ("usePower", C.c_int),
        ("takeLogs", C.c_int),
        ("doubleFFT", C.c_int),
        ("warpAlpha", C.c_double),
        ("warpLowCut", C.c_double),
        ("warpUpCut", C.c_double),
        ("normEnergy", C.c_int),
    ]


class OrcFBank(C.Structure):
    _fields_ = [
        ("frameSize", C.c_int),
        ("numChans", C.c_int),
        ("fftN", C.c_int),
        ("klo", C.c_int),
        ("khi", C.c_int),
        ("sampPeriod", C.c_double),
        ("fres", C.c_double),
        ("mlo", C.c_double),
        ("mhi", C.c_double),
        ("cf", C.POINTER(C.c_double)),
        ("loChan", C.POINTER(C.c_int)),
        ("loWt", C.POINTER(C.c_double)),
        ("usePower", C.c_int),
        ("takeLogs", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with gcc (no-op when the library is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in ("mfcc_oracle.c", "mfcc_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.orc_mel.restype = C.c_double
    L.orc_mel.argtypes = [C.c_int, C.c_double]
    L.orc_fbank_init.argtypes = [C.POINTER(OrcFBank), C.c_int, C.c_double, C.c_int, C.c_double,
                                 C.c_double]
    L.orc_fbank_init_ex.argtypes = [C.POINTER(OrcFBank), C.c_int, C.c_double, C.c_int, C.c_double,
                                    C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                    C.c_double]
    L.orc_normalise_log_energy.argtypes = [dp, C.c_long, C.c_int, C.c_double, C.c_double]
    L.orc_normalise_log_energy.restype = None
    L.orc_fbank_free.argtypes = [C.POINTER(OrcFBank)]
    L.orc_fbank_passband_text.argtypes = [C.POINTER(OrcFBank), C.c_char_p, C.c_size_t]
    L.orc_fft.argtypes = [dp, C.c_int, C.c_int]
    L.orc_fft.restype = None
    L.orc_realft.argtypes = [dp, C.c_int]
    L.orc_realft.restype = None
    L.orc_wave2fbank.argtypes = [dp, dp, dp, dp, C.POINTER(OrcFBank), dp]
    L.orc_wave2fbank.restype = None
    L.orc_fbank2mfcc.argtypes = [dp, C.c_int, dp, C.c_int]
    L.orc_fbank2mfcc.restype = None
    L.orc_fbank2c0.argtypes = [dp, C.c_int]
    L.orc_fbank2c0.restype = C.c_double
    L.orc_preemphasise.argtypes = [dp, C.c_long, C.c_double]
    L.orc_preemphasise.restype = None
    L.orc_hamming.argtypes = [dp, C.c_int]
    L.orc_hamming.restype = None
    L.orc_zcr.argtypes = [dp, C.c_int]
    L.orc_zcr.restype = C.c_double
    L.orc_brightness.argtypes = [dp, C.c_int]
    L.orc_brightness.restype = C.c_double
    L.orc_subband.argtypes = [dp, C.c_int, dp, C.c_int]
    L.orc_subband.restype = None
    L.orc_regress.argtypes = [dp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int]
    L.orc_regress.restype = None
    L.orc_znorm.argtypes = [dp, C.c_int, C.c_long, C.c_int]
    L.orc_znorm.restype = None
    L.orc_num_frames.argtypes = [C.c_long, C.c_int, C.c_int]
    L.orc_num_frames.restype = C.c_long
    L.orc_feature_dim.argtypes = [C.POINTER(OrcConfig)]
    L.orc_step.argtypes = [C.POINTER(OrcConfig)]
    L.orc_extract.argtypes = [C.POINTER(OrcConfig), C.POINTER(C.c_int32), C.c_int, C.c_long, dp]
    L.orc_extract_pcm_bytes.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_int, C.c_int,
                                        C.c_long, dp]
    L.orc_extract_batch.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, dp, C.c_int]
    L.orc_wav_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_size_t)]
    L.orc_write_text.argtypes = [C.c_char_p, dp, C.c_long, C.c_int, C.c_char]
    _lib = L
    return L


_CFG_DEFAULTS = dict(
    sampleRate=16000, hipassfre=-1.0, lowpassfre=-1.0, preemphasise=0.0, wlen=400, inc=160,
    vecNum=1, fbankFlag=1, bankNum=40, MFCCNum=13, MFCC0thFlag=0, energyFlag=0,
    zeroCrossingFlag=0, brightFlag=0, subBandEFlag=0, fftLength=0, regreOrder=1, delwin=2,
    zeroMeanSigFlag=0, znormFlag=0, usePower=1, takeLogs=1, doubleFFT=0, warpAlpha=1.0,
    warpLowCut=0.0, warpUpCut=0.0, normEnergy=0)


def make_config(**kw):
    vals = dict(_CFG_DEFAULTS)
    for k, v in kw.items():
        if k not in vals:
            raise KeyError(k)
        vals[k] = v
    return OrcConfig(**vals)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def num_frames(n, wlen, inc):
    return int(lib().orc_num_frames(n, wlen, inc))


def step(cfg):
    return int(lib().orc_step(C.byref(cfg)))


def fbank_tables(frame_size, sample_rate, num_chans, lopass, hipass):
    """Returns dict with fftN, klo, khi, mlo, mhi, cf, loChan, loWt and the pass-band text."""
    fb = OrcFBank()
    rc = lib().orc_fbank_init(C.byref(fb), frame_size, 1e7 / float(sample_rate), num_chans,
                              float(lopass), float(hipass))
    assert rc == 0
    half = fb.fftN // 2
    buf = C.create_string_buffer(256)
    lib().orc_fbank_passband_text(C.byref(fb), buf, 256)
    out = dict(
        fftN=fb.fftN, klo=fb.klo, khi=fb.khi, mlo=fb.mlo, mhi=fb.mhi, fres=fb.fres,
        cf=np.array([fb.cf[i] for i in range(num_chans + 1)]),
        loChan=np.array([fb.loChan[i] for i in range(half)], dtype=np.int32),
        loWt=np.array([fb.loWt[i] for i in range(half)]),
        text=buf.value.decode())
    lib().orc_fbank_free(C.byref(fb))
    return out


def realft(x):
    s = np.ascontiguousarray(x, dtype=np.float64).copy()
    lib().orc_realft(_dptr(s), s.size)
    return s


def fft(x, invert=0):
    s = np.ascontiguousarray(x, dtype=np.float64).copy()
    lib().orc_fft(_dptr(s), s.size, invert)
    return s


def wave2fbank(frame, sample_rate, num_chans, lopass, hipass, use_power=1, take_logs=1):
    """Returns (fbank, te, te2, packed spectrum).  use_power / take_logs: FBankInfo's fields
    (mfcc.c:154-157,171-176)."""
    fb = OrcFBank()
    s = np.ascontiguousarray(frame, dtype=np.float64)
    lib().orc_fbank_init(C.byref(fb), s.size, 1e7 / float(sample_rate), num_chans, float(lopass),
                         float(hipass))
    fb.usePower, fb.takeLogs = int(use_power), int(take_logs)
    fbank = np.zeros(num_chans)
    x = np.zeros(fb.fftN)
    te = C.c_double()
    te2 = C.c_double()
    lib().orc_wave2fbank(_dptr(s), _dptr(fbank), C.byref(te), C.byref(te2), C.byref(fb), _dptr(x))
    lib().orc_fbank_free(C.byref(fb))
    return fbank, te.value, te2.value, x


def fbank2mfcc(fbank, n):
    f = np.ascontiguousarray(fbank, dtype=np.float64)
    c = np.zeros(n)
    lib().orc_fbank2mfcc(_dptr(f), f.size, _dptr(c), n)
    return c, float(lib().orc_fbank2c0(_dptr(f), f.size))


def regress(data, v_size, step_, offset, delwin):
    d = np.ascontiguousarray(data, dtype=np.float64).copy()
    n = d.size // step_
    lib().orc_regress(_dptr(d), v_size, n, step_, offset, delwin)
    return d


def znorm(data, step_):
    d = np.ascontiguousarray(data, dtype=np.float64).copy()
    lib().orc_znorm(_dptr(d), step_, d.size // step_, step_)
    return d


def extract(cfg, pcm):
    """pcm: int array [channels, samples] of raw integer PCM.  Returns float64 [T, step]."""
    p = np.ascontiguousarray(pcm, dtype=np.int32)
    if p.ndim == 1:
        p = p[None, :]
    nch, n = p.shape
    T = num_frames(n, cfg.wlen, cfg.inc)
    st = step(cfg)
    out = np.zeros((max(T, 0), st))
    rc = lib().orc_extract(C.byref(cfg), p.ctypes.data_as(C.POINTER(C.c_int32)), nch, n,
                           _dptr(out) if out.size else None)
    if rc != 0:
        raise ValueError("orc_extract failed: %d" % rc)
    return out


def extract_pcm_bytes(cfg, raw, channels, bytes_per_sample):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = raw.size // (channels * bytes_per_sample)
    T = num_frames(n, cfg.wlen, cfg.inc)
    out = np.zeros((max(T, 0), step(cfg)))
    rc = lib().orc_extract_pcm_bytes(C.byref(cfg), raw.ctypes.data, channels, bytes_per_sample, n,
                                     _dptr(out) if out.size else None)
    if rc != 0:
        raise ValueError("orc_extract_pcm_bytes failed: %d" % rc)
    return out


def extract_batch(cfg, raw, channels, bytes_per_sample, byte_off, num_samples, threads=0):
    """Thread-per-file batch.  Returns (list of [T, step] views, threads used)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    byte_off = np.ascontiguousarray(byte_off, dtype=np.int64)
    num_samples = np.ascontiguousarray(num_samples, dtype=np.int64)
    st = step(cfg)
    frames = np.array([num_frames(int(n), cfg.wlen, cfg.inc) for n in num_samples], dtype=np.int64)
    out_off = np.concatenate([[0], np.cumsum(frames * st)]).astype(np.int64)
    out = np.zeros(int(out_off[-1]))
    used = lib().orc_extract_batch(C.byref(cfg), raw.ctypes.data, channels, bytes_per_sample,
                                   len(num_samples), byte_off.ctypes.data, num_samples.ctypes.data,
                                   out_off.ctypes.data, _dptr(out), threads)
    views = [out[out_off[u]:out_off[u + 1]].reshape(-1, st) for u in range(len(num_samples))]
    return views, int(used)


def wav_parse(data):
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    ch, sr, bps = C.c_int(), C.c_int(), C.c_int()
    n = C.c_long()
    off = C.c_size_t()
    rc = lib().orc_wav_parse(b.ctypes.data, b.size, C.byref(ch), C.byref(sr), C.byref(bps),
                             C.byref(n), C.byref(off))
    if rc != 0:
        raise ValueError("orc_wav_parse failed: %d" % rc)
    return dict(channels=ch.value, sampleRate=sr.value, bytesPerSample=bps.value,
                numSamples=n.value, dataOffset=off.value)


def write_text(path, data, step_, save_type):
    d = np.ascontiguousarray(data, dtype=np.float64).ravel()
    rc = lib().orc_write_text(path.encode(), _dptr(d), d.size, step_, save_type.encode())
    if rc != 0:
        raise IOError(path)
