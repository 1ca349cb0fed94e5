"""mfcc_amd -- Python host-side mirror of the C ABI in include/mfcc_amd.h.

The product is libmfcc_amd.so (HIP kernels for gfx950 + C host code).  This package only binds it
with ctypes and uses torch for device memory, streams and torch.distributed; it contains no
feature arithmetic of its own and has no CPU fallback: if the library is missing, or no HIP
device is present when an extraction is requested, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFCC_AMD_LIB lets the profiling tools load a diagnostic build of the same library
LIB_PATH = os.environ.get("MFCC_AMD_LIB") or os.path.join(_HERE, "libmfcc_amd.so")

MAX_PATH = 600
SAVE_F, SAVE_E, SAVE_N, SAVE_B = 0, 1, 2, 3
PREC_F32, PREC_F64 = 0, 1

E_INI_OPEN, E_INI_KEY, E_OPEN, E_WAV, E_SAMPLE_RATE = -1, -2, -3, -4, -5
E_VECNUM, E_WRITE, E_CONFIG, E_DEVICE, E_ARG, E_NOMEM = -6, -7, -8, -9, -10, -11


class MFCCConfig(C.Structure):
    """Field-for-field image of `struct MFCCConfig` (which replaces main.c:41-65)."""
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
        ("saveType", C.c_int),
        ("numThreads", C.c_int),
        ("precision", C.c_int),
        ("fileList", C.c_char * MAX_PATH),
        ("numDevices", C.c_int),
        ("usePower", C.c_int),
        ("takeLogs", C.c_int),
        ("doubleFFT", C.c_int),
        ("warpAlpha", C.c_double),
        ("warpLowCut", C.c_double),
        ("warpUpCut", C.c_double),
        ("normEnergy", C.c_int),
    ]


class MFCCCounters(C.Structure):
    """include/mfcc_amd.h: MFCCCounters."""
    _fields_ = [("extractCalls", C.c_uint64), ("frames", C.c_uint64), ("pcmBytes", C.c_uint64), ("featureBytes", C.c_uint64),
                ("filesOk", C.c_uint64), ("filesFailed", C.c_uint64), ("listSeconds", C.c_double), ("recordingSeconds", C.c_double)]


def counters(reset=False):
    """The library's work counters as a dict (MFCCCountersGet); reset=True zeroes them afterwards."""
    c = MFCCCounters()
    lib().MFCCCountersGet(C.byref(c))
    if reset:
        lib().MFCCCountersReset()
    return {name: getattr(c, name) for name, _ in MFCCCounters._fields_}


class MFCCError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mfcc_amd error %d: %s" % (code, message))
        self.code = code


_lib = None

# every symbol include/mfcc_amd.h declares
EXPORTS = [
    "MFCCLastError", "MFCCFree", "MFCCConfigDefaults", "MFCCConfigFromIni", "MFCCNumFrames",
    "MFCCFeatureDim", "MFCCStep", "MFCCWapper", "MFCCWapperMem", "MFCCWapperList",
    "MFCCWriteFeatures", "MFCCWavInfo", "MFCCHostTables", "MFCCContextCreate", "MFCCContextDestroy",
    "MFCCContextPassbandText", "MFCCContextTables", "MFCCPlanCreate", "MFCCPlanCreateShard",
    "MFCCPlanDestroy", "MFCCShardGeometry", "MFCCPlanTotalRows", "MFCCPlanRowOffset", "MFCCPlanNumFrames",
    "MFCCPlanSampleBegin", "MFCCPlanSampleCount", "MFCCPlanOwnedRowBegin", "MFCCPlanOwnedRows",
    "MFCCExtractDevice", "MFCCExtractRecording", "MFCCColumnSumsNoAlloc", "MFCCColumnSumsNoAllocWarmup", "MFCCShardExchange", "MFCCCountersGet", "MFCCCountersReset", "MFCCShardColumnSums", "MFCCShardNormalize", "MFCCChannelSums",
    "MFCCPlanEnableTiming", "MFCCPlanLastKernelMs", "MFCCDebugTuning", "MFCCSynthPcm", "MFCCSynthPcmAt", "MFCCFormatValue",
    "MFCCFFTDevice", "MFCCRealftDevice", "MFCCWave2FBankDevice", "MFCCFBank2MFCCDevice", "MFCCMel",
    "MFCCFBankTables", "MFCCWave2FBankTablesDevice", "MFCCFBank2MFCCDirectDevice",
]


def lib():
    """Loads libmfcc_amd.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C mfcc_amd/csrc` "
                          "(or __graft_entry__.build())" % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; load it first so that this library binds to the
    # same HIP runtime instance (two runtimes in one process do not both see the GPU).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i64p, dp = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)
    cfgp = C.POINTER(MFCCConfig)
    L.MFCCLastError.restype = C.c_char_p
    L.MFCCFree.argtypes = [vp]
    L.MFCCFree.restype = None
    L.MFCCConfigDefaults.argtypes = [cfgp]
    L.MFCCConfigFromIni.argtypes = [C.c_char_p, cfgp, C.POINTER(C.c_int)]
    L.MFCCNumFrames.argtypes = [C.c_long, C.c_int, C.c_int]
    L.MFCCNumFrames.restype = C.c_long
    L.MFCCFeatureDim.argtypes = [cfgp]
    L.MFCCStep.argtypes = [cfgp]
    L.MFCCWapper.argtypes = [cfgp, C.c_char_p, C.c_char_p, C.POINTER(dp), C.POINTER(C.c_long),
                             C.POINTER(C.c_int)]
    L.MFCCWapperMem.argtypes = [cfgp, vp, C.c_size_t, C.c_char_p, C.POINTER(dp),
                                C.POINTER(C.c_long), C.POINTER(C.c_int)]
    L.MFCCWapperList.argtypes = [C.c_char_p]
    L.MFCCWriteFeatures.argtypes = [C.c_char_p, dp, C.c_long, C.c_int, C.c_int]
    L.MFCCWavInfo.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                              C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_size_t)]
    L.MFCCHostTables.argtypes = [cfgp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), dp,
                                 C.POINTER(C.c_int), dp, dp, C.c_char_p, C.c_size_t]
    L.MFCCContextCreate.argtypes = [cfgp, C.c_int, C.POINTER(vp)]
    L.MFCCContextDestroy.argtypes = [vp]
    L.MFCCContextDestroy.restype = None
    L.MFCCContextPassbandText.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.MFCCContextTables.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    dp, C.POINTER(C.c_int), dp]
    L.MFCCPlanCreate.argtypes = [vp, C.c_int, C.c_int, C.c_int, i64p, i64p, C.POINTER(vp)]
    L.MFCCPlanCreateShard.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, i64p,
                                      C.POINTER(vp)]
    L.MFCCPlanDestroy.argtypes = [vp]
    L.MFCCPlanDestroy.restype = None
    L.MFCCShardGeometry.argtypes = [cfgp, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p,
                                    i64p, C.POINTER(C.c_int)]
    for name in ("MFCCPlanTotalRows", "MFCCPlanSampleBegin", "MFCCPlanSampleCount",
                 "MFCCPlanOwnedRowBegin", "MFCCPlanOwnedRows"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_int64
    for name in ("MFCCPlanRowOffset", "MFCCPlanNumFrames"):
        getattr(L, name).argtypes = [vp, C.c_int]
        getattr(L, name).restype = C.c_int64
    L.MFCCExtractDevice.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
    L.MFCCShardColumnSums.argtypes = [vp, vp, vp, vp, vp]
    L.MFCCExtractRecording.argtypes = [cfgp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, C.POINTER(C.c_int64)]
    L.MFCCColumnSumsNoAlloc.argtypes = [C.c_int, vp, C.c_int64, C.c_int, vp, vp]
    L.MFCCColumnSumsNoAllocWarmup.argtypes = []
    L.MFCCShardExchange.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.MFCCCountersGet.argtypes = [vp]
    L.MFCCCountersGet.restype = None
    L.MFCCCountersReset.argtypes = []
    L.MFCCCountersReset.restype = None
    L.MFCCShardNormalize.argtypes = [vp, vp, vp, vp, vp]
    L.MFCCChannelSums.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp]
    L.MFCCPlanEnableTiming.argtypes = [vp, C.c_int]
    L.MFCCDebugTuning.argtypes = [C.c_char_p, C.c_int]
    L.MFCCPlanLastKernelMs.argtypes = [vp]
    L.MFCCPlanLastKernelMs.restype = C.c_float
    L.MFCCSynthPcm.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_int,
                               C.c_int]
    L.MFCCSynthPcmAt.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_int]
    L.MFCCFormatValue.argtypes = [C.c_double, C.c_int, C.c_char_p, C.c_size_t]
    L.MFCCFFTDevice.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    L.MFCCRealftDevice.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp]
    L.MFCCWave2FBankDevice.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.MFCCFBank2MFCCDevice.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.MFCCMel.argtypes = [C.c_int, C.c_double]
    L.MFCCMel.restype = C.c_double
    L.MFCCFBankTables.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                  C.c_double, C.c_double, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp, C.POINTER(C.c_int), dp,
                                  C.c_char_p, C.c_size_t]
    L.MFCCWave2FBankTablesDevice.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                             vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.MFCCFBank2MFCCDirectDevice.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    _lib = L
    return L


COMPAT_LIB_PATH = os.path.join(_HERE, "libmfcc_compat.so")
# every function include/mfcc_compat.h declares
COMPAT_EXPORTS = ["Mel", "InitFBank", "Wave2FBank", "FBank2MFCC", "FBank2C0", "FFT", "Realft",
                  "MFCCCreateVector", "MFCCCreateIntVec", "MFCCVectorSize", "MFCCFreeVector", "MFCCFreeFBank",
                  "MFCCCompatStatus"]


class FBankInfo(C.Structure):
    """mfcc.h:14-27 as include/mfcc_compat.h declares it."""
    _fields_ = [("frameSize", C.c_int), ("numChans", C.c_int), ("sampPeriod", C.c_double), ("fftN", C.c_int),
                ("klo", C.c_int), ("khi", C.c_int), ("usePower", C.c_int), ("takeLogs", C.c_int),
                ("fres", C.c_double), ("cf", C.POINTER(C.c_double)), ("loChan", C.POINTER(C.c_int)),
                ("loWt", C.POINTER(C.c_double)), ("x", C.POINTER(C.c_double))]


_compat = None


def compat_lib():
    """libmfcc_compat.so: the reference's own function names (mfcc.h:29-74) on 1-based vectors."""
    global _compat
    if _compat is not None:
        return _compat
    lib()  # the main library (and torch's HIP runtime) first
    if not os.path.exists(COMPAT_LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C mfcc_amd/csrc`" % COMPAT_LIB_PATH)
    L = C.CDLL(COMPAT_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.Mel.argtypes = [C.c_int, C.c_double]
    L.Mel.restype = C.c_double
    L.InitFBank.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                            C.c_double, C.c_double, C.c_double]
    L.InitFBank.restype = FBankInfo
    L.Wave2FBank.argtypes = [dp, dp, dp, dp, FBankInfo]
    L.Wave2FBank.restype = None
    L.FBank2MFCC.argtypes = [dp, dp, C.c_int]
    L.FBank2MFCC.restype = None
    L.FBank2C0.argtypes = [dp]
    L.FBank2C0.restype = C.c_double
    L.FFT.argtypes = [dp, C.c_int]
    L.FFT.restype = None
    L.Realft.argtypes = [dp]
    L.Realft.restype = None
    L.MFCCCreateVector.argtypes = [C.c_int]
    L.MFCCCreateVector.restype = dp
    L.MFCCCreateIntVec.argtypes = [C.c_int]
    L.MFCCCreateIntVec.restype = C.POINTER(C.c_int)
    L.MFCCVectorSize.argtypes = [C.c_void_p]
    L.MFCCFreeVector.argtypes = [C.c_void_p]
    L.MFCCFreeVector.restype = None
    L.MFCCFreeFBank.argtypes = [C.POINTER(FBankInfo)]
    L.MFCCFreeFBank.restype = None
    _compat = L
    return L


def _check(rc):
    if rc != 0:
        raise MFCCError(rc, lib().MFCCLastError().decode(errors="replace"))


def tuning(**kw):
    """MFCCDebugTuning: overrides of the plan heuristics for plans created afterwards (tuning() with
    no arguments resets all)."""
    if not kw:
        _check(lib().MFCCDebugTuning(None, 0))
    for k, v in kw.items():
        _check(lib().MFCCDebugTuning(k.encode(), int(v)))


def make_config(**kw):
    """MFCCConfigDefaults() overridden by keyword arguments named like the ini keys."""
    cfg = MFCCConfig()
    _check(lib().MFCCConfigDefaults(C.byref(cfg)))
    names = {f[0] for f in MFCCConfig._fields_}
    for k, v in kw.items():
        if k not in names:
            raise KeyError(k)
        if k == "fileList":
            v = v.encode() if isinstance(v, str) else v
        setattr(cfg, k, v)
    return cfg


def config_from_ini(path):
    cfg = MFCCConfig()
    bad = C.c_int(0)
    rc = lib().MFCCConfigFromIni(os.fsencode(path), C.byref(cfg), C.byref(bad))
    if rc != 0:
        err = MFCCError(rc, lib().MFCCLastError().decode(errors="replace"))
        err.bad_line = bad.value
        raise err
    return cfg


def num_frames(num_samples, wlen, inc):
    return int(lib().MFCCNumFrames(num_samples, wlen, inc))


def feature_dim(cfg):
    return int(lib().MFCCFeatureDim(C.byref(cfg)))


def step(cfg):
    return int(lib().MFCCStep(C.byref(cfg)))


def wav_info(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    ch, sr, bps = C.c_int(), C.c_int(), C.c_int()
    n, off = C.c_long(), C.c_size_t()
    _check(lib().MFCCWavInfo(buf.ctypes.data, buf.size, C.byref(ch), C.byref(sr), C.byref(bps),
                             C.byref(n), C.byref(off)))
    return dict(channels=ch.value, sampleRate=sr.value, bytesPerSample=bps.value,
                numSamples=n.value, dataOffset=off.value)


def write_features(path, data, save_type):
    a = np.ascontiguousarray(data, dtype=np.float64)
    rows, st = a.shape
    _check(lib().MFCCWriteFeatures(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_double)),
                                   rows, st, save_type))


def format_value(x, save_type):
    """One value as the text writers print it ("%f" for SAVE_F, "%e" otherwise)."""
    buf = C.create_string_buffer(400)
    n = lib().MFCCFormatValue(float(x), save_type, buf, 400)
    if n < 0:
        _check(n)
    return buf.raw[:n].decode()


def _prec_of(t):
    import torch
    if t.dtype == torch.float64:
        return PREC_F64
    if t.dtype == torch.float32:
        return PREC_F32
    raise TypeError("float32 or float64 tensor expected")


def fft_device(data, invert=False):
    """In-place batched FFT (mfcc.c:211-259) of a [n, 2*nc] device tensor of interleaved (re, im)."""
    import torch
    assert data.is_contiguous() and data.dim() == 2
    st = torch.cuda.current_stream(data.device).cuda_stream
    _check(lib().MFCCFFTDevice(_prec_of(data), data.data_ptr(), data.shape[0], data.shape[1], int(invert),
                               C.c_void_p(st)))
    return data


def realft_device(data):
    """In-place batched Realft (mfcc.c:262-291) of a [n, fftN] device tensor."""
    import torch
    assert data.is_contiguous() and data.dim() == 2
    st = torch.cuda.current_stream(data.device).cuda_stream
    _check(lib().MFCCRealftDevice(_prec_of(data), data.data_ptr(), data.shape[0], data.shape[1], C.c_void_p(st)))
    return data


def mel(k, fres):
    return float(lib().MFCCMel(int(k), float(fres)))


def synth_pcm(num_samples, channels=1, bytes_per_sample=2, seed=0x5EED0000, sample_rate=16000,
              tone_hz=0):
    """Interleaved little-endian PCM bytes (uint8 array) from the library's integer generator;
    channel ch uses seed + ch."""
    out = np.zeros(num_samples * channels * bytes_per_sample, dtype=np.uint8)
    for ch in range(channels):
        _check(lib().MFCCSynthPcm(out.ctypes.data, channels, ch, bytes_per_sample, num_samples,
                                  seed + ch, sample_rate, tone_hz))
    return out


def wav_bytes(pcm, channels, bytes_per_sample, sample_rate, extra_chunk=None):
    """Wraps interleaved PCM bytes in a canonical 44-byte RIFF/WAVE header (test helper)."""
    import struct
    pcm = bytes(pcm)
    fmt = struct.pack("<HHIIHH", 1, channels, sample_rate,
                      sample_rate * channels * bytes_per_sample, channels * bytes_per_sample,
                      8 * bytes_per_sample)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt
    if extra_chunk is not None:
        tag, payload = extra_chunk
        body += tag + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    body += b"data" + struct.pack("<I", len(pcm)) + pcm
    return b"RIFF" + struct.pack("<I", len(body)) + body


def wapper(cfg, wav_path, out_path=None, want_features=True):
    """MFCCWapper: returns the [numFrames, step] float64 matrix (or None)."""
    feat = C.POINTER(C.c_double)()
    nf, st = C.c_long(), C.c_int()
    _check(lib().MFCCWapper(C.byref(cfg), os.fsencode(wav_path),
                            os.fsencode(out_path) if out_path else None,
                            C.byref(feat) if want_features else None, C.byref(nf), C.byref(st)))
    if not want_features:
        return None
    try:
        n = nf.value * st.value
        return np.ctypeslib.as_array(feat, shape=(max(n, 1),))[:n].reshape(nf.value, st.value).copy()
    finally:
        lib().MFCCFree(feat)


def wapper_mem(cfg, wav_file_bytes, out_path=None):
    buf = np.frombuffer(bytes(wav_file_bytes), dtype=np.uint8)
    feat = C.POINTER(C.c_double)()
    nf, st = C.c_long(), C.c_int()
    _check(lib().MFCCWapperMem(C.byref(cfg), buf.ctypes.data, buf.size,
                               os.fsencode(out_path) if out_path else None, C.byref(feat),
                               C.byref(nf), C.byref(st)))
    try:
        n = nf.value * st.value
        return np.ctypeslib.as_array(feat, shape=(max(n, 1),))[:n].reshape(nf.value, st.value).copy()
    finally:
        lib().MFCCFree(feat)


def extract_recording(cfg, pcm, channels, bytes_per_sample, num_shards=0, devices=None):
    """MFCCExtractRecording: one long recording (interleaved PCM bytes on the host) over num_shards
    frame ranges, one GPU and host thread each; returns the [numFrames, step] matrix."""
    pcm = np.ascontiguousarray(np.frombuffer(bytes(pcm), dtype=np.uint8) if not isinstance(pcm, np.ndarray) else pcm)
    n = pcm.size // (channels * bytes_per_sample)
    T = num_frames(n, cfg.wlen, cfg.inc)
    out = np.zeros((T, step(cfg)), dtype=np.float64 if cfg.precision == PREC_F64 else np.float32)
    dev = None
    if devices is not None:
        dev = (C.c_int * len(devices))(*devices)
    nf = C.c_int64()
    _check(lib().MFCCExtractRecording(C.byref(cfg), pcm.ctypes.data, channels, bytes_per_sample, n, num_shards,
                                      dev, out.ctypes.data, C.byref(nf)))
    assert nf.value == T
    return out


def wapper_list(ini_path):
    """MFCCWapperList: returns the number of failed files; raises on ini / list errors."""
    rc = lib().MFCCWapperList(os.fsencode(ini_path))
    if rc < 0:
        _check(rc)
    return rc


def host_tables(cfg):
    """MFCCHostTables as a dict: the filterbank / window tables computed on the host, no device."""
    fft_n, klo, khi = C.c_int(), C.c_int(), C.c_int()
    buf = C.create_string_buffer(256)
    _check(lib().MFCCHostTables(C.byref(cfg), C.byref(fft_n), C.byref(klo), C.byref(khi), None, None,
                                None, None, buf, 256))
    half, P = fft_n.value // 2, cfg.bankNum
    cf, lo_wt, win = np.zeros(P + 1), np.zeros(half), np.zeros(cfg.wlen)
    lo_chan = np.zeros(half, dtype=np.int32)
    dp = C.POINTER(C.c_double)
    _check(lib().MFCCHostTables(C.byref(cfg), None, None, None, cf.ctypes.data_as(dp),
                                lo_chan.ctypes.data_as(C.POINTER(C.c_int)), lo_wt.ctypes.data_as(dp),
                                win.ctypes.data_as(dp), None, 0))
    return dict(fftN=fft_n.value, klo=klo.value, khi=khi.value, cf=cf, loChan=lo_chan, loWt=lo_wt,
                window=win, text=buf.value.decode())


def shard_geometry(cfg, total_samples, frame_begin, frame_end):
    """MFCCShardGeometry as a dict (no device needed)."""
    vals = [C.c_int64() for _ in range(5)]
    lead = C.c_int()
    _check(lib().MFCCShardGeometry(C.byref(cfg), total_samples, frame_begin, frame_end,
                                   *[C.byref(x) for x in vals], C.byref(lead)))
    keys = ("sample_begin", "sample_count", "total_rows", "owned_row_begin", "owned_rows")
    out = {k: x.value for k, x in zip(keys, vals)}
    out["lead"] = lead.value
    return out


class Context:
    """MFCCContext: host tables built and uploaded once per option set."""

    def __init__(self, cfg, device=0):
        self.cfg = cfg
        self.device = device
        h = C.c_void_p()
        _check(lib().MFCCContextCreate(C.byref(cfg), device, C.byref(h)))
        self.handle = h
        self.step = step(cfg)
        self.precision = cfg.precision

    def passband_text(self):
        buf = C.create_string_buffer(256)
        _check(lib().MFCCContextPassbandText(self.handle, buf, 256))
        return buf.value.decode()

    def tables(self):
        fft_n, klo, khi = C.c_int(), C.c_int(), C.c_int()
        _check(lib().MFCCContextTables(self.handle, C.byref(fft_n), C.byref(klo), C.byref(khi),
                                       None, None, None))
        half, P = fft_n.value // 2, self.cfg.bankNum
        cf, lo_wt = np.zeros(P + 1), np.zeros(half)
        lo_chan = np.zeros(half, dtype=np.int32)
        _check(lib().MFCCContextTables(self.handle, None, None, None,
                                       cf.ctypes.data_as(C.POINTER(C.c_double)),
                                       lo_chan.ctypes.data_as(C.POINTER(C.c_int)),
                                       lo_wt.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(fftN=fft_n.value, klo=klo.value, khi=khi.value, cf=cf, loChan=lo_chan, loWt=lo_wt)

    # -- the reference's lower-level API, batched: torch tensors on the context's device ----------

    def _dtype(self):
        import torch
        return torch.float64 if self.precision == PREC_F64 else torch.float32

    def wave2fbank(self, frames):
        """frames: [n, wlen] windowed frames -> ([n, bankNum] log filterbank values, [n] energies)."""
        import torch
        frames = frames.contiguous()
        assert frames.dtype == self._dtype() and frames.shape[1] == self.cfg.wlen
        n = frames.shape[0]
        fb = torch.empty((n, self.cfg.bankNum), dtype=frames.dtype, device=frames.device)
        te = torch.empty((n,), dtype=frames.dtype, device=frames.device)
        st = torch.cuda.current_stream(frames.device).cuda_stream
        _check(lib().MFCCWave2FBankDevice(self.handle, frames.data_ptr(), n, fb.data_ptr(), te.data_ptr(),
                                          C.c_void_p(st)))
        return fb, te

    def fbank2mfcc(self, fbank):
        """fbank: [n, bankNum] -> ([n, MFCCNum] cepstra, [n] C0)."""
        import torch
        fbank = fbank.contiguous()
        assert fbank.dtype == self._dtype() and fbank.shape[1] == self.cfg.bankNum
        n = fbank.shape[0]
        m = torch.empty((n, self.cfg.MFCCNum), dtype=fbank.dtype, device=fbank.device)
        c0 = torch.empty((n,), dtype=fbank.dtype, device=fbank.device)
        st = torch.cuda.current_stream(fbank.device).cuda_stream
        _check(lib().MFCCFBank2MFCCDevice(self.handle, fbank.data_ptr(), n, m.data_ptr(), c0.data_ptr(),
                                          C.c_void_p(st)))
        return m, c0

    def close(self):
        if self.handle:
            lib().MFCCContextDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """MFCCPlan for a batch of utterances (byte_off / num_samples) or, with shard=..., for a
    frame range of one long recording."""

    def __init__(self, ctx, channels, bytes_per_sample, byte_off=None, num_samples=None,
                 shard=None):
        self.ctx = ctx
        self.channels, self.bps = channels, bytes_per_sample
        h = C.c_void_p()
        if shard is None:
            off = np.ascontiguousarray(byte_off, dtype=np.int64)
            ns = np.ascontiguousarray(num_samples, dtype=np.int64)
            self.num_utts = len(ns)
            _check(lib().MFCCPlanCreate(ctx.handle, channels, bytes_per_sample, len(ns),
                                        off.ctypes.data_as(C.POINTER(C.c_int64)),
                                        ns.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(h)))
        else:
            total, fb, fe, sums2 = shard
            self.num_utts = 1
            sp = None
            if sums2 is not None:
                s2 = np.ascontiguousarray(sums2, dtype=np.int64)
                sp = s2.ctypes.data_as(C.POINTER(C.c_int64))
            _check(lib().MFCCPlanCreateShard(ctx.handle, channels, bytes_per_sample, total, fb, fe,
                                             sp, C.byref(h)))
        self.handle = h
        L = lib()
        self.total_rows = int(L.MFCCPlanTotalRows(h))
        self.sample_begin = int(L.MFCCPlanSampleBegin(h))
        self.sample_count = int(L.MFCCPlanSampleCount(h))
        self.owned_row_begin = int(L.MFCCPlanOwnedRowBegin(h))
        self.owned_rows = int(L.MFCCPlanOwnedRows(h))

    def row_offset(self, u):
        return int(lib().MFCCPlanRowOffset(self.handle, u))

    def frames(self, u):
        return int(lib().MFCCPlanNumFrames(self.handle, u))

    def enable_timing(self, on=True):
        _check(lib().MFCCPlanEnableTiming(self.handle, 1 if on else 0))

    def last_kernel_ms(self):
        return float(lib().MFCCPlanLastKernelMs(self.handle))

    def _torch_dtype(self):
        import torch
        return torch.float64 if self.ctx.precision == PREC_F64 else torch.float32

    def alloc_output(self):
        import torch
        return torch.empty((self.total_rows, self.ctx.step), dtype=self._torch_dtype(),
                           device="cuda:%d" % self.ctx.device)

    def extract(self, pcm_dev, out=None, stream=None):
        """pcm_dev: uint8 torch tensor on the context's device.  Asynchronous on `stream`
        (default: torch's current stream).  Returns the [total_rows, step] feature tensor."""
        import torch
        if not pcm_dev.is_cuda or pcm_dev.dtype != torch.uint8:
            raise ValueError("pcm_dev must be a uint8 tensor on the GPU")
        if out is None:
            out = self.alloc_output()
        if stream is None:
            stream = torch.cuda.current_stream(pcm_dev.device).cuda_stream
        _check(lib().MFCCExtractDevice(self.ctx.handle, self.handle, pcm_dev.data_ptr(),
                                       pcm_dev.numel(), out.data_ptr(), C.c_void_p(stream)))
        return out

    def shard_column_sums(self, out, stream=None):
        import torch
        sums = torch.empty(2 * self.ctx.step + 1, dtype=torch.float64, device=out.device)
        if stream is None:
            stream = torch.cuda.current_stream(out.device).cuda_stream
        _check(lib().MFCCShardColumnSums(self.ctx.handle, self.handle, out.data_ptr(),
                                         sums.data_ptr(), C.c_void_p(stream)))
        return sums

    def shard_normalize(self, out, sums, stream=None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(out.device).cuda_stream
        _check(lib().MFCCShardNormalize(self.ctx.handle, self.handle, out.data_ptr(),
                                        sums.data_ptr(), C.c_void_p(stream)))
        return out

    def close(self):
        if self.handle:
            lib().MFCCPlanDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def channel_sums(ctx, pcm_dev, channels, bytes_per_sample, sample_begin, count, stream=None):
    """Twice the integer sums of {L, R, mid, side} over a span of instants (int64 tensor of 4)."""
    import torch
    sums = torch.empty(4, dtype=torch.int64, device=pcm_dev.device)
    if stream is None:
        stream = torch.cuda.current_stream(pcm_dev.device).cuda_stream
    _check(lib().MFCCChannelSums(ctx.handle, pcm_dev.data_ptr(), channels, bytes_per_sample,
                                 sample_begin, count, sums.data_ptr(), C.c_void_p(stream)))
    return sums


def pack_utterances(pcm_list, align=16):
    """Concatenates per-utterance PCM byte arrays with `align`-byte aligned starts.
    Returns (uint8 array, byte offsets)."""
    offs, total = [], 0
    for p in pcm_list:
        offs.append(total)
        total += (len(p) + align - 1) // align * align
    buf = np.zeros(max(total, align), dtype=np.uint8)
    for o, p in zip(offs, pcm_list):
        buf[o:o + len(p)] = np.frombuffer(bytes(p), dtype=np.uint8) if not isinstance(p, np.ndarray) else p
    return buf, np.array(offs, dtype=np.int64)
