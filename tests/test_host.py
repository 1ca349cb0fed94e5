"""CPU tests of the product's host side: the C-ABI library loads and exports what include/mfcc_amd.h
declares, the ini reader, frame arithmetic, host tables, containers (WAV header walk, text / npy /
binary writers), and that extraction refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest

import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The reference's example/config.ini, as data: same keys, values, " ;" comments and "];" sections.
EXAMPLE_INI = """[Frame];
sampleRate = 44100 ;
hipassfre = 22050 ;
lowpassfre = 10 ;
preemphasise = 0.0 ;
wlen = 1764 ;
inc = 882 ;
saveType = e ;
vecNum = 2 ;
fileList = fileList.txt ;

[MFCC];
fbankFlag = 1 ;
bankNum = 40 ;
MFCCNum = -1 ;
MFCC0thFlag = 0 ;
         
[Others];
energyFlag = 1 ;
zeroCrossingFlag = 1 ;
brightFlag = 1 ;
subBandEFlag = 8 ;
fftLength = 0 ;
         
[Regression];
regreOrder = 3 ;
delwin = 9 ;
"""


def test_compat_library_exports_the_reference_names():
    """libmfcc_compat.so loads and exports every function include/mfcc_compat.h declares (the
    reference's names of mfcc.h:29-74 plus the vector helpers); no compute call without a GPU."""
    header = open(os.path.join(ROOT, "include", "mfcc_compat.h")).read()
    body = header[header.index("} FBankInfo;"):]
    declared = set(re.findall(r"\b([A-Za-z][A-Za-z0-9]+)\s*\((?:int|Vector|const|FBankInfo|void)", body))
    assert declared == set(ma.COMPAT_EXPORTS), declared ^ set(ma.COMPAT_EXPORTS)
    L = ma.compat_lib()
    for name in ma.COMPAT_EXPORTS:
        assert hasattr(L, name), name
    v = L.MFCCCreateVector(5)
    assert L.MFCCVectorSize(v) == 5 and v[1] == 0.0
    L.MFCCFreeVector(v)
    assert abs(L.Mel(3, 0.01) - 1127 * np.log(1 + 2 * 0.01)) < 1e-12


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mfcc_amd.h")).read()
    declared = set(re.findall(r"\b(MFCC[A-Za-z0-9]+)\s*\(", header))
    declared -= {"MFCCConfig"}
    L = ma.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(ma.EXPORTS), declared ^ set(ma.EXPORTS)


def test_config_struct_layout_matches_header():
    """ctypes image of MFCCConfig: field order / count as in the header."""
    header = open(os.path.join(ROOT, "include", "mfcc_amd.h")).read()
    body = re.search(r"typedef struct MFCCConfig \{(.*?)\} MFCCConfig;", header, re.S).group(1)
    fields = re.findall(r"^\s*(?:int|double|char)\s+(\w+)", body, re.M)
    assert fields == [f[0] for f in ma.MFCCConfig._fields_]


def test_ini_example_config(tmp_path):
    p = tmp_path / "config.ini"
    p.write_text(EXAMPLE_INI)
    cfg = ma.config_from_ini(str(p))
    assert (cfg.sampleRate, cfg.wlen, cfg.inc, cfg.vecNum) == (44100, 1764, 882, 2)
    assert (cfg.hipassfre, cfg.lowpassfre, cfg.preemphasise) == (22050.0, 10.0, 0.0)
    assert cfg.saveType == ma.SAVE_E and cfg.fileList == b"fileList.txt"
    assert (cfg.fbankFlag, cfg.bankNum) == (1, 40)
    assert (cfg.MFCCNum, cfg.MFCC0thFlag) == (40, 0)  # main.c:142-144
    assert (cfg.energyFlag, cfg.zeroCrossingFlag, cfg.brightFlag, cfg.subBandEFlag, cfg.fftLength) == (1, 1, 1, 8, 0)
    assert (cfg.regreOrder, cfg.delwin) == (3, 9)
    assert ma.feature_dim(cfg) == 51 and ma.step(cfg) == 306  # main.c:161,185,254


def test_ini_readme_layout_and_errors(tmp_path):
    """README.md:88,122,126-134: zeroMeanSigFlag, znormFlag, [IO] fileList / saveType / numThreads."""
    p = tmp_path / "c.ini"
    p.write_text("; comment\n[Frame]\nsampleRate=16000\nzeroMeanSigFlag = 1\nwlen : 400\n"
                 "[Regression]\nznormFlag = 1 ; note\n[IO]\nfileList = a b.txt\nsaveType = n\nnumThreads = 4\n"
                 "precision = f64\nnumDevices = 2\n")
    cfg = ma.config_from_ini(str(p))
    assert (cfg.zeroMeanSigFlag, cfg.znormFlag, cfg.saveType, cfg.numThreads) == (1, 1, ma.SAVE_N, 4)
    assert cfg.numDevices == 2 and ma.make_config().numDevices == 0 and ma.make_config().precision == ma.PREC_F64
    assert cfg.fileList == b"a b.txt" and cfg.precision == ma.PREC_F64 and cfg.wlen == 400
    for letter, code in (("f", ma.SAVE_F), ("e", ma.SAVE_E), ("n", ma.SAVE_N), ("b", ma.SAVE_B), ("zz", ma.SAVE_E)):
        p.write_text("[Frame]\nsaveType = %s\n" % letter)  # main.c:76-80: anything else -> e
        assert ma.config_from_ini(str(p)).saveType == code
    p.write_text("[Frame]\nsampleRate = 8000\nnoSuchKey = 1\nwlen = 5\n")
    with pytest.raises(ma.MFCCError) as e:  # main.c:94,135-138: unknown key, first bad line reported
        ma.config_from_ini(str(p))
    assert e.value.code == ma.E_INI_KEY and e.value.bad_line == 3 and "line 3" in str(e.value)
    p.write_text("[Frame]\nthis line has no separator\n")
    with pytest.raises(ma.MFCCError) as e:
        ma.config_from_ini(str(p))
    assert e.value.bad_line == 2
    with pytest.raises(ma.MFCCError) as e:
        ma.config_from_ini(str(tmp_path / "missing.ini"))
    assert e.value.code == ma.E_INI_OPEN


@pytest.mark.parametrize("n,wlen,inc,expect", [
    (0, 400, 160, 0), (399, 400, 160, 0), (400, 400, 160, 1), (559, 400, 160, 1), (560, 400, 160, 2),
    (160000, 400, 160, 998), (1323001, 1764, 882, 1499), (1270080000, 1764, 882, 1439999)])
def test_frame_count_matches_oracle(oracle, n, wlen, inc, expect):
    assert ma.num_frames(n, wlen, inc) == expect == oracle.num_frames(n, wlen, inc)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_dims_match_oracle(oracle, name):
    kw = CONFIGS[name]
    assert ma.step(ma.make_config(**kw)) == oracle.step(oracle.make_config(**kw))


def test_host_tables_known_answer_and_oracle(oracle):
    """The product's own InitFBank replacement: README.md:33-34 text, and tables equal to the
    oracle's bit for bit (both are IEEE double restatements of mfcc.c:39-122)."""
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        ka = json.load(f)["readme_passband"]
    t = ma.host_tables(ma.make_config(**CONFIGS["c1a"]))
    assert t["text"] == ka["text"]
    for wlen, sr, P, lo, hi in [(1764, 44100, 40, 10, 22050), (400, 16000, 40, -1, -1),
                                (400, 16000, 40, 20, 8000), (400, 16000, 26, 64, 8000),
                                (1764, 44100, 40, 10, 16000), (300, 8000, 23, 100, 3800)]:
        got = ma.host_tables(ma.make_config(sampleRate=sr, wlen=wlen, inc=wlen // 2, bankNum=P, MFCCNum=P,
                                            lowpassfre=lo, hipassfre=hi))
        ref = oracle.fbank_tables(wlen, sr, P, lo, hi)
        assert (got["fftN"], got["klo"], got["khi"]) == (ref["fftN"], ref["klo"], ref["khi"])
        assert np.array_equal(got["cf"], ref["cf"]) and np.array_equal(got["loChan"], ref["loChan"])
        assert np.array_equal(got["loWt"], ref["loWt"])
        assert got["text"] == ref["text"]
        w = np.zeros(wlen)
        oracle.lib().orc_hamming(w.ctypes.data_as(C.POINTER(C.c_double)), wlen)
        assert np.array_equal(got["window"], w)


def test_config_validation():
    for bad in (dict(wlen=2), dict(inc=0), dict(vecNum=5), dict(bankNum=0), dict(fbankFlag=0, MFCCNum=41),
                dict(regreOrder=0), dict(regreOrder=2, delwin=0), dict(subBandEFlag=300), dict(wlen=40000)):
        with pytest.raises(ma.MFCCError) as e:
            ma.host_tables(ma.make_config(**bad))
        assert e.value.code == ma.E_CONFIG, bad


def test_wav_info_variants():
    pcm = ma.synth_pcm(1000, 1, 2, seed=5)
    info = ma.wav_info(ma.wav_bytes(pcm, 1, 2, 16000))
    assert info == dict(channels=1, sampleRate=16000, bytesPerSample=2, numSamples=1000, dataOffset=44)
    pcm = ma.synth_pcm(333, 2, 3, seed=6)
    info = ma.wav_info(ma.wav_bytes(pcm, 2, 3, 44100, extra_chunk=(b"LIST", b"odd")))  # README.md:68
    assert (info["numSamples"], info["dataOffset"], info["bytesPerSample"]) == (333, 44 + 8 + 4, 3)
    with pytest.raises(ma.MFCCError) as e:
        ma.wav_info(b"RIFX" + b"\0" * 60)
    assert e.value.code == ma.E_WAV
    trunc = ma.wav_bytes(pcm, 2, 3, 44100)[:-600]  # data size field larger than the file
    assert ma.wav_info(trunc)["numSamples"] == 233
    info = ma.wav_info(ma.wav_bytes(b"\x80" * 100, 1, 1, 8000))  # 8-bit PCM (unsigned) is accepted
    assert (info["bytesPerSample"], info["numSamples"]) == (1, 100)
    twelve_bit = bytearray(ma.wav_bytes(b"\0" * 100, 1, 2, 8000))
    twelve_bit[34:36] = (12).to_bytes(2, "little")
    with pytest.raises(ma.MFCCError):
        ma.wav_info(bytes(twelve_bit))


def test_wav_info_matches_oracle(oracle):
    pcm = ma.synth_pcm(777, 2, 2, seed=9)
    wav = ma.wav_bytes(pcm, 2, 2, 22050, extra_chunk=(b"fact", b"1234"))
    assert ma.wav_info(wav) == oracle.wav_parse(wav)


def test_writers(tmp_path, oracle):
    rng = np.random.default_rng(0)
    d = rng.standard_normal((7, 5)) * 1e3
    d[0, 0], d[1, 1], d[2, 2] = np.inf, -np.inf, np.nan
    for st, letter, fmt in ((ma.SAVE_F, "f", "%f\t"), (ma.SAVE_E, "e", "%e\t")):
        p = str(tmp_path / ("o." + letter))
        ma.write_features(p, d, st)
        want = "".join("".join(fmt % v for v in row) + "\n" for row in d)  # main.c:276-285
        assert open(p).read() == want
        q = str(tmp_path / ("q." + letter))
        oracle.write_text(q, d, 5, letter)
        assert open(q, "rb").read() == open(p, "rb").read()
    p = str(tmp_path / "o.npy")
    ma.write_features(p, d, ma.SAVE_N)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x93NUMPY\x01\x00"
    hlen = raw[8] | (raw[9] << 8)
    assert (10 + hlen) % 16 == 0 and raw[10 + hlen - 1:10 + hlen] == b"\n"  # cnpy.hpp:250-253
    assert raw[10:10 + hlen].startswith(b"{'descr': '<f8', 'fortran_order': False, 'shape': (7, 5), }")
    back = np.load(p)
    assert back.shape == (7, 5) and np.array_equal(back, d, equal_nan=True)
    p = str(tmp_path / "o.bin")
    ma.write_features(p, d, ma.SAVE_B)
    assert np.array_equal(np.fromfile(p, dtype="<f8").reshape(7, 5), d, equal_nan=True)
    with pytest.raises(ma.MFCCError) as e:
        ma.write_features(str(tmp_path / "no" / "dir" / "x"), d, ma.SAVE_F)
    assert e.value.code == ma.E_WRITE


def test_npy_header_padding_when_already_aligned(tmp_path):
    """cnpy pads by a full 16 when preamble + dict is already a multiple of 16 (cnpy.hpp:251-252)."""
    for rows in (1, 12, 123, 1234, 12345, 123456):
        p = str(tmp_path / "a.npy")
        ma.write_features(p, np.zeros((0, 3)) if rows == 0 else np.zeros((rows, 3)), ma.SAVE_N)
        raw = open(p, "rb").read(200)
        hlen = raw[8] | (raw[9] << 8)
        dict_len = len("{'descr': '<f8', 'fortran_order': False, 'shape': (%d, 3), }" % rows)
        assert hlen == dict_len + (16 - (10 + dict_len) % 16)
        assert np.load(p).shape == (rows, 3)


def test_synth_pcm_is_deterministic():
    a = ma.synth_pcm(5000, 2, 3, seed=0x5EED0000, sample_rate=44100, tone_hz=440)
    b = ma.synth_pcm(5000, 2, 3, seed=0x5EED0000, sample_rate=44100, tone_hz=440)
    assert np.array_equal(a, b)
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
    x = ma.synth_pcm(20000, 1, 2, seed=1).view(np.int16).astype(np.float64)
    assert abs(x.mean()) < 300 and 9000 < x.std() < 17000 and np.count_nonzero(x) > 19900


def test_shard_geometry():
    cfg = ma.make_config(wlen=400, inc=160, regreOrder=3, delwin=2)
    total = ma.num_frames(160000, 400, 160)
    g0 = ma.shard_geometry(cfg, 160000, 0, total // 2)
    g1 = ma.shard_geometry(cfg, 160000, total // 2, total)
    assert g0 == dict(sample_begin=0, sample_count=(499 + 4 - 1) * 160 + 400, total_rows=503,
                      owned_row_begin=0, owned_rows=499, lead=0)
    assert g1["lead"] == 1 and g1["owned_row_begin"] == 4 and g1["owned_rows"] == 499
    assert g1["sample_begin"] == (499 - 4) * 160 - 1
    assert g1["sample_begin"] + g1["sample_count"] == (total - 1) * 160 + 400
    e = ma.shard_geometry(cfg, 160000, 10, 10)
    assert e["owned_rows"] == 0 and e["sample_count"] == 0
    no_delta = ma.shard_geometry(ma.make_config(wlen=400, inc=160), 160000, 100, 200)
    assert no_delta["total_rows"] == 100 and no_delta["owned_row_begin"] == 0


def test_extraction_needs_a_gpu(tmp_path):
    """No CPU fallback: without a HIP device every extraction entry point fails with E_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = ma.make_config(**CONFIGS["c2"])
    with pytest.raises(ma.MFCCError) as e:
        ma.Context(cfg, 0)
    assert e.value.code == ma.E_DEVICE
    wav = tmp_path / "a.wav"
    wav.write_bytes(ma.wav_bytes(ma.synth_pcm(2000, 1, 2), 1, 2, 16000))
    with pytest.raises(ma.MFCCError) as e:
        ma.wapper(cfg, str(wav))
    assert e.value.code == ma.E_DEVICE
    with pytest.raises(ma.MFCCError) as e:  # header checks still come first (main.c:192-196)
        ma.wapper(ma.make_config(**dict(CONFIGS["c2"], sampleRate=8000)), str(wav))
    assert e.value.code == ma.E_SAMPLE_RATE
    with pytest.raises(ma.MFCCError) as e:
        ma.wapper(cfg, str(tmp_path / "missing.wav"))
    assert e.value.code == ma.E_OPEN


# ---------------------------------------------------------------------------------------------
# text writers: printf-identical "%f" / "%e" (main.c:276-285) from the exact integer formatter
# ---------------------------------------------------------------------------------------------

def _special_values():
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, 9.9999995, 9.99999949999, 99999.9999995, 999999.5, 0.0078125,
            -0.0234375, 1e-7, 5e-7, 4.999999e-7, 1.5e-6, 2.5e-6, 1e6, 1e7 - 0.5, 123456789.125, 1e14, 1e15,
            1e21, 9.999999e21, 1e22, 1e-15, 9.9999e-16, 1e-300, 5e-324, 1.7976931348623157e308,
            float("inf"), float("-inf"), 2.0 ** -20, 2.0 ** -30, 2.0 ** 52, 2.0 ** 53 + 2, 0.1, 0.2, 0.3,
            1.0 / 3.0, 2.0 / 3.0, 6.02214076e23, 1.602176634e-19, 999999.9999995, 0.9999995, 0.99999949]
    vals += [k / 128.0 for k in range(1, 64, 2)]          # exact ties of "%f"
    vals += [(2 * k + 1) / 2.0 ** 12 for k in range(40)]
    vals += [10.0 ** k for k in range(-15, 22)]
    vals += [float(np.nextafter(10.0 ** k, 0)) for k in range(-14, 22)]
    vals += [float(np.nextafter(10.0 ** k, np.inf)) for k in range(-14, 22)]
    return vals


def test_format_value_matches_printf_on_special_values():
    for x in _special_values():
        assert ma.format_value(x, ma.SAVE_F) == "%f" % x, x
        assert ma.format_value(x, ma.SAVE_E) == "%e" % x, x
        assert ma.format_value(-x, ma.SAVE_F) == "%f" % -x, x
        assert ma.format_value(-x, ma.SAVE_E) == "%e" % -x, x
    assert ma.format_value(float("nan"), ma.SAVE_F) in ("nan", "-nan")


def test_text_writers_match_printf_in_bulk(tmp_path):
    rng = np.random.default_rng(20240611)
    n, step = 6000, 50
    mant = rng.standard_normal(n * step)
    expo = rng.integers(-14, 14, n * step)
    data = (mant * 10.0 ** expo).reshape(n, step)
    data[::7, 3] = np.round(data[::7, 3] * 128) / 128       # dyadic values: exact ties happen
    data[::11, 5] = np.float32(data[::11, 5]).astype(np.float64)  # what the f32 path hands over
    for save_type, spec in ((ma.SAVE_F, "%f"), (ma.SAVE_E, "%e")):
        path = tmp_path / ("bulk." + spec[1])
        ma.write_features(str(path), data, save_type)
        got = path.read_bytes()
        want = "".join("".join(spec % v + "\t" for v in row) + "\n" for row in data).encode()
        assert got == want


def test_wav_rf64_and_extensible():
    import struct
    pcm = bytes(ma.synth_pcm(500, 2, 2, seed=11))
    fmt = struct.pack("<HHIIHH", 1, 2, 48000, 48000 * 4, 4, 16)
    ds64 = struct.pack("<QQQI", 0, len(pcm), 500, 0)
    rf64 = (b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + b"ds64" + struct.pack("<I", len(ds64)) + ds64 +
            b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", 0xFFFFFFFF) + pcm)
    info = ma.wav_info(rf64)
    assert (info["numSamples"], info["channels"], info["sampleRate"]) == (500, 2, 48000)
    assert rf64[info["dataOffset"]:info["dataOffset"] + 8] == pcm[:8]
    # WAVE_FORMAT_EXTENSIBLE with the PCM sub-format is accepted, with the float sub-format refused
    def extensible(sub):
        body = struct.pack("<HHIIHHHHIH", 0xFFFE, 2, 48000, 48000 * 4, 4, 16, 22, 16, 3, sub) + b"\0" * 14
        return (b"RIFF" + struct.pack("<I", 4 + 8 + len(body) + 8 + len(pcm)) + b"WAVE" + b"fmt " +
                struct.pack("<I", len(body)) + body + b"data" + struct.pack("<I", len(pcm)) + pcm)
    assert ma.wav_info(extensible(1))["numSamples"] == 500
    with pytest.raises(ma.MFCCError) as e:
        ma.wav_info(extensible(3))
    assert e.value.code == ma.E_WAV


def test_host_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """`make asan` builds the host sources (ini parser, WAV walk, writers, formatter, tables, PCM
    generator) with -fsanitize=address,undefined around tests/host_checks.cpp; the run must pass its
    own checks and leave both sanitizers silent (either one ends the program with a report)."""
    import subprocess
    src = os.path.join(ROOT, "mfcc_amd", "csrc")
    subprocess.run(["make", "-C", src, "asan"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(src, "build", "host_asan"), str(tmp_path)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_checks: ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
