"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libmfcc_amd.so, against the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and -- at the benchmark's full sizes -- through size-independent properties.

Tolerances (single precision kernels; util.kind_tolerance): log-mel 1e-4 absolute, cepstra 4e-6
and C0 1e-6 relative to the largest such magnitude of the matrix (the stated bar is 1e-5; an
absolute floor covers matrices whose cepstra are all small), ZCR exact, log energy 2e-5,
brightness / sub-band fractions 6e-7 absolute; deltas inherit the tolerance of their static
column; z-normalised columns: the kind's tolerance divided by the column's standard deviation
(util.assert_znorm_close).  Double precision kernels (the library default) must agree to 1e-9.
Frame counts, row layout and indexing must be exact.  The largest error each kind reached is
printed at the end of the run."""
import glob
import json
import os

import numpy as np
import pytest

import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS
import util
from util import assert_static_close, column_kinds, run_product_batch

pytestmark = pytest.mark.gpu


@pytest.fixture
def tuning():
    """ma.tuning for one test: every override is reset afterwards."""
    yield ma.tuning
    ma.tuning()

HERE = os.path.dirname(os.path.abspath(__file__))


def _pcm(n, ch, bps, seed, sr, tone=440):
    if bps == 1:  # 8-bit WAV PCM is unsigned: the top byte of a 16-bit sample, offset 128
        p16 = ma.synth_pcm(n, ch, 2, seed=0x5EED0000 + seed, sample_rate=sr, tone_hz=tone).reshape(-1, 2)
        return (p16[:, 1] ^ 0x80).astype(np.uint8).copy()
    if bps == 4:  # the generator makes 16- and 24-bit samples: widen 24-bit ones, low byte varied
        p24 = ma.synth_pcm(n, ch, 3, seed=0x5EED0000 + seed, sample_rate=sr, tone_hz=tone).reshape(-1, 3)
        out = np.empty((p24.shape[0], 4), dtype=np.uint8)
        out[:, 1:] = p24
        out[:, 0] = p24[:, 0] ^ 0x5A
        return out.reshape(-1)
    return ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + seed, sample_rate=sr, tone_hz=tone)


def _layout(kw):
    return (1 if kw["vecNum"] == 1 else 2), (2 if kw["sampleRate"] == 16000 else 3)


def _compare(got, ref, kw, f64, src=None):
    """Static block by kind; delta blocks with the same per-kind tolerance.  z-normalised matrices:
    per column kind with the kind's tolerance divided by the column's standard deviation
    (util.assert_znorm_close) when src = (oracle, pcm, channels, bytes per sample) lets the
    un-normalised reference be formed; without it, 3e-4 (1e-8 in double) for the whole matrix."""
    assert got.shape == ref.shape
    if kw.get("znormFlag", 0):
        if src is not None:
            orc, pcm, ch, bps = src
            raw = orc.extract_pcm_bytes(orc.make_config(**dict(kw, znormFlag=0)), np.frombuffer(bytes(pcm), dtype=np.uint8), ch, bps)
            util.assert_znorm_close(got, ref, raw, kw, f64)
            return
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        fin = np.isfinite(ref)
        tol = 1e-8 if f64 else 3e-4
        err = float(np.max(np.abs(np.where(fin, got - ref, 0.0)), initial=0.0))
        util._note("znorm %s(z units, whole matrix)" % ("f64 " if f64 else ""), err, tol)
        assert err < tol
        return
    vs = len(column_kinds(kw)) * kw.get("vecNum", 1)
    for o in range(kw.get("regreOrder", 1)):
        assert_static_close(got[:, o * vs:(o + 1) * vs], ref[:, o * vs:(o + 1) * vs], kw, f64, delta=o > 0,
                            static_ref=ref[:, :vs] if o > 0 else None)


@pytest.mark.parametrize("finish", ["utterance", "blocks", "stream"])
@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c1a", "c1b", "c5"])
def test_named_configs_match_oracle(oracle, name, precision, finish, tuning):
    """finish: deltas and z-norm by the whole-utterance-in-LDS kernel (small batches of short
    utterances), by the row-block kernels (small batches with long utterances) or by the streaming
    whole-utterance kernel (wide batches); each forced here through the `finish` tuning key."""
    tuning(finish={"utterance": 1, "blocks": 3, "stream": 2}[finish])
    kw = CONFIGS[name]
    ch, bps = _layout(kw)
    n = kw["sampleRate"] + 123  # ~1 s: several tiles, first (general) and later (fast) unpack paths
    pcm = _pcm(n, ch, bps, 7, kw["sampleRate"])
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
    f64 = precision == "f64"
    cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
    got = run_product_batch(ma, cfg, [pcm.tobytes()], ch, bps)[0]
    _compare(got, ref, kw, f64, src=(oracle, pcm, ch, bps))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))))
def test_golden_fixtures(oracle, path):
    z = np.load(path)
    kw = json.loads(str(z["config"]))
    ch, bps = int(z["channels"]), int(z["bytes_per_sample"])
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [z["pcm"].tobytes()], ch, bps)[0]
        _compare(got, z["expected"], kw, f64, src=(oracle, z["pcm"], ch, bps))


@pytest.mark.parametrize("wlen", [3, 4, 5, 8, 9, 16, 17, 33, 64, 100, 129, 200, 256, 400, 513, 1024,
                                  1200, 2048, 3000, 4096, 5000, 8192, 10000, 16384])
def test_every_fft_size(oracle, wlen):
    """fftN from 4 to 16384 (every kernel instantiation), odd and even inc."""
    inc = max(1, wlen // 3) | 1 if wlen % 2 else max(1, wlen // 2)
    kw = dict(sampleRate=16000, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=wlen, inc=inc,
              vecNum=1, fbankFlag=1, bankNum=min(12, max(1, wlen // 4)), energyFlag=1,
              zeroCrossingFlag=1, brightFlag=1)
    n = wlen + 9 * inc + (inc - 1)  # exactly 10 frames, the tail is dropped
    pcm = _pcm(n, 1, 2, wlen, 16000, tone=0)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    assert ref.shape[0] == 10
    for f64 in (False, True):
        if f64 and wlen > 8192:
            continue
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [pcm.tobytes()], 1, 2)[0]
        _compare(got, ref, kw, f64)


@pytest.mark.parametrize("wlen,bank", [(8, 2), (50, 7), (100, 12), (200, 23), (256, 20), (400, 5), (400, 13),
                                       (400, 23), (400, 40), (400, 64), (400, 80), (512, 26), (800, 40),
                                       (1024, 33), (1600, 40), (2048, 64), (3000, 40), (5000, 24)])
@pytest.mark.parametrize("cepstral", [False, True])
def test_lean_kernel_modes(oracle, wlen, bank, cepstral):
    """Mono 16-bit single-row configurations without the per-frame extras take the lean kernel, and
    from 512- to 2048-point frames its matrix-core mel filterbank: every channel count layout
    (quads cut over 1..4 blocks, more than 16 quads falling back), even frame shifts, tiles that
    end on a partial round, log-mel and cepstral outputs."""
    inc = max(2, (wlen // 5) & ~1)
    kw = dict(sampleRate=16000, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=wlen, inc=inc,
              vecNum=1, fbankFlag=0 if cepstral else 1, bankNum=bank, MFCCNum=min(bank, 13),
              MFCC0thFlag=1 if cepstral else 0, energyFlag=1 if cepstral else 0)
    n = wlen + 36 * inc + 1  # 37 frames: several tiles, the last one ragged
    # noise only: with a strong tone the weakest bins sit at the single-precision FFT's noise floor
    # (1e-7 of the peak), which channels of one or two bins would show as > 1e-4 in the logarithm
    pcm = _pcm(n, 1, 2, wlen + bank, 16000, tone=0)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    assert ref.shape[0] == 37
    # (double precision: frames of at most 512 samples take the pipelined double-precision kernel)
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [pcm.tobytes()], 1, 2)[0]
        _compare(got, ref, kw, f64)


@pytest.mark.parametrize("bank,lo,hi", [(10, 60, 7600), (52, 60, 7600), (15, 60, 7600), (17, -1, -1), (45, -1, -1)])
def test_lean_kernel_filterbanks_without_a_conflict_free_block_placement(oracle, bank, lo, hi):
    """The pipelined 256-point kernel reads its mel operands two bins at a time, so every block of
    the matrix-core filterbank must start on an even bin: filterbanks whose blocks then cannot all
    sit on different LDS banks (placed with conflicts), and filterbanks whose longest block fills
    the 32 steps from an odd bin (no placement: the general kernel takes the plan)."""
    kw = dict(sampleRate=16000, lowpassfre=lo, hipassfre=hi, preemphasise=0.97, wlen=400, inc=160, vecNum=1,
              fbankFlag=1, bankNum=bank, MFCCNum=min(bank, 13), MFCC0thFlag=0, energyFlag=0)
    pcm = _pcm(400 + 160 * 45 + 3, 1, 2, 500 + bank, 16000, tone=0)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    got = run_product_batch(ma, cfg, [pcm.tobytes()], 1, 2)[0]
    _compare(got, ref, kw, False)


@pytest.mark.parametrize("wlen,bank,vec", [(1025, 8, 1), (1100, 64, 2), (1153, 23, 3), (1500, 40, 4), (1665, 3, 2),
                                           (1764, 26, 4), (1792, 40, 1), (1793, 57, 2), (2000, 12, 3), (2047, 40, 2),
                                           (2048, 64, 4)])
@pytest.mark.parametrize("cepstral", [False, True])
@pytest.mark.parametrize("extras", [False, True])
def test_stereo_2048_point_kernel_modes(oracle, wlen, bank, vec, cepstral, extras):
    """24-bit stereo frames of 1025 to 2048 samples take the one-wave-per-frame kernel: every
    count of first-stage inputs per lane (9 to 16), 1 to 4 derived rows, few and many channels (mel
    segments cut into 1 to 10 parts of 15 or more bins), log-mel and cepstral outputs, with and
    without zero crossings / brightness / sub-bands / energy, odd frame shifts, a ragged last
    frame and utterances that start at odd byte offsets."""
    inc = wlen // 3 + 1
    kw = dict(sampleRate=44100, lowpassfre=-1 if bank % 2 else 30, hipassfre=-1 if bank % 3 else 20000,
              preemphasise=0.97, wlen=wlen, inc=inc, vecNum=vec, fbankFlag=0 if cepstral else 1, bankNum=bank,
              MFCCNum=min(bank, 20), MFCC0thFlag=1 if cepstral else 0, energyFlag=1 if extras else 0,
              zeroCrossingFlag=1 if extras else 0, brightFlag=1 if extras else 0,
              subBandEFlag=(8 if bank % 2 else 5) if extras else 0)
    pcms = [_pcm(wlen + k * inc + 7, 2, 3, wlen + bank + k, 44100, tone=0) for k in (9, 1, 0)]
    refs = [oracle.extract_pcm_bytes(oracle.make_config(**kw), p, 2, 3) for p in pcms]
    # (both precisions have a one-wave-per-frame kernel for this class)
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3, pad=3)
        for g, ref in zip(got, refs):
            _compare(g, ref, kw, f64)


@pytest.mark.parametrize("bands", [1, 2, 3, 4, 8, 16, 32])
def test_stereo_2048_point_kernels_every_kind_of_sub_band_count(oracle, bands):
    """Sub-band energies and brightness (main.c:235-236) in the 24-bit stereo kernels: band counts
    whose bands are whole strides of a lane's bins (2, 4, 8, 16: summed from registers, one code
    path per count) and the others (summed from the spectrum in the work area), both precisions."""
    kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=1764, inc=882, vecNum=2,
              fbankFlag=1, bankNum=26, MFCCNum=13, MFCC0thFlag=0, energyFlag=1, zeroCrossingFlag=1, brightFlag=1,
              subBandEFlag=bands)
    pcms = [_pcm(1764 + 882 * k + 11, 2, 3, 1200 + bands + k, 44100, tone=0) for k in (13, 2, 0)]
    ocfg = oracle.make_config(**kw)
    refs = [oracle.extract_pcm_bytes(ocfg, p, 2, 3) for p in pcms]
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3, pad=2)
        for g, ref in zip(got, refs):
            _compare(g, ref, kw, f64)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_frames_wider_than_a_wave_with_debug_columns_and_a_ragged_last_tile(oracle, precision):
    """Mono 24-bit frames of 2048 samples take the general kernel with two waves per frame; with the
    debug magnitude columns (fftLength, main.c:247) every frame slot of a tile -- the ones past a
    ragged tile's last frame too -- must take part in the workgroup's barriers.  (A slot that skipped
    one left the cepstra of the tile's last frame unwritten now and then: several runs.)"""
    kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.95, wlen=2048, inc=512, vecNum=1, fbankFlag=0,
              bankNum=39, MFCCNum=39, MFCC0thFlag=0, energyFlag=0, zeroCrossingFlag=1, brightFlag=1, subBandEFlag=8,
              fftLength=5, regreOrder=4, delwin=2)
    f64 = precision == "f64"
    pcms = [_pcm(2048 + 512 * k + 100, 1, 3, 31 + k, 44100, tone=0) for k in (26, 10, 5)]
    refs = [oracle.extract_pcm_bytes(oracle.make_config(**kw), p, 1, 3) for p in pcms]
    cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
    for _ in range(6):
        got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 1, 3, pad=0)
        for g, ref in zip(got, refs):
            _compare(g, ref, kw, f64)


@pytest.mark.parametrize("frames_per_comb", [1, 2, 3, 4, 7])
def test_stereo_2048_point_kernels_over_many_short_utterances(oracle, frames_per_comb, tuning):
    """The 24-bit stereo kernels hand runs of 12 x C consecutive frames to the workgroups and let the
    waves claim the combs of a run: every C, utterances of 1 to 70 frames (runs that straddle
    utterances, ragged last runs, fewer frames than waves), both precisions."""
    tuning(leanChunk=frames_per_comb)
    kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=1764, inc=882, vecNum=2,
              fbankFlag=0, bankNum=26, MFCCNum=13, MFCC0thFlag=1, energyFlag=1, zeroCrossingFlag=1, brightFlag=1,
              subBandEFlag=8)
    rng = np.random.default_rng(77 + frames_per_comb)
    lens = [int(1764 + 882 * rng.integers(0, 70) + rng.integers(0, 882)) for _ in range(60)] + [1764, 1763]
    pcms = [_pcm(n, 2, 3, 900 + i, 44100, tone=0) for i, n in enumerate(lens)]
    ocfg = oracle.make_config(**kw)
    refs = [oracle.extract_pcm_bytes(ocfg, p, 2, 3) for p in pcms]
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3, pad=1)
        for g, ref in zip(got, refs):
            assert g.shape == ref.shape
            if ref.shape[0]:
                _compare(g, ref, kw, f64)


def test_stereo_2048_point_kernels_with_the_largest_cepstral_table(oracle):
    """64 channels, 64 coefficients + C0, four rows: the double-precision kernel's cepstral table no
    longer fits the LDS beside twelve work areas and is read from memory; the single-precision
    kernel's still fits."""
    kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=2048, inc=1024, vecNum=4,
              fbankFlag=0, bankNum=64, MFCCNum=64, MFCC0thFlag=1, energyFlag=1, zeroCrossingFlag=1, brightFlag=1,
              subBandEFlag=8)
    pcms = [_pcm(2048 + 1024 * k + 5, 2, 3, 700 + k, 44100, tone=0) for k in (7, 2)]
    refs = [oracle.extract_pcm_bytes(oracle.make_config(**kw), p, 2, 3) for p in pcms]
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3, pad=5)
        for g, ref in zip(got, refs):
            _compare(g, ref, kw, f64)


@pytest.mark.parametrize("wlen,inc", [(1025, 1), (1025, 342), (1765, 441), (2047, 700), (2048, 512), (1764, 882)])
def test_stereo_frames_at_the_first_and_last_bytes_of_the_buffer(oracle, wlen, inc):
    """The 24-bit stereo kernels read a frame's bytes straight from the buffer: a frame whose last
    sample is the buffer's last (odd and even frame lengths: the last sample pair then lies half or
    wholly behind the end), frames that start within the buffer's first 8 bytes (shift of one
    sample), in both precisions, alone and behind a neighbour."""
    kw = dict(sampleRate=44100, lowpassfre=-1, hipassfre=-1, preemphasise=0.97, wlen=wlen, inc=inc, vecNum=4,
              fbankFlag=1, bankNum=24, MFCCNum=12, MFCC0thFlag=0, energyFlag=1, zeroCrossingFlag=1, brightFlag=1,
              subBandEFlag=8)
    ocfg = oracle.make_config(**kw)
    for lens in ([wlen + 5 * inc], [wlen + 2 * inc + 3, wlen + 3 * inc], [wlen]):
        pcms = [_pcm(n, 2, 3, wlen + i, 44100, tone=0) for i, n in enumerate(lens)]
        refs = [oracle.extract_pcm_bytes(ocfg, p, 2, 3) for p in pcms]
        for f64 in (False, True):
            cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
            got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3, pad=0)
            for g, ref in zip(got, refs):
                _compare(g, ref, kw, f64)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MFCC_FUZZ_FIRST", "0")),
                                        int(os.environ.get("MFCC_FUZZ_FIRST", "0")) + int(os.environ.get("MFCC_FUZZ_SEEDS", "48"))))
def test_random_configurations(oracle, seed, tuning):
    """Seeded random walk over the option space (frame geometry, channel layouts, feature flags,
    delta orders and windows, z-norm, batches of ragged utterances at odd byte offsets): whichever
    kernel mode, tile geometry and delta path the plan picks must agree with the oracle.  Every
    third seed takes the row-block delta / z-norm kernels instead of the whole-utterance one, every
    third the streaming whole-utterance kernel (where its LDS regions fit; else the plan falls back)."""
    if seed % 3 == 2:
        tuning(noUttFinish=1)
    if seed % 3 == 1:
        tuning(finish=2, streamRows=(0, 16, 8)[seed % 9 // 3])
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
    plain = bool(rng.integers(0, 2))  # half of the draws stay inside the lean kernel's option set
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
    pcms = [_pcm(n, ch, bps, seed * 10 + i, sr, tone=0).tobytes() for i, n in enumerate(lengths)]
    ocfg = oracle.make_config(**kw)
    refs = [oracle.extract_pcm_bytes(ocfg, np.frombuffer(p, dtype=np.uint8), ch, bps) for p in pcms]
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        gots = run_product_batch(ma, cfg, pcms, ch, bps, pad=int(rng.integers(0, 7)))
        for got, ref, p in zip(gots, refs, pcms):
            assert got.shape == ref.shape
            if ref.shape[0] >= 5 or not kw["znormFlag"]:  # z-norm over a few rows magnifies last-bit differences (0/0 vs +-1)
                _compare(got, ref, kw, f64, src=(oracle, p, ch, bps))


@pytest.mark.parametrize("finish", ["utterance", "blocks", "too long", "stream", "stream, long"])
def test_ragged_batch_and_edge_lengths(oracle, finish, tuning):
    """Empty, shorter than a window, exactly one window, one sample short of two frames, long;
    with the whole-utterance finishing kernel, with the row-block kernels forced, with one
    utterance too long for the former (which moves the whole plan to the latter), and with the
    streaming kernel (chunk sizes 8 and 32, utterances shorter than a chunk and of many chunks)."""
    if finish == "blocks":
        tuning(noUttFinish=1)
    if finish.startswith("stream"):
        tuning(finish=2, streamRows=8 if finish == "stream" else 0)
    kw = CONFIGS["c3"]
    lens = [0, 1, 399, 400, 559, 560, 561, 5000, 16000, 401, 33333] + ([400 + 160 * 2500] if finish in ("too long", "stream, long") else [])
    pcms = [_pcm(n, 1, 2, 100 + i, 16000) for i, n in enumerate(lens)]
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 1, 2)
    ocfg = oracle.make_config(**kw)
    for n, p, g in zip(lens, pcms, got):
        assert g.shape[0] == oracle.num_frames(n, 400, 160) == ma.num_frames(n, 400, 160)
        if g.shape[0] > 1:
            _compare(g, oracle.extract_pcm_bytes(ocfg, p, 1, 2), kw, False, src=(oracle, p, 1, 2))
        elif g.shape[0] == 1:  # z-norm of a single frame is 0/0 in both
            assert np.all(np.isnan(g))


@pytest.mark.parametrize("ch,bps", [(1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (2, 3), (1, 4), (2, 4)])
@pytest.mark.parametrize("shift", [0, 1, 2, 3, 6])
def test_pcm_layouts_and_misaligned_offsets(oracle, ch, bps, shift):
    """Every sample size / channel count, with the utterance starting at odd byte offsets."""
    kw = dict(CONFIGS["c2"], vecNum=ch, zeroCrossingFlag=1)
    n = 3000
    if bps == 4:
        x = np.frombuffer(_pcm(n, ch, 3, 40 + shift, 16000).tobytes(), dtype=np.uint8).reshape(-1, 3)
        pcm = np.concatenate([np.full((x.shape[0], 1), 7, np.uint8), x], axis=1).reshape(-1)  # x * 256 + 7
    else:
        pcm = _pcm(n, ch, bps, 40 + shift, 16000)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
    import torch
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    ctx = ma.Context(cfg, 0)
    buf = np.zeros(pcm.size + 64, dtype=np.uint8)
    buf[shift:shift + pcm.size] = pcm
    plan = ma.Plan(ctx, ch, bps, [shift], [n])
    out = plan.extract(torch.from_numpy(buf).cuda())
    torch.cuda.synchronize()
    _compare(out.double().cpu().numpy(), ref, kw, False)


def test_zero_mean_and_stereo_rows(oracle):
    kw = dict(CONFIGS["c1b"], zeroMeanSigFlag=1, preemphasise=0.95)  # vecNum 4: L, R, mid, side
    pcm = _pcm(30000, 2, 3, 11, 44100)
    x = np.frombuffer(pcm.tobytes(), dtype=np.uint8).astype(np.int64).reshape(-1, 2, 3)
    dc = (x[..., 0] | (x[..., 1] << 8) | (x[..., 2] << 16))
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 2, 3)
    for f64 in (False, True):
        cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
        got = run_product_batch(ma, cfg, [pcm.tobytes()], 2, 3)[0]
        _compare(got, ref, kw, f64)
    assert dc.shape[0] == 30000


def test_fft_length_debug_columns(oracle):
    kw = dict(CONFIGS["c2"], fftLength=7, brightFlag=1, subBandEFlag=3)
    pcm = _pcm(4000, 1, 2, 12, 16000)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], 1, 2)[0]
    _compare(got, ref, kw, False)


def test_silence_matches_reference_semantics(oracle):
    """Digital silence: log-mel floored to 0, log(te) = -inf, brightness / sub-bands NaN, deltas of
    those NaN (main.c:243 has no floor)."""
    kw = dict(CONFIGS["c1a"])
    pcm = np.zeros(20000 * 2 * 3, dtype=np.uint8)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 2, 3)
    got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], 2, 3)[0]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    assert np.all(got[fin] == ref[fin])


def test_zcr_is_exact_on_near_cancelling_samples(oracle):
    """x[i] close to k*x[i-1]: the sign of the pre-emphasised sample decides a zero crossing; the
    kernel's separately rounded double arithmetic must reproduce the reference's sign."""
    kw = dict(CONFIGS["c2"], zeroCrossingFlag=1)
    rng = np.random.default_rng(5)
    base = rng.integers(100, 30000, size=6000)
    x = np.empty(6000, dtype=np.int16)
    x[0::2] = base[0::2] // 100 * 100
    x[1::2] = (x[0::2].astype(np.int64) * 97) // 100  # x[i] = 0.97 x[i-1] exactly in integers
    pcm = x.view(np.uint8)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], 1, 2)[0]
    assert np.array_equal(got[:, 40], ref[:, 40].astype(np.float32).astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("vec_num", [2, 4])
@pytest.mark.parametrize("density", [1, 40])
def test_zcr_is_exact_on_near_cancelling_stereo_24_bit_samples(oracle, vec_num, density):
    """The same for 24-bit stereo frames of 2048 points (left, right, mid and side rows): the
    kernel takes the signs from single-precision samples and redoes a frame in double when one of
    its samples is too close to zero to be sure.  density 1: every other sample cancels (every
    frame is redone); density 40: one pair in 40 (a mix of both paths)."""
    kw = dict(CONFIGS["c4"], vecNum=vec_num)
    rng = np.random.default_rng(17 + density)
    n = 40000
    x = rng.integers(-8000000, 8000000, size=(n, 2)).astype(np.int64)
    idx = np.arange(0, n - 1, 2 * density)
    x[idx] = x[idx] // 200 * 200
    x[idx + 1] = x[idx] * 97 // 100  # exactly 0.97 x[i-1] in integers, in both channels (so in mid and side too)
    pcm = np.zeros((n, 2, 3), dtype=np.uint8)
    for b in range(3):
        pcm[:, :, b] = (x >> (8 * b)) & 0xff
    pcm = pcm.reshape(-1)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 2, 3)
    got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], 2, 3)[0]
    kinds = column_kinds(kw)
    d = len(kinds)
    zc = kinds.index("zc")
    cols = [v * d + zc for v in range(vec_num)]
    assert np.array_equal(got[:, cols], ref[:, cols].astype(np.float32).astype(np.float64))


def test_wapper_entry_points_and_writers(oracle, tmp_path):
    """MFCCWapper / MFCCWapperMem / MFCCWapperList end to end: files in, files out."""
    kw = dict(CONFIGS["c3"])
    pcm = _pcm(8000, 1, 2, 21, 16000)
    wav = tmp_path / "a.wav"
    wav.write_bytes(ma.wav_bytes(pcm, 1, 2, 16000, extra_chunk=(b"LIST", b"info")))
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 1, 2)
    cfg = ma.make_config(precision=ma.PREC_F32, saveType=ma.SAVE_N, **kw)
    out = tmp_path / "a.npy"
    feat = ma.wapper(cfg, str(wav), str(out))
    _compare(feat, ref, kw, False)
    assert np.array_equal(np.load(out), feat)
    feat2 = ma.wapper_mem(cfg, wav.read_bytes())
    assert np.array_equal(feat, feat2)
    with pytest.raises(ma.MFCCError) as e:
        ma.wapper(ma.make_config(precision=ma.PREC_F32, **dict(kw, sampleRate=8000)), str(wav))
    assert e.value.code == ma.E_SAMPLE_RATE
    with pytest.raises(ma.MFCCError) as e:
        ma.wapper(ma.make_config(precision=ma.PREC_F32, **dict(kw, vecNum=2)), str(wav))
    assert e.value.code == ma.E_VECNUM
    # whole-run entry point with a list of three files, one of them unreadable
    pcm2 = _pcm(5000, 1, 2, 22, 16000)
    (tmp_path / "b.wav").write_bytes(ma.wav_bytes(pcm2, 1, 2, 16000))
    lst = tmp_path / "list.txt"
    lst.write_text("%s\t%s\n%s\t%s\n%s\t%s\n\nignored\tafter blank line\n" % (
        wav, tmp_path / "a.txt", tmp_path / "nope.wav", tmp_path / "nope.txt", tmp_path / "b.wav", tmp_path / "b.txt"))
    ini = tmp_path / "config.ini"
    ini.write_text("[Frame];\nsampleRate = 16000 ;\nhipassfre = 8000 ;\nlowpassfre = 64 ;\npreemphasise = 0.97 ;\n"
                   "wlen = 400 ;\ninc = 160 ;\nsaveType = e ;\nvecNum = 1 ;\nfileList = %s ;\n[MFCC];\n"
                   "fbankFlag = 0 ;\nbankNum = 26 ;\nMFCCNum = 13 ;\nMFCC0thFlag = 1 ;\n[Others];\n"
                   "energyFlag = 1 ;\nzeroCrossingFlag = 0 ;\nbrightFlag = 0 ;\nsubBandEFlag = 0 ;\n"
                   "fftLength = 0 ;\n[Regression];\nregreOrder = 3 ;\ndelwin = 2 ;\nznormFlag = 1 ;\n"
                   "[IO];\nnumThreads = 2 ;\n" % lst)
    os.environ["MFCC_QUIET"] = "1"
    assert ma.wapper_list(str(ini)) == 1  # one failed file, the run goes on
    for name, p in (("a.txt", pcm), ("b.txt", pcm2)):
        rows = [[float(v) for v in line.split("\t")[:-1]] for line in open(tmp_path / name)]
        r = oracle.extract_pcm_bytes(oracle.make_config(**kw), p, 1, 2)
        got = np.array(rows)
        assert got.shape == r.shape
        _compare(got, r, kw, False)
        assert open(tmp_path / name).readline().endswith("\t\n")  # main.c:277-278 layout
    assert not (tmp_path / "nope.txt").exists()


def test_wapper_list_batches_mixed_layouts(oracle, tmp_path):
    """The whole-run entry point over a list whose PCM layout changes back and forth (every change
    closes a batch and carries the files already read over to the next one), with a missing file
    and a file shorter than one frame in between: every output equals the per-file entry point's."""
    kw = dict(CONFIGS["c3"], vecNum=1)
    cfg = ma.make_config(saveType=ma.SAVE_B, **kw)  # (the ini below names no precision: the default)
    layouts = [(1, 2), (1, 2), (2, 3), (1, 2), (2, 2), (2, 2), (1, 3), (1, 2)]
    lines, expect = [], {}
    for i in range(40):
        ch, bps = layouts[i % len(layouts)]
        n = 100 if i == 13 else 3000 + 517 * i
        pcm = _pcm(n, ch, bps, 100 + i, 16000)
        wav = tmp_path / ("f%02d.wav" % i)
        wav.write_bytes(ma.wav_bytes(pcm, ch, bps, 16000))
        if i == 7:
            wav = tmp_path / "missing.wav"
        lines.append("%s\t%s" % (wav, tmp_path / ("f%02d.bin" % i)))
        if i != 7:
            expect[i] = (pcm, ch, bps)
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(lines) + "\n")
    ini = tmp_path / "config.ini"
    sections = {"Frame": ["sampleRate", "hipassfre", "lowpassfre", "preemphasise", "wlen", "inc", "vecNum",
                          "zeroMeanSigFlag"],
                "MFCC": ["fbankFlag", "bankNum", "MFCCNum", "MFCC0thFlag"],
                "Others": ["energyFlag", "zeroCrossingFlag", "brightFlag", "subBandEFlag", "fftLength"],
                "Regression": ["regreOrder", "delwin", "znormFlag"]}
    text = ""
    for sec, keys in sections.items():
        text += "[%s]\n" % sec + "".join("%s = %s\n" % (k, kw[k]) for k in keys if k in kw)
    assert sum(k in kw for keys in sections.values() for k in keys) == len(kw)
    os.environ["MFCC_QUIET"] = "1"
    step = ma.step(cfg)
    # one device thread, then three (the path a multi-GPU node takes: every thread draws batches)
    for devices in ("", "numDevices = -3\n"):
        ini.write_text(text + "[IO]\nsaveType = b\nnumThreads = 3\n%sfileList = %s\n" % (devices, lst))
        for f in tmp_path.glob("*.bin"):
            f.unlink()
        assert ma.wapper_list(str(ini)) == 1
        assert not (tmp_path / "f07.bin").exists()
        for i, (pcm, ch, bps) in expect.items():
            got = np.fromfile(tmp_path / ("f%02d.bin" % i), dtype=np.float64).reshape(-1, step)
            want = ma.wapper_mem(cfg, ma.wav_bytes(pcm, ch, bps, 16000))
            assert got.shape == want.shape, i
            assert np.array_equal(got, want, equal_nan=True), i
    assert (tmp_path / "f13.bin").stat().st_size == 0  # shorter than one frame: no rows
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), expect[2][0], 2, 3)
    got = np.fromfile(tmp_path / "f02.bin", dtype=np.float64).reshape(-1, step)
    _compare(got, ref, kw, False)


@pytest.mark.parametrize("name,n,ch,bps", [("c5", 60000, 2, 3), ("c3", 20000, 1, 2)])
def test_frame_sharded_recording_matches_whole(oracle, name, n, ch, bps):
    """Shard plans (halo frames, lead instant, split z-norm) for 3 ranks emulated on one GPU."""
    import torch
    from mfcc_amd import sharding
    kw = dict(CONFIGS[name], zeroMeanSigFlag=1)
    pcm = _pcm(n, ch, bps, 31, kw["sampleRate"])
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    ctx = ma.Context(cfg, 0)
    dev = torch.from_numpy(pcm).cuda()
    total = ma.num_frames(n, kw["wlen"], kw["inc"])
    world, bpi = 3, ch * bps
    ranges = sharding.frame_ranges(total, world)
    # whole-recording channel sums from per-rank partial sums (any partition of the instants)
    cuts = [0, n // 3, n // 2, n]
    parts = [ma.channel_sums(ctx, dev, ch, bps, cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]
    sums2 = sum(parts).cpu().numpy()
    x = np.frombuffer(pcm.tobytes(), dtype=np.uint8).astype(np.int64).reshape(-1, ch, bps)
    v = sum(x[..., k] << (8 * k) for k in range(bps))
    v = np.where(v >= 1 << (8 * bps - 1), v - (1 << (8 * bps)), v)
    L = int(v[:, 0].sum())
    R = int(v[:, 1].sum()) if ch == 2 else 0
    assert list(sums2) == [2 * L, 2 * R, L + R, 2 * (L - R)]
    plans, outs, sums = [], [], []
    for fb, fe in ranges:
        g = ma.shard_geometry(cfg, n, fb, fe)
        shard = dev[g["sample_begin"] * bpi:(g["sample_begin"] + g["sample_count"]) * bpi].clone()
        plan = ma.Plan(ctx, ch, bps, shard=(n, fb, fe, sums2))
        out = plan.extract(shard)
        plans.append(plan)
        outs.append(out)
        sums.append(plan.shard_column_sums(out))
    tot = sums[0] + sums[1] + sums[2]  # rank-ordered sum, as sharding.combine_sums does
    pieces = []
    for plan, out in zip(plans, outs):
        plan.shard_normalize(out, tot)
        pieces.append(out[plan.owned_row_begin:plan.owned_row_begin + plan.owned_rows])
    torch.cuda.synchronize()
    got = torch.cat(pieces).double().cpu().numpy()
    _compare(got, ref, kw, False, src=(oracle, pcm, ch, bps))


def test_batch_order_and_shift_invariance_at_full_size():
    """Benchmark-size properties (16 kHz, 10 s utterances, 40-bin log-mel): (1) an utterance's rows
    do not depend on its position in the batch or on its neighbours; (2) delaying the signal by
    one hop shifts the rows by one frame (pre-emphasis off, so no boundary effect)."""
    import torch
    kw = dict(CONFIGS["c2"], preemphasise=0.0)
    n, U = 160000, 64
    pcms = [_pcm(n, 1, 2, 500 + u, 16000, tone=0) for u in range(U)]
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    a = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 1, 2)
    perm = np.random.default_rng(1).permutation(U)
    b = run_product_batch(ma, cfg, [pcms[i].tobytes() for i in perm], 1, 2)
    for j, i in enumerate(perm):
        assert np.array_equal(a[i], b[j])
    assert a[0].shape == (998, 40)
    shifted = np.concatenate([np.zeros(160 * 2, dtype=np.uint8), pcms[0]])[:n * 2]
    c = run_product_batch(ma, cfg, [shifted.tobytes()], 1, 2)[0]
    assert np.array_equal(c[1:], a[0][:-1])


def test_batch_order_and_shift_invariance_of_the_stereo_class_at_full_size():
    """The same two properties at the size of the 2048-point stereo benchmark (44.1 kHz 24-bit stereo,
    60 s files, left and right rows with cepstra, zero crossings, brightness and sub-bands): rows do
    not depend on the file's place in the batch, and a delay of one hop shifts them by one frame."""
    kw = dict(CONFIGS["c4"], preemphasise=0.0)
    n, U, sr = 60 * 44100, 6, 44100
    pcms = [_pcm(n, 2, 3, 700 + u, sr, tone=0) for u in range(U)]
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    a = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 2, 3)
    perm = np.random.default_rng(2).permutation(U)
    b = run_product_batch(ma, cfg, [pcms[i].tobytes() for i in perm], 2, 3)
    for j, i in enumerate(perm):
        assert np.array_equal(a[i], b[j], equal_nan=True)
    assert a[0].shape == (ma.num_frames(n, kw["wlen"], kw["inc"]), 2 * len(column_kinds(kw)))
    hop = kw["inc"] * 2 * 3
    shifted = np.concatenate([np.zeros(hop, dtype=np.uint8), pcms[0]])[:n * 6]
    c = run_product_batch(ma, cfg, [shifted.tobytes()], 2, 3)[0]
    assert np.array_equal(c[1:], a[0][:-1], equal_nan=True)


def test_shards_equal_the_whole_on_a_long_stereo_recording():
    """Frame-sharded extraction of a ten-minute 44.1 kHz 24-bit stereo recording with the example
    option set (306 columns, deltas over 9 frames): without z-norm 2, 3 and 7 frame ranges give the
    bits of one; with z-norm (column sums added in shard order) they agree to 5e-5."""
    n = 600 * 44100
    pcm = _pcm(n, 2, 3, 801, 44100, tone=0)
    kw = dict(CONFIGS["c5"], znormFlag=0)
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    one = ma.extract_recording(cfg, pcm, 2, 3, num_shards=1)
    assert one.shape == (ma.num_frames(n, 1764, 882), 306)
    for shards in (2, 3, 7):
        many = ma.extract_recording(cfg, pcm, 2, 3, num_shards=shards, devices=[0] * shards)
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
    cfg = ma.make_config(precision=ma.PREC_F32, **CONFIGS["c5"])
    one = ma.extract_recording(cfg, pcm, 2, 3, num_shards=1)
    many = ma.extract_recording(cfg, pcm, 2, 3, num_shards=4, devices=[0] * 4)
    assert np.array_equal(np.isnan(one), np.isnan(many))
    assert float(np.nanmax(np.abs(one - many))) < 5e-5


def test_byte_offsets_beyond_two_gib():
    """A PCM buffer larger than 2^31 bytes (2.4 hours of 44.1 kHz 24-bit stereo, the example option
    set): (1) an utterance that starts beyond that offset comes out bit for bit as it does alone;
    (2) the frames of the whole recording that lie beyond it equal those of an excerpt (static
    columns: every frame; delta columns: the frames whose regression window lies inside the excerpt)."""
    import ctypes as C
    import threading
    import torch
    kw = CONFIGS["c1a"]
    sr, inc, wlen, bpi = 44100, kw["inc"], kw["wlen"], 6
    n = 400_000_000
    assert n * bpi > 2 ** 31 + 2 ** 27
    buf = np.empty(n * bpi, dtype=np.uint8)
    L, threads = ma.lib(), 16
    per = (n + threads - 1) // threads

    def fill(t):
        a, b = t * per, min(n, (t + 1) * per)
        for ch in range(2):
            L.MFCCSynthPcmAt(C.c_void_p(buf.ctypes.data + a * bpi), 2, ch, 3, a, b - a, C.c_uint64(0x5EED0777 + ch), sr, 0)

    ths = [threading.Thread(target=fill, args=(t,)) for t in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    T = ma.num_frames(n, wlen, inc)
    f0 = T - 2000                      # first frame of the excerpt
    s0 = f0 * inc                      # its first sample: byte offset 6 s0 > 2^31
    assert s0 * bpi > 2 ** 31
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    ctx = ma.Context(cfg, 0)
    dev = torch.from_numpy(buf).cuda()
    alone = ma.Plan(ctx, 2, 3, [0], [n - s0]).extract(dev[s0 * bpi:].clone()).cpu().numpy()
    assert alone.shape == (2000, 306)
    two = ma.Plan(ctx, 2, 3, [0, s0 * bpi], [s0, n - s0])
    out = two.extract(dev).cpu().numpy()
    assert np.array_equal(out[two.row_offset(1):], alone, equal_nan=True)
    del out
    whole = ma.Plan(ctx, 2, 3, [0], [n]).extract(dev).cpu().numpy()
    assert whole.shape == (T, 306)
    assert np.array_equal(whole[f0:, :102], alone[:, :102], equal_nan=True)
    halo = 2 * kw["delwin"]
    assert np.array_equal(whole[f0 + halo:T - halo, 102:], alone[halo:2000 - halo, 102:], equal_nan=True)
    ctx.close()


def test_ragged_batch_rows_equal_the_lone_utterances_at_size():
    """600 utterances of 2 to 20 s (13 MFCC + C0 + energy, deltas, z-norm: the row-block delta and
    z-norm kernels over 600 000 rows): a sample of them, the shortest and the longest included,
    comes out bit for bit as it does alone."""
    kw = CONFIGS["c3"]
    rng = np.random.default_rng(12)
    lengths = rng.integers(2 * 16000, 20 * 16000 + 1, size=600)
    pcms = [_pcm(int(n), 1, 2, 3000 + u, 16000, tone=0) for u, n in enumerate(lengths)]
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], 1, 2)
    assert sum(g.shape[0] for g in got) == sum(ma.num_frames(int(n), 400, 160) for n in lengths)
    picks = {int(np.argmin(lengths)), int(np.argmax(lengths)), 0, 299, 599} | set(int(i) for i in rng.integers(0, 600, size=7))
    for u in sorted(picks):
        alone = run_product_batch(ma, cfg, [pcms[u].tobytes()], 1, 2)[0]
        assert np.array_equal(got[u], alone, equal_nan=True), u


def test_linearity_of_the_power_path_at_full_size():
    """Scaling the PCM by 2 adds ln 4 to every log-mel value above the floor and to log-energy."""
    kw = dict(CONFIGS["c2"], energyFlag=1)
    n = 160000
    x = np.frombuffer(_pcm(n, 1, 2, 900, 16000, tone=0).tobytes(), dtype=np.int16) // 4
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    a = run_product_batch(ma, cfg, [x.astype(np.int16).tobytes()], 1, 2)[0]
    b = run_product_batch(ma, cfg, [(x * 2).astype(np.int16).tobytes()], 1, 2)[0]
    assert np.all(a[:, :40] > 1.0)
    assert np.max(np.abs(b - a - np.log(4.0))) < 2e-5


def test_kernel_timing_api():
    import torch
    kw = CONFIGS["c2"]
    pcm = _pcm(160000, 1, 2, 1, 16000)
    ctx = ma.Context(ma.make_config(precision=ma.PREC_F32, **kw), 0)
    plan = ma.Plan(ctx, 1, 2, [0], [160000])
    assert plan.last_kernel_ms() < 0
    plan.enable_timing(True)
    plan.extract(torch.from_numpy(pcm).cuda())
    ms = plan.last_kernel_ms()
    assert 0 < ms < 1000


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_lower_level_api_pieces(oracle, precision):
    """FFT, Realft, Wave2FBank, FBank2MFCC / FBank2C0 (mfcc.h:29-74) as batched device calls against
    the oracle's restatements of the same functions."""
    import torch
    f64 = precision == "f64"
    dt = torch.float64 if f64 else torch.float32
    rng = np.random.default_rng(77)
    for n in (2, 4, 16, 512, 2048, 16384):
        x = rng.standard_normal((5, n))
        tol = (1e-11 if f64 else 2e-5) * np.sqrt(n) * 4
        for inv in (0, 1):
            want = np.stack([oracle.fft(r, inv) for r in x])
            got = ma.fft_device(torch.tensor(x, dtype=dt, device="cuda"), bool(inv)).double().cpu().numpy()
            assert np.max(np.abs(got - want)) < tol * (1 if not inv else 2.0 / n + 1e-3), (n, inv)
        if n >= 4:
            want = np.stack([oracle.realft(r) for r in x])
            got = ma.realft_device(torch.tensor(x, dtype=dt, device="cuda")).double().cpu().numpy()
            assert np.max(np.abs(got - want)) < tol, n
            assert np.all(got[:, 1] == 0.0)
    kw = dict(CONFIGS["c3"])
    cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
    ctx = ma.Context(cfg, 0)
    frames = rng.standard_normal((33, kw["wlen"])) * 3000.0
    frames[0] = 0.0                         # silence: every channel at the floor
    frames[1] = 0.0
    frames[1, 7] = 12345.0                  # single impulse
    fb, te = ctx.wave2fbank(torch.tensor(frames, dtype=dt, device="cuda"))
    fb, te = fb.double().cpu().numpy(), te.double().cpu().numpy()
    for i, fr in enumerate(frames):
        wf, wte, _, _ = oracle.wave2fbank(fr, kw["sampleRate"], kw["bankNum"], kw["lowpassfre"], kw["hipassfre"])
        assert np.max(np.abs(fb[i] - wf)) < (1e-10 if f64 else 1e-4), i
        assert abs(te[i] - wte) <= (1e-12 if f64 else 1e-5) * max(wte, 1.0), i
    # a context with usePower = 0 (and, in double, takeLogs = 0) gives this piece those options too
    for up, tl in ((0, 1), (0, 0), (1, 0)):
        if not tl and not f64:
            continue  # sums without logs span 1e15: compared in double only
        ctx2 = ma.Context(ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **dict(kw, usePower=up, takeLogs=tl)), 0)
        fb2, _ = ctx2.wave2fbank(torch.tensor(frames, dtype=dt, device="cuda"))
        fb2 = fb2.double().cpu().numpy()
        for i, fr in enumerate(frames):
            wf, _, _, _ = oracle.wave2fbank(fr, kw["sampleRate"], kw["bankNum"], kw["lowpassfre"], kw["hipassfre"], up, tl)
            assert np.max(np.abs(fb2[i] - wf)) < (1e-10 * max(1.0, float(np.max(np.abs(wf)))) if f64 else 1e-4), (up, tl, i)
        ctx2.close()
    fbank = rng.standard_normal((40, kw["bankNum"])) * 5 + 10
    m, c0 = ctx.fbank2mfcc(torch.tensor(fbank, dtype=dt, device="cuda"))
    m, c0 = m.double().cpu().numpy(), c0.double().cpu().numpy()
    for i, f in enumerate(fbank):
        wm, wc0 = oracle.fbank2mfcc(f, kw["MFCCNum"])
        scale = max(1.0, float(np.max(np.abs(wm))))
        assert np.max(np.abs(m[i] - wm)) < (1e-11 if f64 else 1e-5) * scale
        assert abs(c0[i] - wc0) < (1e-11 if f64 else 1e-5) * max(1.0, abs(wc0))
    ctx.close()
    fres = 1.0e7 / (1e7 / 16000 * 512 * 700.0)
    assert abs(ma.mel(17, fres) - 1127 * np.log(1 + 16 * fres)) < 1e-12


def test_reference_named_functions_on_one_based_vectors(oracle):
    """InitFBank / Wave2FBank / FBank2MFCC / FBank2C0 / FFT / Realft under the reference's own names
    and 1-based Vector convention (include/mfcc_compat.h), against the oracle's restatement of the
    same functions: tables bit-exact, arithmetic to double rounding."""
    import ctypes as C
    L = ma.compat_lib()
    rng = np.random.default_rng(9)

    def vec(a):
        v = L.MFCCCreateVector(len(a))
        for i, x in enumerate(a):
            v[i + 1] = float(x)
        return v

    def arr(v):
        return np.array([v[i + 1] for i in range(L.MFCCVectorSize(v))])

    for (W, sr, P, lo, hi) in [(400, 16000, 40, -1.0, -1.0), (400, 16000, 26, 64.0, 8000.0), (1764, 44100, 40, 10.0, 22050.0)]:
        info = L.InitFBank(W, 1e7 / sr, P, lo, hi, 1, 1, 0, 1.0, 0.0, 0.0)
        assert L.MFCCCompatStatus() == 0
        t = oracle.fbank_tables(W, sr, P, lo, hi)
        assert (info.fftN, info.klo, info.khi) == (t["fftN"], t["klo"], t["khi"])
        nby2 = info.fftN // 2
        assert np.array_equal(np.array([info.cf[i + 1] for i in range(P + 1)]), t["cf"])
        assert np.array_equal(np.array([info.loChan[i + 1] for i in range(nby2)]), t["loChan"])
        assert np.array_equal(np.array([info.loWt[i + 1] for i in range(nby2)]), t["loWt"])
        for kind in ("noise", "silence", "impulse"):
            frame = rng.standard_normal(W) * 4000.0 if kind == "noise" else np.zeros(W)
            if kind == "impulse":
                frame[5] = 20000.0
            s, fb = vec(frame), L.MFCCCreateVector(P)
            te, te2 = C.c_double(), C.c_double()
            L.Wave2FBank(s, fb, C.byref(te), C.byref(te2), info)
            assert L.MFCCCompatStatus() == 0
            wf, wte, wte2, wspec = oracle.wave2fbank(frame, sr, P, lo, hi)
            assert np.max(np.abs(arr(fb) - wf)) < 1e-10
            assert abs(te.value - wte) <= 1e-12 * max(wte, 1.0)
            assert abs(te2.value - wte2) <= 1e-12 * max(wte2, 1.0)
            spec = np.array([info.x[i + 1] for i in range(info.fftN)])
            assert np.max(np.abs(spec - wspec)) <= 1e-11 * max(1.0, float(np.max(np.abs(wspec))))
            c = L.MFCCCreateVector(13)
            L.FBank2MFCC(fb, c, 13)
            wm, wc0 = oracle.fbank2mfcc(wf, 13)
            assert np.max(np.abs(arr(c) - wm)) < 1e-9
            assert abs(L.FBank2C0(fb) - wc0) < 1e-9
            for v in (s, fb, c):
                L.MFCCFreeVector(v)
        L.MFCCFreeFBank(C.byref(info))
    for n in (4, 64, 512, 4096):
        x = rng.standard_normal(n)
        for inv in (0, 1):
            v = vec(x)
            L.FFT(v, inv)
            assert np.max(np.abs(arr(v) - oracle.fft(x, inv))) < 1e-10 * np.sqrt(n)
            L.MFCCFreeVector(v)
        v = vec(x)
        L.Realft(v)
        assert np.max(np.abs(arr(v) - oracle.realft(x))) < 1e-10 * np.sqrt(n)
        L.MFCCFreeVector(v)


def test_file_sharded_whole_run(tmp_path):
    """The whole-run entry point over ranks (files sharded by rank, nothing exchanged): the two
    ranks of a world of 2, run one after the other on this GPU, together write every output, each
    equal to the per-file entry point's."""
    from mfcc_amd import sharding
    kw = dict(CONFIGS["c2"])
    cfg = ma.make_config(saveType=ma.SAVE_B, **kw)  # (the ini below names no precision: the default)
    lines, pcms = [], []
    for i in range(9):
        pcm = _pcm(2000 + 900 * i, 1, 2, 300 + i, 16000)
        pcms.append(pcm)
        (tmp_path / ("u%d.wav" % i)).write_bytes(ma.wav_bytes(pcm, 1, 2, 16000))
        lines.append("%s\t%s" % (tmp_path / ("u%d.wav" % i), tmp_path / ("u%d.bin" % i)))
    lst = tmp_path / "all.list"
    lst.write_text("\n".join(lines) + "\n")
    ini = tmp_path / "config.ini"
    ini.write_text("[Frame]\nsampleRate = 16000\nhipassfre = %s\nlowpassfre = %s\npreemphasise = %s\nwlen = %d\n"
                   "inc = %d\nvecNum = 1\n[MFCC]\nfbankFlag = 1\nbankNum = %d\n[Others]\nenergyFlag = %d\n"
                   "[Regression]\nregreOrder = 1\n[IO]\nsaveType = b\nnumThreads = 2\nfileList = %s\n" % (
                       kw["hipassfre"], kw["lowpassfre"], kw["preemphasise"], kw["wlen"], kw["inc"], kw["bankNum"],
                       kw.get("energyFlag", 0), lst))
    os.environ["MFCC_QUIET"] = "1"
    for rank in range(2):
        rini, mine = sharding.split_file_list(str(ini), rank, 2, str(tmp_path / "ranks"))
        assert len(mine) in (4, 5)
        assert ma.wapper_list(rini) == 0
    step = ma.step(cfg)
    for i, pcm in enumerate(pcms):
        got = np.fromfile(tmp_path / ("u%d.bin" % i), dtype=np.float64).reshape(-1, step)
        assert np.array_equal(got, ma.wapper_mem(cfg, ma.wav_bytes(pcm, 1, 2, 16000)))
    assert sharding.wapper_list_sharded(ma, str(ini), str(tmp_path / "one")) == (0, 0)


def test_init_fbank_with_warping_and_double_fft(oracle):
    """InitFBank's options that main.c:168 leaves off (doubleFFT, alpha with warpLowCut/warpUpCut,
    mfcc.c:17-36,55-56,83-92) and FBankInfo's usePower = 0 / takeLogs = 0 (mfcc.c:154-157,171-176),
    through the reference-named functions, against the oracle's InitFBank and Wave2FBank."""
    import ctypes as C
    L = ma.compat_lib()
    O = oracle.lib()
    W, sr, P, lo, hi, alpha, wl, wu = 400, 16000, 24, 60.0, 7600.0, 1.1, 300.0, 5000.0
    sp = 1e7 / sr
    info = L.InitFBank(W, sp, P, lo, hi, 0, 0, 1, alpha, wl, wu)
    assert L.MFCCCompatStatus() == 0 and info.fftN == 1024
    fb = oracle.OrcFBank()
    assert O.orc_fbank_init_ex(C.byref(fb), W, sp, P, lo, hi, 0, 0, 1, alpha, wl, wu) == 0
    nby2 = fb.fftN // 2
    assert (info.fftN, info.klo, info.khi) == (fb.fftN, fb.klo, fb.khi)
    got_cf = np.array([info.cf[i + 1] for i in range(P + 1)])
    ref_cf = np.array([fb.cf[i] for i in range(P + 1)])
    assert np.max(np.abs(got_cf - ref_cf)) < 1e-9
    lo_chan = np.array([info.loChan[i + 1] for i in range(nby2)])
    lo_wt = np.array([info.loWt[i + 1] for i in range(nby2)])
    assert np.array_equal(lo_chan, np.array([fb.loChan[i] for i in range(nby2)]))
    assert np.max(np.abs(lo_wt - np.array([fb.loWt[i] for i in range(nby2)]))) < 1e-12
    # magnitude mode without logs
    rng = np.random.default_rng(3)
    frame = rng.standard_normal(W) * 2000.0
    s = L.MFCCCreateVector(W)
    for i, x in enumerate(frame):
        s[i + 1] = float(x)
    fbv = L.MFCCCreateVector(P)
    te, te2 = C.c_double(), C.c_double()
    L.Wave2FBank(s, fbv, C.byref(te), C.byref(te2), info)
    assert L.MFCCCompatStatus() == 0
    dp = C.POINTER(C.c_double)
    want = np.zeros(P)
    x = np.zeros(fb.fftN)
    rte, rte2 = C.c_double(), C.c_double()
    fr = np.ascontiguousarray(frame)
    O.orc_wave2fbank(fr.ctypes.data_as(dp), want.ctypes.data_as(dp), C.byref(rte), C.byref(rte2), C.byref(fb),
                     x.ctypes.data_as(dp))
    got = np.array([fbv[i + 1] for i in range(P)])
    assert np.max(np.abs(got - want)) < 1e-8 * max(1.0, float(np.max(want)))
    assert abs(te.value - rte.value) < 1e-9 * rte.value and abs(te2.value - rte2.value) < 1e-8 * rte2.value
    L.MFCCFreeVector(s)
    L.MFCCFreeVector(fbv)
    L.MFCCFreeFBank(C.byref(info))
    O.orc_fbank_free(C.byref(fb))


_WARP = dict(warpAlpha=1.1, warpLowCut=300.0, warpUpCut=5000.0, lowpassfre=60.0, hipassfre=7600.0)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("opts", [dict(usePower=0), dict(doubleFFT=1), _WARP, dict(takeLogs=0),
                                  dict(_WARP, usePower=0, doubleFFT=1, warpAlpha=0.9),
                                  dict(normEnergy=1, energyFlag=1, regreOrder=3),
                                  dict(normEnergy=1, energyFlag=1, regreOrder=2, znormFlag=1, doubleFFT=1)],
                         ids=["magnitude", "double-fft", "warp", "no-logs", "all-three", "norm-energy", "norm-energy-znorm"])
@pytest.mark.parametrize("name", ["c3", "c1a", "c2"])
def test_options_the_reference_leaves_off_in_the_extraction_path(oracle, name, opts, precision):
    """usePower = 0, takeLogs = 0, doubleFFT, the alpha warping (InitFBank arguments main.c:168
    fixes, mfcc.c:55-56,83-92,154-157,171-176) and NormaliseLogEnergy (commented call main.c:255)
    as options of the whole per-file path, against the oracle's extraction with the same options."""
    kw = dict(CONFIGS[name], **opts)
    f64 = precision == "f64"
    if not f64 and not kw.get("takeLogs", 1):
        pytest.skip("sums without logs span 1e15: compared in double only")
    ch, bps = _layout(kw)
    pcms = [_pcm(n, ch, bps, 11 + i, kw["sampleRate"]) for i, n in enumerate((kw["sampleRate"] // 2 + 57, 3 * kw["wlen"]))]
    cfg = ma.make_config(precision=ma.PREC_F64 if f64 else ma.PREC_F32, **kw)
    got = run_product_batch(ma, cfg, [p.tobytes() for p in pcms], ch, bps)
    for g, p in zip(got, pcms):
        ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), p, ch, bps)
        _compare(g, ref, kw, f64, src=(oracle, p, ch, bps))


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("name", ["c3", "c1a"])
def test_an_utterance_gives_the_same_bits_alone_and_beside_a_long_neighbour(name, precision, tuning):
    """The delta / z-norm kernel is chosen from the batch (whole-utterance kernel in registers or in
    two sweeps, the row-block kernels, or -- wide batches -- the streaming kernel): one file must come out
    bit-identical whatever its batch neighbours are (MFCCWapperList batches, MFCCWapper does not).
    One exception is measured here instead: double-precision z-normalised rows, whose column sums
    the variants add in different orders (and E[x^2] - mean^2 amplifies the last bits), agree to 1e-9
    of the column's largest magnitude."""
    kw = CONFIGS[name]
    ch, bps = _layout(kw)
    sr = kw["sampleRate"]
    short = _pcm(sr // 2 + 77, ch, bps, 31, sr)
    mid = _pcm(3 * sr, ch, bps, 32, sr)
    long_ = _pcm(40 * sr if name == "c3" else 12 * sr, ch, bps, 33, sr)
    cfg = ma.make_config(precision=ma.PREC_F64 if precision == "f64" else ma.PREC_F32, **kw)
    alone = run_product_batch(ma, cfg, [short.tobytes()], ch, bps)[0]
    alone_mid = run_product_batch(ma, cfg, [mid.tobytes()], ch, bps)[0]
    def same(a, b):
        if precision == "f64" and kw.get("znormFlag", 0):
            scale = np.maximum(np.nanmax(np.abs(b), axis=0, keepdims=True), 1.0)
            err = float(np.nanmax(np.abs(a - b) / scale))
            util._note("f64 z-norm, alone vs batch", err, 1e-9)
            return np.array_equal(np.isnan(a), np.isnan(b)) and err < 1e-9
        return np.array_equal(a, b, equal_nan=True)

    for batch in ([short, mid], [short, mid, long_]):
        got = run_product_batch(ma, cfg, [b.tobytes() for b in batch], ch, bps)
        assert same(got[0], alone)
        assert same(got[1], alone_mid)
    tuning(finish=2)  # what a wide batch takes
    got = run_product_batch(ma, cfg, [b.tobytes() for b in (short, mid, long_)], ch, bps)
    assert same(got[0], alone)
    assert same(got[1], alone_mid)


@pytest.mark.parametrize("name,frames,force_sweeps", [("c3", 1000, False), ("c3", 1100, False), ("c3", 97, True),
                                                      ("c1a", 100, False), ("c1a", 60, True), ("c3", 1400, False)])
def test_whole_utterance_finish_variants(oracle, name, frames, force_sweeps, tuning):
    """Both forms of the whole-utterance delta / z-norm kernel: deltas kept in registers (at most
    32 rows per lane: c3 1000 frames, c1a 100), the two-sweep form that longer utterances take (c3
    1100 frames) or the noFinishReg tuning key forces, and the row-block kernels once the utterance no longer
    fits the LDS budget (c3 1400 frames); ragged batch with a one-frame and an empty utterance."""
    if force_sweeps:
        tuning(noFinishReg=1)
    kw = dict(CONFIGS[name], znormFlag=1)
    ch, bps = _layout(kw)
    lens = [kw["wlen"] + (frames - 1) * kw["inc"] + 3, kw["wlen"], kw["wlen"] - 1, kw["wlen"] + 7 * kw["inc"]]
    pcms = [_pcm(n, ch, bps, 500 + i, kw["sampleRate"]) for i, n in enumerate(lens)]
    got = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [p.tobytes() for p in pcms], ch, bps)
    ocfg = oracle.make_config(**kw)
    assert got[0].shape[0] == frames and got[2].shape[0] == 0
    for p, g in zip(pcms, got):
        if g.shape[0] >= 5:
            _compare(g, oracle.extract_pcm_bytes(ocfg, p, ch, bps), kw, False, src=(oracle, p, ch, bps))


def test_wapper_from_many_threads_with_more_configs_than_cache_slots():
    """The reference's model is a thread per file (README.md:19,134): 8 threads call MFCCWapperMem
    with 12 different option sets (the context cache holds 8, so entries are pushed out while
    other threads still run on them); every result must equal the single-threaded one bit for bit."""
    import threading
    variants = []
    for i in range(12):
        kw = dict(CONFIGS["c3" if i % 2 else "c2"])
        kw["bankNum"] = 24 + i  # 12 distinct table sets
        variants.append(kw)
    pcm = _pcm(16000 * 2 + 37, 1, 2, 77, 16000)
    wav = ma.wav_bytes(pcm, 1, 2, 16000)
    cfgs = [ma.make_config(precision=ma.PREC_F32, **kw) for kw in variants]
    want = [ma.wapper_mem(cfg, wav) for cfg in cfgs]
    errors = []

    def work(t):
        try:
            for rep in range(3):
                for j in range(len(cfgs)):
                    i = (j * 5 + t + rep) % len(cfgs)
                    got = ma.wapper_mem(cfgs[i], wav)
                    if not np.array_equal(got.view(np.uint64), want[i].view(np.uint64)):
                        errors.append("thread %d config %d differs" % (t, i))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def _tonal_pcm(n, ch, bps, sr, seed):
    """A near-full-scale 1 kHz tone plus noise 90 dB below full scale: the input with the widest
    per-frame dynamic range the PCM format allows (weak channels sit at the noise floor)."""
    rng = np.random.default_rng(seed)
    fs = float(2 ** (8 * bps - 1) - 1)
    t = np.arange(n) / sr
    cols = []
    for c in range(ch):
        x = 0.98 * fs * np.sin(2 * np.pi * (1000.0 + 37.0 * c) * t + 0.3 * c) + rng.normal(0.0, fs * 10 ** (-90 / 20), n)
        cols.append(np.clip(np.rint(x), -fs - 1, fs).astype(np.int64))
    inter = np.stack(cols, axis=1).reshape(-1)
    raw = np.zeros((inter.size, bps), dtype=np.uint8)
    for b in range(bps):
        raw[:, b] = (inter >> (8 * b)) & 0xFF
    return raw.reshape(-1)


@pytest.mark.parametrize("name", ["c2", "c1a"])
def test_tonal_input_error_envelope(oracle, name):
    """The worst input a PCM format allows for the 1e-4 log-mel tolerance: a near-full-scale tone
    over a noise floor 90 dB down, so that weak channels sit far below the strongest one of their
    frame (the rounding noise of a single-precision FFT is relative to the frame's energy).
    Asserted: the precision bench.py reports for the configuration's class (presets.BENCH_PRECISION:
    single for 16-bit input, double for 24-bit input) meets 1e-4 on EVERY channel; double (the
    library's default) agrees to 1e-9.  For 24-bit input the single-precision error is printed by
    level and bounded by 1e-4 within 60 dB of the frame's strongest channel and by 5e-3 anywhere."""
    from mfcc_amd.presets import BENCH_PRECISION
    kw = dict(CONFIGS[name], regreOrder=1, energyFlag=0, zeroCrossingFlag=0, brightFlag=0, subBandEFlag=0)
    ch, bps = _layout(kw)
    sr = kw["sampleRate"]
    pcm = _tonal_pcm(sr // 2, ch, bps, sr, 11)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
    P, V = kw["bankNum"], kw["vecNum"]
    assert ref.shape[1] == P * V
    got64 = run_product_batch(ma, ma.make_config(**kw), [pcm.tobytes()], ch, bps)[0]  # default precision
    assert ma.make_config(**kw).precision == ma.PREC_F64
    assert np.max(np.abs(got64 - ref)) < 1e-9
    got32 = run_product_batch(ma, ma.make_config(precision=ma.PREC_F32, **kw), [pcm.tobytes()], ch, bps)[0]
    benched = BENCH_PRECISION["c2" if name == "c2" else "c1"]
    got_b = got64 if benched == "f64" else got32
    err_b = float(np.max(np.abs(got_b - ref)))
    util._note("tonal %s, benched precision %s (all channels)" % (name, benched), err_b, 1e-4)
    assert err_b <= 1e-4
    err = np.abs(got32 - ref).reshape(-1, V, P)
    lev = ref.reshape(-1, V, P)
    below = np.max(lev, axis=2, keepdims=True) - lev  # log units below the frame's strongest channel
    edges = [0, 4.6, 9.2, 13.8, 18.4, 23.0, 27.6, 1e9]  # 0, 20, 40, ... dB
    print("\n%s: single-precision log-mel error by level below the frame's strongest channel" % name)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (below >= lo) & (below < hi)
        if m.any():
            e = float(err[m].max())
            print("  %5.0f .. %5.0f dB: %7d values, max error %.3g" % (lo * 4.343, min(hi, 40) * 4.343, int(m.sum()), e))
            util._note("tonal %s f32, %3.0f+ dB below the frame's peak" % (name, lo * 4.343), e, 1e-4 if hi <= 13.8 else 5e-3)
    assert float(err[below < 13.8].max()) <= 1e-4
    assert float(err.max()) <= 5e-3


@pytest.mark.parametrize("zero_mean", [0, 1])
def test_recording_over_several_shards_from_one_process(oracle, zero_mean):
    """MFCCExtractRecording (the C entry point that drives one GPU and one host thread per frame
    range): three shards on the one GPU of this box against the oracle on the whole recording, and
    -- without z-norm, where nothing depends on the summation order -- bit for bit against one shard."""
    kw = dict(CONFIGS["c5"], zeroMeanSigFlag=zero_mean)
    n = 44100 * 3 + 555
    pcm = _pcm(n, 2, 3, 91, 44100)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 2, 3)
    for prec, f64 in ((ma.PREC_F32, False), (ma.PREC_F64, True)):
        cfg = ma.make_config(precision=prec, **kw)
        got = ma.extract_recording(cfg, pcm, 2, 3, num_shards=3, devices=[0, 0, 0])
        assert got.shape == ref.shape
        _compare(got.astype(np.float64), ref, kw, f64, src=(oracle, pcm, 2, 3))
    kw2 = dict(kw, znormFlag=0)
    cfg = ma.make_config(precision=ma.PREC_F32, **kw2)
    one = ma.extract_recording(cfg, pcm, 2, 3, num_shards=1)
    for shards in (2, 5):
        many = ma.extract_recording(cfg, pcm, 2, 3, num_shards=shards, devices=[0] * shards)
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
    # more shards than frames, and a recording shorter than one frame
    tiny = ma.extract_recording(cfg, pcm[:6 * 1764 * 2], 2, 3, num_shards=4, devices=[0] * 4)
    assert tiny.shape[0] == ma.num_frames(1764 * 2, 1764, 882)
    assert ma.extract_recording(cfg, pcm[:600], 2, 3, num_shards=2, devices=[0, 0]).shape[0] == 0


def test_recording_exchange_over_rccl_equals_the_host_exchange(oracle, tuning):
    """The exchanges of MFCCExtractRecording (integer channel sums, z-norm column sums) through the
    RCCL branch -- ncclCommInitAll, ncclAllGather on device buffers, rank-ordered sum kernel --
    forced on for the one shard this box's one GPU allows, bit for bit against the host-memory
    branch and within tolerance of the oracle; two shards on one GPU are refused in that mode."""
    kw = dict(CONFIGS["c5"], zeroMeanSigFlag=1)
    n = 44100 * 2 + 321
    pcm = _pcm(n, 2, 3, 92, 44100)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, 2, 3)
    for prec, f64 in ((ma.PREC_F32, False), (ma.PREC_F64, True)):
        cfg = ma.make_config(precision=prec, **kw)
        tuning(rccl=-1)
        host = ma.extract_recording(cfg, pcm, 2, 3, num_shards=1)
        tuning(rccl=1)
        over = ma.extract_recording(cfg, pcm, 2, 3, num_shards=1)
        assert np.array_equal(host.view(np.uint8), over.view(np.uint8))
        _compare(over.astype(np.float64), ref, kw, f64, src=(oracle, pcm, 2, 3))
    with pytest.raises(ma.MFCCError):
        ma.extract_recording(cfg, pcm, 2, 3, num_shards=2, devices=[0, 0])


def test_shard_exchange_entry_point_on_a_one_rank_communicator():
    """MFCCShardExchange as a multi-process caller uses it: a communicator made by the caller (here
    ncclCommInitAll over the one device, through librccl's C ABI), values in device memory; with
    one rank the totals equal the rank's own values, for doubles and for 64-bit integers."""
    import ctypes as C
    import torch
    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    L = ma.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vals = torch.arange(613, dtype=torch.float64, device="cuda") * 0.37 - 5.0
    want = vals.clone()
    gather = torch.zeros(613, dtype=torch.float64, device="cuda")
    assert L.MFCCShardExchange(comm, 0, C.c_void_p(vals.data_ptr()), 613, C.c_void_p(gather.data_ptr()), st) == 0, L.MFCCLastError()
    torch.cuda.synchronize()
    assert torch.equal(vals, want) and torch.equal(gather, want)
    iv = torch.tensor([2 ** 40 + 7, -3, 0, 2 ** 62], dtype=torch.int64, device="cuda")
    ig = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert L.MFCCShardExchange(comm, 1, C.c_void_p(iv.data_ptr()), 4, C.c_void_p(ig.data_ptr()), st) == 0
    torch.cuda.synchronize()
    assert iv.tolist() == [2 ** 40 + 7, -3, 0, 2 ** 62] == ig.tolist()
    assert L.MFCCShardExchange(None, 0, C.c_void_p(vals.data_ptr()), 613, C.c_void_p(gather.data_ptr()), st) == ma.E_ARG
    rccl.ncclCommDestroy(comm)


def test_library_counters_follow_the_work(tmp_path):
    """MFCCCountersGet: frames, PCM bytes and feature bytes of the extractions, files through the
    per-file and list entry points, wall time of the list run."""
    kw = CONFIGS["c3"]
    cfg = ma.make_config(**kw)  # double precision: 8-byte features
    ma.counters(reset=True)
    pcm = _pcm(16000 + 40, 1, 2, 77, 16000)
    wav = ma.wav_bytes(pcm, 1, 2, 16000)
    feat = ma.wapper_mem(cfg, wav)
    c = ma.counters()
    assert c["extractCalls"] == 1 and c["filesOk"] == 1 and c["filesFailed"] == 0
    assert c["frames"] == feat.shape[0] == ma.num_frames(16040, 400, 160)
    assert c["pcmBytes"] == 16040 * 2 and c["featureBytes"] == feat.size * 8
    with pytest.raises(ma.MFCCError):
        ma.wapper_mem(cfg, wav[:20])
    assert ma.counters()["filesFailed"] == 1
    lines = []
    for i in range(3):
        (tmp_path / ("a%d.wav" % i)).write_bytes(wav)
        lines.append("%s\t%s" % (tmp_path / ("a%d.wav" % i), tmp_path / ("a%d.txt" % i)))
    lines.append("%s\t%s" % (tmp_path / "missing.wav", tmp_path / "m.txt"))
    (tmp_path / "l.txt").write_text("\n".join(lines) + "\n")
    ini = tmp_path / "c.ini"
    ini.write_text("[Frame]\nsampleRate = 16000\nwlen = 400\ninc = 160\nsaveType = b\nfileList = %s\n[MFCC]\nMFCCNum = 12\n" % (tmp_path / "l.txt"))
    os.environ["MFCC_QUIET"] = "1"
    try:
        ma.counters(reset=True)
        assert ma.wapper_list(str(ini)) == 1
    finally:
        del os.environ["MFCC_QUIET"]
    c = ma.counters()
    assert c["filesOk"] == 3 and c["filesFailed"] == 1 and c["frames"] == 3 * feat.shape[0] and c["listSeconds"] > 0


def test_command_line_program_matches_the_per_file_entry_point(tmp_path):
    """The drop-in program (`mfcc config.ini`, README.md:26) on an example/config.ini-shaped run:
    its files equal MFCCWapper's byte for byte, and its console carries the reference's lines
    (main.c:140,180,184-187,212,252,274; mfcc.c:74-75) unless MFCC_QUIET is set."""
    import subprocess
    kw = CONFIGS["c1a"]
    exe = os.path.join(os.path.dirname(HERE), "tools", "mfcc")
    lines = []
    for i in range(3):
        pcm = _pcm(44100 // 2 + 777 * i, 2, 3, 40 + i, 44100)
        (tmp_path / ("a%d.wav" % i)).write_bytes(ma.wav_bytes(pcm, 2, 3, 44100))
        lines.append("%s\t%s" % (tmp_path / ("a%d.wav" % i), tmp_path / ("a%d.fbank" % i)))
    (tmp_path / "fileList.txt").write_text("\n".join(lines) + "\n")
    ini = tmp_path / "config.ini"
    ini.write_text("[Frame];\nsampleRate = 44100 ;\nhipassfre = 22050 ;\nlowpassfre = 10 ;\npreemphasise = 0 ;\n"
                   "wlen = 1764 ;\ninc = 882 ;\nsaveType = f ;\nvecNum = 2 ;\nfileList = %s ;\n"
                   "[MFCC];\nfbankFlag = 1 ;\nbankNum = 40 ;\nMFCCNum = -1 ;\nMFCC0thFlag = 0 ;\n"
                   "[Others];\nenergyFlag = 1 ;\nzeroCrossingFlag = 1 ;\nbrightFlag = 1 ;\nsubBandEFlag = 8 ;\nfftLength = 0 ;\n"
                   "[Regression];\nregreOrder = 3 ;\ndelwin = 9 ;\n" % (tmp_path / "fileList.txt"))
    env = {k: v for k, v in os.environ.items() if k != "MFCC_QUIET"}
    r = subprocess.run([exe, str(ini)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    for text in ("finish reading the config.ini", "FFT passband 2 to 1024 out of 1 to 1024", "Mel passband 15.986084 to 3923.357581",
                 "Convert %s to %s" % (tmp_path / "a0.wav", tmp_path / "a0.fbank"), "the frame feature dimension is 306",
                 "order 3", "start...", "total coef size: %d" % (306 * ma.num_frames(44100 // 2, 1764, 882)),
                 "post-processing...", "writing the doc..."):
        assert text in r.stdout, text
    cfg = ma.make_config(saveType=ma.SAVE_F, **kw)
    for i in range(3):
        ma.wapper(cfg, str(tmp_path / ("a%d.wav" % i)), str(tmp_path / ("w%d.fbank" % i)), want_features=False)
        assert (tmp_path / ("a%d.fbank" % i)).read_bytes() == (tmp_path / ("w%d.fbank" % i)).read_bytes()
    r = subprocess.run([exe, str(ini)], capture_output=True, text=True, env=dict(env, MFCC_QUIET="1"), timeout=120)
    assert r.returncode == 0 and r.stdout == ""
