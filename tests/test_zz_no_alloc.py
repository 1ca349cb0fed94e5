"""GPU test of MFCCColumnSumsNoAlloc, the entry point for callers that cannot allocate device
memory: through the C ABI, against MFCCShardColumnSums (the entry point it is a variant of) on the
same matrix, on two calls in a row, the second with a smaller input; plus the refusal of matrices
whose partial sums do not fit the fixed scratch array.  Both add the same double-precision terms,
in different fixed orders (row chunks of 256 here, the plan's segments there): the row count must
be equal, every sum must agree to 1e-12 of the sum of its terms' magnitudes."""
import ctypes as C

import numpy as np
import pytest

import mfcc_amd as ma
from mfcc_amd.presets import CONFIGS

pytestmark = pytest.mark.gpu


def _sums_with_plan(ctx, ctx_plain, ch, bps, dev_pcm, n):
    """The existing entry point: a shard plan that owns every frame of the recording.  Returns the
    un-normalised matrix (from the same plan without z-norm: a shard plan with z-norm leaves the
    matrix to its normalisation call) and the plan's sums."""
    frames = ma.num_frames(n, ctx.cfg.wlen, ctx.cfg.inc)
    plan = ma.Plan(ctx, ch, bps, shard=(n, 0, frames, None))
    out = plan.extract(dev_pcm)
    sums = plan.shard_column_sums(out)
    plain = ma.Plan(ctx_plain, ch, bps, shard=(n, 0, frames, None))
    return plain.extract(dev_pcm), sums


def _sums_no_alloc(out, step, prec):
    import torch
    got = torch.full((2 * step + 1,), float("nan"), dtype=torch.float64, device=out.device)
    stream = torch.cuda.current_stream(out.device).cuda_stream
    ma._check(ma.lib().MFCCColumnSumsNoAlloc(prec, C.c_void_p(out.data_ptr()), out.shape[0], step,
                                              C.c_void_p(got.data_ptr()), C.c_void_p(stream)))
    return got


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_no_alloc_column_sums_match_the_plan_entry_point(precision):
    import torch
    kw = dict(CONFIGS["c5"])
    prec = ma.PREC_F64 if precision == "f64" else ma.PREC_F32
    cfg = ma.make_config(precision=prec, **kw)
    ctx = ma.Context(cfg, 0)
    ctx_plain = ma.Context(ma.make_config(precision=prec, **dict(kw, znormFlag=0)), 0)
    ch, bps = 2, 3
    # two calls in a row, the second with a smaller input (fewer rows, fewer chunks)
    for n in (44100 * 7 + 311, 44100 * 2 + 17):
        pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + n, sample_rate=44100, tone_hz=440)
        dev = torch.from_numpy(pcm).cuda()
        out, want = _sums_with_plan(ctx, ctx_plain, ch, bps, dev, n)
        got = _sums_no_alloc(out, ctx.step, prec)
        torch.cuda.synchronize()
        w, g = want.cpu().numpy(), got.cpu().numpy()
        assert g[-1] == out.shape[0] == w[-1]
        x = out.double().cpu().numpy()
        mag = np.concatenate([np.abs(x).sum(axis=0), (x * x).sum(axis=0), [0.0]])
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.isinf(g), np.isinf(w))
        fin = np.isfinite(w)
        assert np.all(np.abs(g - w)[fin] <= 1e-12 * mag[fin])


def test_no_alloc_column_sums_edges():
    import torch
    step = 40
    sums = torch.full((2 * step + 1,), 7.0, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    L = ma.lib()
    # no rows: zeros, count 0
    assert L.MFCCColumnSumsNoAlloc(ma.PREC_F32, None, 0, step, C.c_void_p(sums.data_ptr()), C.c_void_p(stream)) == 0
    torch.cuda.synchronize()
    assert float(sums.abs().sum()) == 0.0
    # a small matrix against numpy (double accumulation of exactly representable values)
    x = torch.arange(300 * step, dtype=torch.float32, device="cuda").reshape(300, step) % 97
    assert L.MFCCColumnSumsNoAlloc(ma.PREC_F32, C.c_void_p(x.data_ptr()), 300, step, C.c_void_p(sums.data_ptr()),
                                   C.c_void_p(stream)) == 0
    torch.cuda.synchronize()
    xn = x.cpu().numpy().astype(np.float64)
    assert np.array_equal(sums[:step].cpu().numpy(), xn.sum(axis=0))
    assert np.array_equal(sums[step:2 * step].cpu().numpy(), (xn * xn).sum(axis=0))
    assert float(sums[-1]) == 300.0
    # beyond the fixed scratch (4 MiB of partial sums): refused, with a message
    rows = 256 * (4 * 1024 * 1024 // (2 * step * 8) + 1)
    rc = L.MFCCColumnSumsNoAlloc(ma.PREC_F32, C.c_void_p(x.data_ptr()), rows, step, C.c_void_p(sums.data_ptr()),
                                 C.c_void_p(stream))
    assert rc == ma.E_ARG
    assert b"scratch" in L.MFCCLastError()


def test_no_alloc_column_sums_replayed_from_a_captured_graph():
    """What the entry point is for: after the eager warm-up call, a call made while its stream is
    being captured only enqueues kernels; the graph is then replayed on new contents of the same
    buffers and must give the eager result of those contents, bit for bit."""
    import torch
    L = ma.lib()
    step, rows = 45, 3000
    assert L.MFCCColumnSumsNoAllocWarmup() == 0
    x = torch.randn(rows, step, dtype=torch.float32, device="cuda")
    sums = torch.zeros(2 * step + 1, dtype=torch.float64, device="cuda")
    eager = torch.zeros_like(sums)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        rc = L.MFCCColumnSumsNoAlloc(ma.PREC_F32, C.c_void_p(x.data_ptr()), rows, step, C.c_void_p(sums.data_ptr()), C.c_void_p(st))
    assert rc == 0, L.MFCCLastError()
    for seed in (1, 2):
        torch.manual_seed(seed)
        x.copy_(torch.randn(rows, step, dtype=torch.float32, device="cuda") * (10.0 ** seed))
        sums.zero_()
        g.replay()
        torch.cuda.synchronize()
        st = torch.cuda.current_stream().cuda_stream
        assert L.MFCCColumnSumsNoAlloc(ma.PREC_F32, C.c_void_p(x.data_ptr()), rows, step, C.c_void_p(eager.data_ptr()), C.c_void_p(st)) == 0
        torch.cuda.synchronize()
        assert float(sums[-1]) == rows
        assert np.array_equal(sums.cpu().numpy().view(np.uint64), eager.cpu().numpy().view(np.uint64))
