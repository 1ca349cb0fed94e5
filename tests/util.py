"""Shared helpers of the test-suite: named option sets and the GPU-vs-oracle comparison."""
import numpy as np

from mfcc_amd.presets import CONFIGS  # noqa: E402,F401


def oracle_cfg(ob, **kw):
    return ob.make_config(**{k: v for k, v in kw.items()})


def product_cfg(ma, precision=0, **kw):
    return ma.make_config(precision=precision, **kw)


def column_kinds(kw):
    """Per static column of one vec: 'mel' (log-mel), 'cep', 'c0', 'energy', 'zc', 'bright',
    'sub', 'fft'."""
    fb = kw.get("fbankFlag", 1)
    kinds = (["mel"] * kw["bankNum"]) if fb else (["cep"] * kw["MFCCNum"])
    if not fb and kw.get("MFCC0thFlag", 0):
        kinds.append("c0")
    if kw.get("energyFlag", 0):
        kinds.append("energy")
    if kw.get("zeroCrossingFlag", 0):
        kinds.append("zc")
    if kw.get("brightFlag", 0):
        kinds.append("bright")
    kinds += ["sub"] * kw.get("subBandEFlag", 0)
    kinds += ["fft"] * kw.get("fftLength", 0)
    return kinds


# Largest error every comparison of this run achieved, per column kind: {kind: (error, tolerance)}
# with the largest error / tolerance ratio; printed at the end of the run (conftest.py).
ACHIEVED = {}


def _note(kind, err, tol):
    old = ACHIEVED.get(kind)
    if old is None or err / tol > old[0] / old[1]:
        ACHIEVED[kind] = (err, tol)


def assert_static_close(got, ref, kw, f64=False, delta=False, static_ref=None):
    """Tolerances for the static block: log-mel / log-energy 1e-4 absolute; cepstra 1e-5 relative
    to the largest cepstral magnitude of the matrix and C0 1e-5 relative to the largest C0; ZCR exact (the same crossing count,
    the quotient rounded to single precision); brightness / sub-band
    fractions 1e-5 absolute; debug FFT magnitudes 1e-5 relative to the largest magnitude.
    In f64 mode everything must agree to 1e-9 (relative to magnitude).  delta=True: the block holds
    regression deltas; they inherit the tolerances, except that a ZCR delta is 1e-6 absolute, and
    the relative ones (cepstra, magnitudes) stay relative to the static block `static_ref` the
    deltas were formed from (a delta is a difference of those values divided by a constant >= 2)."""
    kinds = column_kinds(kw) * kw.get("vecNum", 1)
    assert got.shape == ref.shape
    n = len(kinds)
    g, r = got[:, :n], ref[:, :n]
    sr = r if static_ref is None else np.where(np.isfinite(static_ref[:, :n]), static_ref[:, :n], 0.0)
    assert np.array_equal(np.isnan(g), np.isnan(r))
    assert np.array_equal(np.isinf(g), np.isinf(r))
    fin = np.isfinite(r)
    g = np.where(fin, g, 0.0)
    r = np.where(fin, r, 0.0)
    kinds = np.array(kinds)
    if f64:
        # relative to the column's largest magnitude: deltas and near-cancelling bins are
        # differences of much larger numbers
        scale = np.maximum(np.max(np.abs(r), axis=0, keepdims=True, initial=0.0), 1.0)
        rel = float(np.max(np.abs(g - r) / scale, initial=0.0))
        _note("f64 (relative)", rel, 1e-9)
        assert rel < 1e-9
        return
    # cepstra: 1e-5 of the largest cepstral magnitude, but never tighter than a fifth of what the
    # 1e-4 log-mel tolerance implies for a sum of P channels scaled by sqrt(2/P)
    cep_floor = 2e-5 * float(np.sqrt(2.0 * kw.get("bankNum", 1)))
    for kind, atol_of in (("mel", lambda x: 1e-4), ("energy", lambda x: 1e-4),
                          ("cep", lambda x: max(1e-5 * x, cep_floor)), ("c0", lambda x: max(1e-5 * x, cep_floor)),
                          ("zc", lambda x: 1e-6), ("bright", lambda x: 1e-5), ("sub", lambda x: 1e-5),
                          ("fft", lambda x: 1e-5 * max(x, 1.0))):
        m = kinds == kind
        if not m.any() or g.shape[0] == 0:
            continue
        mx = float(np.max(np.abs(sr[:, m])))
        if kind == "zc" and not delta:  # exact: the same count / (wlen - 1), rounded once to single precision
            assert np.array_equal(g[:, m], r[:, m].astype(np.float32).astype(np.float64))
            continue
        err = float(np.max(np.abs(g[:, m] - r[:, m])))
        _note(kind + (" delta" if delta else ""), err, atol_of(mx))
        assert err <= atol_of(mx), "%s%s: max error %g > %g" % (kind, " (delta block)" if delta else "", err, atol_of(mx))


def run_product_batch(ma, cfg, pcm_list, channels, bps, device=0, pad=None):
    """Runs the batch through the C ABI on the GPU; returns a list of float64 [T, step].
    pad: pack the utterances `pad` bytes apart (odd byte offsets) instead of 16-byte aligned."""
    import torch
    ctx = ma.Context(cfg, device)
    if pad is None:
        buf, offs = ma.pack_utterances(pcm_list)
    else:
        offs, total = [], pad
        for p in pcm_list:
            offs.append(total)
            total += len(p) + pad
        buf = np.zeros(max(total, 16), dtype=np.uint8)
        for o, p in zip(offs, pcm_list):
            buf[o:o + len(p)] = np.frombuffer(bytes(p), dtype=np.uint8)
        offs = np.array(offs, dtype=np.int64)
    ns = np.array([len(p) // (channels * bps) for p in pcm_list], dtype=np.int64)
    plan = ma.Plan(ctx, channels, bps, offs, ns)
    dev = torch.from_numpy(buf).to("cuda:%d" % device)
    out = plan.extract(dev)
    torch.cuda.synchronize()
    host = out.double().cpu().numpy()
    res = [host[plan.row_offset(u):plan.row_offset(u) + plan.frames(u)] for u in range(len(ns))]
    plan.close()
    ctx.close()
    return res
