"""CPU tests of the multi-GPU host logic (mfcc_amd/sharding.py): file sharding, frame ranges, and
a world_size-2 gloo run of the frame-sharded long-recording flow.  On CPU each rank's share of the
arithmetic is stood in for by the oracle applied to exactly the instants the shard plan asks for;
what is under test is the shard geometry (halo frames, lead instant), the statistics exchange and
the final normalisation -- the same calls the GPU path makes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mfcc_amd as ma
from mfcc_amd import sharding
from mfcc_amd.presets import CONFIGS


def test_shard_files_balances_and_covers():
    rng = np.random.default_rng(0)
    frames = rng.integers(200, 2000, size=1000)
    for world in (1, 2, 4, 8):
        parts = sharding.shard_files(frames, world)
        assert sorted(i for p in parts for i in p) == list(range(1000))
        loads = [int(frames[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= frames.max()
    assert sharding.shard_files([5, 5, 5], 4) == [[0], [1], [2], []]


def test_frame_ranges():
    for total, world in ((1439999, 8), (10, 3), (0, 2), (5, 8)):
        r = sharding.frame_ranges(total, world)
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_shard(ob, kw, pcm_all, ch, bps, total, fb, fe):
    """What one rank computes: static + delta features of its extended frame range from the
    instants the plan names, with the pre-emphasis predecessor supplied by the lead instant."""
    cfg = ma.make_config(precision=ma.PREC_F32, **kw)
    g = ma.shard_geometry(cfg, total, fb, fe)
    bpi = ch * bps
    raw = pcm_all[g["sample_begin"] * bpi:(g["sample_begin"] + g["sample_count"]) * bpi]
    ints = np.zeros((ch, g["sample_count"]), dtype=np.int64)
    b = raw.reshape(-1, ch, bps).astype(np.int64)
    for k in range(bps):
        ints += (b[:, :, k] << (8 * k)).T
    ints = np.where(ints >= 1 << (8 * bps - 1), ints - (1 << (8 * bps)), ints)
    # the oracle pre-emphasises from the buffer's first instant; with a lead instant that first
    # output sample is wrong by construction and is dropped, exactly as the kernel never frames it
    kw2 = dict(kw, regreOrder=1, znormFlag=0)
    static = ob.extract(ob.make_config(**kw2), ints.astype(np.int32))
    wl, inc = kw["wlen"], kw["inc"]
    if g["lead"]:
        # frames of the shard start one instant later than the oracle's framing of the buffer:
        # re-run on the buffer without the lead but with the true predecessor folded in
        x = ints.astype(np.float64)
        rows = [x[0]] if ch == 1 else [x[0], x[1], 0.5 * (x[0] + x[1]), x[0] - x[1]]
        pre = [np.concatenate([[0.0], r[1:] - kw["preemphasise"] * r[:-1]]) for r in rows]
        static = _oracle_static_from_rows(ob, kw2, [p[1:] for p in pre], g["total_rows"])
    D = static.shape[1]
    order = kw["regreOrder"]
    full = np.zeros((g["total_rows"], D * order))
    full[:, :D] = static[:g["total_rows"]]
    for o in range(2, order + 1):
        full = ob.regress(full, D, D * order, D, kw["delwin"]).reshape(-1, D * order) if o == 2 else \
            _regress_block(ob, full, D, order, o, kw["delwin"])
    return full, g


def _regress_block(ob, full, D, order, o, delwin):
    step = D * order
    flat = full.copy().reshape(-1)
    shifted = flat[(o - 2) * D:]
    pad = np.concatenate([shifted, np.zeros((o - 2) * D)])
    res = ob.regress(pad, D, step, D, delwin)
    out = flat.copy()
    n = full.shape[0]
    r = res.reshape(n, step)
    out = out.reshape(n, step)
    out[:, (o - 1) * D:o * D] = r[:, D:2 * D]
    return out


def _oracle_static_from_rows(ob, kw, rows, nframes):
    """Static features of already pre-emphasised rows (stage functions of the oracle)."""
    import ctypes as C
    L = ob.lib()
    wl, inc, P = kw["wlen"], kw["inc"], kw["bankNum"]
    ham = np.zeros(wl)
    L.orc_hamming(ham.ctypes.data_as(C.POINTER(C.c_double)), wl)
    out = []
    for t in range(nframes):
        row = []
        for v in range(kw["vecNum"]):
            fr = np.ascontiguousarray(rows[v][t * inc:t * inc + wl])
            zc = L.orc_zcr(fr.ctypes.data_as(C.POINTER(C.c_double)), wl)
            fb, te, _, x = ob.wave2fbank(fr * ham, kw["sampleRate"], P, kw["lowpassfre"], kw["hipassfre"])
            if kw.get("fbankFlag", 1):
                row += list(fb)
            else:
                c, c0 = ob.fbank2mfcc(fb, kw["MFCCNum"])
                row += list(c) + ([c0] if kw.get("MFCC0thFlag", 0) else [])
            if kw.get("energyFlag", 0):
                row.append(np.log(te))
            if kw.get("zeroCrossingFlag", 0):
                row.append(zc)
            if kw.get("brightFlag", 0):
                row.append(L.orc_brightness(x.ctypes.data_as(C.POINTER(C.c_double)), x.size))
            S = kw.get("subBandEFlag", 0)
            if S:
                sub = np.zeros(S)
                L.orc_subband(x.ctypes.data_as(C.POINTER(C.c_double)), x.size,
                              sub.ctypes.data_as(C.POINTER(C.c_double)), S)
                row += list(sub)
        out.append(row)
    return np.array(out)


def _worker(rank, world, port, kw, n, ch, bps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + 5000, sample_rate=kw["sampleRate"], tone_hz=440)
    total = ma.num_frames(n, kw["wlen"], kw["inc"])
    fb, fe = sharding.frame_ranges(total, world)[rank]
    full, g = _oracle_shard(ob, kw, pcm, ch, bps, n, fb, fe)
    own = full[g["owned_row_begin"]:g["owned_row_begin"] + g["owned_rows"]]
    # statistics exchange, as MFCCShardColumnSums / combine_sums / MFCCShardNormalize do it
    sums = torch.from_numpy(np.concatenate([own.sum(0), (own * own).sum(0), [own.shape[0]]]))
    tot = sharding.combine_sums(sums, dist).numpy()
    step = own.shape[1]
    cnt = tot[2 * step]
    mean = tot[:step] / cnt
    std = np.sqrt(tot[step:2 * step] / cnt - mean * mean)
    q.put((rank, (own - mean) / std, tot))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,n,ch,bps", [("c3", 6000, 1, 2), ("c5", 30000, 2, 3)])
def test_frame_sharded_flow_world2_gloo(oracle, name, n, ch, bps):
    kw = CONFIGS[name]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, n, ch, bps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, feat, tot = q.get(timeout=120)
        got[r] = (feat, tot)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][1], got[1][1])  # identical totals on every rank
    pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + 5000, sample_rate=kw["sampleRate"], tone_hz=440)
    ref = oracle.extract_pcm_bytes(oracle.make_config(**kw), pcm, ch, bps)
    both = np.concatenate([got[0][0], got[1][0]])
    assert both.shape == ref.shape
    assert np.allclose(both, ref, rtol=1e-9, atol=1e-9)


def test_split_file_list_covers_every_line_once(tmp_path):
    """The per-rank lists of a sharded run: every line up to the first blank one lands on exactly
    one rank, balanced by file size, and the per-rank ini differs from the original only in its
    fileList line (comment suffix kept) and in `[IO] numDevices = 1` at its end, which replaces any
    numDevices of the original: a rank drives the one GPU of its process, not every visible one."""
    lines = []
    for i in range(11):
        src = tmp_path / ("s%02d.wav" % i)
        if i != 4:  # one missing file: counted as empty, still assigned
            src.write_bytes(b"x" * (100 + 37 * i * i))
        lines.append("%s\t%s" % (src, tmp_path / ("d%02d.txt" % i)))
    lst = tmp_path / "all.list"
    lst.write_text("\n".join(lines) + "\n\nignored\tafter the blank line\n")
    ini = tmp_path / "config.ini"
    body = "[Frame];\nsampleRate = 16000 ;\nfileList = %s ; the list\nwlen = 400 ;\n" % lst
    ini.write_text(body + "[IO]\nnumDevices = 4 ;\n")
    seen, loads = [], []
    for rank in range(3):
        rini, mine = sharding.split_file_list(str(ini), rank, 3, str(tmp_path / "ranks"))
        seen += mine
        got = open(tmp_path / "ranks" / ("rank%d.list" % rank)).read().splitlines()
        assert got == [lines[i] for i in mine]
        text = open(rini).read()
        assert "fileList = %s ; the list" % (tmp_path / "ranks" / ("rank%d.list" % rank)) in text
        assert text.replace(str(tmp_path / "ranks" / ("rank%d.list" % rank)), str(lst)) == body + "[IO]\n[IO]\nnumDevices = 1\n"
        cfg = ma.config_from_ini(rini) if hasattr(ma, "config_from_ini") else None
        assert cfg is None or cfg.numDevices == 1
        loads.append(sum(os.path.getsize(lines[i].split("\t")[0]) for i in mine if i != 4))
    assert sorted(seen) == list(range(11))
    assert max(loads) - min(loads) <= 100 + 37 * 100  # within one largest file
