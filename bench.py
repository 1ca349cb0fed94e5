#!/usr/bin/env python3
"""Throughput benchmark of the MI355X extraction path.

    python bench.py --gpus N --steps K --warmup W [--workload c1|c2|c3|c4|c5] [--precision auto|f32|f64]

A step is one pass of the device-resident hot path (static-feature kernel, then deltas / z-norm
when the option set has them) over one batch of synthetic audio that already sits in HBM.
The line is in the precision that meets the stated tolerance on the workload's PCM format (single
for the 16-bit workloads c2 / c3, double for the 24-bit ones); the same workload timed in the
other precision is the side record "f64" / "f32" of the same line.

Workloads (the five benchmark configurations):
  c2 (default)  16 kHz mono 25 ms / 10 ms, 40-bin log-mel, 10 000 utterances of 10 s per GPU
  c3            16 kHz 13 MFCC + C0 + energy, deltas + delta-deltas + z-norm, 10 000 utterances per
                GPU with seeded lengths of 2 .. 20 s (ragged batch)
  c4            44.1 kHz 24-bit stereo, 40 MFCC + ZCR / brightness / 8 sub-bands, 256 files of 60 s
  c1            the reference's example/config.ini on one 1 323 001-sample 24-bit stereo file
  c5            one 8-hour 44.1 kHz 24-bit stereo recording, c1's options + z-norm
File lists (c1 .. c4) shard by utterance: every rank owns its own batch, nothing is exchanged on
the data path (weak scaling).  The long recording (c5) shards by frame range: every rank owns
T / N frames (plus the halo frames its deltas need) and the ranks exchange the z-norm column sums
over the NCCL backend (RCCL) between the statistics and the normalisation pass (strong scaling).
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E

WORKLOADS = {
    # name: (key in mfcc_amd.presets.CONFIGS, channels, bytes/sample, utterances per GPU, description)
    "c2": ("c2", 1, 2, 10000,
           "16 kHz mono 25ms/10ms, 40-bin log-mel FBank (fbankFlag=1), 10000 x 10 s utterances per GPU"),
    "c3": ("c3", 1, 2, 10000,
           "16 kHz 13 MFCC+C0+energy, regreOrder=3 (delwin=2) + znorm, 10000 utterances of 2..20 s (seeded) per GPU"),
    "c4": ("c4", 2, 3, 256,
           "44.1 kHz 24-bit stereo vecNum=2, 40 MFCC + ZCR/brightness/8 sub-bands, 256 x 60 s files per GPU"),
    "c1": ("c1a", 2, 3, 1,
           "example/config.ini (44.1 kHz 24-bit stereo, 306 columns) on one 1323001-sample file per GPU"),
    "c5": ("c5", 2, 3, 1,
           "one 8 h 44.1 kHz 24-bit stereo recording (1270080000 samples per channel), example/config.ini + znorm, "
           "frames sharded over the GPUs"),
}
C5_SAMPLES = 1270080000


def utterance_lengths(name, kw, n_utts, rank):
    """Samples per utterance of the rank's batch."""
    sr = kw["sampleRate"]
    if name == "c2":
        return np.full(n_utts, 10 * sr, dtype=np.int64)
    if name == "c3":
        rng = np.random.default_rng(0x5EED0000 + 3000 + rank)
        return rng.integers(2 * sr, 20 * sr + 1, size=n_utts, dtype=np.int64)
    if name == "c4":
        return np.full(n_utts, 60 * sr, dtype=np.int64)
    return np.full(n_utts, 1323001, dtype=np.int64)  # c1: README.md:48


def synth_range(ma, buf_addr, channels, bps, first, count, seed_base, sample_rate, threads=16):
    """Samples first .. first+count-1 of the stream with seeds seed_base + ch into the buffer at
    buf_addr, split over threads."""
    import ctypes as C
    L = ma.lib()
    per = (count + threads - 1) // threads

    def work(t):
        a, b = t * per, min(count, (t + 1) * per)
        for ch in range(channels):
            if b > a:
                L.MFCCSynthPcmAt(C.c_void_p(buf_addr + a * channels * bps), channels, ch, bps, first + a, b - a,
                                 C.c_uint64(seed_base + ch), sample_rate, 0)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


def synth_batch(ma, lengths, channels, bps, sample_rate, seed_base, threads=16):
    """Packed PCM of the utterances (16-byte aligned starts) from the library's integer generator;
    utterance u, channel ch uses seed seed_base + 8*u + ch."""
    import ctypes as C
    sizes = (lengths * channels * bps + 15) // 16 * 16
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    buf = np.zeros(int(sizes.sum()), dtype=np.uint8)
    L = ma.lib()
    base = buf.ctypes.data

    def work(t):
        for u in range(t, len(lengths), threads):
            for ch in range(channels):
                L.MFCCSynthPcm(C.c_void_p(base + int(offs[u])), channels, ch, bps, int(lengths[u]),
                               C.c_uint64(seed_base + 8 * u + ch), sample_rate, 0)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return buf, offs


def physical_cores():
    """(physical cores, logical CPUs) of the host, from /proc/cpuinfo."""
    try:
        cores, logical, phys, core = set(), 0, None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    logical += 1
                elif line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                    cores.add((phys, core))
        return (len(cores) or logical), logical
    except OSError:
        return None, os.cpu_count()


def cpu_baseline(kw, buf, offs, lengths, channels, bps):
    """Times the CPU oracle on a bounded sample of the same utterances: one OpenMP thread per file
    on the threads this process may use, then a single thread."""
    from oracle import binding as ob
    import mfcc_amd as ma
    ob.build()
    cfg = ob.make_config(**kw)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    frames = np.array([ma.num_frames(int(n), kw["wlen"], kw["inc"]) for n in lengths], dtype=np.int64)
    if len(offs) == 1:
        # one long file: the oracle works on whole files, so time a slice of it (ten minutes of
        # audio at most) per thread count
        per = int(min(lengths[0], 600 * kw["sampleRate"]))
        n_slices = int(min(threads, max(1, lengths[0] // per)))
        offs = np.arange(n_slices, dtype=np.int64) * per * channels * bps
        lengths = np.full(n_slices, per, dtype=np.int64)
        frames = np.array([ma.num_frames(per, kw["wlen"], kw["inc"])] * n_slices, dtype=np.int64)
    probe = min(len(offs), threads)
    t0 = time.perf_counter()
    _, used = ob.extract_batch(cfg, buf, channels, bps, offs[:probe], lengths[:probe], threads)
    dt = time.perf_counter() - t0
    rate = float(frames[:probe].sum()) / dt
    # about 8 s of CPU work with all threads: a prefix of the batch, or that prefix several times over
    want_frames = 8.0 * rate
    csum = np.cumsum(frames)
    n = int(min(len(offs), max(probe, np.searchsorted(csum, want_frames) + 1)))
    reps = max(1, int(round(want_frames / float(csum[n - 1]))))
    t0 = time.perf_counter()
    for _ in range(reps):
        _, used = ob.extract_batch(cfg, buf, channels, bps, offs[:n], lengths[:n], threads)
    dt = time.perf_counter() - t0
    value = reps * float(csum[n - 1]) / dt
    # single thread: about 4 s
    n1 = int(max(1, min(len(offs), np.searchsorted(csum, 4.0 * value / max(used, 1)) + 1)))
    t0 = time.perf_counter()
    ob.extract_batch(cfg, buf, channels, bps, offs[:n1], lengths[:n1], 1)
    dt1 = time.perf_counter() - t0
    phys, logical = physical_cores()
    return {"value": value, "unit": "frames/s", "cores": used, "kind": "port",
            "one_thread": float(csum[n1 - 1]) / dt1, "host_physical_cores": phys, "host_logical_cpus": logical,
            "sample": "first %d utterance(s) / slices of the same batch, %d pass(es) (%d frames), oracle C port in double "
                      "precision, one OpenMP thread per file, %.1f s; one_thread: first %d on a single thread, %.1f s"
                      % (n, reps, reps * int(csum[n - 1]), dt, n1, dt1)}


VALU_PEAK = {"f32": 157.3e12, "f64": 78.6e12}  # vector FLOP/s of one MI355X (no matrix cores)


def profile_numbers(workload, n_utts, precision):
    """HBM traffic and unit-busy fractions of the static-feature kernel from the committed
    rocprofv3 counter passes of this workload and precision (measured in a profiling run of this
    same command, not in this run: counters cannot be read from inside the timed region)."""
    res = {"traffic": None}
    for tname in ("r03_%s_%s_traffic.json" % (workload, precision), "traffic_%s.json" % workload):
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):
            continue
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("utts") == n_utts and tj.get("precision") == precision:
                res["traffic"] = tj.get("hbm_bytes_per_launch")
                res["traffic_source"] = "profiles/%s (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command)" % tname
                break
        except Exception:
            pass
    for cname in ("r03_%s_%s_counters.json" % (workload, precision),) + (("r02_%s_counters.json" % workload,) if precision == "f32" else ()):
        cpath = os.path.join(ROOT, "profiles", cname)
        if not os.path.exists(cpath):
            continue
        try:
            with open(cpath) as f:
                cj = json.load(f)
            for k in ("valu_frac", "lds_frac", "mfma_frac"):
                if k in cj:
                    res[k] = cj[k]
            pf = cj.get("per_frame", {})
            if "SQ_INSTS_VALU" in pf:
                res["valu_instr_per_frame"] = pf["SQ_INSTS_VALU"]
            if "SQ_LDS_IDX_ACTIVE" in pf:
                # an LDS-active cycle moves at most 128 bytes per CU
                res["lds_cycles_per_frame"] = pf["SQ_LDS_IDX_ACTIVE"]
                res["lds_bytes_per_frame_at_peak"] = 128.0 * pf["SQ_LDS_IDX_ACTIVE"]
            res["counters_source"] = "profiles/" + cname
            break
        except Exception:
            pass
    return res


def flops_per_frame(kw):
    """Algorithmic FLOP of one frame (all vecNum rows): FFT 2.5 F log2(F/2), split + power + mel +
    centroid 4 F, pre-emphasis + window + energy + ZCR 6 W, cepstra 2 M P, logs 20 P."""
    W, P = kw["wlen"], kw["bankNum"]
    F = 2
    while F < W:
        F *= 2
    M = 0 if kw.get("fbankFlag", 1) else kw.get("MFCCNum", 0) + kw.get("MFCC0thFlag", 0)
    per_row = 2.5 * F * (np.log2(F) - 1) + 4 * F + 6 * W + 2 * M * P + 20 * P
    return float(per_row * kw.get("vecNum", 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--utts", type=int, default=0, help="utterances per GPU (0 = workload default)")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "f64"],
                    help="auto: the precision that meets the stated tolerance on the workload's PCM format "
                         "(mfcc_amd.presets.BENCH_PRECISION); the other one is reported beside it")
    ap.add_argument("--tune", default="", help="plan-heuristic overrides for measurements, key=value,... (MFCCDebugTuning)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-other", action="store_true", help="skip the leg in the other precision")
    args = ap.parse_args()

    import torch
    import mfcc_amd as ma
    from mfcc_amd.presets import CONFIGS, BENCH_PRECISION

    if args.precision == "auto":
        args.precision = BENCH_PRECISION[args.workload]
    if args.tune:
        ma.tuning(**{k: int(v) for k, v in (item.split("=") for item in args.tune.split(",") if item)})
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    # Rehearsal aid for a one-GPU box: MFCC_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # synchronises over gloo (RCCL refuses two ranks on one device).
    one_device = os.environ.get("MFCC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    key, channels, bps, n_utts, desc = WORKLOADS[args.workload]
    if args.utts:
        n_utts = args.utts
    kw = CONFIGS[key]
    sr, bpi = kw["sampleRate"], channels * bps
    zm, zn = kw.get("zeroMeanSigFlag", 0), kw.get("znormFlag", 0)
    sharded = args.workload == "c5"
    device = "cuda:%d" % local_rank
    from mfcc_amd import sharding

    # the batch (the same bytes for both precisions)
    cfg0 = ma.make_config(precision=ma.PREC_F32, **kw)
    if sharded:
        total = C5_SAMPLES if not args.utts else args.utts * sr * 60  # --utts N: an N-minute recording instead
        T = ma.num_frames(total, kw["wlen"], kw["inc"])
        fb, fe = T * rank // world, T * (rank + 1) // world
        g = ma.shard_geometry(cfg0, total, fb, fe)
        buf = np.zeros(g["sample_count"] * bpi, dtype=np.uint8)
        synth_range(ma, buf.ctypes.data, channels, bps, g["sample_begin"], g["sample_count"], 0x5EED0000 + 5000, sr)
        offs, lengths = np.zeros(1, dtype=np.int64), np.array([g["sample_count"]], dtype=np.int64)
    else:
        lengths = utterance_lengths(args.workload, kw, n_utts, rank)
        buf, offs = synth_batch(ma, lengths, channels, bps, sr, 0x5EED0000 + 1000 * 2 + 100000 * rank)
    pcm_dev = torch.from_numpy(buf).to(device)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(precision, steps, warmup):
        """Times `steps` passes in one precision; returns (record for the JSON line, one_pass, out)."""
        prec = ma.PREC_F64 if precision == "f64" else ma.PREC_F32
        cfg = ma.make_config(precision=prec, **kw)
        el = 8 if precision == "f64" else 4
        step, vsize = ma.step(cfg), ma.step(cfg) // kw.get("regreOrder", 1)
        # algorithmic bytes per frame: the PCM of one frame shift read once (twice with zero-mean), every
        # output element written once (and read + rewritten by the z-norm); the static-feature kernel
        # alone reads the PCM and writes its own vSize columns
        pass_bytes = bpi * kw["inc"] * (1 + zm) + step * el * (1 + 2 * zn)
        kernel_bytes = bpi * kw["inc"] + vsize * el
        ctx = ma.Context(cfg, local_rank)
        if sharded:
            plan = ma.Plan(ctx, channels, bps, shard=(total, fb, fe, None))
            frames, total_frames = fe - fb, T
        else:
            plan = ma.Plan(ctx, channels, bps, offs, lengths)
            frames = plan.total_rows
            total_frames = frames * world
        out = plan.alloc_output()
        plan.enable_timing(True)

        def one_pass():
            plan.extract(pcm_dev, out)
            if sharded and zn:
                sums = plan.shard_column_sums(out)
                sums = sharding.combine_sums(sums, dist)  # all_gather over RCCL + rank-ordered sum
                plan.shard_normalize(out, sums)

        if rank == 0:
            print("bench: %s %s, %d frames on this rank, inputs resident" % (args.workload, precision, frames), file=sys.stderr, flush=True)
        for _ in range(warmup):
            one_pass()
        sync_all()
        kernel_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
            kernel_ms.append(plan.last_kernel_ms())  # waits for the kernel's stop event
        sync_all()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        value = total_frames * steps / elapsed
        kavg = float(np.mean(kernel_ms)) * 1e-3
        achieved = frames * kernel_bytes / kavg
        prof = profile_numbers(args.workload, n_utts, precision)
        flops = flops_per_frame(kw)
        alu_frac = frames * flops / kavg / VALU_PEAK[precision]
        # which unit limits the kernel: the memory system if its fraction is the largest, else the
        # busiest on-chip units (vector ALUs / LDS busy fractions from the committed counter passes)
        units = {"hbm": achieved / HBM_PEAK}
        if "valu_frac" in prof:
            units["valu"] = prof["valu_frac"]
        if "lds_frac" in prof:
            units["lds"] = prof["lds_frac"]
        top = max(units.values())
        bound = "+".join(k for k in ("hbm", "lds", "valu") if k in units and units[k] >= 0.8 * top)
        if len(units) == 1:
            bound = "hbm (nominal: no counter pass of this workload and precision is committed)"
        roofline = {"bound": bound, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK,
                    "kernel": "static-feature kernel (f32: k_lean256 c2, c3; k_wide1024 c1, c4, c5; f64: k_pipe64 c2, c3; k_pipe64w c1, c4, c5)",
                    "kernel_ms": kavg * 1e3,
                    "algorithmic_bytes_per_frame": kernel_bytes,
                    "algorithmic_flop_per_frame": flops,
                    "alu_frac": alu_frac, "alu_peak_tflops": VALU_PEAK[precision] / 1e12,
                    "whole_pass": {"algorithmic_bytes_per_frame": pass_bytes,
                                   "frac": total_frames * pass_bytes * steps / elapsed / (HBM_PEAK * world)}}
        roofline.update(prof)
        rec = {"value": value, "ms_per_step": elapsed / steps * 1e3, "frames": int(frames), "roofline": roofline}
        return rec, one_pass, out, plan, ctx

    rec, one_pass, out, plan, ctx = measure(args.precision, args.steps, args.warmup)
    frames = rec["frames"]
    line = None
    if rank == 0:
        value = rec["value"]
        line = {
            "metric": "frames/sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "audio_hours_per_s": value * kw["inc"] / sr / 3600.0,
            "config": {"workload": desc, "name": args.workload,
                       "utterances_per_gpu": 1 if sharded else n_utts, "frames_per_gpu": int(frames),
                       "parallelism": ("frame ranges sharded by rank, halo frames recomputed, z-norm column sums exchanged "
                                       "(all_gather + rank-ordered sum)") if sharded
                       else "utterances sharded by rank, no collective"},
            "roofline": rec["roofline"],
        }
    if not args.no_host and world == 1:
        # Same pass with the batch starting and ending in pinned host memory (what a caller
        # of the host-buffer entry points sees); reported beside `value`, never as it.
        pcm_host = torch.from_numpy(buf).pin_memory()
        out_host = torch.empty(out.shape, dtype=out.dtype).pin_memory()
        reps = 3
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            pcm_dev.copy_(pcm_host, non_blocking=True)
            one_pass()
            out_host.copy_(out, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        line["host_buffers"] = {"value": frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                                "note": "pinned host PCM -> HBM -> kernels -> pinned host features"}
        del pcm_host, out_host
    del out, one_pass
    plan.close()
    ctx.close()
    if not args.no_other:
        # the same workload, steps and timing in the other precision, beside the headline
        other = "f64" if args.precision == "f32" else "f32"
        rec2, _, out2, plan2, ctx2 = measure(other, args.steps, args.warmup)
        del out2
        plan2.close()
        ctx2.close()
        if rank == 0:
            r2 = rec2["roofline"]
            line[other] = {"value": rec2["value"], "ms_per_step": rec2["ms_per_step"], "kernel_ms": r2["kernel_ms"],
                           "frac": r2["frac"], "alu_frac": r2["alu_frac"], "bound": r2["bound"],
                           "algorithmic_bytes_per_frame": r2["algorithmic_bytes_per_frame"],
                           "whole_pass_frac": r2["whole_pass"]["frac"],
                           "note": ("double precision: the library default and the reference's arithmetic (agreement 1e-11)"
                                    if other == "f64" else
                                    "single precision: misses the 1e-4 log-mel tolerance on 24-bit input in channels 60 dB "
                                    "or more below the frame's strongest one")}
    if rank == 0:
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(kw, buf, offs, lengths, channels, bps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
