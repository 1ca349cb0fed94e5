#!/usr/bin/env python3
"""Throughput benchmark of the MI355X extraction path.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the device-resident hot path (static-feature kernel, then deltas / z-norm
when the option set has them) over one batch of synthetic utterances that already sit in HBM.
Default workload: 16 kHz mono, 25 ms / 10 ms frames, 40-bin log-mel, 10 000 utterances of 10 s
per GPU (file lists shard by utterance: every rank owns its own batch, no collective on the data
path, weak scaling).  Rank 0 prints one JSON line.
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
    # name: (key in mfcc_amd.presets.CONFIGS, channels, bytes/sample, seconds, utterances,
    #        algorithmic bytes per frame of the whole pass, of the static-feature kernel alone
    #        (PCM read + its own rows written), description)
    "c2": ("c2", 1, 2, 10.0, 10000, 480, 480,
           "16 kHz mono 25ms/10ms, 40-bin log-mel FBank (fbankFlag=1), 10000 x 10 s utterances per GPU"),
    "c3": ("c3", 1, 2, 10.0, 10000, 860, 380,
           "16 kHz 13 MFCC+C0+energy, regreOrder=3 (delwin=2) + znorm, 10000 x 10 s utterances per GPU"),
    "c4": ("c4", 2, 3, 60.0, 256, 5692, 5692,
           "44.1 kHz 24-bit stereo vecNum=2, 40 MFCC + ZCR/brightness/8 sub-bands, 256 x 60 s files per GPU"),
}


def synth_batch(ma, n_utts, n_samples, channels, bps, sample_rate, seed_base, threads=16):
    """Packed PCM of n_utts utterances (16-byte aligned starts) from the library's integer
    generator; utterance u, channel ch uses seed seed_base + 8*u + ch."""
    import ctypes as C
    per = (n_samples * channels * bps + 15) // 16 * 16
    buf = np.zeros(n_utts * per, dtype=np.uint8)
    offs = np.arange(n_utts, dtype=np.int64) * per
    L = ma.lib()
    base = buf.ctypes.data

    def work(t):
        for u in range(t, n_utts, threads):
            for ch in range(channels):
                L.MFCCSynthPcm(C.c_void_p(base + int(offs[u])), channels, ch, bps, n_samples,
                               C.c_uint64(seed_base + 8 * u + ch), sample_rate, 0)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return buf, offs


def cpu_baseline(kw, buf, offs, n_samples, channels, bps, frames_per_utt):
    """Times the CPU oracle (thread per file) on a bounded sample of the same utterances."""
    from oracle import binding as ob
    ob.build()
    cfg = ob.make_config(**kw)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    ns = np.full(len(offs), n_samples, dtype=np.int64)
    probe = min(len(offs), 2 * threads)
    t0 = time.perf_counter()
    _, used = ob.extract_batch(cfg, buf, channels, bps, offs[:probe], ns[:probe], threads)
    dt = time.perf_counter() - t0
    rate = probe * frames_per_utt / dt
    # about 12 s of CPU work: a prefix of the batch, or the whole batch several times over
    want = 12.0 * rate / frames_per_utt
    n = int(min(len(offs), max(probe, want)))
    reps = max(1, int(round(want / n)))
    t0 = time.perf_counter()
    for _ in range(reps):
        _, used = ob.extract_batch(cfg, buf, channels, bps, offs[:n], ns[:n], threads)
    dt = time.perf_counter() - t0
    return {"value": reps * n * frames_per_utt / dt, "unit": "frames/s", "cores": used, "kind": "port",
            "sample": "first %d utterances of the same batch, %d pass(es) (%d frames), oracle C port, "
                      "one OpenMP thread per file, %.1f s" % (n, reps, reps * n * frames_per_utt, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--utts", type=int, default=0, help="utterances per GPU (0 = workload default)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    args = ap.parse_args()

    import torch
    import mfcc_amd as ma
    from mfcc_amd.presets import CONFIGS

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

    key, channels, bps, seconds, n_utts, pass_bytes_per_frame, bytes_per_frame, desc = WORKLOADS[args.workload]
    if args.utts:
        n_utts = args.utts
    kw = CONFIGS[key]
    prec = ma.PREC_F64 if args.precision == "f64" else ma.PREC_F32
    cfg = ma.make_config(precision=prec, **kw)
    n_samples = int(seconds * kw["sampleRate"])
    frames_per_utt = ma.num_frames(n_samples, kw["wlen"], kw["inc"])

    buf, offs = synth_batch(ma, n_utts, n_samples, channels, bps, kw["sampleRate"],
                            0x5EED0000 + 1000 * 2 + 100000 * rank)
    ctx = ma.Context(cfg, local_rank)
    plan = ma.Plan(ctx, channels, bps, offs, np.full(n_utts, n_samples, dtype=np.int64))
    pcm_dev = torch.from_numpy(buf).to("cuda:%d" % local_rank)
    out = plan.alloc_output()
    frames = plan.total_rows
    assert frames == n_utts * frames_per_utt
    plan.enable_timing(True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.extract(pcm_dev, out)
    sync_all()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.extract(pcm_dev, out)
        kernel_ms.append(plan.last_kernel_ms())  # waits for the kernel's stop event
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = frames * world
        value = total_frames * args.steps / elapsed
        kavg = float(np.mean(kernel_ms)) * 1e-3
        achieved = frames * bytes_per_frame / kavg
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("utts") == n_utts and tj.get("precision") == args.precision:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "frames/sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "audio_hours_per_s": value * kw["inc"] / kw["sampleRate"] / 3600.0,
            "config": {"workload": desc, "name": args.workload, "utterances_per_gpu": n_utts,
                       "frames_per_gpu": frames, "parallelism": "utterances sharded by rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": "k_front (static features)", "kernel_ms": kavg * 1e3,
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "whole_pass": {"algorithmic_bytes_per_frame": pass_bytes_per_frame,
                                        "frac": frames * pass_bytes_per_frame * args.steps / elapsed / HBM_PEAK}},
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
                plan.extract(pcm_dev, out)
                out_host.copy_(out, non_blocking=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            line["host_buffers"] = {"value": frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                                    "note": "pinned host PCM -> HBM -> kernels -> pinned host features"}
            del pcm_host, out_host
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(kw, buf, offs, n_samples, channels, bps, frames_per_utt)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
