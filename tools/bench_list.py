#!/usr/bin/env python3
"""End-to-end rate of the drop-in list entry point (MFCCWapperList = the reference's main loop,
main.c:173-289): WAV files on disk in, feature files on disk out, nothing resident beforehand.

  python tools/bench_list.py [--files 2000] [--seconds 10] [--save b,n,f] [--preset c2|mfcc39] [--precision f64|f32]
                             [--dir <repo>/.bench_list_tmp]

Writes synthetic 16-bit mono WAV files, one ini per save type, and prints one JSON line per save
type with files/s and frames/s (file reads, H2D, kernels, D2H and writers all inside the timed
region).  The scratch directory (inside the repository tree by default) is removed afterwards.
"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, REPO)
import mfcc_amd as ma  # noqa: E402


def write_wav(path, pcm, rate):
    import struct
    data = pcm.astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", len(data))
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2000)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--save", default="b,n,f")
    ap.add_argument("--dir", default=os.path.join(REPO, ".bench_list_tmp"))
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--preset", default="mfcc39", choices=["mfcc39", "c2"],
                    help="mfcc39: 12 MFCC + C0, deltas and delta-deltas; c2: the 40-bin log-mel benchmark configuration")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"])
    args = ap.parse_args()
    rate, wlen, inc = 16000, 400, 160
    n = int(args.seconds * rate)
    shutil.rmtree(args.dir, ignore_errors=True)
    os.makedirs(os.path.join(args.dir, "wav"))
    os.makedirs(os.path.join(args.dir, "out"))
    rng = np.random.default_rng(1)
    base = (rng.standard_normal(n + 4096) * 3000).astype(np.int16)
    try:
        for i in range(args.files):
            k = i % 4096
            write_wav(os.path.join(args.dir, "wav", "%06d.wav" % i), base[k:k + n], rate)
        frames = (n - wlen) // inc + 1
        lst = os.path.join(args.dir, "list.txt")
        for save in args.save.split(","):
            with open(lst, "w") as f:
                for i in range(args.files):
                    f.write("%s\t%s\n" % (os.path.join(args.dir, "wav", "%06d.wav" % i),
                                          os.path.join(args.dir, "out", "%06d.%s" % (i, save))))
            ini = os.path.join(args.dir, "config_%s.ini" % save)
            with open(ini, "w") as f:
                if args.preset == "c2":
                    body = ("[Frame]\nsampleRate = %d\nwlen = %d\ninc = %d\nlowpassfre = 20\nhipassfre = 8000\n"
                            "preemphasise = 0.97\nsaveType = %s\nfileList = %s\n[MFCC]\nfbankFlag = 1\nbankNum = 40\n"
                            % (rate, wlen, inc, save, lst))
                else:
                    body = ("[Frame]\nsampleRate = %d\nwlen = %d\ninc = %d\nsaveType = %s\nfileList = %s\n"
                            "[MFCC]\nMFCCNum = 12\nMFCC0thFlag = 1\n[Regression]\nregreOrder = 3\ndelwin = 2\n"
                            % (rate, wlen, inc, save, lst))
                f.write(body + "[IO]\nnumThreads = %d\nprecision = %s\n" % (args.threads, args.precision))
            os.environ["MFCC_QUIET"] = "1"
            best = None
            for _ in range(args.repeat):
                t0 = time.perf_counter()
                rc = ma.lib().MFCCWapperList(ini.encode())
                dt = time.perf_counter() - t0
                assert rc == 0, (rc, ma.lib().MFCCLastError())
                best = dt if best is None else min(best, dt)
            out0 = os.path.join(args.dir, "out", "%06d.%s" % (0, save))
            print(json.dumps({"entry": "MFCCWapperList", "preset": args.preset, "precision": args.precision, "saveType": save, "files": args.files,
                              "seconds_per_file": args.seconds, "wall_s": round(best, 4),
                              "files_per_s": round(args.files / best, 1),
                              "frames_per_s": round(args.files * frames / best, 1),
                              "out_bytes_per_file": os.path.getsize(out0), "threads": args.threads}),
                  flush=True)
    finally:
        shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
