#!/usr/bin/env python3
"""GPU box: the command-line program built with host AddressSanitizer (tools/asan/mfcc_asan, linked
with a host-sanitized build of the library) over a mixed file list: 16- and 24-bit, mono and
stereo, every save type, deltas and z-norm, a file shorter than a window, a broken file.
Prints the program's exit codes and any sanitizer report."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mfcc_amd as ma  # noqa: E402  (only for wav_bytes: no GPU work in this process)

tmp = "/tmp/asan_run"
os.makedirs(tmp, exist_ok=True)
rng = np.random.default_rng(3)
exe = os.path.join(ROOT, "tools", "asan", "mfcc_asan")
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0", MFCC_QUIET="1")
rc_all = 0
for case, (sr, ch, bps, wlen, inc, save, extra) in enumerate([
        (16000, 1, 2, 400, 160, "f", "fbankFlag = 1 ;\nbankNum = 40 ;\n"),
        (16000, 1, 2, 400, 160, "n", "fbankFlag = 0 ;\nbankNum = 26 ;\nMFCCNum = 13 ;\nMFCC0thFlag = 1 ;\n"),
        (44100, 2, 3, 1764, 882, "b", "fbankFlag = 1 ;\nbankNum = 40 ;\n"),
        (44100, 2, 3, 2048, 512, "e", "fbankFlag = 0 ;\nbankNum = 30 ;\nMFCCNum = 12 ;\nMFCC0thFlag = 0 ;\n")]):
    lines = []
    for i, n in enumerate([sr * 2 + 17, sr // 3, wlen - 1, wlen, sr + 5]):
        x = (rng.standard_normal((n, ch)) * 3000).astype(np.int32)
        if bps == 2:
            raw = x.astype("<i2").tobytes()
        else:
            raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in (x * 200).reshape(-1))
        path = "%s/c%d_%d.wav" % (tmp, case, i)
        open(path, "wb").write(ma.wav_bytes(np.frombuffer(raw, dtype=np.uint8), ch, bps, sr))
        lines.append("%s\t%s/c%d_%d.out" % (path, tmp, case, i))
    open("%s/c%d_bad.wav" % (tmp, case), "wb").write(b"RIFFxxxxWAVEjunk")
    for prec in ("f64", "f32"):
        ini = "%s/c%d_%s.ini" % (tmp, case, prec)
        open("%s/list%d.txt" % (tmp, case), "w").write("\n".join(lines) + "\n")
        open(ini, "w").write("[Frame];\nsampleRate = %d ;\nhipassfre = -1 ;\nlowpassfre = -1 ;\npreemphasise = 0.97 ;\nwlen = %d ;\ninc = %d ;\n"
                             "saveType = %s ;\nvecNum = %d ;\nfileList = %s/list%d.txt ;\n[MFCC];\n%s[Others];\nenergyFlag = 1 ;\n"
                             "zeroCrossingFlag = 1 ;\nbrightFlag = 1 ;\nsubBandEFlag = %d ;\nfftLength = 0 ;\n[Regression];\nregreOrder = 3 ;\n"
                             "delwin = 2 ;\nznormFlag = 1 ;\n[IO];\nprecision = %s ;\nnumThreads = 8 ;\n"
                             % (sr, wlen, inc, save, ch if ch == 1 else 2, tmp, case, extra, 8 if ch == 2 else 0, prec))
        r = subprocess.run([exe, ini], capture_output=True, text=True, env=env, timeout=300)
        bad = "AddressSanitizer" in r.stderr or "AddressSanitizer" in r.stdout
        print("case %d %s: exit %d%s" % (case, prec, r.returncode, "  SANITIZER REPORT" if bad else ""), flush=True)
        if bad or r.returncode not in (0, 1):
            print(r.stderr[-3000:])
            rc_all = 1
    # a list with a broken file: the program must report it, not crash
    open("%s/listbad%d.txt" % (tmp, case), "w").write("%s/c%d_bad.wav\t%s/bad.out\n" % (tmp, case, tmp))
    ini = "%s/c%d_bad.ini" % (tmp, case)
    open(ini, "w").write(open("%s/c%d_f64.ini" % (tmp, case)).read().replace("list%d.txt" % case, "listbad%d.txt" % case))
    r = subprocess.run([exe, ini], capture_output=True, text=True, env=env, timeout=300)
    bad = "AddressSanitizer" in r.stderr
    print("case %d broken file: exit %d%s" % (case, r.returncode, "  SANITIZER REPORT" if bad else ""), flush=True)
    if bad:
        print(r.stderr[-3000:])
        rc_all = 1
sys.exit(rc_all)
