#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference ships no feature vectors (its example wav / fbank blobs are missing and it has no
tests), and its mfcc.c cannot be built here (hmath.h absent), so these fixtures are SELF-GOLDEN:
inputs from the library's integer PCM generator, expected outputs from the CPU oracle
(oracle/mfcc_oracle.c).  They pin the oracle against accidental change and give the GPU tests a
second, file-based reference.  The only value that comes from the reference itself is the
pass-band text of README.md:33-34, stored in known_answers.json.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import mfcc_amd as ma  # noqa: E402  (only the host-side PCM generator is used)
from mfcc_amd.presets import CONFIGS  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (config key, channels, bytes/sample, samples, tone Hz, overrides)
    "c2_mono16": ("c2", 1, 2, 8000 + 37, 440, {}),
    "c3_mono16": ("c3", 1, 2, 8000 + 37, 1000, {}),
    "c4_stereo24": ("c4", 2, 3, 22050 + 5, 440, {}),
    "c1a_stereo24": ("c1a", 2, 3, 22050 + 5, 1000, {}),
    "c1b_stereo24": ("c1b", 2, 3, 22050 + 5, 1000, {}),
    "c5_stereo24": ("c5", 2, 3, 22050 + 5, 440, {}),
    "c3_zeromean": ("c3", 1, 2, 4000, 440, {"zeroMeanSigFlag": 1}),
    "c2_exact_window": ("c2", 1, 2, 400, 0, {}),            # N == wlen: one frame
    "c2_short": ("c2", 1, 2, 399, 0, {}),                   # N < wlen: no frame
    "c2_tail": ("c2", 1, 2, 400 + 160 - 1, 0, {}),          # N == wlen + inc - 1: still one frame
}


def main():
    for name, (key, ch, bps, n, tone, over) in CASES.items():
        kw = dict(CONFIGS[key], **over)
        pcm = ma.synth_pcm(n, ch, bps, seed=0x5EED0000 + sum(map(ord, name)), sample_rate=kw["sampleRate"],
                           tone_hz=tone)
        ref = ob.extract_pcm_bytes(ob.make_config(**kw), pcm, ch, bps)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pcm=pcm, expected=ref,
                            channels=ch, bytes_per_sample=bps, config=json.dumps(kw))
        print(name, ref.shape)
    known = {
        "readme_passband": {
            "source": "README.md:33-34 (printed by mfcc.c:74-75 for example/config.ini)",
            "config": {"wlen": 1764, "sampleRate": 44100, "bankNum": 40, "lowpassfre": 10, "hipassfre": 22050},
            "text": "FFT passband 2 to 1024 out of 1 to 1024\nMel passband 15.986084 to 3923.357581\n",
        },
        "readme_example_dims": {
            "source": "example/config.ini + main.c:161,211,254 (snapshot arithmetic); README.md:48 sample count",
            "numSamples": 1323001, "frames": 1499, "step": 306,
        },
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)


if __name__ == "__main__":
    main()
