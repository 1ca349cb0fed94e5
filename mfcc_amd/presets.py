"""Named option sets: the five benchmark configurations of this build (plus variants)."""

# Option sets named after the five benchmark configurations (plus variants).
CONFIGS = {
    # 16 kHz mono 25 ms / 10 ms, 40-bin log-mel
    "c2": dict(sampleRate=16000, lowpassfre=20, hipassfre=8000, preemphasise=0.97, wlen=400,
               inc=160, vecNum=1, fbankFlag=1, bankNum=40),
    # 16 kHz 13 MFCC + C0 + energy, deltas (delwin 2) + znorm
    "c3": dict(sampleRate=16000, lowpassfre=64, hipassfre=8000, preemphasise=0.97, wlen=400,
               inc=160, vecNum=1, fbankFlag=0, bankNum=26, MFCCNum=13, MFCC0thFlag=1, energyFlag=1,
               regreOrder=3, delwin=2, znormFlag=1),
    # 44.1 kHz stereo vecNum 2, 40 MFCC + ZCR / brightness / 8 sub-bands, 2048-point FFT
    "c4": dict(sampleRate=44100, lowpassfre=10, hipassfre=22050, preemphasise=0.97, wlen=1764,
               inc=882, vecNum=2, fbankFlag=0, bankNum=40, MFCCNum=40, zeroCrossingFlag=1,
               brightFlag=1, subBandEFlag=8),
    # the reference's example/config.ini verbatim (306 columns)
    "c1a": dict(sampleRate=44100, lowpassfre=10, hipassfre=22050, preemphasise=0.0, wlen=1764,
                inc=882, vecNum=2, fbankFlag=1, bankNum=40, MFCCNum=-1, energyFlag=1,
                zeroCrossingFlag=1, brightFlag=1, subBandEFlag=8, regreOrder=3, delwin=9),
    # README's 408-column variant: vecNum 4, regreOrder 2
    "c1b": dict(sampleRate=44100, lowpassfre=10, hipassfre=22050, preemphasise=0.0, wlen=1764,
                inc=882, vecNum=4, fbankFlag=1, bankNum=40, MFCCNum=-1, energyFlag=1,
                zeroCrossingFlag=1, brightFlag=1, subBandEFlag=8, regreOrder=2, delwin=9),
}
CONFIGS["c5"] = dict(CONFIGS["c1a"], znormFlag=1)

# Precision bench.py reports each workload in: single precision wherever it meets the stated
# tolerance (1e-4 absolute on log-mel) on the worst input of the workload's PCM format -- 16-bit
# input does -- and double precision for the 24-bit workloads, where a single-precision FFT's
# rounding noise (relative to the frame's energy) exceeds 1e-4 in channels 60 dB or more below the
# frame's strongest one.  The other precision is reported beside it.
BENCH_PRECISION = {"c2": "f32", "c3": "f32", "c1": "f64", "c4": "f64", "c5": "f64"}
