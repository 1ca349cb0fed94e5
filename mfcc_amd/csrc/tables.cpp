// One-time host tables, computed in double and uploaded by the context:
//   filterbank set-up (replaces InitFBank / Mel, mfcc.c:11-14,39-122; alpha == 1.0 as at the
//   only call site main.c:168), Hamming window (GenHamWindow, call site main.c:171), DCT matrix
//   (the cos() of mfcc.c:192 hoisted out of the frame loop) and FFT twiddles.
#include <math.h>

#include "mfcc_host.h"

namespace mfcc {

static double mel_of_bin(int k, double fres) { return 1127 * log(1 + (k - 1) * fres); } // mfcc.c:13

// WarpFreq (mfcc.c:17-36): piecewise-linear vocal-tract-length warping of a centre frequency.
static double warp_freq(double fcl, double fcu, double freq, double minFreq, double maxFreq, double alpha)
{
    if (alpha == 1.0) return freq;
    const double scale = 1.0 / alpha;
    const double cu = fcu * 2 / (1 + scale), cl = fcl * 2 / (1 + scale);
    const double au = (maxFreq - cu * scale) / (maxFreq - cu), al = (cl * scale - minFreq) / (cl - minFreq);
    if (freq > cu) return au * (freq - cu) + scale * cu;
    if (freq < cl) return al * (freq - minFreq) + minFreq;
    return scale * freq;
}

// InitFBank (mfcc.c:39-122).  The extraction path calls it as main.c:168 does (doubleFFT = 0,
// alpha = 1.0); the other values serve callers of the lower-level API.
int build_fbank(int frameSize, double sampPeriod, int numChans, double lopass, double hipass, Tables &t,
                int doubleFFT, double alpha, double warpLowCut, double warpUpCut)
{
    const int P = numChans, maxChan = P + 1;
    t.fftN = 2;
    while (frameSize > t.fftN) t.fftN *= 2; // mfcc.c:53-54
    if (doubleFFT) t.fftN *= 2;             // mfcc.c:55-56
    t.nc = t.fftN / 2;
    t.log2nc = 0;
    while ((1 << t.log2nc) < t.nc) ++t.log2nc;
    const int half = t.nc;
    t.fres = 1.0E7 / (sampPeriod * t.fftN * 700.0); // mfcc.c:58

    // pass band, mfcc.c:61-73
    t.klo = 2;
    t.khi = half;
    t.mlo = 0;
    t.mhi = mel_of_bin(half + 1, t.fres);
    if (lopass >= 0.0) {
        t.mlo = 1127 * log(1 + lopass / 700.0);
        t.klo = (int)((lopass * sampPeriod * 1.0e-7 * t.fftN) + 2.5);
        if (t.klo < 2) t.klo = 2;
    }
    if (hipass >= 0.0) {
        t.mhi = 1127 * log(1 + hipass / 700.0);
        t.khi = (int)((hipass * sampPeriod * 1.0e-7 * t.fftN) + 0.5);
        if (t.khi > half) t.khi = half;
    }

    // centre frequencies, mfcc.c:77-82
    t.cf.assign(maxChan, 0.0);
    const double ms = t.mhi - t.mlo;
    for (int chan = 1; chan <= maxChan; ++chan) {
        double cf = ((double)chan / (double)maxChan) * ms + t.mlo;
        if (alpha != 1.0) { // mfcc.c:83-92: scale assuming scaling starts at lopass
            const double minFreq = 700.0 * (exp(t.mlo / 1127.0) - 1.0), maxFreq = 700.0 * (exp(t.mhi / 1127.0) - 1.0);
            cf = 700 * (exp(cf / 1127.0) - 1.0);
            cf = 1127.0 * log(1.0 + warp_freq(warpLowCut, warpUpCut, cf, minFreq, maxFreq, alpha) / 700.0);
        }
        t.cf[chan - 1] = cf;
    }

    // lower channel of each bin, mfcc.c:96-104
    t.loChan.assign(half, -1);
    for (int k = 1, chan = 1; k <= half; ++k) {
        if (k < t.klo || k > t.khi) continue;
        const double melk = mel_of_bin(k, t.fres);
        while (chan <= maxChan && t.cf[chan - 1] < melk) ++chan;
        t.loChan[k - 1] = chan - 1;
    }
    // lower-channel weights, mfcc.c:107-118
    t.loWt.assign(half, 0.0);
    for (int k = t.klo; k <= t.khi && k <= half; ++k) {
        const int chan = t.loChan[k - 1];
        const double melk = mel_of_bin(k, t.fres);
        if (chan > 0)
            t.loWt[k - 1] = (t.cf[chan] - melk) / (t.cf[chan] - t.cf[chan - 1]);
        else
            t.loWt[k - 1] = (t.cf[0] - melk) / (t.cf[0] - t.mlo);
    }
    return MFCC_OK;
}

int build_tables(const MFCCConfig &cfg, Tables &t)
{
    const double sampPeriod = 1e7 / (double)cfg.sampleRate; // main.c:141
    const int P = cfg.bankNum;
    build_fbank(cfg.wlen, sampPeriod, P, cfg.lowpassfre, cfg.hipassfre, t, cfg.doubleFFT, cfg.warpAlpha, cfg.warpLowCut,
                cfg.warpUpCut); // main.c:168 passes 0, 1.0, 0, 0
    // Hamming window: HTK GenHamWindow, w[i] = 0.54 - 0.46 cos(2 pi (i-1) / (W-1))
    t.window.resize(cfg.wlen);
    {
        const double a = kHtkTwoPi / (cfg.wlen - 1);
        for (int i = 0; i < cfg.wlen; ++i) t.window[i] = 0.54 - 0.46 * cos(a * i);
    }

    // DCT-II rows 1..M, k-major; scale sqrt(2/P) applied after the sum as mfcc.c:193 does
    const int M = eff_mfccnum(cfg);
    t.mfnorm = sqrt(2.0 / (double)P);
    t.dct.assign((size_t)P * M, 0.0);
    {
        const double piFactor = kHtkPi / (double)P;
        for (int j = 1; j <= M; ++j) {
            const double x = (double)j * piFactor;
            for (int k = 1; k <= P; ++k) t.dct[(size_t)(k - 1) * M + (j - 1)] = cos(x * (k - 0.5));
        }
    }

    // FFT twiddles exp(-2 pi i m / nc) and real-FFT split factors (cos, sin)(pi a / nc)
    t.tw.resize(2 * (size_t)t.nc);
    for (int m = 0; m < t.nc; ++m) {
        const double ang = 2.0 * M_PI * (double)m / (double)t.nc;
        t.tw[2 * m] = cos(ang);
        t.tw[2 * m + 1] = -sin(ang);
    }
    const int nsplit = t.nc / 2 > 0 ? t.nc / 2 : 1;
    t.splitTw.resize(2 * (size_t)nsplit);
    for (int a = 0; a < nsplit; ++a) {
        const double ang = M_PI * (double)a / (double)t.nc;
        t.splitTw[2 * a] = cos(ang);
        t.splitTw[2 * a + 1] = sin(ang);
    }
    return MFCC_OK;
}

} // namespace mfcc

using namespace mfcc;

extern "C" int MFCCHostTables(const MFCCConfig *cfgIn, int *fftN, int *klo, int *khi, double *cf,
                              int *loChan, double *loWt, double *window, char *passband, size_t cap)
{
    if (!cfgIn) return set_error("NULL argument"), MFCC_E_ARG;
    MFCCConfig cfg = *cfgIn;
    if (cfg.fbankFlag) {
        cfg.MFCCNum = cfg.bankNum;
        cfg.MFCC0thFlag = 0;
    }
    int rc = validate_config(cfg);
    if (rc) return rc;
    Tables t;
    if ((rc = build_tables(cfg, t))) return rc;
    if (fftN) *fftN = t.fftN;
    if (klo) *klo = t.klo;
    if (khi) *khi = t.khi;
    if (cf) std::copy(t.cf.begin(), t.cf.end(), cf);
    if (loChan) std::copy(t.loChan.begin(), t.loChan.end(), loChan);
    if (loWt) std::copy(t.loWt.begin(), t.loWt.end(), loWt);
    if (window) std::copy(t.window.begin(), t.window.end(), window);
    if (passband)
        snprintf(passband, cap, "FFT passband %d to %d out of 1 to %d\nMel passband %f to %f\n", t.klo,
                 t.khi, t.nc, t.mlo, t.mhi); /* mfcc.c:74-75 */
    return MFCC_OK;
}

