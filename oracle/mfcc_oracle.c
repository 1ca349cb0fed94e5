/*
 * TEST INFRASTRUCTURE ONLY -- see mfcc_oracle.h for the rules and the parity status
 * (filterbank set-up pinned by README.md:33-34; everything else PARITY UNPINNED).
 *
 * Scalar IEEE-double restatement of the reference's per-file extraction body.  Arithmetic follows
 * the reference expression by expression so that a build of the real sources would agree to the
 * last bit wherever the sources exist; build with -ffp-contract=off.
 */
#include "mfcc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* HTK's hmath.h constants (the reference's hmath.h is absent; README.md:143 says it follows HTK). */
#define ORC_PI 3.14159265358979
#define ORC_TPI 6.28318530717959

/* ---------------------------------------------------------------- filterbank set-up */

double orc_mel(int k, double fres) /* mfcc.c:11-14, k is the reference's 1-based bin index */
{
    return 1127 * log(1 + (k - 1) * fres);
}

double orc_warp_freq(double fcl, double fcu, double freq, double minFreq, double maxFreq,
                     double alpha) /* mfcc.c:17-36 */
{
    if (alpha == 1.0) return freq;
    const double scale = 1.0 / alpha;
    const double cu = fcu * 2 / (1 + scale);
    const double cl = fcl * 2 / (1 + scale);
    const double au = (maxFreq - cu * scale) / (maxFreq - cu);
    const double al = (cl * scale - minFreq) / (cl - minFreq);
    if (freq > cu) return au * (freq - cu) + scale * cu;
    if (freq < cl) return al * (freq - minFreq) + minFreq;
    return scale * freq;
}

int orc_fbank_init(orc_fbank *fb, int frameSize, double sampPeriod, int numChans, double lopass,
                   double hipass) /* the call of main.c:168 */
{
    return orc_fbank_init_ex(fb, frameSize, sampPeriod, numChans, lopass, hipass, 1, 1, 0, 1.0, 0.0, 0.0);
}

int orc_fbank_init_ex(orc_fbank *fb, int frameSize, double sampPeriod, int numChans, double lopass,
                      double hipass, int usePower, int takeLogs, int doubleFFT, double alpha,
                      double warpLowCut, double warpUpCut)
{
    memset(fb, 0, sizeof *fb);
    fb->frameSize = frameSize;
    fb->numChans = numChans;
    fb->sampPeriod = sampPeriod;
    fb->usePower = usePower; /* mfcc.c:52 */
    fb->takeLogs = takeLogs;
    fb->fftN = 2; /* mfcc.c:53-54 */
    while (frameSize > fb->fftN) fb->fftN *= 2;
    if (doubleFFT) fb->fftN *= 2; /* mfcc.c:55-56 */
    const int half = fb->fftN / 2;
    fb->fres = 1.0E7 / (sampPeriod * fb->fftN * 700.0); /* mfcc.c:58 */
    const int maxChan = numChans + 1;
    fb->klo = 2; /* mfcc.c:61-73 */
    fb->khi = half;
    fb->mlo = 0;
    fb->mhi = orc_mel(half + 1, fb->fres);
    if (lopass >= 0.0) {
        fb->mlo = 1127 * log(1 + lopass / 700.0);
        fb->klo = (int)((lopass * sampPeriod * 1.0e-7 * fb->fftN) + 2.5);
        if (fb->klo < 2) fb->klo = 2;
    }
    if (hipass >= 0.0) {
        fb->mhi = 1127 * log(1 + hipass / 700.0);
        fb->khi = (int)((hipass * sampPeriod * 1.0e-7 * fb->fftN) + 0.5);
        if (fb->khi > half) fb->khi = half;
    }
    fb->cf = (double *)calloc((size_t)maxChan + 1, sizeof(double));
    fb->loChan = (int *)calloc((size_t)half, sizeof(int));
    fb->loWt = (double *)calloc((size_t)half, sizeof(double));
    if (!fb->cf || !fb->loChan || !fb->loWt) return -1;
    const double ms = fb->mhi - fb->mlo; /* mfcc.c:78-93 */
    for (int chan = 1; chan <= maxChan; chan++) {
        if (alpha == 1.0) {
            fb->cf[chan - 1] = ((double)chan / (double)maxChan) * ms + fb->mlo;
        } else { /* scale assuming scaling starts at lopass */
            const double minFreq = 700.0 * (exp(fb->mlo / 1127.0) - 1.0);
            const double maxFreq = 700.0 * (exp(fb->mhi / 1127.0) - 1.0);
            double cf = ((double)chan / (double)maxChan) * ms + fb->mlo;
            cf = 700 * (exp(cf / 1127.0) - 1.0);
            fb->cf[chan - 1] =
                1127.0 * log(1.0 + orc_warp_freq(warpLowCut, warpUpCut, cf, minFreq, maxFreq, alpha) / 700.0);
        }
    }
    /* mfcc.c:96-104: lower channel of every bin, -1 outside the pass band */
    for (int k = 1, chan = 1; k <= half; k++) {
        const double melk = orc_mel(k, fb->fres);
        if (k < fb->klo || k > fb->khi)
            fb->loChan[k - 1] = -1;
        else {
            while (chan <= maxChan && fb->cf[chan - 1] < melk) ++chan;
            fb->loChan[k - 1] = chan - 1;
        }
    }
    /* mfcc.c:107-118: weight of the lower channel */
    for (int k = 1; k <= half; k++) {
        const int chan = fb->loChan[k - 1];
        if (k < fb->klo || k > fb->khi)
            fb->loWt[k - 1] = 0.0;
        else if (chan > 0)
            fb->loWt[k - 1] =
                (fb->cf[chan] - orc_mel(k, fb->fres)) / (fb->cf[chan] - fb->cf[chan - 1]);
        else
            fb->loWt[k - 1] = (fb->cf[0] - orc_mel(k, fb->fres)) / (fb->cf[0] - fb->mlo);
    }
    return 0;
}

void orc_fbank_free(orc_fbank *fb)
{
    free(fb->cf);
    free(fb->loChan);
    free(fb->loWt);
    memset(fb, 0, sizeof *fb);
}

int orc_fbank_passband_text(const orc_fbank *fb, char *buf, size_t cap) /* mfcc.c:74-75 */
{
    return snprintf(buf, cap, "FFT passband %d to %d out of 1 to %d\nMel passband %f to %f\n",
                    fb->klo, fb->khi, fb->fftN / 2, fb->mlo, fb->mhi);
}

/* ---------------------------------------------------------------- FFT */

/* mfcc.c:211-259.  s holds n doubles = n/2 interleaved complex values. */
void orc_fft(double *s, int n, int invert)
{
    const int nc = n / 2; /* complex points */
    /* bit-reversal reordering, mfcc.c:219-231 (complex index j walks the reversed counter) */
    for (int i = 0, j = 0; i < nc; i++) {
        if (j > i) {
            double tr = s[2 * j], ti = s[2 * j + 1];
            s[2 * j] = s[2 * i];
            s[2 * j + 1] = s[2 * i + 1];
            s[2 * i] = tr;
            s[2 * i + 1] = ti;
        }
        int m = nc / 2;
        while (m >= 1 && j >= m) {
            j -= m;
            m /= 2;
        }
        j += m;
    }
    /* butterflies with the trigonometric recurrence, mfcc.c:232-254; forward sign e^{+i theta} */
    for (int limit = 2; limit < n; limit *= 2) {
        double theta = 2 * ORC_PI / limit;
        if (invert) theta = -theta;
        const double x = sin(0.5 * theta);
        const double wpr = -2.0 * x * x, wpi = sin(theta);
        double wr = 1.0, wi = 0.0;
        const int half = limit / 2; /* butterflies per group, in complex points */
        for (int q = 0; q < half; q++) {
            for (int a = q; a < nc; a += limit) {
                const int b = a + half;
                const double xre = wr * s[2 * b] - wi * s[2 * b + 1];
                const double xri = wr * s[2 * b + 1] + wi * s[2 * b];
                s[2 * b] = s[2 * a] - xre;
                s[2 * b + 1] = s[2 * a + 1] - xri;
                s[2 * a] = s[2 * a] + xre;
                s[2 * a + 1] = s[2 * a + 1] + xri;
            }
            const double wx = wr;
            wr = wr * wpr - wi * wpi + wr;
            wi = wi * wpr + wx * wpi + wi;
        }
    }
    if (invert)
        for (int i = 0; i < n; i++) s[i] = s[i] / nc;
}

/* mfcc.c:262-291.  n real inputs -> n/2 complex bins (DC .. n/2-1), Nyquist discarded. */
void orc_realft(double *s, int n)
{
    const int nc = n / 2, n2 = nc / 2;
    const double theta = ORC_PI / nc;
    orc_fft(s, n, 0);
    const double x = sin(0.5 * theta);
    const double yr2 = -2.0 * x * x, yi2 = sin(theta);
    double yr = 1.0 + yr2, yi = yi2;
    for (int a = 1; a < n2; a++) { /* reference i = a+1 runs 2..n2; partner bin is nc-a */
        const int b = nc - a;
        const double wrs = yr, wis = yi;
        const double xr1 = (s[2 * a] + s[2 * b]) / 2.0, xi1 = (s[2 * a + 1] - s[2 * b + 1]) / 2.0;
        const double xr2 = (s[2 * a + 1] + s[2 * b + 1]) / 2.0, xi2 = (s[2 * b] - s[2 * a]) / 2.0;
        s[2 * a] = xr1 + wrs * xr2 - wis * xi2;
        s[2 * a + 1] = xi1 + wrs * xi2 + wis * xr2;
        s[2 * b] = xr1 - wrs * xr2 + wis * xi2;
        s[2 * b + 1] = -xi1 + wrs * xi2 + wis * xr2;
        const double yr0 = yr;
        yr = yr * yr2 - yi * yi2 + yr;
        yi = yi * yr2 + yr0 * yi2 + yi;
    }
    const double xr1 = s[0];
    s[0] = xr1 + s[1];
    s[1] = 0.0;
}

/* ---------------------------------------------------------------- per-frame analysis */

/* mfcc.c:125-177.  x is the fftN
 * workspace and keeps the packed spectrum on return, as info.x does in the reference. */
void orc_wave2fbank(const double *s, double *fbank, double *te, double *te2, const orc_fbank *fb,
                    double *x)
{
    const double melfloor = 1.0;
    if (te) {
        *te = 0.0;
        for (int k = 0; k < fb->frameSize; k++) *te += (s[k] * s[k]);
    }
    for (int k = 0; k < fb->frameSize; k++) x[k] = s[k];
    for (int k = fb->frameSize; k < fb->fftN; k++) x[k] = 0.0;
    orc_realft(x, fb->fftN);
    for (int b = 0; b < fb->numChans; b++) fbank[b] = 0.0;
    for (int k = fb->klo; k <= fb->khi; k++) { /* 1-based bin k: re x[2k-2], im x[2k-1] */
        double t1 = x[2 * k - 2], t2 = x[2 * k - 1];
        const double ek = fb->usePower ? t1 * t1 + t2 * t2 : sqrt(t1 * t1 + t2 * t2); /* mfcc.c:154-157 */
        const int bin = fb->loChan[k - 1];
        t1 = fb->loWt[k - 1] * ek;
        if (bin > 0) fbank[bin - 1] += t1;
        if (bin < fb->numChans) fbank[bin] += ek - t1;
    }
    if (te2) {
        *te2 = 0.0;
        for (int k = 1; k <= fb->fftN / 2; k++)
            *te2 += x[2 * k - 2] * x[2 * k - 2] + x[2 * k - 1] * x[2 * k - 1];
    }
    if (fb->takeLogs) /* mfcc.c:171-176 */
        for (int b = 0; b < fb->numChans; b++) {
            double t1 = fbank[b];
            if (t1 < melfloor) t1 = melfloor;
            fbank[b] = log(t1);
        }
}

void orc_fbank2mfcc(const double *fbank, int numChan, double *c, int n) /* mfcc.c:180-195 */
{
    const double mfnorm = sqrt(2.0 / (double)numChan);
    const double pi_factor = ORC_PI / (double)numChan;
    for (int j = 1; j <= n; j++) {
        const double x = (double)j * pi_factor;
        double acc = 0.0;
        for (int k = 1; k <= numChan; k++) acc += fbank[k - 1] * cos(x * (k - 0.5));
        c[j - 1] = acc * mfnorm;
    }
}

double orc_fbank2c0(const double *fbank, int numChan) /* mfcc.c:197-208 */
{
    const double mfnorm = sqrt(2.0 / (double)numChan);
    double sum = 0.0;
    for (int k = 0; k < numChan; k++) sum += fbank[k];
    return sum * mfnorm;
}

/* HTK HSigP PreEmphasise, applied by main.c:208 to the whole signal before framing. */
void orc_preemphasise(double *s, long n, double k)
{
    if (n < 1) return;
    for (long i = n - 1; i >= 1; i--) s[i] -= s[i - 1] * k;
    s[0] *= 1.0 - k;
}

/* HTK HSigP GenHamWindow (call site main.c:171); Ham (main.c:229) is the element-wise product. */
void orc_hamming(double *w, int n)
{
    const double a = ORC_TPI / (n - 1);
    for (int i = 0; i < n; i++) w[i] = 0.54 - 0.46 * cos(a * i);
}

/* main.c:227 "average zero crossing rate" (README.md:111) of the un-windowed frame.  Source
 * absent; definition used: sign changes between neighbours divided by the neighbour pairs. */
double orc_zcr(const double *s, int n)
{
    if (n < 2) return 0.0;
    int cnt = 0;
    for (int i = 0; i + 1 < n; i++)
        if (s[i] * s[i + 1] < 0.0) cnt++;
    return (double)cnt / (double)(n - 1);
}

/* main.c:235; source absent.  main.c:234 says centroid and sub-band energies "are fractions":
 * power-weighted mean 1-based bin index over the fftN/2 packed bins, divided by fftN/2. */
double orc_brightness(const double *x, int fftN)
{
    const int half = fftN / 2;
    double num = 0.0, den = 0.0;
    for (int k = 1; k <= half; k++) {
        const double p = x[2 * k - 2] * x[2 * k - 2] + x[2 * k - 1] * x[2 * k - 1];
        num += (double)k * p;
        den += p;
    }
    return num / den / (double)half;
}

/* main.c:236; source absent.  nsub contiguous groups of (fftN/2)/nsub bins, the last group also
 * takes the remainder, each divided by the total power. */
void orc_subband(const double *x, int fftN, double *sub, int nsub)
{
    const int half = fftN / 2, width = half / nsub;
    double total = 0.0;
    for (int g = 0; g < nsub; g++) {
        const int lo = g * width, hi = (g == nsub - 1) ? half : lo + width;
        double e = 0.0;
        for (int k = lo; k < hi; k++) e += x[2 * k] * x[2 * k] + x[2 * k + 1] * x[2 * k + 1];
        sub[g] = e;
        total += e;
    }
    for (int g = 0; g < nsub; g++) sub[g] /= total;
}

/* HTK HSigP Regress with head = tail = 0, simpleDiffs = 0 (main.c:259-260): for every frame and
 * column, sum_t t*(x[min(i+t,n)] - x[max(i-t,1)]) / (2 sum t^2), written `offset` columns on. */
void orc_regress(double *data, int vSize, long n, int step, int offset, int delwin)
{
    double sigmaT2 = 0.0;
    for (int t = 1; t <= delwin; t++) sigmaT2 += t * t;
    sigmaT2 *= 2.0;
    for (long i = 0; i < n; i++) {
        for (int j = 0; j < vSize; j++) {
            long back = i, forw = i;
            double sum = 0.0;
            for (int t = 1; t <= delwin; t++) {
                if (i - t >= 0) back--;
                if (i + t <= n - 1) forw++;
                sum += t * (data[forw * step + j] - data[back * step + j]);
            }
            data[i * step + j + offset] = sum / sigmaT2;
        }
    }
}

/* README.md:122 z-norm "in each dimension within the single audio"; the snapshot only holds the
 * commented call FZeroMean/FNormalize(&dp[1], step, rowNum, step) (main.c:264).  Definition used:
 * per column, subtract the mean over the n frames, divide by the population standard deviation. */
void orc_znorm(double *data, int ncol, long n, int step)
{
    for (int j = 0; j < ncol; j++) {
        double sum = 0.0;
        for (long i = 0; i < n; i++) sum += data[i * step + j];
        const double mean = sum / (double)n;
        double sq = 0.0;
        for (long i = 0; i < n; i++) {
            data[i * step + j] -= mean;
            sq += data[i * step + j] * data[i * step + j];
        }
        const double sd = sqrt(sq / (double)n);
        for (long i = 0; i < n; i++) data[i * step + j] /= sd;
    }
}

/* ---------------------------------------------------------------- per-file body */

/* HTK HSigP NormaliseLogEnergy: the largest value of the column becomes 1, values more than
 * silFloor dB below it are raised to that floor, differences are scaled by escale. */
void orc_normalise_log_energy(double *data, long n, int step, double silFloor, double escale)
{
    double max = data[0];
    for (long i = 1; i < n; i++)
        if (data[(size_t)i * step] > max) max = data[(size_t)i * step];
    const double min = max - (silFloor * log(10.0)) / 10.0;
    for (long i = 0; i < n; i++) {
        double *p = data + (size_t)i * step;
        if (*p < min) *p = min;
        *p = 1.0 - (max - *p) * escale;
    }
}

long orc_num_frames(long numSamples, int wlen, int inc) /* main.c:211 == trip count of :217 */
{
    if (numSamples < wlen) return 0;
    return (numSamples - (wlen - inc)) / inc;
}

int orc_feature_dim(const orc_config *c) /* main.c:142-144,161 */
{
    const int m = c->fbankFlag ? c->bankNum : c->MFCCNum;
    const int c0 = c->fbankFlag ? 0 : c->MFCC0thFlag;
    return m + c0 + c->energyFlag + c->zeroCrossingFlag + c->brightFlag + c->subBandEFlag +
           c->fftLength;
}

int orc_step(const orc_config *c) { return orc_feature_dim(c) * c->vecNum * c->regreOrder; }

int orc_extract(const orc_config *c, const int32_t *pcm, int numChannels, long numSamples,
                double *out)
{
    const int rows = numChannels == 2 ? 4 : numChannels; /* main.c:198-199 */
    if (c->vecNum > rows || c->vecNum < 1) return -2;    /* main.c:218-222 */
    const int M = c->fbankFlag ? c->bankNum : c->MFCCNum;
    const int c0 = c->fbankFlag ? 0 : c->MFCC0thFlag;
    const int D = orc_feature_dim(c), vSize = D * c->vecNum, step = orc_step(c);
    const long T = orc_num_frames(numSamples, c->wlen, c->inc);
    const long N = numSamples;

    orc_fbank fb;
    if (orc_fbank_init_ex(&fb, c->wlen, 1e7 / (double)c->sampleRate, c->bankNum, c->lowpassfre,
                          c->hipassfre, c->usePower, c->takeLogs, c->doubleFFT, c->warpAlpha,
                          c->warpLowCut, c->warpUpCut))
        return -1;
    double *d1 = (double *)malloc(sizeof(double) * (size_t)rows * (size_t)(N > 0 ? N : 1));
    double *ham = (double *)malloc(sizeof(double) * (size_t)c->wlen);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)c->wlen);
    double *x = (double *)malloc(sizeof(double) * (size_t)fb.fftN);
    double *fbank = (double *)malloc(sizeof(double) * (size_t)c->bankNum);
    double *d3 = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    double *sub = (double *)malloc(sizeof(double) * (size_t)(c->subBandEFlag + 1));
    orc_hamming(ham, c->wlen);

    for (int r = 0; r < numChannels; r++) /* main.c:200-203 */
        for (long i = 0; i < N; i++) d1[(size_t)r * N + i] = (double)pcm[(size_t)r * N + i];
    if (numChannels == 2) /* main.c:204-207 */
        for (long i = 0; i < N; i++) {
            d1[2 * (size_t)N + i] = 0.5 * (d1[i] + d1[(size_t)N + i]);
            d1[3 * (size_t)N + i] = d1[i] - d1[(size_t)N + i];
        }
    if (c->zeroMeanSigFlag) /* README.md:88; no code in the snapshot: per-row mean removal */
        for (int r = 0; r < rows; r++) {
            double sum = 0.0;
            for (long i = 0; i < N; i++) sum += d1[(size_t)r * N + i];
            const double mean = N > 0 ? sum / (double)N : 0.0;
            for (long i = 0; i < N; i++) d1[(size_t)r * N + i] -= mean;
        }
    for (int r = 0; r < rows; r++) orc_preemphasise(d1 + (size_t)r * N, N, c->preemphasise);

    size_t pos = 0;
    for (long j = 0; j <= N - c->wlen; j += c->inc) { /* main.c:217 */
        for (int v = 0; v < c->vecNum; v++) {         /* main.c:223 */
            for (int i = 0; i < c->wlen; i++) d2[i] = d1[(size_t)v * N + j + i];
            const double zc = orc_zcr(d2, c->wlen);
            for (int i = 0; i < c->wlen; i++) d2[i] *= ham[i];
            double te, te2;
            orc_wave2fbank(d2, fbank, &te, &te2, &fb, x);
            const double bright = orc_brightness(x, fb.fftN);
            if (c->subBandEFlag) orc_subband(x, fb.fftN, sub, c->subBandEFlag);
            if (c->fbankFlag)
                memcpy(d3, fbank, sizeof(double) * (size_t)M);
            else
                orc_fbank2mfcc(fbank, c->bankNum, d3, M);
            for (int q = 0; q < M; q++) out[pos++] = d3[q]; /* main.c:241-247 */
            if (c0) out[pos++] = orc_fbank2c0(fbank, c->bankNum);
            if (c->energyFlag) out[pos++] = log(te);
            if (c->zeroCrossingFlag) out[pos++] = zc;
            if (c->brightFlag) out[pos++] = bright;
            for (int q = 0; q < c->subBandEFlag; q++) out[pos++] = sub[q];
            for (int q = 1; q <= c->fftLength; q++)
                out[pos++] = sqrt(x[2 * q - 2] * x[2 * q - 2] + x[2 * q - 1] * x[2 * q - 1]);
        }
        for (int q = 0; q < vSize * (c->regreOrder - 1); q++) out[pos++] = 0.0; /* main.c:249 */
    }
    /* main.c:255 (commented out there): the column behind the first row's coefficients */
    if (c->normEnergy && T > 0) orc_normalise_log_energy(out + M, T, step, 50.0, 0.1);
    if (c->regreOrder >= 2) orc_regress(out, vSize, T, step, vSize, c->delwin);
    if (c->regreOrder >= 3) orc_regress(out + vSize, vSize, T, step, vSize, c->delwin);
    if (c->znormFlag && T > 0) orc_znorm(out, step, T, step);

    free(d1), free(ham), free(d2), free(x), free(fbank), free(d3), free(sub);
    orc_fbank_free(&fb);
    return 0;
}

static int32_t pcm_sample(const uint8_t *p, int bytesPerSample)
{
    if (bytesPerSample == 1) return (int32_t)p[0] - 128; /* 8-bit WAV PCM is unsigned, offset 128 */
    if (bytesPerSample == 2) return (int16_t)(p[0] | (p[1] << 8));
    if (bytesPerSample == 3) {
        int32_t v = p[0] | (p[1] << 8) | (p[2] << 16);
        return (v & 0x800000) ? v - 0x1000000 : v;
    }
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
                     ((uint32_t)p[3] << 24));
}

int orc_extract_pcm_bytes(const orc_config *c, const uint8_t *bytes, int numChannels,
                          int bytesPerSample, long numSamples, double *out)
{
    int32_t *pcm = (int32_t *)malloc(sizeof(int32_t) * (size_t)numChannels *
                                     (size_t)(numSamples > 0 ? numSamples : 1));
    if (!pcm) return -1;
    for (long i = 0; i < numSamples; i++)
        for (int ch = 0; ch < numChannels; ch++)
            pcm[(size_t)ch * numSamples + i] =
                pcm_sample(bytes + ((size_t)i * numChannels + ch) * bytesPerSample, bytesPerSample);
    const int rc = orc_extract(c, pcm, numChannels, numSamples, out);
    free(pcm);
    return rc;
}

int orc_extract_batch(const orc_config *c, const uint8_t *bytes, int numChannels,
                      int bytesPerSample, int nUtts, const int64_t *byteOff,
                      const int64_t *numSamples, const int64_t *outOff, double *out,
                      int numThreads)
{
    int used = 1;
#ifdef _OPENMP
    if (numThreads > 0) omp_set_num_threads(numThreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int u = 0; u < nUtts; u++)
        orc_extract_pcm_bytes(c, bytes + byteOff[u], numChannels, bytesPerSample,
                              (long)numSamples[u], out + outOff[u]);
    return used;
}

/* ---------------------------------------------------------------- containers */

static uint32_t rd32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int orc_wav_parse(const uint8_t *f, size_t len, int *channels, int *sampleRate,
                  int *bytesPerSample, long *numSamples, size_t *dataOffset)
{
    if (len < 12 || memcmp(f, "RIFF", 4) || memcmp(f + 8, "WAVE", 4)) return -1;
    size_t pos = 12;
    int haveFmt = 0, blockAlign = 0;
    while (pos + 8 <= len) {
        const uint32_t sz = rd32(f + pos + 4);
        if (!memcmp(f + pos, "fmt ", 4)) {
            if (sz < 16 || pos + 8 + 16 > len) return -2;
            const int tag = f[pos + 8] | (f[pos + 9] << 8);
            if (tag != 1 && tag != 0xFFFE) return -3;
            *channels = f[pos + 10] | (f[pos + 11] << 8);
            *sampleRate = (int)rd32(f + pos + 12);
            blockAlign = f[pos + 20] | (f[pos + 21] << 8);
            const int bits = f[pos + 22] | (f[pos + 23] << 8);
            *bytesPerSample = bits / 8;
            haveFmt = 1;
        } else if (!memcmp(f + pos, "data", 4)) {
            if (!haveFmt || blockAlign <= 0) return -4;
            size_t avail = len - (pos + 8), bytes = sz;
            if (bytes > avail) bytes = avail;
            *numSamples = (long)(bytes / (size_t)blockAlign);
            *dataOffset = pos + 8;
            return 0;
        }
        pos += 8 + (size_t)sz + (sz & 1);
    }
    return -5;
}

int orc_write_text(const char *path, const double *data, long n, int step, char saveType)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    for (long i = 1; i <= n; i++) { /* main.c:276-285 */
        fprintf(fp, saveType == 'f' ? "%f\t" : "%e\t", data[i - 1]);
        if (i % step == 0) fprintf(fp, "\n");
    }
    fclose(fp);
    return 0;
}
