/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference extraction path.
 *
 * Nothing in the product (mfcc_amd/, include/, tools/) may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it, as the checker.
 *
 * Parity status: the filterbank set-up is pinned by the one known answer the reference holds
 * (README.md:33-34, the two pass-band lines printed by mfcc.c:74-75).  Everything else is
 * PARITY UNPINNED: the reference ships no test vectors, its example wav/fbank blobs are missing
 * (.MISSING_LARGE_BLOBS) and mfcc.c cannot be compiled here because hmath.h (BasicAudioToolBox,
 * not vendored, no pinned version) is absent.  The functions below restate mfcc.c / main.c
 * operation by operation (same arithmetic order, IEEE double, 0-based indices instead of the
 * reference's 1-based vectors) and, for the absent sigProcess/WAVE sources, the HTK HSigP
 * algorithms the README says they follow (README.md:143) or a documented definition.
 */
#ifndef MFCC_ORACLE_H
#define MFCC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Option set of main.c:41-65 plus the README-only keys (README.md:88,122). */
typedef struct {
    int sampleRate;
    double hipassfre;
    double lowpassfre;
    double preemphasise;
    int wlen;
    int inc;
    int vecNum;
    int fbankFlag;
    int bankNum;
    int MFCCNum;
    int MFCC0thFlag;
    int energyFlag;
    int zeroCrossingFlag;
    int brightFlag;
    int subBandEFlag;
    int fftLength;
    int regreOrder;
    int delwin;
    int zeroMeanSigFlag;
    int znormFlag;
    /* Options the reference carries but main.c leaves at fixed values or commented out: the
     * InitFBank arguments of main.c:168 (usePower = 1, takeLogs = 1, doubleFFT = 0, alpha = 1.0,
     * warp cut-offs 0) and the NormaliseLogEnergy call of main.c:255 (normEnergy = 0: as shipped). */
    int usePower;
    int takeLogs;
    int doubleFFT;
    double warpAlpha;
    double warpLowCut;
    double warpUpCut;
    int normEnergy;
} orc_config;

/* FBankInfo of mfcc.h:14-27 with 0-based tables: entry [k-1] holds the reference's [k]. */
typedef struct {
    int frameSize, numChans, fftN, klo, khi;
    double sampPeriod, fres, mlo, mhi;
    double *cf;    /* numChans+1 centre frequencies (mel) */
    int *loChan;   /* fftN/2 */
    double *loWt;  /* fftN/2 */
    int usePower, takeLogs; /* mfcc.c:52 */
} orc_fbank;

double orc_mel(int k, double fres);                                   /* mfcc.c:11-14 */
int orc_fbank_init(orc_fbank *fb, int frameSize, double sampPeriod, int numChans,
                   double lopass, double hipass);                     /* mfcc.c:39-122 */
/* All eleven arguments of InitFBank (mfcc.c:39-42); WarpFreq is mfcc.c:17-36. */
int orc_fbank_init_ex(orc_fbank *fb, int frameSize, double sampPeriod, int numChans,
                      double lopass, double hipass, int usePower, int takeLogs, int doubleFFT,
                      double alpha, double warpLowCut, double warpUpCut);
double orc_warp_freq(double fcl, double fcu, double freq, double minFreq, double maxFreq,
                     double alpha);                                   /* mfcc.c:17-36 */
void orc_fbank_free(orc_fbank *fb);
/* The two lines mfcc.c:74-75 print, formatted into buf. */
int orc_fbank_passband_text(const orc_fbank *fb, char *buf, size_t cap);

void orc_fft(double *s, int n, int invert);                           /* mfcc.c:211-259 */
void orc_realft(double *s, int n);                                    /* mfcc.c:262-291 */
void orc_wave2fbank(const double *s, double *fbank, double *te, double *te2,
                    const orc_fbank *fb, double *x);                  /* mfcc.c:125-177 */
void orc_fbank2mfcc(const double *fbank, int numChan, double *c, int n); /* mfcc.c:180-195 */
double orc_fbank2c0(const double *fbank, int numChan);                /* mfcc.c:197-208 */

void orc_preemphasise(double *s, long n, double k);                   /* call site main.c:208 */
void orc_hamming(double *w, int n);                                   /* call site main.c:171 */
double orc_zcr(const double *s, int n);                               /* call site main.c:227 */
double orc_brightness(const double *x, int fftN);                     /* call site main.c:235 */
void orc_subband(const double *x, int fftN, double *sub, int nsub);   /* call site main.c:236 */
void orc_regress(double *data, int vSize, long n, int step, int offset, int delwin); /* main.c:259-260 */
void orc_znorm(double *data, int ncol, long n, int step);             /* commented call main.c:264 */
/* Commented call main.c:255; its source is absent from the snapshot: HTK HSigP's definition
 * (the README names HTK as the model, README.md:143).  PARITY UNPINNED. */
void orc_normalise_log_energy(double *data, long n, int step, double silFloor, double escale);

long orc_num_frames(long numSamples, int wlen, int inc);              /* main.c:211,217 */
int orc_feature_dim(const orc_config *c);                             /* main.c:161,185: per-vec D */
int orc_step(const orc_config *c);                                    /* main.c:254 */

/* Whole per-file body main.c:197-266.  pcm: numChannels rows of numSamples raw integer samples,
 * row-major.  out: orc_num_frames() x orc_step() doubles.  Returns 0, or <0 on a config error. */
int orc_extract(const orc_config *c, const int32_t *pcm, int numChannels, long numSamples,
                double *out);

/* Same, starting from interleaved little-endian PCM bytes (16- or 24-bit). */
int orc_extract_pcm_bytes(const orc_config *c, const uint8_t *bytes, int numChannels,
                          int bytesPerSample, long numSamples, double *out);

/* Thread-per-file batch (README.md:134) over n utterances of one packed 16-bit mono/stereo or
 * 24-bit buffer: byteOff[u], numSamples[u] in, outOff[u] (in doubles) in.  Returns threads used. */
int orc_extract_batch(const orc_config *c, const uint8_t *bytes, int numChannels,
                      int bytesPerSample, int nUtts, const int64_t *byteOff,
                      const int64_t *numSamples, const int64_t *outOff, double *out,
                      int numThreads);

/* RIFF/WAVE header walk (README.md:68).  Returns 0 and fills the fields, <0 on malformed input. */
int orc_wav_parse(const uint8_t *file, size_t len, int *channels, int *sampleRate,
                  int *bytesPerSample, long *numSamples, size_t *dataOffset);

/* Text layouts of main.c:275-286 ('f' -> "%f\t", otherwise "%e\t"; "\n" after each `step`). */
int orc_write_text(const char *path, const double *data, long n, int step, char saveType);

#ifdef __cplusplus
}
#endif
#endif
