/*
 * mfcc_amd -- MI355X (gfx950) feature extraction behind the reference's C-style interface.
 *
 * Plain C ABI: pointers, sizes and int status codes only.  Every function returns 0 on success
 * or a negative MFCC_E_* code; MFCCLastError() gives the message of the calling thread's last
 * failure.  Nothing here exits the process or waits for a key press (the reference's error path,
 * main.c:130-139,182,192-196,218-222,273, does both).
 *
 * The product never computes features on the CPU: every extraction entry point needs a HIP
 * device and fails with MFCC_E_DEVICE when none is usable.
 */
#ifndef MFCC_AMD_H
#define MFCC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFCC_MAX_PATH 600 /* maxBuffLength, main.c:13: line buffer of the list and of fileList */

enum {
    MFCC_OK = 0,
    MFCC_E_INI_OPEN = -1,    /* main.c:130-134 "Can't load '.ini'" */
    MFCC_E_INI_KEY = -2,     /* main.c:135-139 unknown key / malformed line */
    MFCC_E_OPEN = -3,        /* main.c:182 "open .wav failed", main.c:160 list open */
    MFCC_E_WAV = -4,         /* malformed or unsupported RIFF/WAVE */
    MFCC_E_SAMPLE_RATE = -5, /* main.c:192-196 "Sample rate not right" */
    MFCC_E_VECNUM = -6,      /* main.c:218-222 vecNum > rows derived from the channels */
    MFCC_E_WRITE = -7,       /* main.c:273 "open result.dat failed" */
    MFCC_E_CONFIG = -8,      /* option out of range for the kernels (sizes, flags) */
    MFCC_E_DEVICE = -9,      /* no usable HIP device / HIP runtime error */
    MFCC_E_ARG = -10,        /* NULL or inconsistent argument */
    MFCC_E_NOMEM = -11
};

enum { MFCC_SAVE_F = 0, MFCC_SAVE_E = 1, MFCC_SAVE_N = 2, MFCC_SAVE_B = 3 };
enum { MFCC_PREC_F32 = 0, MFCC_PREC_F64 = 1 };

/*
 * Option set.  Replaces `Config` (main.c:41-65) and carries the README-only keys
 * zeroMeanSigFlag (README.md:88), znormFlag (README.md:122), numThreads (README.md:134) and the
 * saveType values n / b (README.md:133).  `precision` has no counterpart in the reference (which
 * is double throughout): it selects the arithmetic type of the HIP kernels.
 */
typedef struct MFCCConfig {
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
    int subBandEFlag; /* a count, not a flag (main.c:163,246) */
    int fftLength;
    int regreOrder;
    int delwin;
    int zeroMeanSigFlag;
    int znormFlag;
    int saveType;   /* MFCC_SAVE_* ; ini letters f / e / n / b (main.c:76-80, README.md:133) */
    int numThreads; /* host threads used by MFCCWapperList for file reading and writing */
    int precision;  /* MFCC_PREC_* */
    char fileList[MFCC_MAX_PATH];
    int numDevices; /* GPUs MFCCWapperList spreads its batches over ([IO] numDevices): 0 = all visible;
                       -k = k device threads round-robin over the visible GPUs (several streams per GPU) */
    /* Options the reference carries but fixes in its one call of InitFBank (main.c:168) or leaves
     * commented out; MFCCConfigDefaults sets the reference's values (1, 1, 0, 1.0, 0, 0, 0).  ini keys
     * of the same names in [MFCC] (normEnergy in [Regression]). */
    int usePower;      /* 0: magnitude instead of power into the filterbank (mfcc.c:154-157) */
    int takeLogs;      /* 0: filterbank sums without floor and log (mfcc.c:171-176) */
    int doubleFFT;     /* 1: FFT of twice the usual length (mfcc.c:55-56) */
    double warpAlpha;  /* vocal tract length warping factor of the channel centres (mfcc.c:17-36,83-92) */
    double warpLowCut; /* its lower and upper cut-off frequencies in Hz */
    double warpUpCut;
    int normEnergy;    /* 1: NormaliseLogEnergy(column MFCCNum, 50 dB, 0.1) before the regression (main.c:255) */
} MFCCConfig;

/* Thread-local message of the last failed call ("" if none). */
const char *MFCCLastError(void);

/* Work counters of the process (the reference only prints progress lines, main.c:180-187,212,252,274):
 * what the extraction entry points have done since the process started or the last reset.
 * frames / pcmBytes / featureBytes: rows produced, PCM bytes consumed and feature bytes produced by
 * MFCCExtractDevice (every entry point extracts through it; a shard plan counts the rows it owns);
 * filesOk / filesFailed: files through MFCCWapper, MFCCWapperMem and MFCCWapperList;
 * listSeconds / recordingSeconds: wall time spent inside MFCCWapperList / MFCCExtractRecording, so
 * that frames / listSeconds is the end-to-end rate of a list run.  Thread-safe. */
typedef struct MFCCCounters {
    uint64_t extractCalls, frames, pcmBytes, featureBytes, filesOk, filesFailed;
    double listSeconds, recordingSeconds;
} MFCCCounters;
void MFCCCountersGet(MFCCCounters *out);
void MFCCCountersReset(void);

/* Releases any buffer this library handed out (feature matrices). */
void MFCCFree(void *p);

/* Fills *cfg with neutral defaults (all flags 0, regreOrder 1, numThreads 1, all GPUs, f64). */
int MFCCConfigDefaults(MFCCConfig *cfg);

/*
 * Reads config.ini.  Replaces ini_parse + handler (main.c:67-96,129).  Accepts the snapshot's
 * layout and the README's: saveType / fileList in [Frame] or [IO], trailing " ;" comments, section
 * lines ending "];".  Keys of the snapshot that are missing stay at their defaults.  Returns
 * MFCC_E_INI_OPEN, or MFCC_E_INI_KEY with *badLine (may be NULL) set to the first offending line.
 * Applies main.c:141-144: fbankFlag forces MFCCNum = bankNum and MFCC0thFlag = 0.
 */
int MFCCConfigFromIni(const char *iniPath, MFCCConfig *cfg, int *badLine);

/* Frame count of main.c:211 (C integer division), 0 when the file is shorter than one window. */
long MFCCNumFrames(long numSamples, int wlen, int inc);
/* Per-vec feature width (main.c:161 + MFCCNum) and row width `step` (main.c:254). */
int MFCCFeatureDim(const MFCCConfig *cfg);
int MFCCStep(const MFCCConfig *cfg);

/*
 * Per-file entry point.  Replaces the body of the list loop, main.c:178-289, which the README
 * (README.md:142) calls MFCCWapper.  Reads wavPath, extracts on the current HIP device and
 *   - writes outPath in cfg->saveType layout when outPath != NULL,
 *   - returns the [numFrames x step] row-major matrix in *features when features != NULL
 *     (release with MFCCFree).
 */
int MFCCWapper(const MFCCConfig *cfg, const char *wavPath, const char *outPath, double **features,
               long *numFrames, int *step);

/* Same as MFCCWapper on a WAV file image already in memory. */
int MFCCWapperMem(const MFCCConfig *cfg, const void *wavFile, size_t wavLen, const char *outPath,
                  double **features, long *numFrames, int *step);

/*
 * Whole-run entry point.  Replaces main() (main.c:98-300): reads the ini, then every
 * "src<TAB>dst" line of cfg.fileList until the first blank line (main.c:173-177).  A file that
 * fails is reported and skipped; the return value is the number of failed files (0 = all good)
 * or a negative code when the ini or the list cannot be read.  Files are batched onto the GPU.
 */
int MFCCWapperList(const char *iniPath);

/* Output layouts of main.c:272-287 (f, e) and cnpy.hpp:88-131,235-264 (n); b = raw little-endian
 * doubles, row-major, no header. */
int MFCCWriteFeatures(const char *path, const double *data, long numFrames, int step, int saveType);

/* One value as the text writers print it: "%f" (saveType f) or "%e" (anything else), main.c:276-285,
 * without the trailing tab.  buf needs at least 400 bytes; returns the length (>= 0) or an error. */
int MFCCFormatValue(double x, int saveType, char *buf, size_t cap);

/* RIFF/WAVE header walk (README.md:68; replaces loadWAVEFile, main.c:190-191). */
int MFCCWavInfo(const void *wavFile, size_t wavLen, int *channels, int *sampleRate,
                int *bytesPerSample, long *numSamples, size_t *dataOffset);

/* The one-time host tables without a device: what InitFBank (mfcc.c:39-122) and GenHamWindow
 * (main.c:171) compute, plus the two pass-band lines mfcc.c:74-75 print.  cf[bankNum+1],
 * loChan[fftN/2], loWt[fftN/2] (entry [k-1] = the reference's 1-based [k]), window[wlen]; any
 * pointer may be NULL. */
int MFCCHostTables(const MFCCConfig *cfg, int *fftN, int *klo, int *khi, double *cf, int *loChan,
                   double *loWt, double *window, char *passband, size_t cap);

/* ------------------------------------------------------------------------------------------
 * Device-resident batched path: PCM bytes already in HBM, features left in HBM.
 * ---------------------------------------------------------------------------------------- */

typedef struct MFCCContext MFCCContext; /* tables + device for one option set */
typedef struct MFCCPlan MFCCPlan;       /* one batch: utterance table, tiles, scratch */

/* Builds the host tables (InitFBank mfcc.c:39-122, GenHamWindow main.c:171, the DCT matrix of
 * mfcc.c:186-193) in double and uploads them to `device`. */
int MFCCContextCreate(const MFCCConfig *cfg, int device, MFCCContext **ctx);
void MFCCContextDestroy(MFCCContext *ctx);
/* The two lines InitFBank prints (mfcc.c:74-75), formatted into buf. */
int MFCCContextPassbandText(const MFCCContext *ctx, char *buf, size_t cap);
/* Copies the filterbank tables out (any pointer may be NULL): cf[bankNum+1], loChan[fftN/2],
 * loWt[fftN/2] with entry [k-1] holding the reference's 1-based [k]. */
int MFCCContextTables(const MFCCContext *ctx, int *fftN, int *klo, int *khi, double *cf,
                      int *loChan, double *loWt);

/*
 * Plans a batch of nUtts independent utterances that share one packed PCM buffer of interleaved
 * little-endian samples: utterance u starts byteOff[u] bytes into the buffer and has
 * numSamples[u] sample instants of `channels` x `bytesPerSample` bytes.  Its feature rows start
 * at row MFCCPlanRowOffset(plan, u) of the output matrix; rows are `step` wide.
 */
int MFCCPlanCreate(MFCCContext *ctx, int channels, int bytesPerSample, int nUtts,
                   const int64_t *byteOff, const int64_t *numSamples, MFCCPlan **plan);

/*
 * Plans frames [frameBegin, frameEnd) of ONE recording of totalSamples instants for a rank that
 * holds only part of it.  The rank must supply the PCM bytes of instants
 * [MFCCPlanSampleBegin, MFCCPlanSampleBegin + MFCCPlanSampleCount): its frames, the delta halo
 * of (regreOrder-1)*delwin frames on each interior side and one lead instant for pre-emphasis.
 * The output matrix holds MFCCPlanTotalRows rows; the rank's own frames start at row
 * MFCCPlanOwnedRowBegin.  chanSum2 (may be NULL unless zeroMeanSigFlag): twice the integer sum of
 * every derived row {L, R, mid, side} over the WHOLE recording, used for the zero-mean option.
 */
int MFCCPlanCreateShard(MFCCContext *ctx, int channels, int bytesPerSample, int64_t totalSamples,
                        int64_t frameBegin, int64_t frameEnd, const int64_t *chanSum2,
                        MFCCPlan **plan);
void MFCCPlanDestroy(MFCCPlan *plan);

/* The arithmetic of MFCCPlanCreateShard without a device (host-only, usable for planning a
 * distribution): which instants a rank must hold for frames [frameBegin, frameEnd) and where its
 * own rows sit in its output matrix.  lead = 1 when the first instant is only the pre-emphasis
 * predecessor of the second.  Any out-pointer may be NULL. */
int MFCCShardGeometry(const MFCCConfig *cfg, int64_t totalSamples, int64_t frameBegin,
                      int64_t frameEnd, int64_t *sampleBegin, int64_t *sampleCount,
                      int64_t *totalRows, int64_t *ownedRowBegin, int64_t *ownedRows, int *lead);

int64_t MFCCPlanTotalRows(const MFCCPlan *plan);
int64_t MFCCPlanRowOffset(const MFCCPlan *plan, int utt);
int64_t MFCCPlanNumFrames(const MFCCPlan *plan, int utt);
int64_t MFCCPlanSampleBegin(const MFCCPlan *plan);
int64_t MFCCPlanSampleCount(const MFCCPlan *plan);
int64_t MFCCPlanOwnedRowBegin(const MFCCPlan *plan);
int64_t MFCCPlanOwnedRows(const MFCCPlan *plan);

/*
 * Runs the planned batch on `stream` (a hipStream_t, NULL = default stream), asynchronously:
 * static features -> regression deltas -> (batch plans only) per-utterance z-norm.
 * pcmDev: device pointer to the packed PCM bytes (pcmBytes of them are readable).
 * outDev: device matrix [MFCCPlanTotalRows x step] of float (MFCC_PREC_F32) or double.
 * No allocation and no host synchronisation happens here except when zeroMeanSigFlag is set on a
 * batch plan (the row means are then reduced on the device first, still asynchronously).
 */
int MFCCExtractDevice(MFCCContext *ctx, MFCCPlan *plan, const void *pcmDev, size_t pcmBytes,
                      void *outDev, void *stream);

/* Shard plans: z-norm is split so the caller can all-reduce the statistics between the two.
 * MFCCShardColumnSums writes 2*step+1 doubles to sumsDev: sum x, sum x^2 per column over the
 * rank's own rows, then the row count.  MFCCShardNormalize applies (x - mean) / std given the
 * all-reduced vector of the same layout.  Both belong to the plan's latest MFCCExtractDevice call.
 * With znormFlag a shard plan's MFCCExtractDevice need not fill outDev: it keeps the static features
 * with the plan and leaves the sums of the rows they give as per-segment partial sums there;
 * MFCCShardColumnSums adds those up (outDev is not read), and MFCCShardNormalize forms the rows
 * from the kept static features and writes them normalised (outDev is written, not read): the
 * matrix travels to memory once.  Without znormFlag MFCCExtractDevice fills outDev as for a batch
 * plan. */
int MFCCShardColumnSums(MFCCContext *ctx, MFCCPlan *plan, const void *outDev, double *sumsDev,
                        void *stream);
int MFCCShardNormalize(MFCCContext *ctx, MFCCPlan *plan, void *outDev, const double *sumsDev,
                       void *stream);
/* Twice the integer sum of each derived row {L, R, mid, side} over instants
 * [sampleBegin, sampleBegin + count) of the bytes at pcmDev (instant 0 = first byte): 4 int64
 * written to sums2Dev.  Ranks add these up to feed MFCCPlanCreateShard's chanSum2. */
int MFCCChannelSums(MFCCContext *ctx, const void *pcmDev, int channels, int bytesPerSample,
                    int64_t sampleBegin, int64_t count, int64_t *sums2Dev, void *stream);

/*
 * One very long recording over several GPUs, driven from this one process: the per-file body of
 * main.c:198-266 (rows, pre-emphasis, frames, deltas, README's zero-mean and z-norm) with the frames
 * split into numShards contiguous ranges, one host thread and one GPU per range (devices[g], or the
 * visible GPUs in turn when devices is NULL; numShards 0 = one per visible GPU; shards may share a
 * GPU).  Each range recomputes the halo frames its deltas need; the only exchange is of the integer
 * channel sums and of the z-norm column sums (2*step+1 doubles), added by every shard in shard order.
 * pcm: host pointer to the interleaved PCM payload (numSamples instants); features: host matrix
 * [MFCCNumFrames x MFCCStep] of float (MFCC_PREC_F32) or double, filled in frame order.
 */
int MFCCExtractRecording(const MFCCConfig *cfg, const void *pcm, int channels, int bytesPerSample,
                         int64_t numSamples, int numShards, const int *devices, void *features,
                         int64_t *numFrames);

/* The exchange step of a frame-sharded recording between processes (one process per GPU, each with
 * its own RCCL communicator of the same clique: ncclCommInitRank on the caller's side): valuesDev
 * holds this rank's `count` partial sums -- doubles (the 2*step+1 z-norm column sums of
 * MFCCShardColumnSums) or, with isInt64, 64-bit integers (the 4 channel sums of MFCCChannelSums) --
 * and receives the totals: ncclAllGather into gatherDev (world * count * 8 bytes of device
 * memory), then one kernel adds the ranks' vectors in rank order, so every rank holds
 * bit-identical totals.  comm: an ncclComm_t.  Asynchronous on the stream.  The reference
 * normalises one file in one thread (commented calls main.c:263-266; README.md:122); this is the
 * step that lets one recording span GPUs.  MFCCExtractRecording uses it between its shards when
 * they sit on distinct GPUs.  librccl.so.1 is loaded on first use. */
int MFCCShardExchange(void *comm, int isInt64, void *valuesDev, int count, void *gatherDev, void *stream);

/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the column sums of
 * MFCCShardColumnSums -- sum x, sum x^2 per column, then the row count, 2*step+1 doubles to
 * sumsDev, the same double-precision terms added in a fixed order of its own -- of a [rows x step] matrix of float (MFCC_PREC_F32) or double in
 * device memory, on the current device, without a context or a plan and without any hipMalloc /
 * hipFree.  The per-chunk partial sums go to a 4 MiB device array inside the library (one copy
 * per GPU, held until the process ends): ceil(rows / 256) * 2 * step * 8 bytes must fit, larger
 * matrices are refused with MFCC_E_ARG.  All calls on one GPU share that array: calls that
 * overlap there (two streams or threads) race; calls on one stream are ordered.  The stream must
 * belong to the current device (MFCC_E_ARG otherwise).
 * Under stream capture: call MFCCColumnSumsNoAllocWarmup() once per device before the capture
 * begins.  It does eagerly what the first call on a device would otherwise do inside the capture
 * (the look-up of the array's device address and the loading of the kernels) and synchronises;
 * afterwards a call only enqueues kernels on the stream. */
int MFCCColumnSumsNoAlloc(int precision, const void *dataDev, int64_t rows, int step, double *sumsDev,
                          void *stream);
int MFCCColumnSumsNoAllocWarmup(void);

/* ------------------------------------------------------------------------------------------
 * The reference's lower-level API (mfcc.h:29-74), batched on the device: numFrames vectors back
 * to back in device memory, 0-based (the reference's 1-based Vector element [i] is element
 * [i-1] here), float or double as `precision` / the context says; asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------ */

/* FFT (mfcc.c:211-259) in place on numFrames vectors of n reals = n/2 complex points
 * (re, im interleaved); forward sign e^{+i theta}; invert != 0 divides by n/2. */
int MFCCFFTDevice(int precision, void *dataDev, int numFrames, int n, int invert, void *stream);
/* Realft (mfcc.c:262-291) in place on numFrames vectors of n reals: packed spectrum, element 0 =
 * Re X0 + Im X0 as the reference leaves it, element 1 = 0. */
int MFCCRealftDevice(int precision, void *dataDev, int numFrames, int n, void *stream);
/* Wave2FBank (mfcc.c:125-177): numFrames windowed frames of cfg.wlen samples -> numFrames x bankNum
 * log filterbank values (usePower = 1, floor 1.0) and, if teDev != NULL, the frame energies. */
int MFCCWave2FBankDevice(MFCCContext *ctx, const void *framesDev, int numFrames, void *fbankDev,
                         void *teDev, void *stream);
/* The same with explicit tables and every output of the reference's function: InitFBank's results
 * for any argument set come from MFCCFBankTables (host, double; doubleFFT mfcc.c:55-56 and the
 * alpha / warpLowCut / warpUpCut warping of mfcc.c:17-36,83-92 included; entry [k-1] = the reference's
 * [k]; loChan/loWt are uploaded by the caller), usePower / takeLogs as in FBankInfo (mfcc.c:154-157,171-176); te2Dev
 * (mfcc.c:164-168) and specDev (numFrames x fftN: the packed spectrum the reference leaves in info.x,
 * read by main.c:235-236) may be NULL. */
int MFCCFBankTables(int frameSize, double sampPeriod, int numChans, double lopass, double hipass,
                    int doubleFFT, double alpha, double warpLowCut, double warpUpCut,
                    int *fftN, int *klo, int *khi, double *fres, double *cf, int *loChan, double *loWt,
                    char *passband, size_t cap);
int MFCCWave2FBankTablesDevice(int precision, const void *framesDev, int numFrames, int frameSize,
                               int fftN, int klo, int khi, int numChans, const int *loChanDev,
                               const void *loWtDev, int usePower, int takeLogs, void *fbankDev,
                               void *teDev, void *te2Dev, void *specDev, void *stream);
/* FBank2MFCC / FBank2C0 for any channel count without a context: the cosines of mfcc.c:192 are
 * evaluated in the kernel.  numFrames x numChans -> numFrames x n (and numFrames C0 values if
 * c0Dev != NULL). */
int MFCCFBank2MFCCDirectDevice(int precision, const void *fbankDev, int numFrames, int numChans, int n,
                               void *mfccDev, void *c0Dev, void *stream);
/* FBank2MFCC (mfcc.c:180-195) and, if c0Dev != NULL, FBank2C0 (mfcc.c:197-208): numFrames x bankNum
 * -> numFrames x MFCCNum values (C0: numFrames values).  The context must not be in filterbank mode. */
int MFCCFBank2MFCCDevice(MFCCContext *ctx, const void *fbankDev, int numFrames, void *mfccDev,
                         void *c0Dev, void *stream);
/* Mel (mfcc.c:11-15): mel value of the 1-based FFT bin k. */
double MFCCMel(int k, double fres);

/* Elapsed milliseconds the dominant (static feature) kernel took in the most recent
 * MFCCExtractDevice on this plan, measured with HIP events on the launch stream; synchronises
 * on the stop event.  Negative when timing was not enabled with MFCCPlanEnableTiming. */
int MFCCPlanEnableTiming(MFCCPlan *plan, int on);
float MFCCPlanLastKernelMs(MFCCPlan *plan);

/* No counterpart in the reference.  Debug / tuning override of one plan heuristic for the plans
 * created afterwards in this process (measurement scripts, tests that must reach one kernel
 * variant): keys noLean, pplog, tileFrames, tileTables, tileBlocks, melPieces, noMelMatrix,
 * blocksPerCu, noUttFinish, noFinishReg, noFusedDelta, leanChunk, debug; key NULL resets all.  The
 * library reads no tuning from the environment. */
int MFCCDebugTuning(const char *key, int value);

/* Deterministic synthetic PCM used by the benchmark and the tests (integer-only generator, so the
 * bytes are identical on every machine): triangular-pdf noise from a 64-bit LCG seeded with
 * `seed`, optional integer-table tone of `toneHz` at -12 dBFS added; 16- or 24-bit samples.
 * Host code; writes numSamples * bytesPerSample bytes (one channel) to dst with stride
 * channels * bytesPerSample starting at channel `ch`. */
int MFCCSynthPcm(void *dst, int channels, int ch, int bytesPerSample, int64_t numSamples,
                 uint64_t seed, int sampleRate, int toneHz);
/* The same stream from sample firstSample on (numSamples of them, written from dst on): ranges of
 * one long recording can be synthesised independently. */
int MFCCSynthPcmAt(void *dst, int channels, int ch, int bytesPerSample, int64_t firstSample,
                   int64_t numSamples, uint64_t seed, int sampleRate, int toneHz);

#ifdef __cplusplus
}
#endif
#endif
