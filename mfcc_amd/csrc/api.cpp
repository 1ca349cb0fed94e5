// C ABI: contexts, batch / shard plans, device extraction, and the reference-shaped entry points
// MFCCWapper (per file, main.c:178-289) and MFCCWapperList (whole run, main.c:98-300).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <array>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rocprofiler-sdk-roctx/roctx.h>

#include "mfcc_host.h"

using namespace mfcc;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_error("HIP error %d (%s) at %s:%d", (int)e_, hipGetErrorString(e_), __FILE__, \
                      __LINE__);                                                            \
            return MFCC_E_DEVICE;                                                           \
        }                                                                                   \
    } while (0)

// Makes `device` current for the enclosing scope and restores the caller's device afterwards: a
// library call must not leave the calling thread on another GPU.
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1; // nothing to restore
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
// Named range in rocprofv3 --marker-trace timelines (host side: the enclosed calls are asynchronous).
struct TraceRange {
    explicit TraceRange(const char *name) { roctxRangePush(name); }
    ~TraceRange() { roctxRangePop(); }
};

#define DEVICE_SCOPE(dev)            \
    DeviceScope deviceScope_(dev);   \
    HIP_TRY(deviceScope_.err)

// ------------------------------------------------------------------------------------------

struct MFCCContext {
    MFCCConfig cfg;
    Tables tab;
    int device = 0;
    int M = 0, c0 = 0, D = 0, vSize = 0, step = 0;
    void *dWindow = nullptr, *dTw = nullptr, *dSplitTw = nullptr, *dLoWt = nullptr, *dDct = nullptr;
    int dctPitch = 0, dctElems = 0;
    int32_t *dLoChan = nullptr; // bin -> lower channel, for the batched Wave2FBank entry point
};

struct MFCCPlan {
    MFCCContext *ctx = nullptr;
    bool shard = false;
    int channels = 0, bps = 0, numUtts = 0;
    std::vector<UttDesc> utts;
    int64_t totalRows = 0;
    int numTiles = 0, numChunks = 0;
    int framesPerTile = 0, rowPitch = 0, rawOrZElems = 0, teamElems = 0, numRegBlocks = 0;
    int tabElems = 0, tabPitch = 0, winPad = 0, featElems = 0, fbkElems = 0, rowTail = 0;
    int blocksPerCu = 1, numCus = 256, pplog = 4;
    int activeTeams = 1;
    int deltaSubRows = 0, deltaRedOffset = 0; // fused delta kernel geometry (0: not used)
    size_t deltaLds = 0;
    double *dBlockPartial = nullptr;
    // whole-utterance finishing pass (deltas + z-norm from compact static rows), see finish.hip
    bool uttFinish = false;
    bool finStream = false;     // ... by the segment-streaming kernels (any utterance length) instead of the whole-utterance-in-LDS one
    int32_t *dSegs = nullptr, *dSegBegin = nullptr; // segments {utterance, first row}, first segment of every utterance
    double *dSegPartial = nullptr, *dSegStats = nullptr;
    int numSegs = 0;
    int strColBlock = 0, strColBlocks = 0, strChunkRows = 0, strWaveElems = 0, strRedOffset = 0, strFast = 0, strStage = 0;
    int regRows = kRegRows; // rows per block of the regression / fused delta kernels
    void *dStatics = nullptr; // totalRows x vSize
    int finSubRows = 0, finStaticElems = 0, finRedOffset = 0, finRegRows = 0;
    size_t finLds = 0;
    int mode = 0; // kernel mode: 0 general, 1 lean, 2 lean + matrix-core mel
    bool lean256 = false; // mode 2 run by the pipelined 256-point kernel
    bool wide1024 = false; // 24-bit stereo, 1024 complex points: run by k_wide1024
    int pipe64 = 0;        // double precision, run by k_pipe64: 8 (mono 16-bit, 256 complex points) or 10 (24-bit stereo, 1024)
    int leanWaves = 0;    // ... with this many waves per workgroup
    int leanChunk = 0;    // ... tiles per chunk override (0: heuristic, -1: fixed ranges)
    std::vector<float> melW;
    std::vector<double> melW64; // k_pipe64's matrix-core filterbank (256-point class)
    double *dMelW64 = nullptr;
    int melSlotSteps[4] = {0, 0, 0, 0};
    std::vector<int32_t> melMeta;
    int melSteps = 0;
    float *dMelW = nullptr;
    int32_t *dMelMeta = nullptr;
    unsigned *dWork = nullptr; // k_lean256's chunk counter and finished-wave counter (zero between launches)
    size_t ldsBytes = 0;
    UttDesc *dUtts = nullptr;
    int32_t *dTileUtt = nullptr, *dChunkUtt = nullptr, *dLaneSeg = nullptr, *dRegUtt = nullptr;
    TileDesc *dTiles = nullptr;
    void *dStageTw = nullptr;
    std::vector<double> stageTw;
    int maxSeg = 1, zPitch = 0, melPieces = 1;
    std::vector<int32_t> laneSeg;
    double *dPartial = nullptr, *dStats = nullptr;
    unsigned long long *dSums2 = nullptr;
    int64_t sampleBegin = 0, sampleCount = 0, ownedRowBegin = 0, ownedRows = 0;
    int64_t planEnd = 0; // one past the last PCM byte the plan owns
    bool timing = false, timed = false;
    hipEvent_t evStart = nullptr, evStop = nullptr;
};

template <typename T>
static int upload(const std::vector<double> &src, void **dst)
{
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = (T)src[i];
    const size_t bytes = std::max<size_t>(tmp.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(dst, bytes));
    if (!tmp.empty()) HIP_TRY(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return MFCC_OK;
}

static int upload_tab(int precision, const std::vector<double> &src, void **dst)
{
    return precision == MFCC_PREC_F64 ? upload<double>(src, dst) : upload<float>(src, dst);
}

extern "C" int MFCCContextCreate(const MFCCConfig *cfgIn, int device, MFCCContext **out)
{
    if (!cfgIn || !out) return set_error("NULL argument"), MFCC_E_ARG;
    *out = nullptr;
    MFCCConfig cfg = *cfgIn;
    if (cfg.fbankFlag) { /* main.c:142-144 */
        cfg.MFCCNum = cfg.bankNum;
        cfg.MFCC0thFlag = 0;
    }
    int rc = validate_config(cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error("no HIP device available (the library has no CPU path)"), MFCC_E_DEVICE;
    if (device < 0 || device >= ndev) return set_error("device %d out of range", device), MFCC_E_DEVICE;
    DEVICE_SCOPE(device);
    std::unique_ptr<MFCCContext, void (*)(MFCCContext *)> c(new MFCCContext, MFCCContextDestroy);
    c->cfg = cfg;
    c->device = device;
    rc = build_tables(cfg, c->tab);
    if (rc) return rc;
    if (c->tab.log2nc < 1 || c->tab.log2nc > kMaxLog2Nc)
        return set_error("fftN %d is outside the supported range", c->tab.fftN), MFCC_E_CONFIG;
    c->M = eff_mfccnum(cfg);
    c->c0 = eff_c0(cfg);
    c->D = MFCCFeatureDim(&cfg);
    c->vSize = c->D * cfg.vecNum;
    c->step = c->vSize * cfg.regreOrder;
    const int p = cfg.precision;
    {
        // Half the Hamming values: the half-size transform of the half-scaled frame makes the
        // real-FFT split (mfcc.c:274-287) come out unscaled without a halving of its own; the
        // kernel multiplies the frame energy by 4.  Scaling by a power of two is exact.
        std::vector<double> padded(c->tab.window);
        for (double &w : padded) w *= 0.5;
        padded.resize((size_t)c->tab.fftN + 8, 0.0);
        if ((rc = upload_tab(p, padded, &c->dWindow))) return rc;
    }
    if ((rc = upload_tab(p, c->tab.tw, &c->dTw))) return rc;
    if ((rc = upload_tab(p, c->tab.splitTw, &c->dSplitTw))) return rc;
    if ((rc = upload_tab(p, c->tab.loWt, &c->dLoWt))) return rc;
    {
        // j-major copy of the DCT table plus a row of ones (C0 = scaled plain sum, mfcc.c:197-208);
        // rows zero-filled to an odd pitch >= P + 3: lanes read different rows at the same k (an
        // odd pitch spreads them over the LDS banks) and up to 3 elements past P
        const int P = cfg.bankNum, M = c->M, JW = M + c->c0;
        c->dctPitch = (P + 3) | 1;
        c->dctElems = cfg.fbankFlag ? 0 : JW * c->dctPitch;
        std::vector<double> tr((size_t)std::max(c->dctElems, 1), 0.0);
        if (!cfg.fbankFlag)
            for (int j = 0; j < JW; ++j)
                for (int k = 0; k < P; ++k)
                    tr[(size_t)j * c->dctPitch + k] = j < M ? c->tab.dct[(size_t)k * M + j] : 1.0;
        if ((rc = upload_tab(p, tr, &c->dDct))) return rc;
    }
    HIP_TRY(hipMalloc((void **)&c->dLoChan, std::max<size_t>(c->tab.loChan.size(), 1) * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(c->dLoChan, c->tab.loChan.data(), c->tab.loChan.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    *out = c.release();
    return MFCC_OK;
}

extern "C" void MFCCContextDestroy(MFCCContext *c)
{
    if (!c) return;
    DeviceScope scope(c->device);
    (void)hipFree(c->dWindow);
    (void)hipFree(c->dTw);
    (void)hipFree(c->dSplitTw);
    (void)hipFree(c->dLoWt);
    (void)hipFree(c->dDct);
    (void)hipFree(c->dLoChan);
    delete c;
}

extern "C" int MFCCContextPassbandText(const MFCCContext *c, char *buf, size_t cap)
{
    if (!c || !buf) return set_error("NULL argument"), MFCC_E_ARG;
    snprintf(buf, cap, "FFT passband %d to %d out of 1 to %d\nMel passband %f to %f\n", c->tab.klo,
             c->tab.khi, c->tab.nc, c->tab.mlo, c->tab.mhi); /* mfcc.c:74-75 */
    return MFCC_OK;
}

extern "C" int MFCCContextTables(const MFCCContext *c, int *fftN, int *klo, int *khi, double *cf,
                                 int *loChan, double *loWt)
{
    if (!c) return set_error("NULL argument"), MFCC_E_ARG;
    if (fftN) *fftN = c->tab.fftN;
    if (klo) *klo = c->tab.klo;
    if (khi) *khi = c->tab.khi;
    if (cf) std::copy(c->tab.cf.begin(), c->tab.cf.end(), cf);
    if (loChan) std::copy(c->tab.loChan.begin(), c->tab.loChan.end(), loChan);
    if (loWt) std::copy(c->tab.loWt.begin(), c->tab.loWt.end(), loWt);
    return MFCC_OK;
}

// ------------------------------------------------------------------------------------------
// Debug / tuning overrides of the plan heuristics.  Process-wide, set through MFCCDebugTuning only
// (never read from the environment, so a caller's environment cannot change which kernel runs):
// measurement scripts and the tests that must reach a specific kernel variant use it.
// ------------------------------------------------------------------------------------------

namespace {
struct Tuning {
    int noLean = 0;        // never take the lean instantiations
    int pplog = 0;         // complex points per lane = 2^pplog (0: heuristic)
    int tileFrames = 0, tileTables = 0, tileBlocks = 0; // general kernel tile: frames, tables in LDS, workgroups per CU
    int melPieces = 0;     // parts a mel segment is cut into (0: heuristic)
    int noMelMatrix = 0;   // no matrix-core mel filterbank
    int leanWaves = 0;     // pipelined 256-point kernel: 8 = skip the 12-wave form
    int blocksPerCu = 0;   // resident workgroups per CU (0: heuristic)
    int noUttFinish = 0;   // deltas / z-norm by the row-block kernels even for short utterances
    int noFinishReg = 0;   // whole-utterance finishing kernel without its register variant
    int finish = 0;        // batch plans' deltas / z-norm: 0 heuristic, 1 whole-utterance-in-LDS kernel if it fits, 2 streaming kernel, 3 row blocks
    int streamRows = 0;    // rows per chunk of the streaming finishing kernel (0: heuristic)
    int noFusedDelta = 0;  // one regression pass per order
    int leanChunk = 0;     // tiles per chunk of the pipelined 256-point kernel (0: heuristic, -1: fixed ranges)
    int pinHost = 0;       // MFCCExtractRecording page-locks the caller's PCM and feature buffers for the call (hipHostRegister): 0 = from 64 MiB of PCM on, 1 = always, -1 = never
    int rccl = 0;          // MFCCExtractRecording's sums exchange: 0 = RCCL when the shards sit on distinct GPUs, 1 = always RCCL, -1 = always host memory
    int debug = 0;         // print the plan's choices to stderr
};
Tuning g_tune;
std::mutex g_tuneMu;
Tuning tuning()
{
    std::lock_guard<std::mutex> g(g_tuneMu);
    return g_tune;
}
} // namespace

extern "C" int MFCCDebugTuning(const char *key, int value)
{
    std::lock_guard<std::mutex> g(g_tuneMu);
    if (!key) {
        g_tune = Tuning();
        return MFCC_OK;
    }
    const struct { const char *name; int Tuning::*field; } keys[] = {
        {"noLean", &Tuning::noLean}, {"pplog", &Tuning::pplog}, {"tileFrames", &Tuning::tileFrames},
        {"tileTables", &Tuning::tileTables}, {"tileBlocks", &Tuning::tileBlocks}, {"melPieces", &Tuning::melPieces},
        {"noMelMatrix", &Tuning::noMelMatrix}, {"leanWaves", &Tuning::leanWaves}, {"blocksPerCu", &Tuning::blocksPerCu}, {"noUttFinish", &Tuning::noUttFinish},
        {"noFinishReg", &Tuning::noFinishReg}, {"finish", &Tuning::finish}, {"streamRows", &Tuning::streamRows}, {"noFusedDelta", &Tuning::noFusedDelta}, {"leanChunk", &Tuning::leanChunk},
        {"rccl", &Tuning::rccl}, {"pinHost", &Tuning::pinHost}, {"debug", &Tuning::debug}};
    for (const auto &k : keys)
        if (!strcmp(key, k.name)) {
            g_tune.*(k.field) = value;
            return MFCC_OK;
        }
    return set_error("MFCCDebugTuning: unknown key %s", key), MFCC_E_ARG;
}

// ------------------------------------------------------------------------------------------
// Process-wide work counters (the reference only prints progress, main.c:180-187,212,252,274):
// what the extraction entry points have been asked to do since the process started or since the
// last MFCCCountersReset.
// ------------------------------------------------------------------------------------------

namespace {
struct Counters {
    std::atomic<uint64_t> extractCalls{0}, frames{0}, pcmBytes{0}, featureBytes{0}, filesOk{0}, filesFailed{0};
    std::atomic<uint64_t> listNanos{0}, recordingNanos{0};
};
Counters g_count;
} // namespace

extern "C" void MFCCCountersGet(MFCCCounters *out)
{
    if (!out) return;
    out->extractCalls = g_count.extractCalls.load();
    out->frames = g_count.frames.load();
    out->pcmBytes = g_count.pcmBytes.load();
    out->featureBytes = g_count.featureBytes.load();
    out->filesOk = g_count.filesOk.load();
    out->filesFailed = g_count.filesFailed.load();
    out->listSeconds = (double)g_count.listNanos.load() * 1e-9;
    out->recordingSeconds = (double)g_count.recordingNanos.load() * 1e-9;
}

extern "C" void MFCCCountersReset(void)
{
    g_count.extractCalls = 0, g_count.frames = 0, g_count.pcmBytes = 0, g_count.featureBytes = 0;
    g_count.filesOk = 0, g_count.filesFailed = 0, g_count.listNanos = 0, g_count.recordingNanos = 0;
}

// ------------------------------------------------------------------------------------------
// plans
// ------------------------------------------------------------------------------------------

// k_pipe64's mel sums: every segment (the bins whose lower channel is s, s = 0 .. P) is cut into
// at most 10 parts of at most partBins bins; part i goes to lane i % G of a frame in round i / G
// (G = 16 lanes and at most 64 parts, or 64 lanes and at most 128 parts).  partBins is the value
// with the fewest bin steps per frame (rounds x partBins).  Table: 128 x {first bin, end bin, part or
// -1, 0}, then per channel {first part, parts} of the segment whose upper shares it takes (segment
// ch) and of the one whose lower shares it takes (segment ch + 1).  false: the filterbank has more
// segments than the table has parts.
static bool pipe64_part_table(MFCCPlan *p, int cls, bool debug)
{
    const MFCCContext *c = p->ctx;
    const int S1 = c->cfg.bankNum + 1, nc = c->tab.nc, G = pipe64_lanes(cls), maxParts = G == 16 ? 64 : 128;
    std::vector<int> segB(S1, 0), segE(S1, 0), partBegin(S1 + 2, 0);
    for (int k = 0; k < nc; ++k) {
        const int ch = c->tab.loChan[k];
        if (ch < 0 || ch >= S1) continue;
        if (segE[ch] == 0) segB[ch] = k;
        segE[ch] = k + 1;
    }
    int longest = 0;
    for (int sIdx = 0; sIdx < S1; ++sIdx) longest = std::max(longest, segE[sIdx] - segB[sIdx]);
    int partBins = 0, bestCost = 0;
    for (int cand = 1; cand <= std::max(nc, 1); ++cand) {
        int total = 0;
        for (int sIdx = 0; sIdx < S1; ++sIdx) total += std::max(1, (segE[sIdx] - segB[sIdx] + cand - 1) / cand);
        if ((longest + cand - 1) / cand <= 10 && total <= maxParts) {
            const int cost = ((total + G - 1) / G) * ((cand + 3) / 4 * 4 + 2); // bin steps (in fours) + the round's fixed price
            if (!partBins || cost < bestCost) partBins = cand, bestCost = cost;
        }
        if (cand >= longest) break;
    }
    if (!partBins) return false;
    p->laneSeg.assign((size_t)(128 + 64) * 4, 0);
    for (int i = 0; i < 128; ++i) p->laneSeg[4 * i + 2] = -1;
    int parts = 0;
    for (int sIdx = 0; sIdx < S1; ++sIdx) {
        const int len = segE[sIdx] - segB[sIdx], n = std::max(1, (len + partBins - 1) / partBins);
        partBegin[sIdx] = parts;
        for (int j = 0; j < n; ++j, ++parts) {
            p->laneSeg[4 * parts + 0] = segB[sIdx] + (int)((long)len * j / n);
            p->laneSeg[4 * parts + 1] = segB[sIdx] + (int)((long)len * (j + 1) / n);
            p->laneSeg[4 * parts + 2] = parts;
        }
    }
    partBegin[S1] = partBegin[S1 + 1] = parts;
    for (int ch = 0; ch < c->cfg.bankNum; ++ch) {
        int32_t *e = &p->laneSeg[(size_t)(128 + ch) * 4];
        e[0] = partBegin[ch], e[1] = partBegin[ch + 1] - partBegin[ch];
        e[2] = partBegin[ch + 1], e[3] = partBegin[ch + 2] - partBegin[ch + 1];
    }
    p->maxSeg = (parts + G - 1) / G;
    if (debug) fprintf(stderr, "mfcc plan: %d mel parts of at most %d bins\n", parts, partBins);
    return true;
}

// k_pipe64, 256-point class: the mel filterbank (mfcc.c:152-163) of a wave's four frames as
// 4 x 4 x 4 double-precision matrix-core products.  Channels are taken in quads; a quad's bins are
// those of the segments that reach its channels.  The instruction works on four independent blocks
// (lane / 4 % 4), so four quads are worked on at a time: a "slot".  Quads are dealt to (slot, block)
// by descending run length; a slot takes as many steps of 4 bins as its longest run needs.  Lane
// (j, blk, k) = (lane % 4, lane / 4 % 4, lane / 16) reads the power of frame j at bin
// r0[blk] + 4 t + k; the four frames' values sit 264 doubles apart, 8 LDS banks, so the 32 lanes
// of one LDS pass hit 32 different 8-byte bank pairs when the blocks' r0 are 0, 2, 4, 6 modulo 8
// in some order: the order with the fewest steps is taken.  Tables: weights [step][lane]; per lane
// and slot {element offset of its power operand inside the wave's work areas, channel of its result
// or -1}.  false: too many quads or steps.
static bool pipe64_mel_matrix(MFCCPlan *p, bool debug)
{
    const MFCCContext *c = p->ctx;
    const int P = c->cfg.bankNum, nc = c->tab.nc, NQ = (P + 3) / 4;
    if (NQ > 16) return false;
    std::vector<int> sb(P + 1, 0), se(P + 1, 0);
    for (int k = 0; k < nc; ++k) {
        const int ch = c->tab.loChan[k];
        if (ch < 0 || ch > P) continue;
        if (se[ch] == 0) sb[ch] = k;
        se[ch] = k + 1;
    }
    std::vector<int> q0(NQ, 0), q1(NQ, 0), order(NQ);
    for (int q = 0; q < NQ; ++q) {
        int k0 = nc, k1 = 0;
        for (int sIdx = 4 * q; sIdx <= std::min(4 * q + 4, P); ++sIdx)
            if (se[sIdx] > sb[sIdx]) k0 = std::min(k0, sb[sIdx]), k1 = std::max(k1, se[sIdx]);
        if (k1 < k0) k0 = k1 = 0;
        q0[q] = k0, q1[q] = k1;
        order[q] = q;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return q1[x] - q0[x] > q1[y] - q0[y]; });
    int quadOf[4][4], r0[4][4], total = 0;
    for (int sl = 0; sl < 4; ++sl) {
        p->melSlotSteps[sl] = 0;
        for (int b = 0; b < 4; ++b) quadOf[sl][b] = sl * 4 + b < NQ ? order[sl * 4 + b] : -1, r0[sl][b] = 2 * b;
        if (sl * 4 >= NQ) continue;
        int perm[4] = {0, 1, 2, 3}, best = 0, bestPerm[4] = {0, 1, 2, 3};
        auto start_of = [&](int b, int res) { // largest start <= the run's first bin with start = res (mod 8)
            const int q = quadOf[sl][b];
            if (q < 0) return res;
            return q0[q] - (((q0[q] - res) % 8 + 8) % 8);
        };
        do {
            int steps = 1;
            for (int b = 0; b < 4; ++b)
                if (quadOf[sl][b] >= 0) steps = std::max(steps, (q1[quadOf[sl][b]] - start_of(b, 2 * perm[b]) + 3) / 4);
            if (!best || steps < best) {
                best = steps;
                std::copy(perm, perm + 4, bestPerm);
            }
        } while (std::next_permutation(perm, perm + 4));
        p->melSlotSteps[sl] = best;
        for (int b = 0; b < 4; ++b) {
            int st = start_of(b, 2 * bestPerm[b]);
            // every read stays inside the wave's work areas: start + 4 steps <= 280 (the last frame's
            // values begin at element 800 of 1088)
            while (st + 4 * best > 280) st -= 8;
            if (st < -8) return false;
            r0[sl][b] = st;
        }
        total += best;
    }
    if (total > 96) return false;
    p->melSteps = total;
    p->melW64.assign((size_t)total * 64, 0.0);
    p->melMeta.assign(64 * 8, 0);
    int tbase = 0;
    for (int sl = 0; sl < 4; ++sl) {
        for (int ln = 0; ln < 64; ++ln) {
            const int j = ln & 3, blk = (ln >> 2) & 3, k = ln >> 4;
            const int q = quadOf[sl][blk], ch = q >= 0 ? 4 * q + j : -1;
            p->melMeta[8 * ln + 2 * sl] = 8 + j * 264 + r0[sl][blk] + k;
            p->melMeta[8 * ln + 2 * sl + 1] = (ch >= 0 && ch < P) ? ch : -1;
            if (ch < 0 || ch >= P) continue;
            for (int t = 0; t < p->melSlotSteps[sl]; ++t) {
                const int bin = r0[sl][blk] + 4 * t + k;
                if (bin < q0[q] || bin >= q1[q] || bin < 0 || bin >= nc) continue;
                const int sIdx = c->tab.loChan[bin]; // mfcc.c:158-163: loWt to channel sIdx - 1, the rest to channel sIdx (0-based)
                double w = 0.0;
                if (sIdx == ch + 1) w = c->tab.loWt[bin];
                if (sIdx == ch) w = 1.0 - c->tab.loWt[bin];
                p->melW64[(size_t)(tbase + t) * 64 + ln] = w;
            }
        }
        tbase += p->melSlotSteps[sl];
    }
    if (debug)
        fprintf(stderr, "mfcc plan: matrix-core mel (double), %d steps of 4 bins in slots of %d / %d / %d / %d\n", total, p->melSlotSteps[0],
                p->melSlotSteps[1], p->melSlotSteps[2], p->melSlotSteps[3]);
    return true;
}

// LDS geometry of the static-feature kernel for this plan: frames per tile, whether the window /
// twiddle / split tables get an LDS copy, and how many workgroups fit a CU.
static int choose_tile(MFCCPlan *p, bool allowWide = true)
{
    const MFCCContext *c = p->ctx;
    const Tuning tune = tuning();
    const int V = c->cfg.vecNum, log2nc = c->tab.log2nc, prec = c->cfg.precision;
    const int nc = c->tab.nc, P = c->cfg.bankNum;
    // complex points per lane = 2^pplog: for the general kernel 8 points (radix-8 stages) measured
    // fastest on MI355X where that variant exists, 16 otherwise
    // Lean candidates (see k_front) run 16 points per lane at two workgroups per CU: measured
    // slightly ahead of 8 points at three (6.46 vs 6.63 ms on 10,000 ten-second utterances).
    // (magnitude instead of power and sums without logs: the general kernel only)
    const bool plainFbank = c->cfg.usePower && c->cfg.takeLogs;
    const bool leanCand = plainFbank && prec == MFCC_PREC_F32 && p->channels == 1 && p->bps == 2 && V == 1 &&
                          c->cfg.inc % 2 == 0 && !c->cfg.zeroCrossingFlag && !c->cfg.brightFlag &&
                          !c->cfg.subBandEFlag && !c->cfg.fftLength && !c->cfg.zeroMeanSigFlag &&
                          !tune.noLean;
    // 24-bit stereo with 1024 complex points: k_wide1024 (16 points per lane, wave-sized teams)
    const bool wideCand = plainFbank && allowWide && prec == MFCC_PREC_F32 && !c->cfg.zeroMeanSigFlag && !c->cfg.fftLength &&
                          p->totalRows < 0x7fffffffLL && !tune.noLean &&
                          wide1024_applies(log2nc, p->channels, p->bps, c->cfg.wlen) && c->cfg.bankNum <= 64;
    // double precision, one of the two benchmark input classes: k_pipe64 (raw PCM bytes in LDS, wave-sized teams)
    const int pipeCls = (plainFbank && allowWide && prec == MFCC_PREC_F64 && !c->cfg.zeroMeanSigFlag && !c->cfg.fftLength &&
                         p->totalRows < 0x7fffffffLL && !tune.noLean && c->cfg.bankNum <= 64 &&
                         c->cfg.subBandEFlag <= nc / 16)
                            ? pipe64_class(log2nc, p->channels, p->bps, c->cfg.wlen, c->cfg.inc)
                            : 0;
    int pplog = (front_variant_exists(log2nc, 3) && !leanCand && !wideCand && !pipeCls) ? 3 : 4;
    // (double precision, 256 complex points: 16 points per lane with 8-frame tiles measured 6 % ahead
    // of 8 points with 2-frame tiles, 17.9 vs 19.0 ms on 10,000 ten-second utterances)
    if (prec == MFCC_PREC_F64 && log2nc == 8) pplog = 4;
    if (tune.pplog > 0 && front_variant_exists(log2nc, tune.pplog) && !wideCand && !pipeCls) pplog = tune.pplog;
    p->pplog = pplog;
    const int teams = front_teams(log2nc, pplog);
    const int groups = front_groups(log2nc, pplog) / teams; // frames in flight per team
    const int lanes = front_lanes(log2nc, pplog);
    // work area of one frame: nc complex points with one pad point per 16, or the nc (t1, ek-t1)
    // pairs, the 2 (P+1) segment sums and the 2 S sub-band sums behind them, whichever is larger
    // Mel segments longer than a few bins are cut into up to 4 parts when the frame has lanes to
    // spare (one round must still hold every part): the parts are summed by different lanes.
    int pieces = 1;
    while (pieces < 4 && (pieces + 1) * (c->cfg.bankNum + 1) <= lanes) ++pieces;
    if (tune.melPieces > 0) pieces = std::max(1, std::min(4, tune.melPieces));
    p->melPieces = pieces;
    int zPitch = (std::max(2 * (nc + nc / 16), 2 * nc + std::max(2 * pieces * (c->cfg.bankNum + 1), 16) + 2 * c->cfg.subBandEFlag) + 3) & ~3;
    p->zPitch = zPitch;
    {
        // Per-stage twiddle blocks, laid out as the kernel's fft_stages reads them: for a stage
        // with sub-transform size NS and radix R (R = 2^min(rbmax, remaining bits)), entries
        // [r-1][m] = exp(-2 pi i r m / (NS R)), m < NS; the first stage (NS = 1) has none.
        const int rbmax = std::min(pplog, 4);
        p->stageTw.clear();
        for (int logns = 0; logns < log2nc;) {
            const int rb = std::min(rbmax, log2nc - logns), R = 1 << rb, NS = 1 << logns;
            if (logns > 0)
                for (int r = 1; r < R; ++r)
                    for (int m = 0; m < NS; ++m) {
                        const double ang = 2.0 * M_PI * (double)r * (double)m / ((double)NS * R);
                        p->stageTw.push_back(cos(ang));
                        p->stageTw.push_back(-sin(ang));
                    }
            logns += rb;
        }
        p->stageTw.resize(2 * (size_t)nc + 8, 0.0); // the kernel stages 2 nc elements into LDS
    }
    {
        // Mel segments: segment s (0..P) = the bins whose lower channel is s (contiguous, because
        // loChan never decreases over the pass band); a segment without bins still gets a slot, so
        // that its sums are written as zeros.
        const int S1 = c->cfg.bankNum + 1;
        std::vector<int> segB(S1, 0), segE(S1, 0);
        for (int k = 0; k < nc; ++k) {
            const int ch = c->tab.loChan[k];
            if (ch < 0 || ch >= S1) continue;
            if (segE[ch] == 0) segB[ch] = k;
            segE[ch] = k + 1;
        }
        // part j of segment s is entry j * S1 + s (the slot its sums are stored in)
        const int P1 = S1 * pieces;
        std::vector<int> sb(P1, 0), se(P1, 0);
        for (int sIdx = 0; sIdx < S1; ++sIdx) {
            const int len = segE[sIdx] - segB[sIdx];
            for (int j = 0; j < pieces; ++j) {
                sb[j * S1 + sIdx] = segB[sIdx] + (int)((long)len * j / pieces);
                se[j * S1 + sIdx] = segB[sIdx] + (int)((long)len * (j + 1) / pieces);
            }
        }
        // Every round each lane sums one segment, 4 bins per step; the lanes of a frame read
        // 8-byte entries from 32 LDS banks, so segments of one round whose first bins share a bank
        // collide on every step.  Longest first, each segment goes to the round where it adds the
        // fewest LDS cycles (steps x worst bank multiplicity), one spare round allowed.
        const int minRounds = (P1 + lanes - 1) / lanes;
        const int rounds = minRounds + 1;
        std::vector<int> order(P1);
        for (int i = 0; i < P1; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(),
                         [&](int x, int y) { return se[x] - sb[x] > se[y] - sb[y]; });
        std::vector<std::vector<int>> inRound(rounds);
        auto roundCost = [&](const std::vector<int> &segs) {
            int cost = 0;
            for (int step = 0;; ++step) {
                int mult[32] = {0}, worst = 0;
                for (int sg : segs) {
                    const int len = se[sg] - sb[sg];
                    if (len > 4 * step) worst = std::max(worst, ++mult[(sb[sg] + 4 * step) & 31]);
                }
                if (!worst) break;
                cost += 4 * worst;
            }
            return cost + 6; // fixed price of a round: its table read, its two stores
        };
        for (int sg : order) {
            int best = -1, bestDelta = 0;
            for (int r = 0; r < rounds; ++r) {
                if ((int)inRound[r].size() >= lanes) continue;
                const int before = inRound[r].empty() ? 0 : roundCost(inRound[r]);
                inRound[r].push_back(sg);
                const int delta = roundCost(inRound[r]) - before;
                inRound[r].pop_back();
                if (best < 0 || delta < bestDelta) best = r, bestDelta = delta;
            }
            inRound[best].push_back(sg);
        }
        int used = 0;
        for (int r = 0; r < rounds; ++r)
            if (!inRound[r].empty()) used = r + 1;
        p->maxSeg = std::max(used, 1);
        p->laneSeg.assign((size_t)lanes * p->maxSeg * 4, 0);
        for (size_t i = 0; i < (size_t)lanes * p->maxSeg; ++i) p->laneSeg[4 * i + 2] = -1;
        for (int r = 0; r < p->maxSeg; ++r)
            for (size_t ln = 0; ln < inRound[r].size(); ++ln) {
                const int sg = inRound[r][ln];
                int32_t *e = &p->laneSeg[(ln * p->maxSeg + r) * 4];
                e[0] = sb[sg];
                e[1] = se[sg];
                e[2] = sg;
            }
        if (tune.debug) {
            int total = 0;
            for (int r = 0; r < p->maxSeg; ++r) total += roundCost(inRound[r]);
            fprintf(stderr, "mfcc plan: mel rounds %d, modelled LDS cycles per frame pair %d\n", p->maxSeg, total);
        }
    }
    // Matrix-core mel filterbank (kernel mode 2): 16 blocks of 4 lanes, each with 4 consecutive
    // channels ("quad") and a contiguous run of the bins that touch them.  Possible when a frame fills at most a quarter wave
    // (256..1024 complex points at 16 points per lane) and every run fits kMelSteps.
    p->melSteps = 0;
    int leanSteps = 0; // the same filterbank placed for k_lean256's 8-byte reads
    std::vector<float> leanW;
    std::vector<int32_t> leanMeta;
    if (leanCand && pplog == 4 && log2nc >= 8 && log2nc <= 10 && !tune.noMelMatrix) {
        const int P1 = P + 1, NQ = (P + 3) / 4;
        std::vector<int> sb(P1, 0), se(P1, 0);
        for (int k = 0; k < nc; ++k) {
            const int ch = c->tab.loChan[k];
            if (ch < 0 || ch >= P1) continue;
            if (se[ch] == 0) sb[ch] = k;
            se[ch] = k + 1;
        }
        // Runs: quad q's bins are those of segments 4q .. 4q+4 (they reach channels 4q .. 4q+3).
        // A run may be cut into up to 4 equal parts on neighbouring blocks; the smallest part
        // length that needs at most 16 blocks wins.
        struct Piece { int quad, k0, k1, pos, len; }; // pos / len: place in its quad's chain of blocks
        std::vector<int> q0(NQ, 0), q1(NQ, 0);
        for (int q = 0; q < NQ; ++q) {
            int k0 = nc, k1 = 0;
            for (int sIdx = 4 * q; sIdx <= std::min(4 * q + 4, P); ++sIdx)
                if (se[sIdx] > sb[sIdx]) k0 = std::min(k0, sb[sIdx]), k1 = std::max(k1, se[sIdx]);
            if (k1 < k0) k0 = k1 = 0;
            q0[q] = k0, q1[q] = k1;
        }
        // A chain must stay inside one row of 4 blocks (16 lanes), so that its partial sums can be
        // added by row-wise lane shifts: chains are packed into the 4 rows longest first.
        int part = 0;
        std::vector<int> chainStart(NQ, 0); // first block of quad q's chain
        for (int cand = 1; cand <= nc && !part && NQ <= 16; ++cand) {
            std::vector<int> order(NQ), len(NQ);
            bool ok = true;
            for (int q = 0; q < NQ; ++q) {
                len[q] = std::max(1, (q1[q] - q0[q] + cand - 1) / cand);
                ok = ok && len[q] <= 4;
                order[q] = q;
            }
            if (!ok) continue;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len[x] > len[y]; });
            int fill[4] = {0, 0, 0, 0};
            for (int q : order) {
                int row = -1;
                for (int r = 0; r < 4 && row < 0; ++r)
                    if (fill[r] + len[q] <= 4) row = r;
                if (row < 0) {
                    ok = false;
                    break;
                }
                chainStart[q] = 4 * row + fill[row];
                fill[row] += len[q];
            }
            if (ok) part = cand;
        }
        // pieces[b]: what block b does (quad < 0: nothing)
        std::vector<Piece> pieces(part ? 16 : 0, Piece{-1, 0, 0, 0, 1});
        for (int q = 0; q < NQ && part; ++q) {
            const int len = q1[q] - q0[q], n = std::max(1, (len + part - 1) / part);
            for (int i = 0; i < n; ++i) // equal parts, the first ones one bin longer
                pieces[chainStart[q] + i] = {q, q0[q] + (int)((long)len * i / n), q0[q] + (int)((long)len * (i + 1) / n), i, n};
        }
        // Every block reads `steps` consecutive bins from r0 <= k0 (r0 + steps >= k1, inside the
        // frame).  Rule 0 (k_front: single-word reads, 32 banks, frame slots 8 banks apart): the 32
        // lanes of a half wave (8 blocks x 4 frame slots) hit 32 different LDS banks when the 8
        // blocks' r0 differ modulo 8.  Rule 1 (k_lean256: 8-byte reads of two consecutive bins, 64
        // banks, frame slots 16 banks apart): r0 even and the 8 blocks' r0 / 2 different modulo 8.
        // Either way the smallest `steps` for which such starts exist wins (first fit, blocks with
        // the fewest possible starts first); rule 1 falls back to even starts with conflicts.
        int longest = 0;
        for (const Piece &pc : pieces) longest = std::max(longest, pc.k1 - pc.k0);
        auto place = [&](int rule, bool conflictFree, std::vector<int> &r0) {
            r0.assign(pieces.size(), 0);
            const int unit = rule ? 2 : 1;
            for (int trial = std::max(longest, 1); trial <= std::min(kMelSteps, nc) && part; ++trial) {
                if (rule && (trial & 1)) continue;
                std::vector<size_t> order(pieces.size());
                for (size_t i = 0; i < order.size(); ++i) order[i] = i;
                auto choices = [&](size_t b) {
                    if (pieces[b].quad < 0) return nc; // an idle block may start anywhere: placed last
                    return std::min(pieces[b].k0, nc - trial) - std::max(0, pieces[b].k1 - trial);
                };
                std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return choices(x) < choices(y); });
                char used[2][8] = {{0}};
                bool ok = true;
                for (size_t b : order) {
                    const bool idle = pieces[b].quad < 0;
                    const int lo = idle ? 0 : std::max(0, pieces[b].k1 - trial), hi = idle ? std::min(7 * unit, nc - trial) : std::min(pieces[b].k0, nc - trial);
                    int pick = -1;
                    for (int r = hi; r >= lo && pick < 0; --r)
                        if (r % unit == 0 && (!conflictFree || !used[b / 8][(r / unit) & 7])) pick = r;
                    if (pick < 0) {
                        ok = false;
                        break;
                    }
                    used[b / 8][(pick / unit) & 7] = 1;
                    r0[b] = pick;
                }
                if (ok) return trial;
            }
            return 0;
        };
        // tables of one placement: weights [step][lane], per-lane {r0, chain flags, channel, 0}
        auto build = [&](int steps, const std::vector<int> &r0, std::vector<float> &W, std::vector<int32_t> &meta) {
            W.assign((size_t)steps * 64, 0.0f);
            meta.assign(64 * 4, 0);
            for (int ln = 0; ln < 64; ++ln) meta[4 * ln + 2] = -1;
            for (size_t b = 0; b < pieces.size(); ++b) {
                const Piece &pc = pieces[b];
                const bool owner = pc.pos == pc.len - 1;
                for (int j = 0; j < 4; ++j) {
                    const int ln = 4 * (int)b + j, ch = 4 * pc.quad + j;
                    meta[4 * ln + 0] = r0[b];
                    if (pc.quad < 0) continue;
                    meta[4 * ln + 1] = (pc.pos >= 1 ? 1 : 0) | (owner ? 2 : 0) | (pc.pos >= 2 ? 4 : 0) | (pc.pos >= 3 ? 8 : 0);
                    meta[4 * ln + 2] = (owner && ch < P) ? ch : -1;
                    if (ch >= P) continue;
                    for (int t = 0; t < steps; ++t) {
                        const int k = r0[b] + t;
                        if (k < pc.k0 || k >= pc.k1) continue;
                        const int sIdx = c->tab.loChan[k]; // mfcc.c:158-163: loWt to channel s, the rest to s+1
                        double w = 0.0;
                        if (sIdx == ch + 1) w = c->tab.loWt[k];
                        if (sIdx == ch) w = 1.0 - c->tab.loWt[k];
                        W[(size_t)t * 64 + ln] = (float)w;
                    }
                }
            }
        };
        std::vector<int> r0;
        const int steps = place(0, true, r0);
        if (steps >= 1) {
            p->melSteps = steps;
            build(steps, r0, p->melW, p->melMeta);
            if (log2nc == 8) {
                std::vector<int> r1;
                int stepsL = place(1, true, r1);
                bool clean = stepsL >= 1;
                if (!clean) stepsL = place(1, false, r1);
                if (stepsL >= 1) {
                    leanSteps = stepsL;
                    build(stepsL, r1, leanW, leanMeta);
                }
                if (tune.debug) fprintf(stderr, "mfcc plan: matrix-core mel with 8-byte reads, %d steps%s\n", stepsL, clean ? "" : " (bank conflicts)");
            }
            // frame slots 8 banks apart: work-area pitch = 8 (mod 32) elements
            zPitch = ((zPitch + 23) & ~31) + 8;
            p->zPitch = zPitch;
            if (tune.debug) fprintf(stderr, "mfcc plan: matrix-core mel, %d steps (longest run %d)\n", steps, longest);
        }
    }
    const size_t el = prec == MFCC_PREC_F64 ? 8 : 4;
    const int bpi = p->channels * p->bps;
    const int base = (groups + V - 1) / V;
    const size_t kLds = 160 * 1024;
    const int rowTail = c->tab.fftN - c->cfg.wlen + 2; // zero-filled, so frame loads need no bounds test
    p->rowTail = rowTail;
    const int wavesCap = front_waves_limit(prec, log2nc, pplog); // workgroups per CU the registers allow
    p->winPad = (c->tab.fftN + 2 + 3) & ~3; // window values, then zeros up to fftN + 2
    const size_t tabElems = (((size_t)p->winPad + 2 * (size_t)nc + (size_t)std::max(nc, 2) + (size_t)nc +
                              4 * (size_t)lanes * p->maxSeg * 4 / el + (size_t)c->dctElems) + 3) & ~(size_t)3;
    int teamEl = 0, fbkEl = 0, useTeams = teams;
    auto geometry = [&](int ft, bool tabLds, int &rowPitch, int &rawOrZ) {
        const long ninst = (long)(ft - 1) * c->cfg.inc + c->cfg.wlen + 1;
        rowPitch = (int)((ninst + rowTail + 3) & ~3L);
        const size_t rawBytes = (size_t)ninst * bpi + 48;
        size_t n = std::max((size_t)groups * zPitch, (rawBytes + el - 1) / el);
        rawOrZ = (int)((n + 3) & ~(size_t)3);
        // staging area: the tile's rows plus one 16-byte vector of slack for the alignment shift
        const size_t feat = (((size_t)ft * c->vSize + 3) & ~(size_t)3) + 4;
        fbkEl = c->cfg.fbankFlag ? 0 : (int)(((size_t)ft * V * P + 3) & ~(size_t)3);
        teamEl = (int)((size_t)V * rowPitch + rawOrZ + feat + fbkEl + 16 + (teams > 1 ? 64 : 256)); // + sum slots + prefetch sink
        const size_t elems = (tabLds ? tabElems : 0) + (size_t)useTeams * teamEl;
        return elems * el;
    };
    // Prefer the occupancy the registers allow, then LDS tables, then longer tiles.
    int bestFt = 0, bestBlocks = 0;
    bool bestTab = false;
    for (int blocks = wavesCap; blocks >= 1 && !bestFt; --blocks)
        for (int tab = 1; tab >= 0 && !bestFt; --tab)
            for (int m = (teams > 1 ? 4 : 2); m >= 1 && !bestFt; --m) {
                int rp, rz;
                // Measured on MI355X: three 54,208-byte workgroups are not co-resident on a CU
                // while three 49,072-byte ones are, so budget in 4 KiB steps.
                const size_t need = (geometry(base * m, tab != 0, rp, rz) + 4095) & ~(size_t)4095;
                if (need * blocks <= kLds) {
                    bestFt = base * m;
                    bestTab = tab != 0;
                    bestBlocks = blocks;
                }
            }
    // Last resort: one workgroup per CU, shorter tiles, then fewer of the workgroup's teams at work.
    for (int at = teams; at >= 1 && !bestFt; at /= 2) {
        useTeams = at;
        for (int ft = base; ft >= 1 && !bestFt; --ft) {
            int rp, rz;
            if (geometry(ft, false, rp, rz) <= kLds) {
                bestFt = ft;
                bestBlocks = 1;
            }
        }
    }
    if (!bestFt) useTeams = teams;
    if (tune.tileFrames > 0 && tune.tileBlocks > 0) {
        bestFt = tune.tileFrames;
        bestTab = tune.tileTables != 0;
        bestBlocks = tune.tileBlocks;
    }
    if (!bestFt)
        return set_error("wlen %d / fftN %d needs more than 160 KiB of LDS per workgroup",
                         c->cfg.wlen, c->tab.fftN),
               MFCC_E_CONFIG;
    p->framesPerTile = bestFt;
    p->tabElems = bestTab ? (int)tabElems : 0;
    p->blocksPerCu = bestBlocks;
    p->ldsBytes = geometry(bestFt, bestTab, p->rowPitch, p->rawOrZElems);
    p->teamElems = teamEl;
    p->activeTeams = useTeams;
    p->fbkElems = fbkEl;
    p->featElems = (int)((((size_t)bestFt * c->vSize + 3) & ~(size_t)3) + 4);
    // Lean instantiation: mono 16-bit input, one row, even frame shift, none of the per-frame extras.
    p->mode = (leanCand && (pplog == 3 || pplog == 4) && bestTab) ? (p->melSteps > 0 ? 2 : 1) : 0;
    p->leanChunk = tune.leanChunk;
    p->wide1024 = false;
    p->pipe64 = 0;
    if (pipeCls) {
        // a wave's LDS region: the tile's raw bytes, then one work area per frame in flight; in front
        // of the regions the filterbank weights and (256-point class) window, twiddles, split factors
        // and cepstral table
        const int rawEl = pipe64_raw_bytes(pipeCls, c->cfg.wlen, c->cfg.inc) / 8;
        const int gt = pipe64_frames_per_tile(pipeCls);
        const int teamL = rawEl + gt * pipe64_area_elems(pipeCls);
        p->melSteps = 0;
        if (!(pipeCls == 8 ? pipe64_mel_matrix(p, tune.debug != 0) : pipe64_part_table(p, pipeCls, tune.debug != 0))) return choose_tile(p, false);
        int tabL = pipe64_tab_elems(pipeCls, c->cfg.wlen, c->dctElems, p->melSteps);
        const bool fits12 = ((size_t)tabL + (size_t)12 * teamL) * sizeof(double) <= kLds;
        const int pipeWaves = pipe64_waves(pipeCls, fits12);
        size_t bytes = ((size_t)tabL + (size_t)pipeWaves * teamL) * sizeof(double);
        if (bytes > kLds) return choose_tile(p, false);
        int dctLdsEl = 0; // 1024-point class: the cepstral table shares the LDS when there is room
        if (pipeCls == 10 && c->dctElems > 0 && bytes + (((size_t)c->dctElems + 1) & ~(size_t)1) * sizeof(double) <= kLds) {
            dctLdsEl = (c->dctElems + 1) & ~1;
            tabL += dctLdsEl;
            bytes += (size_t)dctLdsEl * sizeof(double);
        }
        p->pipe64 = pipeCls;
        p->framesPerTile = gt;
        p->rowPitch = pipeCls == 10 ? dctLdsEl : rawEl; // (1024-point class: elements of the table's LDS copy)
        p->rawOrZElems = gt * pipe64_area_elems(pipeCls);
        p->fbkElems = 0;
        p->featElems = 0;
        p->teamElems = teamL;
        p->tabElems = tabL;
        p->activeTeams = pipeWaves;
        p->ldsBytes = bytes;
        p->blocksPerCu = 1;
        p->mode = 0;
        if (tune.debug)
            fprintf(stderr, "mfcc plan: double-precision pipelined kernel, %d complex points, %d waves per workgroup, LDS %zu B\n", 1 << pipeCls,
                    pipeWaves, bytes);
        return MFCC_OK;
    }
    if (wideCand) {
        // a wave's LDS region: one work area, the frame's log-mel rows; shared: window pairs and
        // the lanes' mel part tables
        const int workEl = 2 * (nc + nc / 16); // the padded complex points; every later use of the area is smaller
        const int P4 = (P + 3) & ~3;
        const int fbkL = c->cfg.fbankFlag ? 0 : V * P4; // log-mel values [channel][row], channels padded to 4
        const int teamL = workEl + fbkL + 4; // (+ the one point the split reads past the work area)
        const int sharedL = wide1024_shared_elems(c->cfg.wlen);
        size_t bytes = ((size_t)wide1024_waves() * teamL + sharedL) * sizeof(float);
        if (bytes > kLds) return choose_tile(p, false);
        // the cepstral table shares the LDS when there is room: rows zero-padded to 4 channels, odd pitch
        size_t dctL = 0;
        const int dctPitchL = P4 + 1;
        if (!c->cfg.fbankFlag) {
            const size_t el = ((size_t)(c->M + c->c0) * dctPitchL + 3) & ~(size_t)3;
            if (bytes + el * sizeof(float) <= kLds) dctL = el;
        }
        const bool dctLds = dctL > 0;
        if (dctLds) bytes += dctL * sizeof(float);
        p->wide1024 = true;
        p->framesPerTile = 1;
        p->rowPitch = (int)dctL; // (this kernel: elements of the cepstral table's LDS copy)
        p->rawOrZElems = workEl;
        p->fbkElems = fbkL;
        p->featElems = 0;
        p->teamElems = teamL;
        p->tabElems = sharedL + (int)dctL;
        p->tabPitch = dctPitchL;
        p->activeTeams = wide1024_waves();
        p->ldsBytes = bytes;
        p->blocksPerCu = 1;
        p->mode = 0;
        if (tune.debug)
            fprintf(stderr, "mfcc plan: 1024-point stereo kernel, %d waves per workgroup, LDS %zu B\n", wide1024_waves(), bytes);
        return MFCC_OK;
    }
    // The pipelined kernel for 256 complex points (lean256.hip): a wave is a team and a tile one
    // round of 4 frames; the shared part of the LDS holds the cepstral table, the mel weights and
    // the window pairs; a wave's region holds its sample row and four work areas.  12 waves per
    // workgroup when the registers and the LDS allow, else 8; one workgroup per CU.
    p->lean256 = false;
    if (p->mode == 2 && p->totalRows < 0x7fffffffLL && c->cfg.wlen >= 8 && lean256_applies(log2nc, p->mode, c->cfg.inc, c->cfg.wlen, leanSteps)) {
        const int nz = lean256_nz(c->cfg.wlen);
        const int ft = 4;
        const int ninst = (ft - 1) * c->cfg.inc + c->cfg.wlen + 1;
        // the row ends where the last frame's zero-window reads and the last fetched vector end
        const int rowPitchL = (std::max((ft - 1) * c->cfg.inc + 32 * nz, ((ninst + 7) & ~7)) + 3) & ~3;
        // work area of a frame: its padded complex points only; pitch = 48 (mod 64) elements puts the
        // four frame slots 16 banks apart and the two slots of a half wave 32 banks apart
        const int zBase = 2 * (nc + nc / 16);
        const int zl = zBase + ((48 - zBase % 64) + 64) % 64;
        const int teamL = rowPitchL + 4 * zl;
        const int tabL = (((c->dctElems + 1) & ~1) + lean256_mel_steps(leanSteps) * 64 + 2 * nz * 16 + 3) & ~3;
        for (int pass = 0; pass < 2 && !p->lean256; ++pass) {
            if (pass == 0 && tune.leanWaves == 8) continue;
            const int waves = lean256_waves(c->cfg.wlen, pass == 0);
            if (pass == 1 && waves == lean256_waves(c->cfg.wlen, true)) break;
            const size_t bytes = ((size_t)tabL + (size_t)waves * teamL) * sizeof(float);
            if (bytes > kLds) continue;
            p->lean256 = true;
            p->melSteps = leanSteps;
            p->melW.swap(leanW);
            p->melMeta.swap(leanMeta);
            p->leanWaves = waves;
            p->framesPerTile = ft;
            p->rowPitch = rowPitchL;
            p->rowTail = 0;
            p->rawOrZElems = 4 * zl;
            p->zPitch = zl;
            p->fbkElems = 0;
            p->featElems = 0;
            p->teamElems = teamL;
            p->tabElems = tabL;
            p->activeTeams = waves;
            p->ldsBytes = bytes;
            p->blocksPerCu = 1;
            if (tune.debug)
                fprintf(stderr, "mfcc plan: pipelined 256-point kernel, %d waves per workgroup, LDS %zu B\n", waves, bytes);
        }
        if (p->lean256) return MFCC_OK;
    }
    const int occ = front_occupancy(prec, log2nc, pplog, bestTab, p->mode, p->ldsBytes);
    if (tune.debug)
        fprintf(stderr, "mfcc plan: points/lane %d, teams %d, mode %d, framesPerTile %d, tables in LDS %d, LDS %zu B, workgroups/CU planned %d, runtime says %d\n",
                1 << pplog, teams, p->mode, bestFt, (int)bestTab, p->ldsBytes, p->blocksPerCu, occ);
    if (occ > 0 && occ < p->blocksPerCu) p->blocksPerCu = occ;
    if (tune.blocksPerCu > 0) p->blocksPerCu = tune.blocksPerCu;
    return MFCC_OK;
}

static int finish_plan(MFCCPlan *p)
{
    MFCCContext *c = p->ctx;
    const Tuning tune = tuning();
    int rc = choose_tile(p);
    if (rc) return rc;
    // Rows per block of the regression / fused delta kernels: 256, or fewer (down to 32) while the
    // plan has fewer blocks than two per CU, so that one file alone still spreads over the GPU.  (Every
    // LDS sub-tile of a block carries its own halo rows, so smaller blocks cost no extra reads.)
    {
        DEVICE_SCOPE(c->device);
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0)
            p->numCus = cus;
        p->regRows = kRegRows;
        while (p->regRows > 32 && p->totalRows / p->regRows < 2 * (int64_t)p->numCus) p->regRows /= 2;
    }
    std::vector<int32_t> tileUtt, chunkUtt, regUtt;
    std::vector<TileDesc> tileDesc;
    int64_t tiles = 0, chunks = 0, regBlocks = 0;
    const int64_t bpi = (int64_t)p->channels * p->bps;
    if (p->framesPerTile > 4095) return set_error("internal: tile longer than 4095 frames"), MFCC_E_CONFIG;
    for (int u = 0; u < p->numUtts; ++u) {
        UttDesc &d = p->utts[u];
        const int nt = (d.numFrames + p->framesPerTile - 1) / p->framesPerTile;
        const int ncb = (d.ownCount + kStatRows - 1) / kStatRows;
        const int nrb = (d.numFrames + p->regRows - 1) / p->regRows;
        if (tiles + nt > 0x7fffffffLL || chunks + ncb > 0x7fffffffLL)
            return set_error("batch too large: more than 2^31 tiles"), MFCC_E_ARG;
        d.regBegin = (int32_t)regBlocks;
        d.statBegin = (int32_t)chunks;
        d.tileBegin = (int32_t)tiles;
        tiles += nt;
        chunks += ncb;
        regBlocks += nrb;
    }
    tileUtt.reserve((size_t)tiles);
    tileDesc.reserve((size_t)tiles);
    chunkUtt.reserve((size_t)chunks);
    regUtt.reserve((size_t)regBlocks);
    int64_t planEnd = 0; // one past the last PCM byte the plan owns
    for (const UttDesc &d : p->utts)
        if (d.numFrames > 0) planEnd = std::max<int64_t>(planEnd, d.byteOff + (int64_t)d.numInstants * bpi);
    for (int u = 0; u < p->numUtts; ++u) {
        const UttDesc &d = p->utts[u];
        for (int f0 = 0; f0 < d.numFrames; f0 += p->framesPerTile) {
            const int nf = std::min(p->framesPerTile, d.numFrames - f0);
            const int64_t first = (int64_t)f0 * c->cfg.inc + d.lead; // first instant of the first frame
            const int64_t first0 = first > 0 ? first - 1 : 0;         // plus its pre-emphasis predecessor
            const uint64_t off = (uint64_t)(d.byteOff + first0 * bpi);
            if (off > kTileOffMask) return set_error("PCM buffer larger than 2^48 bytes"), MFCC_E_ARG;
            TileDesc td;
            if (p->lean256 || p->wide1024 || p->pipe64) {
                // k_lean256's / k_wide1024's / k_pipe64's form: offset of the first sample itself; bit 60: it starts the signal;
                // bit 61: a 16-byte vector of the fetch (which begins one sample earlier) would
                // leave the planned bytes
                const int64_t src = d.byteOff + first * bpi;
                const int64_t vecs = (((int64_t)(nf - 1) * c->cfg.inc + c->cfg.wlen + 1) * bpi + 15) / 16;
                const bool edge = p->wide1024      ? src - 8 < 0 || src + wide1024_reach_bytes(c->cfg.wlen) > planEnd
                                  : p->pipe64 == 10 ? src - 8 < 0 || src + pipe64_reach_bytes(c->cfg.wlen) > planEnd
                                                    : src - bpi < 0 || src - bpi + vecs * 16 > planEnd;
                if ((uint64_t)src > kTileOffMask) return set_error("PCM buffer larger than 2^48 bytes"), MFCC_E_ARG;
                td.src = (uint64_t)src | ((uint64_t)nf << 48) | ((uint64_t)(first > 0 ? 0 : 1) << 60) | ((uint64_t)(edge ? 1 : 0) << 61);
                td.outRow = d.outRow + f0;
                tileDesc.push_back(td);
                continue;
            }
            td.src = off | ((uint64_t)nf << 48) | ((uint64_t)(first > 0 ? 1 : 0) << 60);
            if (p->mode == 2 && first > 0) {
                const int64_t elems = (int64_t)(nf - 1) * c->cfg.inc + c->cfg.wlen + 1; // with the predecessor
                const int64_t vecs = (elems + 7) / 8;
                if (vecs <= 3 * 64 && (int64_t)off + vecs * 16 <= planEnd) td.src |= 1ull << 61;
            }
            td.outRow = d.outRow + f0;
            tileDesc.push_back(td);
            tileUtt.push_back(u);
        }
        chunkUtt.insert(chunkUtt.end(), (size_t)((d.ownCount + kStatRows - 1) / kStatRows), u);
        regUtt.insert(regUtt.end(), (size_t)((d.numFrames + p->regRows - 1) / p->regRows), u);
    }
    p->planEnd = planEnd;
    p->numRegBlocks = (int)regBlocks;
    p->numTiles = (int)tiles;
    p->numChunks = (int)chunks;
    DEVICE_SCOPE(c->device);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0)
            p->numCus = cus;
    }
    const size_t nu = std::max<size_t>(p->utts.size(), 1);
    HIP_TRY(hipMalloc((void **)&p->dUtts, nu * sizeof(UttDesc)));
    if (!p->utts.empty())
        HIP_TRY(hipMemcpy(p->dUtts, p->utts.data(), p->utts.size() * sizeof(UttDesc), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&p->dTiles, std::max<size_t>(tileDesc.size(), 1) * sizeof(TileDesc)));
    if (!tileDesc.empty())
        HIP_TRY(hipMemcpy(p->dTiles, tileDesc.data(), tileDesc.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    if (c->cfg.zeroMeanSigFlag) {
        HIP_TRY(hipMalloc((void **)&p->dTileUtt, std::max<size_t>(tileUtt.size(), 1) * sizeof(int32_t)));
        if (!tileUtt.empty())
            HIP_TRY(hipMemcpy(p->dTileUtt, tileUtt.data(), tileUtt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (c->cfg.regreOrder > 1) {
        HIP_TRY(hipMalloc((void **)&p->dRegUtt, std::max<size_t>(regUtt.size(), 1) * sizeof(int32_t)));
        if (!regUtt.empty())
            HIP_TRY(hipMemcpy(p->dRegUtt, regUtt.data(), regUtt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {
        const int rc2 = upload_tab(c->cfg.precision, p->stageTw, &p->dStageTw);
        if (rc2) return rc2;
    }
    if (p->wide1024) {
        // k_wide1024's mel sums: every segment (the bins whose lower channel is s, s = 0 .. P) is cut
        // into at most 10 parts of at most partBins (>= 15, odd) bins; part i goes to lane i % 64 in round
        // i / 64 (partBins is raised until two rounds hold the filterbank).  Table: 128 x {first bin, end bin, part
        // or -1, 0}, then per channel {first part, parts} of the segment whose upper shares it takes
        // (segment ch) and of the one whose lower shares it takes (segment ch + 1).
        const int S1 = c->cfg.bankNum + 1, nc = c->tab.nc;
        std::vector<int> segB(S1, 0), segE(S1, 0), partBegin(S1 + 1, 0);
        for (int k = 0; k < nc; ++k) {
            const int ch = c->tab.loChan[k];
            if (ch < 0 || ch >= S1) continue;
            if (segE[ch] == 0) segB[ch] = k;
            segE[ch] = k + 1;
        }
        p->laneSeg.assign((size_t)(128 + 64) * 4, 0);
        for (int i = 0; i < 128; ++i) p->laneSeg[4 * i + 2] = -1;
        // bins per part: an odd number (the parts of a long segment then start in different LDS
        // banks), 15 or the next odd number with which the longest segment has at most 10 parts
        int longest = 0;
        for (int sIdx = 0; sIdx < S1; ++sIdx) longest = std::max(longest, segE[sIdx] - segB[sIdx]);
        int partBins = 15;
        for (;; partBins += 2) { // (ends: with 17 bins or more, 64 channels make at most 65 + 1024 / 17 parts)
            int total = 0;
            for (int sIdx = 0; sIdx < S1; ++sIdx) total += std::max(1, (segE[sIdx] - segB[sIdx] + partBins - 1) / partBins);
            if ((longest + partBins - 1) / partBins <= 10 && total <= 128) break;
        }
        p->melPieces = (partBins + 7) / 8;
        int parts = 0;
        for (int sIdx = 0; sIdx < S1; ++sIdx) {
            const int len = segE[sIdx] - segB[sIdx], n = std::max(1, (len + partBins - 1) / partBins);
            partBegin[sIdx] = parts;
            for (int j = 0; j < n; ++j, ++parts) {
                if (parts >= 128) return set_error("internal: more than 128 mel segment parts"), MFCC_E_CONFIG;
                p->laneSeg[4 * parts + 0] = segB[sIdx] + (int)((long)len * j / n);
                p->laneSeg[4 * parts + 1] = segB[sIdx] + (int)((long)len * (j + 1) / n);
                p->laneSeg[4 * parts + 2] = parts;
            }
        }
        partBegin[S1] = parts;
        if (c->cfg.bankNum > 64) return set_error("internal: more than 64 channels in the 1024-point stereo kernel"), MFCC_E_CONFIG;
        for (int ch = 0; ch < c->cfg.bankNum; ++ch) {
            int32_t *e = &p->laneSeg[(size_t)(128 + ch) * 4];
            e[0] = partBegin[ch], e[1] = partBegin[ch + 1] - partBegin[ch];
            e[2] = partBegin[ch + 1], e[3] = partBegin[ch + 2] - partBegin[ch + 1];
        }
        p->maxSeg = (parts + 63) / 64;
    }
    HIP_TRY(hipMalloc((void **)&p->dLaneSeg, std::max<size_t>(p->laneSeg.size(), 4) * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p->dLaneSeg, p->laneSeg.data(), p->laneSeg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (p->lean256 || p->wide1024 || p->pipe64) {
        HIP_TRY(hipMalloc((void **)&p->dWork, 2 * sizeof(unsigned)));
        HIP_TRY(hipMemset(p->dWork, 0, 2 * sizeof(unsigned)));
    }
    if (p->pipe64 == 8) {
        HIP_TRY(hipMalloc((void **)&p->dMelW64, p->melW64.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(p->dMelW64, p->melW64.data(), p->melW64.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&p->dMelMeta, p->melMeta.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->dMelMeta, p->melMeta.data(), p->melMeta.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (p->mode == 2) {
        HIP_TRY(hipMalloc((void **)&p->dMelW, p->melW.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(p->dMelW, p->melW.data(), p->melW.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&p->dMelMeta, p->melMeta.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->dMelMeta, p->melMeta.data(), p->melMeta.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // Whole-utterance finishing pass: batch plans whose longest utterance's static rows fit the
    // LDS budget (two workgroups per CU) beside the sub-tile buffers.
    p->uttFinish = false;
    p->finStream = false;
    const bool wantFinish = c->cfg.regreOrder <= 3 && (c->cfg.regreOrder >= 2 || c->cfg.znormFlag) &&
                            p->totalRows > 0 && !tune.noUttFinish && tune.finish != 3;
    if (wantFinish) {
        const size_t elb = c->cfg.precision == MFCC_PREC_F64 ? 8 : 4;
        int64_t maxFrames = 1;
        for (const UttDesc &d : p->utts) maxFrames = std::max<int64_t>(maxFrames, d.numFrames);
        // Segment-streaming kernels (any utterance length; the partition of an utterance's rows does not
        // depend on the batch): the default whenever their LDS regions fit and the window is short.
        size_t strLds = 0;
        const bool strFits = stream_finish_geometry(c->vSize, c->cfg.regreOrder, c->cfg.delwin, (int)elb, tune.streamRows, &p->strColBlock,
                                                    &p->strColBlocks, &p->strChunkRows, &p->strWaveElems, &p->strRedOffset, &strLds, &p->strFast, &p->strStage);
        // (regression windows beyond the register path's -- delwin > 3 -- run faster through the row-block
        // kernels, or the whole-utterance one where it fits)
        const bool strGood = p->strFast || c->cfg.regreOrder == 1;
        if (strFits && (tune.finish == 2 || (tune.finish == 0 && strGood))) {
            p->uttFinish = p->finStream = true;
            p->finLds = strLds;
            const int segRows = stream_finish_seg_chunks() * p->strChunkRows;
            std::vector<int32_t> segs, segBegin(nu + 1, 0);
            for (size_t u = 0; u < nu; ++u) {
                segBegin[u] = (int32_t)(segs.size() / 2);
                for (int r = 0; r < p->utts[u].numFrames; r += segRows) {
                    segs.push_back((int32_t)u);
                    segs.push_back(r);
                }
            }
            segBegin[nu] = (int32_t)(segs.size() / 2);
            p->numSegs = segBegin[nu];
            HIP_TRY(hipMalloc((void **)&p->dSegs, std::max<size_t>(segs.size(), 2) * sizeof(int32_t)));
            if (!segs.empty()) HIP_TRY(hipMemcpy(p->dSegs, segs.data(), segs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&p->dSegBegin, segBegin.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(p->dSegBegin, segBegin.data(), segBegin.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            if (c->cfg.znormFlag) {
                HIP_TRY(hipMalloc((void **)&p->dSegPartial, std::max<size_t>((size_t)p->numSegs, 1) * 2 * c->step * sizeof(double)));
                HIP_TRY(hipMalloc((void **)&p->dSegStats, std::max<size_t>(nu, 1) * 2 * c->step * sizeof(double)));
            }
        }
        // Whole-utterance-in-LDS kernel: batch plans whose longest utterance's static rows fit the LDS
        // budget (two workgroups per CU) beside the sub-tile buffers.
        const size_t statics = (size_t)maxFrames * c->vSize;
        for (int sr = 64; sr >= 8 && !p->uttFinish && !p->shard && c->vSize <= kBlock; sr /= 2) {
            const size_t elems = statics + (size_t)(sr + 2 * c->cfg.delwin) * c->vSize + (size_t)sr * c->step;
            const size_t redOff = (elems * elb + 15) & ~(size_t)15;
            const size_t total = redOff + (2 * (size_t)kFinishBlock + 2 * (size_t)c->step) * sizeof(double);
            if (total <= 80 * 1024) {
                p->uttFinish = true;
                p->finSubRows = sr;
                p->finStaticElems = (int)statics;
                p->finRedOffset = (int)redOff;
                p->finLds = total;
            }
        }
        if (p->uttFinish && !p->finStream) { // register variant: sub-tiles of rowLanes rows, at most kFinishRegRows per lane
            const int rowLanes = kFinishBlock / std::min(kFinishBlock, c->vSize);
            p->finRegRows = (maxFrames <= (int64_t)kFinishRegRows * rowLanes && !tune.noFinishReg) ? 1 : 0;
            // (its two first-delta buffers of rowLanes + 2 delwin rows must fit where the sub-tile buffers are)
            if (p->finRegRows && 2 * (rowLanes + 2 * c->cfg.delwin) * c->vSize >
                                     (p->finSubRows + 2 * c->cfg.delwin) * c->vSize + p->finSubRows * c->step)
                p->finRegRows = 0;
        }
        if (tune.debug)
            fprintf(stderr, "mfcc plan: deltas / z-norm by %s (longest utterance %lld frames, %zu B of LDS)\n",
                    !p->uttFinish ? "row blocks"
                                  : (p->finStream ? "segment-streaming kernels"
                                                  : (p->finRegRows ? "whole-utterance kernel, registers" : "whole-utterance kernel, two sweeps")),
                    (long long)maxFrames, p->finLds);
        if (p->uttFinish)
            HIP_TRY(hipMalloc(&p->dStatics, std::max<size_t>((size_t)p->totalRows * c->vSize * elb, 16)));
    }
    // Fused delta (+ statistics) pass: orders 2 and 3, rows of at most one workgroup of columns,
    // two LDS tiles of (subRows + 2 halo) rows within 64 KiB.
    p->deltaSubRows = 0;
    if (!p->uttFinish && c->cfg.regreOrder >= 2 && c->vSize <= kBlock && !tune.noFusedDelta) {
        const size_t elb = c->cfg.precision == MFCC_PREC_F64 ? 8 : 4;
        const int halo = (std::min(c->cfg.regreOrder, 3) - 1) * c->cfg.delwin;
        for (int sr = 256; sr >= 8 && !p->deltaSubRows; sr /= 2) {
            const size_t tiles = (2 * (size_t)(sr + 2 * halo) * c->vSize * elb + 15) & ~(size_t)15;
            if (tiles + 2 * kBlock * sizeof(double) <= 64 * 1024 && sr >= c->cfg.delwin) {
                p->deltaSubRows = sr;
                p->deltaRedOffset = (int)tiles;
                p->deltaLds = tiles + 2 * kBlock * sizeof(double);
            }
        }
    }
    if (p->deltaSubRows && c->cfg.znormFlag && !p->shard && c->cfg.regreOrder <= 3) // sums of every column
        HIP_TRY(hipMalloc((void **)&p->dBlockPartial, std::max<size_t>(regUtt.size(), 1) * 2 * c->step * sizeof(double)));
    const bool needStats = c->cfg.znormFlag != 0 && !p->uttFinish;
    if (needStats) {
        HIP_TRY(hipMalloc((void **)&p->dChunkUtt, std::max<size_t>(chunkUtt.size(), 1) * sizeof(int32_t)));
        if (!chunkUtt.empty())
            HIP_TRY(hipMemcpy(p->dChunkUtt, chunkUtt.data(), chunkUtt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&p->dPartial, std::max<size_t>(chunkUtt.size(), 1) * 2 * c->step * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&p->dStats, nu * 2 * c->step * sizeof(double)));
    }
    if (c->cfg.zeroMeanSigFlag && !p->shard)
        HIP_TRY(hipMalloc((void **)&p->dSums2, nu * 4 * sizeof(unsigned long long)));
    return MFCC_OK;
}

extern "C" int MFCCPlanCreate(MFCCContext *c, int channels, int bytesPerSample, int nUtts,
                              const int64_t *byteOff, const int64_t *numSamples, MFCCPlan **out)
{
    if (!c || !out || nUtts < 0 || (nUtts > 0 && (!byteOff || !numSamples)))
        return set_error("bad argument to MFCCPlanCreate"), MFCC_E_ARG;
    *out = nullptr;
    if (channels < 1 || channels > 2 || bytesPerSample < 1 || bytesPerSample > 4)
        return set_error("unsupported PCM layout: %d channels x %d bytes", channels, bytesPerSample), MFCC_E_WAV;
    const int rows = channels == 2 ? 4 : channels; /* main.c:198-199 */
    if (c->cfg.vecNum > rows)
        return set_error("vecnum %d > num_channels %d", c->cfg.vecNum, channels), MFCC_E_VECNUM; /* main.c:218-222 */
    std::unique_ptr<MFCCPlan> p(new MFCCPlan);
    p->ctx = c;
    p->channels = channels;
    p->bps = bytesPerSample;
    p->numUtts = nUtts;
    p->utts.resize((size_t)nUtts);
    int64_t row = 0;
    for (int u = 0; u < nUtts; ++u) {
        if (numSamples[u] < 0 || numSamples[u] > 0x7fffffffLL || byteOff[u] < 0)
            return set_error("utterance %d: bad offset or length", u), MFCC_E_ARG;
        UttDesc &d = p->utts[u];
        memset(&d, 0, sizeof d);
        d.byteOff = byteOff[u];
        d.outRow = row;
        d.numInstants = (int32_t)numSamples[u];
        d.numFrames = (int32_t)MFCCNumFrames((long)numSamples[u], c->cfg.wlen, c->cfg.inc);
        d.ownBegin = 0;
        d.ownCount = d.numFrames;
        row += d.numFrames;
    }
    p->totalRows = row;
    p->ownedRows = row;
    int rc = finish_plan(p.get());
    if (rc) {
        MFCCPlanDestroy(p.release());
        return rc;
    }
    *out = p.release();
    return MFCC_OK;
}

extern "C" int MFCCShardGeometry(const MFCCConfig *cfg, int64_t totalSamples, int64_t frameBegin,
                                 int64_t frameEnd, int64_t *sampleBegin, int64_t *sampleCount,
                                 int64_t *totalRows, int64_t *ownedRowBegin, int64_t *ownedRows,
                                 int *lead)
{
    if (!cfg || totalSamples < 0 || cfg->wlen < 1 || cfg->inc < 1)
        return set_error("bad argument to MFCCShardGeometry"), MFCC_E_ARG;
    const int64_t W = cfg->wlen, H = cfg->inc;
    const int64_t total = totalSamples < W ? 0 : (totalSamples - (W - H)) / H;
    frameBegin = std::max<int64_t>(0, std::min(frameBegin, total));
    frameEnd = std::max(frameBegin, std::min(frameEnd, total));
    // Regress reads delwin frames on each side; every further delta order reads delwin more
    const int64_t halo = (int64_t)(cfg->regreOrder - 1) * cfg->delwin;
    const int64_t eb = std::max<int64_t>(0, frameBegin - halo), ee = std::min(total, frameEnd + halo);
    const bool empty = frameEnd == frameBegin;
    const int ld = (!empty && eb * H > 0) ? 1 : 0; // one lead instant for pre-emphasis
    const int64_t sb = empty ? 0 : eb * H - ld;
    if (sampleBegin) *sampleBegin = sb;
    if (sampleCount) *sampleCount = empty ? 0 : (ee - 1) * H + W - sb;
    if (totalRows) *totalRows = empty ? 0 : ee - eb;
    if (ownedRowBegin) *ownedRowBegin = empty ? 0 : frameBegin - eb;
    if (ownedRows) *ownedRows = frameEnd - frameBegin;
    if (lead) *lead = ld;
    return MFCC_OK;
}

extern "C" int MFCCPlanCreateShard(MFCCContext *c, int channels, int bytesPerSample,
                                   int64_t totalSamples, int64_t frameBegin, int64_t frameEnd,
                                   const int64_t *chanSum2, MFCCPlan **out)
{
    if (!c || !out || totalSamples < 0) return set_error("bad argument to MFCCPlanCreateShard"), MFCC_E_ARG;
    *out = nullptr;
    if (channels < 1 || channels > 2 || bytesPerSample < 1 || bytesPerSample > 4)
        return set_error("unsupported PCM layout"), MFCC_E_WAV;
    const int rows = channels == 2 ? 4 : channels;
    if (c->cfg.vecNum > rows) return set_error("vecnum %d > num_channels %d", c->cfg.vecNum, channels), MFCC_E_VECNUM;
    if (c->cfg.zeroMeanSigFlag && !chanSum2)
        return set_error("zeroMeanSigFlag needs chanSum2 for a shard plan"), MFCC_E_ARG;
    if (c->cfg.normEnergy)
        return set_error("normEnergy needs the whole recording's maximum: not available for frame shards"), MFCC_E_CONFIG;
    std::unique_ptr<MFCCPlan> p(new MFCCPlan);
    p->ctx = c;
    p->shard = true;
    p->channels = channels;
    p->bps = bytesPerSample;
    p->numUtts = 1;
    p->utts.resize(1);
    UttDesc &d = p->utts[0];
    memset(&d, 0, sizeof d);
    int64_t rowsTotal = 0;
    int lead = 0;
    int rc = MFCCShardGeometry(&c->cfg, totalSamples, frameBegin, frameEnd, &p->sampleBegin,
                               &p->sampleCount, &rowsTotal, &p->ownedRowBegin, &p->ownedRows, &lead);
    if (rc) return rc;
    if (p->sampleCount > 0x7fffffffLL) return set_error("shard longer than 2^31 instants"), MFCC_E_ARG;
    d.byteOff = 0;
    d.outRow = 0;
    d.numInstants = (int32_t)p->sampleCount;
    d.numFrames = (int32_t)rowsTotal;
    d.lead = lead;
    d.ownBegin = (int32_t)p->ownedRowBegin;
    d.ownCount = (int32_t)p->ownedRows;
    if (c->cfg.zeroMeanSigFlag && totalSamples > 0)
        for (int r = 0; r < 4; ++r) d.mean[r] = (double)chanSum2[r] / (2.0 * (double)totalSamples);
    p->totalRows = d.numFrames;
    rc = finish_plan(p.get());
    if (rc) {
        MFCCPlanDestroy(p.release());
        return rc;
    }
    *out = p.release();
    return MFCC_OK;
}

extern "C" void MFCCPlanDestroy(MFCCPlan *p)
{
    if (!p) return;
    DeviceScope scope(p->ctx ? p->ctx->device : 0);
    (void)hipFree(p->dUtts);
    (void)hipFree(p->dTileUtt);
    (void)hipFree(p->dTiles);
    (void)hipFree(p->dRegUtt);
    (void)hipFree(p->dMelW);
    (void)hipFree(p->dMelW64);
    (void)hipFree(p->dBlockPartial);
    (void)hipFree(p->dStatics);
    (void)hipFree(p->dSegs);
    (void)hipFree(p->dSegBegin);
    (void)hipFree(p->dSegPartial);
    (void)hipFree(p->dSegStats);
    (void)hipFree(p->dMelMeta);
    (void)hipFree(p->dWork);
    (void)hipFree(p->dChunkUtt);
    (void)hipFree(p->dLaneSeg);
    (void)hipFree(p->dStageTw);
    (void)hipFree(p->dPartial);
    (void)hipFree(p->dStats);
    (void)hipFree(p->dSums2);
    if (p->evStart) (void)hipEventDestroy(p->evStart);
    if (p->evStop) (void)hipEventDestroy(p->evStop);
    delete p;
}

extern "C" int64_t MFCCPlanTotalRows(const MFCCPlan *p) { return p ? p->totalRows : 0; }
extern "C" int64_t MFCCPlanRowOffset(const MFCCPlan *p, int u)
{
    return (p && u >= 0 && u < p->numUtts) ? p->utts[u].outRow : -1;
}
extern "C" int64_t MFCCPlanNumFrames(const MFCCPlan *p, int u)
{
    return (p && u >= 0 && u < p->numUtts) ? p->utts[u].numFrames : -1;
}
extern "C" int64_t MFCCPlanSampleBegin(const MFCCPlan *p) { return p ? p->sampleBegin : 0; }
extern "C" int64_t MFCCPlanSampleCount(const MFCCPlan *p) { return p ? p->sampleCount : 0; }
extern "C" int64_t MFCCPlanOwnedRowBegin(const MFCCPlan *p) { return p ? p->ownedRowBegin : 0; }
extern "C" int64_t MFCCPlanOwnedRows(const MFCCPlan *p) { return p ? p->ownedRows : 0; }

extern "C" int MFCCPlanEnableTiming(MFCCPlan *p, int on)
{
    if (!p) return set_error("NULL plan"), MFCC_E_ARG;
    if (on && !p->evStart) {
        DEVICE_SCOPE(p->ctx->device);
        HIP_TRY(hipEventCreate(&p->evStart));
        HIP_TRY(hipEventCreate(&p->evStop));
    }
    p->timing = on != 0;
    p->timed = false;
    return MFCC_OK;
}

extern "C" float MFCCPlanLastKernelMs(MFCCPlan *p)
{
    if (!p || !p->timing || !p->timed) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(p->evStop) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, p->evStart, p->evStop) != hipSuccess) return -1.0f;
    return ms;
}

// ------------------------------------------------------------------------------------------
// extraction
// ------------------------------------------------------------------------------------------

template <typename T>
static int run_front(MFCCContext *c, MFCCPlan *p, const void *pcmDev, void *outDev, int outStep, void *stream)
{
    FrontArgs<T> a;
    memset(&a, 0, sizeof a);
    a.pcm = static_cast<const unsigned char *>(pcmDev);
    a.tiles = p->dTiles;
    a.utts = p->dUtts;
    a.tileUtt = p->dTileUtt;
    a.zeroMean = c->cfg.zeroMeanSigFlag ? 1 : 0;
    a.lean = p->mode;
    a.melW = p->dMelW;
    a.melMeta = p->dMelMeta;
    a.melSteps = p->melSteps;
    a.melW64 = p->dMelW64;
    for (int sl = 0; sl < 4; ++sl) a.melSlotSteps[sl] = p->melSlotSteps[sl];
    a.out = static_cast<T *>(outDev);
    a.numTiles = p->numTiles;
    a.wlen = c->cfg.wlen;
    a.inc = c->cfg.inc;
    a.framesPerTile = p->framesPerTile;
    a.V = c->cfg.vecNum;
    a.channels = p->channels;
    a.bps = p->bps;
    a.P = c->cfg.bankNum;
    a.M = c->M;
    a.D = c->D;
    a.vSize = c->vSize;
    a.step = outStep; // row pitch of outDev: the full row, or vSize for the compact static buffer
    a.c0Flag = c->c0;
    a.energyFlag = c->cfg.energyFlag;
    a.usePower = c->cfg.usePower ? 1 : 0;
    a.takeLogs = c->cfg.takeLogs ? 1 : 0;
    a.zcFlag = c->cfg.zeroCrossingFlag;
    a.brightFlag = c->cfg.brightFlag;
    a.S = c->cfg.subBandEFlag;
    a.fftLength = c->cfg.fftLength;
    a.fbankFlag = c->cfg.fbankFlag ? 1 : 0;
    a.preK = c->cfg.preemphasise;
    a.mfnorm = (T)c->tab.mfnorm;
    a.window = static_cast<const T *>(c->dWindow);
    a.tw = static_cast<const T *>(p->dStageTw);
    a.splitTw = static_cast<const T *>(c->dSplitTw);
    a.loWt = static_cast<const T *>(c->dLoWt);
    a.laneSeg = p->dLaneSeg;
    a.maxSeg = p->maxSeg;
    a.melPieces = p->melPieces;
    a.pairAligned = (c->cfg.inc % 2 == 0) ? 1 : 0;
    a.zPitch = p->zPitch;
    a.dctT = static_cast<const T *>(c->dDct);
    a.dctPitch = c->dctPitch;
    a.dctElems = c->dctElems;
    a.rowPitch = p->rowPitch;
    a.rowTail = p->rowTail;
    a.rawOrZElems = p->rawOrZElems;
    a.teamElems = p->teamElems;
    a.activeTeams = p->activeTeams;
    a.tabElems = p->tabElems;
    a.tabPitch = p->tabPitch;
    a.winPad = p->winPad;
    a.featElems = p->featElems;
    a.fbkElems = p->fbkElems;
    a.stamps = nullptr;
    a.skip = 0;
#ifdef MFCC_STAMPS
    {
        static unsigned long long *dStamps = nullptr;
        if (!dStamps) {
            HIP_TRY(hipMalloc((void **)&dStamps, 12 * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(dStamps, 0, 12 * sizeof(unsigned long long)));
        }
        if (getenv("MFCC_STAMPS_DUMP")) {
            unsigned long long h[12];
            HIP_TRY(hipMemcpy(h, dStamps, sizeof h, hipMemcpyDeviceToHost));
            if (h[1])
                fprintf(stderr, "mfcc stamps: interval [%s -> %s]: %.1f cycles x %.2f per tile = %.2f%% of the wave lifetimes\n",
                        getenv("MFCC_STAMP_BEGIN") ? getenv("MFCC_STAMP_BEGIN") : "?",
                        getenv("MFCC_STAMP_END") ? getenv("MFCC_STAMP_END") : "?", (double)h[0] / (double)h[1],
                        (double)h[1] / std::max(1, p->numTiles), h[2] ? 100.0 * (double)h[0] / (double)h[2] : 0.0);
            if (h[6])
                fprintf(stderr, "mfcc stamps: waves %llu, average lifetime %.1f us, first start to last end %.1f us\n", h[6],
                        (double)h[5] / (double)h[6] / 100.0, (double)(h[4] - h[3]) / 100.0);
            HIP_TRY(hipMemset(dStamps, 0, 12 * sizeof(unsigned long long)));
            const unsigned long long big = ~0ull;
            HIP_TRY(hipMemcpy(dStamps + 3, &big, sizeof big, hipMemcpyHostToDevice));
        }
        a.stamps = dStamps;
        if (const char *e = getenv("MFCC_SKIP")) a.skip = atoi(e);
        a.stampBegin = getenv("MFCC_STAMP_BEGIN") ? (int)strtol(getenv("MFCC_STAMP_BEGIN"), nullptr, 0) : 0;
        a.stampEnd = getenv("MFCC_STAMP_END") ? atoi(getenv("MFCC_STAMP_END")) : -1;
    }
#endif
    // persistent workgroups: as many as stay resident, each walks tiles with a grid stride
    const int teams = p->activeTeams;
    const int blocks = (int)std::min<int64_t>((p->numTiles + teams - 1) / teams, (int64_t)p->numCus * p->blocksPerCu);
    if constexpr (sizeof(T) == 4) {
        if (p->wide1024) {
            // runs of waves x chunk consecutive frames go to the workgroups in turn; the waves of a
            // workgroup claim the combs of its runs (frames w, w + waves, ...) from a counter in LDS.
            // Measured on C4: 1 .. 6 frames per comb 4.62 ms, 8: 4.74, 31: 5.4 (the runs are not
            // re-balanced between workgroups); 4 keeps the fetch of shared bytes lowest.
            const int waves = p->activeTeams;
            const int64_t teamsMax = (int64_t)p->numCus * waves;
            int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(4, p->numTiles / (4 * teamsMax)));
            if (p->leanChunk > 0) chunk = p->leanChunk;
            const int64_t chunks = (p->numTiles + chunk - 1) / chunk;
            const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((chunks + waves - 1) / waves, p->numCus));
            a.numUtts = p->numUtts;
            a.planEnd = p->planEnd;
            a.tilesPerTeam = chunk;
            a.work = p->dWork;
            const int rcw = launch_wide1024(a, nb, p->ldsBytes, stream);
            if (rcw) return set_error("static-feature kernel launch failed (HIP error %d)", rcw), MFCC_E_DEVICE;
            return MFCC_OK;
        }
        if (p->lean256) {
            // waves draw chunks of consecutive tiles: 32 tiles when there is plenty of work (measured
            // on 2.5 M tiles: 16..96 tiles 3.34 ms, 8 tiles 3.74, 128 tiles 3.51, fixed ranges 3.7),
            // fewer when that would leave a wave fewer than 8 chunks, one when there are more waves
            // than tiles
            const int waves = p->activeTeams;
            const int64_t teamsMax = (int64_t)p->numCus * waves;
            int chunk = p->numTiles <= teamsMax ? 1 : (int)std::max<int64_t>(2, std::min<int64_t>(32, p->numTiles / (8 * teamsMax)));
            if (p->leanChunk > 0) chunk = p->leanChunk;
            if (p->leanChunk < 0) chunk = (int)((p->numTiles + teamsMax - 1) / teamsMax);
            const int64_t chunks = (p->numTiles + chunk - 1) / chunk;
            const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((chunks + waves - 1) / waves, p->numCus));
            a.numUtts = p->numUtts;
            a.planEnd = p->planEnd;
            a.tilesPerTeam = chunk;
            a.work = p->dWork;
            const int rcl = launch_lean256(a, nb, p->ldsBytes, p->leanWaves, stream);
            if (rcl) return set_error("static-feature kernel launch failed (HIP error %d)", rcl), MFCC_E_DEVICE;
            return MFCC_OK;
        }
    }
    if constexpr (sizeof(T) == 8) {
        if (p->pipe64) {
            const int waves = p->activeTeams;
            const int64_t teamsMax = (int64_t)p->numCus * waves;
            int chunk = p->numTiles <= teamsMax ? 1 : (int)std::max<int64_t>(2, std::min<int64_t>(32, p->numTiles / (8 * teamsMax)));
            // (1024-point class: combs of a workgroup's runs claimed from a counter in LDS, see k_wide1024's launch)
            if (p->pipe64 == 10) chunk = (int)std::max<int64_t>(1, std::min<int64_t>(4, p->numTiles / (4 * teamsMax)));
            if (p->leanChunk > 0) chunk = p->leanChunk;
            const int64_t chunks = (p->numTiles + chunk - 1) / chunk;
            const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((chunks + waves - 1) / waves, p->numCus));
            a.numUtts = p->numUtts;
            a.planEnd = p->planEnd;
            a.tilesPerTeam = chunk;
            a.work = p->dWork;
            const int rcp = launch_pipe64(a, p->pipe64, waves, nb, p->ldsBytes, stream);
            if (rcp) return set_error("static-feature kernel launch failed (HIP error %d)", rcp), MFCC_E_DEVICE;
            return MFCC_OK;
        }
    }
    const int rc = launch_front(c->cfg.precision, c->tab.log2nc, p->pplog, &a, blocks, p->ldsBytes, stream);
    if (rc) return set_error("static-feature kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

template <typename T>
static int run_regress(MFCCContext *c, MFCCPlan *p, void *outDev, void *stream)
{
    double sigmaT2 = 0.0;
    for (int t = 1; t <= c->cfg.delwin; ++t) sigmaT2 += (double)t * t;
    sigmaT2 *= 2.0;
    for (int order = 2; order <= std::min(c->cfg.regreOrder, 3); ++order) { /* main.c:259-260: two calls at most */
        RegressArgs<T> a;
        memset(&a, 0, sizeof a);
        a.data = static_cast<T *>(outDev);
        a.utts = p->dUtts;
        a.blockUtt = p->dRegUtt;
        a.numBlocks = p->numRegBlocks;
        a.vSize = c->vSize;
        a.step = c->step;
        a.srcCol = (order - 2) * c->vSize;
        a.dstCol = (order - 1) * c->vSize;
        a.delwin = c->cfg.delwin;
        a.regRows = p->regRows;
        a.sigmaT2 = sigmaT2;
        a.invSigmaT2 = (T)(1.0 / sigmaT2);
        const int rc = launch_regress(c->cfg.precision, &a, p->numRegBlocks, stream);
        if (rc) return set_error("regression kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    }
    return MFCC_OK;
}

// All regression orders in one pass (and the z-norm partial sums of batch plans).
template <typename T>
static int run_delta(MFCCContext *c, MFCCPlan *p, void *outDev, void *stream)
{
    double sigmaT2 = 0.0;
    for (int t = 1; t <= c->cfg.delwin; ++t) sigmaT2 += (double)t * t;
    DeltaArgs<T> a;
    memset(&a, 0, sizeof a);
    a.data = static_cast<T *>(outDev);
    a.utts = p->dUtts;
    a.blockUtt = p->dRegUtt;
    a.numBlocks = p->numRegBlocks;
    a.vSize = c->vSize;
    a.step = c->step;
    a.delwin = c->cfg.delwin;
    a.subRows = p->deltaSubRows;
    a.regRows = p->regRows;
    a.redOffset = p->deltaRedOffset;
    a.sigmaT2 = 2.0 * sigmaT2;
    a.partial = p->dBlockPartial;
    const int rc = launch_delta(c->cfg.precision, &a, std::min(c->cfg.regreOrder, 3), p->deltaLds, stream);
    if (rc) return set_error("delta kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

// phase (segment-streaming kernels): 0 batch plan; 1 shard plan's extraction; 2 shard plan's normalisation
template <typename T>
static int run_finish(MFCCContext *c, MFCCPlan *p, void *outDev, void *stream, int phase = 0)
{
    double sigmaT2 = 0.0;
    for (int t = 1; t <= c->cfg.delwin; ++t) sigmaT2 += (double)t * t;
    FinishArgs<T> a;
    memset(&a, 0, sizeof a);
    a.statics = static_cast<const T *>(p->dStatics);
    a.out = static_cast<T *>(outDev);
    a.utts = p->dUtts;
    a.vSize = c->vSize;
    a.step = c->step;
    a.delwin = c->cfg.delwin;
    a.subRows = p->finSubRows;
    a.staticElems = p->finStaticElems;
    a.redOffset = p->finRedOffset;
    a.znorm = c->cfg.znormFlag ? 1 : 0;
    a.regRows = p->finRegRows;
    a.sigmaT2 = 2.0 * sigmaT2;
    if (p->finStream) {
        StreamArgs<T> sa;
        memset(&sa, 0, sizeof sa);
        sa.statics = a.statics, sa.out = a.out, sa.utts = a.utts, sa.segs = p->dSegs, sa.segBegin = p->dSegBegin;
        sa.numSegs = p->numSegs, sa.numUtts = p->numUtts;
        sa.vSize = a.vSize, sa.step = a.step, sa.delwin = a.delwin, sa.znorm = a.znorm;
        sa.colBlock = p->strColBlock, sa.numColBlocks = p->strColBlocks, sa.chunkRows = p->strChunkRows;
        sa.waveElems = p->strWaveElems, sa.redOffset = p->strRedOffset, sa.sigmaT2 = a.sigmaT2;
        sa.fastPath = p->strFast, sa.stageOut = p->strStage, sa.phase = phase;
        sa.partial = p->dSegPartial, sa.stats = p->dSegStats;
        const int rc = launch_seg_finish(c->cfg.precision, &sa, c->cfg.regreOrder, p->finLds, stream);
        if (rc) return set_error("segment finishing kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
        return MFCC_OK;
    }
    const int rc = launch_utt_finish(c->cfg.precision, &a, c->cfg.regreOrder, p->numUtts, p->finLds, stream);
    if (rc) return set_error("finishing kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

template <typename T>
static StatArgs<T> stat_args(MFCCContext *c, MFCCPlan *p, void *outDev)
{
    StatArgs<T> a;
    memset(&a, 0, sizeof a);
    a.data = static_cast<T *>(outDev);
    a.utts = p->dUtts;
    a.chunkUtt = p->dChunkUtt;
    a.numChunks = p->numChunks;
    a.step = c->step;
    a.partial = p->dPartial;
    a.stats = p->dStats;
    return a;
}

static int run_colstats(MFCCContext *c, MFCCPlan *p, void *outDev, void *stream)
{
    int rc;
    if (c->cfg.precision == MFCC_PREC_F64) {
        StatArgs<double> a = stat_args<double>(c, p, outDev);
        rc = launch_colstats(1, &a, p->numChunks, stream);
    } else {
        StatArgs<float> a = stat_args<float>(c, p, outDev);
        rc = launch_colstats(0, &a, p->numChunks, stream);
    }
    if (rc) return set_error("z-norm statistics kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

static int run_normalize(MFCCContext *c, MFCCPlan *p, void *outDev, bool shard, const double *sums,
                         void *stream)
{
    int rc;
    if (c->cfg.precision == MFCC_PREC_F64) {
        StatArgs<double> a = stat_args<double>(c, p, outDev);
        rc = launch_normalize(1, &a, p->numChunks, shard, sums, stream);
    } else {
        StatArgs<float> a = stat_args<float>(c, p, outDev);
        rc = launch_normalize(0, &a, p->numChunks, shard, sums, stream);
    }
    if (rc) return set_error("z-norm kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCExtractDevice(MFCCContext *c, MFCCPlan *p, const void *pcmDev, size_t pcmBytes,
                                 void *outDev, void *stream)
{
    if (!c || !p || p->ctx != c) return set_error("plan does not belong to this context"), MFCC_E_ARG;
    if (p->numTiles == 0) return MFCC_OK;
    if (!pcmDev || !outDev) return set_error("NULL device buffer"), MFCC_E_ARG;
    const int64_t bpi = (int64_t)p->channels * p->bps;
    for (const UttDesc &d : p->utts) // every planned byte must lie inside the caller's buffer
        if (d.numFrames > 0 && (uint64_t)(d.byteOff + (int64_t)d.numInstants * bpi) > (uint64_t)pcmBytes)
            return set_error("PCM buffer of %zu bytes is shorter than the plan needs", pcmBytes), MFCC_E_ARG;
    DEVICE_SCOPE(c->device);
    TraceRange range("MFCCExtractDevice");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        uint64_t pcm = 0;
        for (const UttDesc &d : p->utts) pcm += (uint64_t)d.numInstants * (uint64_t)bpi;
        g_count.extractCalls += 1;
        g_count.frames += (uint64_t)p->ownedRows;
        g_count.pcmBytes += pcm;
        g_count.featureBytes += (uint64_t)p->ownedRows * (uint64_t)c->step * (c->cfg.precision == MFCC_PREC_F64 ? 8u : 4u);
    }
    int rc;
    if (c->cfg.zeroMeanSigFlag && !p->shard) {
        HIP_TRY(hipMemsetAsync(p->dSums2, 0, (size_t)p->numUtts * 4 * sizeof(unsigned long long), st));
        rc = launch_uttmeans(static_cast<const unsigned char *>(pcmDev), p->dUtts, p->numUtts,
                             p->channels, p->bps, p->dSums2, stream);
        if (rc) return set_error("row-mean kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    }
    const bool f64 = c->cfg.precision == MFCC_PREC_F64;
    // The reference fills the delta and delta-delta slots only (main.c:259-260); slots of higher
    // orders keep their initial value, taken as zero here.
    if (c->cfg.regreOrder > 3)
        HIP_TRY(hipMemsetAsync(outDev, 0, (size_t)p->totalRows * c->step * (f64 ? 8 : 4), st));
    if (p->timing) HIP_TRY(hipEventRecord(p->evStart, st));
    void *frontOut = p->uttFinish ? p->dStatics : outDev;
    const int frontStep = p->uttFinish ? c->vSize : c->step;
    if (p->uttFinish && tuning().debug == 2) // diagnostic: static features no kernel writes show as NaN
        HIP_TRY(hipMemsetAsync(p->dStatics, 0xFF, (size_t)p->totalRows * c->vSize * (f64 ? 8 : 4), st));
    rc = f64 ? run_front<double>(c, p, pcmDev, frontOut, frontStep, stream)
             : run_front<float>(c, p, pcmDev, frontOut, frontStep, stream);
    if (rc) return rc;
    if (p->timing) {
        HIP_TRY(hipEventRecord(p->evStop, st));
        p->timed = true;
    }
    if (c->cfg.normEnergy) { // main.c:255: the first row's column behind its coefficients, 50 dB, 0.1
        const int e = launch_energy_norm(c->cfg.precision, frontOut, p->dUtts, p->numUtts, frontStep, c->M, 50.0, 0.1, stream);
        if (e) return set_error("energy normalisation kernel launch failed (HIP error %d)", e), MFCC_E_DEVICE;
    }
    if (p->uttFinish) {
        const int phase = p->shard ? 1 : 0;
        return f64 ? run_finish<double>(c, p, outDev, stream, phase) : run_finish<float>(c, p, outDev, stream, phase);
    }
    const bool fused = p->deltaSubRows > 0;
    if (fused)
        rc = f64 ? run_delta<double>(c, p, outDev, stream) : run_delta<float>(c, p, outDev, stream);
    else
        rc = f64 ? run_regress<double>(c, p, outDev, stream) : run_regress<float>(c, p, outDev, stream);
    if (rc) return rc;
    if (c->cfg.znormFlag && !p->shard) {
        const bool byBlock = fused && p->dBlockPartial;
        if (!byBlock && (rc = run_colstats(c, p, outDev, stream))) return rc;
        StatArgs<float> fa = stat_args<float>(c, p, nullptr);
        if (byBlock) fa.partial = p->dBlockPartial;
        const int e = launch_colfinal(fa, p->numUtts, byBlock ? p->regRows : 0, stream);
        if (e) return set_error("z-norm finalise kernel launch failed (HIP error %d)", e), MFCC_E_DEVICE;
        if ((rc = run_normalize(c, p, outDev, false, nullptr, stream))) return rc;
    }
    return MFCC_OK;
}

extern "C" int MFCCShardColumnSums(MFCCContext *c, MFCCPlan *p, const void *outDev, double *sumsDev,
                                   void *stream)
{
    if (!c || !p || p->ctx != c || !p->shard || !sumsDev) return set_error("bad argument"), MFCC_E_ARG;
    if (!c->cfg.znormFlag) return set_error("znormFlag is off"), MFCC_E_CONFIG;
    DEVICE_SCOPE(c->device);
    if (p->finStream) {
        // the segments' sums are what the plan's last MFCCExtractDevice left beside the rows
        if (p->numSegs == 0) {
            HIP_TRY(hipMemsetAsync(sumsDev, 0, (size_t)(2 * c->step + 1) * sizeof(double), static_cast<hipStream_t>(stream)));
            return MFCC_OK;
        }
        const int e = launch_seg_shard_sums(p->dSegPartial, p->dUtts, p->numSegs, sumsDev, c->step, stream);
        if (e) return set_error("shard statistics kernel launch failed (HIP error %d)", e), MFCC_E_DEVICE;
        return MFCC_OK;
    }
    if (p->numChunks == 0) {
        HIP_TRY(hipMemsetAsync(sumsDev, 0, (size_t)(2 * c->step + 1) * sizeof(double), static_cast<hipStream_t>(stream)));
        return MFCC_OK;
    }
    if (!outDev) return set_error("NULL device buffer"), MFCC_E_ARG;
    int rc = run_colstats(c, p, const_cast<void *>(outDev), stream);
    if (rc) return rc;
    StatArgs<float> fa = stat_args<float>(c, p, nullptr);
    const int e = launch_shardfinal(fa, p->numChunks, 0, sumsDev, stream);
    if (e) return set_error("shard statistics kernel launch failed (HIP error %d)", e), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCShardNormalize(MFCCContext *c, MFCCPlan *p, void *outDev, const double *sumsDev,
                                  void *stream)
{
    if (!c || !p || p->ctx != c || !p->shard || !sumsDev) return set_error("bad argument"), MFCC_E_ARG;
    if (p->finStream) {
        // the rows are formed again from the plan's static features and leave normalised: the
        // matrix is written, not read
        if (p->numSegs == 0) return MFCC_OK;
        if (!outDev) return set_error("NULL device buffer"), MFCC_E_ARG;
        DEVICE_SCOPE(c->device);
        const int e = launch_totals_to_stats(sumsDev, p->dSegStats, c->step, stream);
        if (e) return set_error("statistics kernel launch failed (HIP error %d)", e), MFCC_E_DEVICE;
        return c->cfg.precision == MFCC_PREC_F64 ? run_finish<double>(c, p, outDev, stream, 2) : run_finish<float>(c, p, outDev, stream, 2);
    }
    if (p->numChunks == 0) return MFCC_OK;
    if (!outDev) return set_error("NULL device buffer"), MFCC_E_ARG;
    DEVICE_SCOPE(c->device);
    return run_normalize(c, p, outDev, true, sumsDev, stream);
}

extern "C" int MFCCChannelSums(MFCCContext *c, const void *pcmDev, int channels, int bytesPerSample,
                               int64_t sampleBegin, int64_t count, int64_t *sums2Dev, void *stream)
{
    if (!c || !sums2Dev || (count > 0 && !pcmDev)) return set_error("bad argument"), MFCC_E_ARG;
    if (channels < 1 || channels > 2 || bytesPerSample < 1 || bytesPerSample > 4)
        return set_error("unsupported PCM layout"), MFCC_E_WAV;
    DEVICE_SCOPE(c->device);
    HIP_TRY(hipMemsetAsync(sums2Dev, 0, 4 * sizeof(int64_t), static_cast<hipStream_t>(stream)));
    const int rc = launch_chansums(static_cast<const unsigned char *>(pcmDev), channels, bytesPerSample,
                                   sampleBegin * channels * bytesPerSample, count,
                                   reinterpret_cast<unsigned long long *>(sums2Dev), stream);
    if (rc) return set_error("channel-sum kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

// ------------------------------------------------------------------------------------------
// the reference's lower-level API, batched on the device
// ------------------------------------------------------------------------------------------

static bool pow2_between(int n, int lo, int hi) { return n >= lo && n <= hi && (n & (n - 1)) == 0; }

extern "C" int MFCCFFTDevice(int precision, void *dataDev, int numFrames, int n, int invert, void *stream)
{
    if (numFrames < 0 || (numFrames > 0 && !dataDev) || !pow2_between(n, 2, 2 << kMaxLog2Nc))
        return set_error("bad argument to MFCCFFTDevice (n must be a power of two, 2..%d)", 2 << kMaxLog2Nc), MFCC_E_ARG;
    const int rc = launch_fft_batch(precision == MFCC_PREC_F64, dataDev, numFrames, n, invert ? 1 : 0, 0, stream);
    if (rc) return set_error("FFT kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCRealftDevice(int precision, void *dataDev, int numFrames, int n, void *stream)
{
    if (numFrames < 0 || (numFrames > 0 && !dataDev) || !pow2_between(n, 4, 2 << kMaxLog2Nc))
        return set_error("bad argument to MFCCRealftDevice (n must be a power of two, 4..%d)", 2 << kMaxLog2Nc), MFCC_E_ARG;
    const int rc = launch_fft_batch(precision == MFCC_PREC_F64, dataDev, numFrames, n, 0, 1, stream);
    if (rc) return set_error("Realft kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCWave2FBankDevice(MFCCContext *c, const void *framesDev, int numFrames, void *fbankDev,
                                    void *teDev, void *stream)
{
    if (!c || numFrames < 0 || (numFrames > 0 && (!framesDev || !fbankDev)))
        return set_error("bad argument to MFCCWave2FBankDevice"), MFCC_E_ARG;
    DEVICE_SCOPE(c->device);
    const int rc = launch_wave2fbank_batch(c->cfg.precision == MFCC_PREC_F64, framesDev, fbankDev, teDev, nullptr,
                                           nullptr, numFrames, c->cfg.wlen, c->tab.fftN, c->tab.klo, c->tab.khi,
                                           c->cfg.bankNum, c->dLoChan, c->dLoWt, c->cfg.usePower ? 1 : 0,
                                           c->cfg.takeLogs ? 1 : 0, stream);
    if (rc) return set_error("Wave2FBank kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCFBank2MFCCDevice(MFCCContext *c, const void *fbankDev, int numFrames, void *mfccDev,
                                    void *c0Dev, void *stream)
{
    if (!c || numFrames < 0 || (numFrames > 0 && (!fbankDev || !mfccDev)))
        return set_error("bad argument to MFCCFBank2MFCCDevice"), MFCC_E_ARG;
    if (c->cfg.fbankFlag) return set_error("the context is in filterbank mode (fbankFlag): it has no cepstral table"), MFCC_E_CONFIG;
    DEVICE_SCOPE(c->device);
    const int rc = launch_fbank2mfcc_batch(c->cfg.precision == MFCC_PREC_F64, fbankDev, mfccDev, c0Dev, numFrames,
                                           c->cfg.bankNum, c->M, c->dDct, c->dctPitch, c->tab.mfnorm, stream);
    if (rc) return set_error("FBank2MFCC kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCFBankTables(int frameSize, double sampPeriod, int numChans, double lopass, double hipass,
                               int doubleFFT, double alpha, double warpLowCut, double warpUpCut,
                               int *fftN, int *klo, int *khi, double *fres, double *cf, int *loChan, double *loWt,
                               char *passband, size_t cap)
{
    if (frameSize < 1 || numChans < 1 || !(sampPeriod > 0.0) || !(alpha > 0.0))
        return set_error("bad argument to MFCCFBankTables"), MFCC_E_ARG;
    Tables t;
    build_fbank(frameSize, sampPeriod, numChans, lopass, hipass, t, doubleFFT, alpha, warpLowCut, warpUpCut);
    if (t.fftN > (2 << kMaxLog2Nc)) return set_error("fftN %d is outside the supported range", t.fftN), MFCC_E_CONFIG;
    if (fftN) *fftN = t.fftN;
    if (klo) *klo = t.klo;
    if (khi) *khi = t.khi;
    if (fres) *fres = t.fres;
    if (cf) std::copy(t.cf.begin(), t.cf.end(), cf);
    if (loChan) std::copy(t.loChan.begin(), t.loChan.end(), loChan);
    if (loWt) std::copy(t.loWt.begin(), t.loWt.end(), loWt);
    if (passband)
        snprintf(passband, cap, "FFT passband %d to %d out of 1 to %d\nMel passband %f to %f\n", t.klo, t.khi, t.nc,
                 t.mlo, t.mhi); /* mfcc.c:74-75 */
    return MFCC_OK;
}

extern "C" int MFCCWave2FBankTablesDevice(int precision, const void *framesDev, int numFrames, int frameSize,
                                          int fftN, int klo, int khi, int numChans, const int *loChanDev,
                                          const void *loWtDev, int usePower, int takeLogs, void *fbankDev,
                                          void *teDev, void *te2Dev, void *specDev, void *stream)
{
    if (numFrames < 0 || (numFrames > 0 && (!framesDev || !fbankDev)) || !loChanDev || !loWtDev || numChans < 1 ||
        frameSize < 1 || frameSize > fftN || !pow2_between(fftN, 4, 2 << kMaxLog2Nc) || klo < 1 || khi > fftN / 2)
        return set_error("bad argument to MFCCWave2FBankTablesDevice"), MFCC_E_ARG;
    const int rc = launch_wave2fbank_batch(precision == MFCC_PREC_F64, framesDev, fbankDev, teDev, te2Dev, specDev,
                                           numFrames, frameSize, fftN, klo, khi, numChans, loChanDev, loWtDev,
                                           usePower ? 1 : 0, takeLogs ? 1 : 0, stream);
    if (rc) return set_error("Wave2FBank kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCFBank2MFCCDirectDevice(int precision, const void *fbankDev, int numFrames, int numChans, int n,
                                          void *mfccDev, void *c0Dev, void *stream)
{
    if (numFrames < 0 || numChans < 1 || n < 0 || (numFrames > 0 && (!fbankDev || (n > 0 && !mfccDev))))
        return set_error("bad argument to MFCCFBank2MFCCDirectDevice"), MFCC_E_ARG;
    const int rc = launch_fbank2mfcc_direct(precision == MFCC_PREC_F64, fbankDev, mfccDev, c0Dev, numFrames, numChans,
                                            n, stream);
    if (rc) return set_error("FBank2MFCC kernel launch failed (HIP error %d)", rc), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" double MFCCMel(int k, double fres) { return 1127 * log(1 + (k - 1) * fres); } /* mfcc.c:11-15 */

// ------------------------------------------------------------------------------------------
// file-level entry points
// ------------------------------------------------------------------------------------------

extern "C" void MFCCFree(void *p) { free(p); }

namespace {

// Contexts are reused across MFCCWapper calls with the same numeric options.  A context is
// immutable once created (host tables + device tables), so any number of threads may run plans on
// it at once (the reference's thread-per-file model, README.md:19,134).  A caller holds a shared
// reference for the length of its call: an entry pushed out of the cache is only destroyed when
// its last user lets go.
typedef std::shared_ptr<MFCCContext> CtxRef;
struct CtxCache {
    std::mutex mu;
    std::vector<std::pair<std::string, CtxRef>> items;
};
CtxCache g_cache;
constexpr size_t kCtxCacheSize = 8;

std::string cfg_key(const MFCCConfig &c, int device)
{
    MFCCConfig k = c;
    memset(k.fileList, 0, sizeof k.fileList);
    k.saveType = 0;
    k.numThreads = 0;
    std::string s(reinterpret_cast<const char *>(&k), sizeof k);
    s.append(reinterpret_cast<const char *>(&device), sizeof device);
    return s;
}

int cached_context(const MFCCConfig &cfgIn, CtxRef *out)
{
    MFCCConfig cfg;
    memset(&cfg, 0, sizeof cfg); // zero the padding so the key is well defined
    cfg.sampleRate = cfgIn.sampleRate, cfg.hipassfre = cfgIn.hipassfre, cfg.lowpassfre = cfgIn.lowpassfre;
    cfg.preemphasise = cfgIn.preemphasise, cfg.wlen = cfgIn.wlen, cfg.inc = cfgIn.inc, cfg.vecNum = cfgIn.vecNum;
    cfg.fbankFlag = cfgIn.fbankFlag, cfg.bankNum = cfgIn.bankNum, cfg.MFCCNum = cfgIn.MFCCNum;
    cfg.MFCC0thFlag = cfgIn.MFCC0thFlag, cfg.energyFlag = cfgIn.energyFlag;
    cfg.zeroCrossingFlag = cfgIn.zeroCrossingFlag, cfg.brightFlag = cfgIn.brightFlag;
    cfg.subBandEFlag = cfgIn.subBandEFlag, cfg.fftLength = cfgIn.fftLength, cfg.regreOrder = cfgIn.regreOrder;
    cfg.delwin = cfgIn.delwin, cfg.zeroMeanSigFlag = cfgIn.zeroMeanSigFlag, cfg.znormFlag = cfgIn.znormFlag;
    cfg.precision = cfgIn.precision;
    cfg.usePower = cfgIn.usePower, cfg.takeLogs = cfgIn.takeLogs, cfg.doubleFFT = cfgIn.doubleFFT;
    cfg.warpAlpha = cfgIn.warpAlpha, cfg.warpLowCut = cfgIn.warpLowCut, cfg.warpUpCut = cfgIn.warpUpCut;
    cfg.normEnergy = cfgIn.normEnergy;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return set_error("no HIP device available (the library has no CPU path)"), MFCC_E_DEVICE;
    const std::string key = cfg_key(cfg, device);
    std::lock_guard<std::mutex> lock(g_cache.mu);
    for (auto &it : g_cache.items)
        if (it.first == key) {
            *out = it.second;
            return MFCC_OK;
        }
    MFCCContext *c = nullptr;
    const int rc = MFCCContextCreate(&cfg, device, &c);
    if (rc) return rc;
    if (g_cache.items.size() >= kCtxCacheSize) g_cache.items.erase(g_cache.items.begin()); // oldest out
    g_cache.items.emplace_back(key, CtxRef(c, MFCCContextDestroy));
    *out = g_cache.items.back().second;
    return MFCC_OK;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
};

// Runs one packed batch (same channel count / sample size) and returns the rows as doubles.
int extract_packed(MFCCContext *c, const unsigned char *packed, size_t packedBytes, int channels,
                   int bps, const std::vector<int64_t> &off, const std::vector<int64_t> &ns,
                   std::vector<double> &rowsOut, std::vector<int64_t> &rowOff, std::vector<int64_t> &frames)
{
    MFCCPlan *plan = nullptr;
    int rc = MFCCPlanCreate(c, channels, bps, (int)off.size(), off.data(), ns.data(), &plan);
    if (rc) return rc;
    std::unique_ptr<MFCCPlan, void (*)(MFCCPlan *)> guard(plan, MFCCPlanDestroy);
    const int64_t rows = MFCCPlanTotalRows(plan);
    rowOff.resize(off.size());
    frames.resize(off.size());
    for (size_t u = 0; u < off.size(); ++u) {
        rowOff[u] = MFCCPlanRowOffset(plan, (int)u);
        frames[u] = MFCCPlanNumFrames(plan, (int)u);
    }
    rowsOut.assign((size_t)rows * c->step, 0.0);
    if (rows == 0) return MFCC_OK;
    const size_t el = c->cfg.precision == MFCC_PREC_F64 ? 8 : 4;
    DevBuf dPcm, dOut;
    HIP_TRY(hipMalloc(&dPcm.p, packedBytes + 16));
    HIP_TRY(hipMalloc(&dOut.p, (size_t)rows * c->step * el));
    HIP_TRY(hipMemcpy(dPcm.p, packed, packedBytes, hipMemcpyHostToDevice));
    rc = MFCCExtractDevice(c, plan, dPcm.p, packedBytes, dOut.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (el == 8) {
        HIP_TRY(hipMemcpy(rowsOut.data(), dOut.p, rowsOut.size() * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(rowsOut.size());
        HIP_TRY(hipMemcpy(tmp.data(), dOut.p, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) rowsOut[i] = (double)tmp[i];
    }
    return MFCC_OK;
}

} // namespace

static int wapper_mem(const MFCCConfig *cfg, const void *wavFile, size_t wavLen, const char *outPath, double **features,
                      long *numFrames, int *step);

extern "C" int MFCCWapperMem(const MFCCConfig *cfg, const void *wavFile, size_t wavLen,
                             const char *outPath, double **features, long *numFrames, int *step)
{
    const int rc = wapper_mem(cfg, wavFile, wavLen, outPath, features, numFrames, step);
    (rc == MFCC_OK ? g_count.filesOk : g_count.filesFailed) += 1;
    return rc;
}

static int wapper_mem(const MFCCConfig *cfg, const void *wavFile, size_t wavLen, const char *outPath, double **features,
                      long *numFrames, int *step)
{
    if (features) *features = nullptr;
    if (numFrames) *numFrames = 0;
    if (!cfg || !wavFile) return set_error("NULL argument"), MFCC_E_ARG;
    WavInfo w;
    int rc = wav_parse(static_cast<const unsigned char *>(wavFile), wavLen, w);
    if (rc) return rc;
    if (cfg->sampleRate != w.sampleRate) /* main.c:192-196 */
        return set_error("Sample rate not right (config %d, file %d)", cfg->sampleRate, w.sampleRate),
               MFCC_E_SAMPLE_RATE;
    CtxRef cref;
    if ((rc = cached_context(*cfg, &cref))) return rc;
    MFCCContext *c = cref.get();
    if (step) *step = c->step;
    std::vector<int64_t> off(1, 0), ns(1, w.numSamples), rowOff, frames;
    std::vector<double> rows;
    const unsigned char *data = static_cast<const unsigned char *>(wavFile) + w.dataOffset;
    const size_t dataBytes = (size_t)w.numSamples * w.channels * w.bytesPerSample;
    rc = extract_packed(c, data, dataBytes, w.channels, w.bytesPerSample, off, ns, rows, rowOff, frames);
    if (rc) return rc;
    if (numFrames) *numFrames = (long)frames[0];
    if (outPath && (rc = MFCCWriteFeatures(outPath, rows.data(), (long)frames[0], c->step, cfg->saveType)))
        return rc;
    if (features) {
        double *buf = static_cast<double *>(malloc(std::max<size_t>(rows.size(), 1) * sizeof(double)));
        if (!buf) return set_error("out of memory"), MFCC_E_NOMEM;
        if (!rows.empty()) memcpy(buf, rows.data(), rows.size() * sizeof(double));
        *features = buf;
    }
    return MFCC_OK;
}

extern "C" int MFCCWapper(const MFCCConfig *cfg, const char *wavPath, const char *outPath,
                          double **features, long *numFrames, int *step)
{
    if (features) *features = nullptr;
    if (numFrames) *numFrames = 0;
    if (!cfg || !wavPath) return set_error("NULL argument"), MFCC_E_ARG;
    std::vector<unsigned char> file;
    const int rc = read_file(wavPath, file);
    if (rc) {
        g_count.filesFailed += 1;
        return rc;
    }
    return MFCCWapperMem(cfg, file.data(), file.size(), outPath, features, numFrames, step);
}

namespace {

struct ListItem {
    std::string src, dst;
    std::vector<unsigned char> file;
    WavInfo w;
    int status = MFCC_OK;
    bool announced = false;
    std::string message;
};

// Runs fn(i) for i in [0, n) on up to nth threads (the calling thread is one of them).
template <typename F>
void parallel_for(size_t n, int nth, F fn)
{
    nth = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(nth, 1), n));
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nth; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

// Grow-only page-locked host buffer / device buffer
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MFCC_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return MFCC_OK;
    }
    ~PinnedBuf()
    {
        if (p) (void)hipHostFree(p);
    }
};
struct GrowDevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MFCC_OK;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return MFCC_OK;
    }
    ~GrowDevBuf()
    {
        if (p) (void)hipFree(p);
    }
};

// One batch of list items of a single PCM layout on its way through the stages
// read+pack -> device -> write.
struct ListBatch {
    std::vector<size_t> idx; // positions in the item list
    int channels = 0, bps = 0;
    std::vector<int64_t> off, ns, rowOff, frames;
    size_t pcmBytes = 0;
    PinnedBuf hPcm, hOut;
    int rc = MFCC_OK;
    std::string message;
};

template <typename T>
class BlockingQueue {
public:
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            q_.push_back(v);
        }
        cv_.notify_one();
    }
    T pop()
    {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        return v;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Device stage of one batch: upload, extract, download (asynchronous copies from and to
// page-locked memory on the stage's own stream).
int list_device_stage(MFCCContext *c, ListBatch &b, GrowDevBuf &dPcm, GrowDevBuf &dOut, hipStream_t st)
{
    MFCCPlan *plan = nullptr;
    int rc = MFCCPlanCreate(c, b.channels, b.bps, (int)b.off.size(), b.off.data(), b.ns.data(), &plan);
    if (rc) return rc;
    std::unique_ptr<MFCCPlan, void (*)(MFCCPlan *)> guard(plan, MFCCPlanDestroy);
    const size_t n = b.off.size();
    b.rowOff.resize(n);
    b.frames.resize(n);
    for (size_t u = 0; u < n; ++u) {
        b.rowOff[u] = MFCCPlanRowOffset(plan, (int)u);
        b.frames[u] = MFCCPlanNumFrames(plan, (int)u);
    }
    const int64_t rows = MFCCPlanTotalRows(plan);
    if (rows == 0) return MFCC_OK;
    const size_t el = c->cfg.precision == MFCC_PREC_F64 ? 8 : 4;
    const size_t outBytes = (size_t)rows * c->step * el;
    if ((rc = b.hOut.ensure(outBytes))) return rc;
    if ((rc = dPcm.ensure(b.pcmBytes + 16))) return rc;
    if ((rc = dOut.ensure(outBytes))) return rc;
    {
        TraceRange r("mfcc list: H2D");
        HIP_TRY(hipMemcpyAsync(dPcm.p, b.hPcm.p, b.pcmBytes, hipMemcpyHostToDevice, st));
    }
    if ((rc = MFCCExtractDevice(c, plan, dPcm.p, b.pcmBytes, dOut.p, st))) return rc;
    {
        TraceRange r("mfcc list: D2H + wait");
        HIP_TRY(hipMemcpyAsync(b.hOut.p, dOut.p, outBytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return MFCC_OK;
}

} // namespace

// The reference's whole run (main.c:98-300) as a three-stage pipeline over batches of files:
// the calling thread reads and parses files (numThreads readers) and packs their PCM payloads
// into page-locked memory; a device thread uploads, extracts and downloads; a writer thread
// formats and writes the feature files (numThreads writers).  Three batches are in flight, so
// file reads, PCIe copies, kernels and writers of neighbouring batches overlap.
extern "C" int MFCCWapperList(const char *iniPath)
{
    MFCCConfig cfg;
    int bad = 0;
    int rc = MFCCConfigFromIni(iniPath, &cfg, &bad);
    if (rc) return rc;
    const bool quiet = getenv("MFCC_QUIET") != nullptr;
    const bool timing = getenv("MFCC_LIST_TIMING") != nullptr;
    const double tList0 = now_s();
    if (!quiet) printf("finish reading the config.ini\n"); /* main.c:140 */
    CtxRef cref; // held until the list is done
    if ((rc = cached_context(cfg, &cref))) return rc;
    MFCCContext *c = cref.get();
    if (!quiet) {
        char text[256];
        MFCCContextPassbandText(c, text, sizeof text);
        fputs(text, stdout);
    }
    FILE *fl = fopen(cfg.fileList, "r");
    if (!fl) return set_error("open file list %s failed", cfg.fileList), MFCC_E_OPEN;
    std::vector<ListItem> items;
    char line[MFCC_MAX_PATH];
    while (fgets(line, MFCC_MAX_PATH, fl)) { /* main.c:173-177 */
        if (line[0] == '\n') break;
        char *src = strtok(line, "\t");
        char *dst = strtok(nullptr, "\n");
        ListItem it;
        it.src = src ? src : "";
        it.dst = dst ? dst : "";
        while (!it.dst.empty() && (it.dst.back() == '\r' || it.dst.back() == '\n')) it.dst.pop_back();
        if (!src || !dst || it.dst.empty()) {
            it.status = MFCC_E_ARG;
            it.message = "list line is not \"src<TAB>dst\"";
        }
        items.push_back(std::move(it));
    }
    fclose(fl);

    const int nth = std::max(1, cfg.numThreads);
    const size_t el = c->cfg.precision == MFCC_PREC_F64 ? 8 : 4;
    const size_t kBatchPcm = (size_t)64 << 20, kBatchOut = (size_t)256 << 20, kBatchFiles = 4096;
    // One device thread per GPU ([IO] numDevices, default: every visible one): each takes the next
    // loaded batch, so the files spread over the GPUs by themselves and nothing is exchanged
    // (every file's result is independent of the device and of the batch it travels in).
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) visible = 1;
    // numDevices < 0: that many device threads spread round-robin over the visible GPUs (several
    // streams per GPU, whose copies and kernels overlap)
    const int numDev = cfg.numDevices < 0 ? std::min(-cfg.numDevices, 64)
                                          : std::max(1, std::min(cfg.numDevices > 0 ? cfg.numDevices : visible, visible));
    const int kSlots = 2 * numDev + 1;
    std::vector<ListBatch> slots((size_t)kSlots);
    BlockingQueue<ListBatch *> freeQ, loadedQ, computedQ;
    for (auto &sl : slots) freeQ.push(&sl);
    double tRead = 0, tPack = 0, tDev = 0, tWrite = 0;
    std::mutex tMu;

    std::vector<std::thread> devThreads;
    for (int d = 0; d < numDev; ++d)
        devThreads.emplace_back([&, d]() {
            // device d of the visible ones, starting at the calling thread's current device
            const int dev = (c->device + d) % visible;
            GrowDevBuf dPcm, dOut;
            hipStream_t st = nullptr;
            CtxRef dref;
            bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreate(&st) == hipSuccess;
            if (ok) ok = (dev == c->device) ? (dref = cref, true) : cached_context(cfg, &dref) == MFCC_OK;
            std::string why = ok ? "" : (MFCCLastError()[0] ? MFCCLastError() : "no HIP device available (the library has no CPU path)");
            for (;;) {
                ListBatch *b = loadedQ.pop();
                if (!b) break;
                const double t0 = now_s();
                if (!ok) {
                    set_error("%s", why.c_str());
                    b->rc = MFCC_E_DEVICE;
                } else {
                    b->rc = list_device_stage(dref.get(), *b, dPcm, dOut, st);
                }
                if (b->rc) b->message = MFCCLastError();
                {
                    std::lock_guard<std::mutex> g(tMu);
                    tDev += now_s() - t0;
                }
                computedQ.push(b);
            }
            if (st) (void)hipStreamDestroy(st);
            computedQ.push(nullptr);
        });

    std::thread writeThread([&]() {
        for (int finished = 0; finished < numDev;) {
            ListBatch *b = computedQ.pop();
            if (!b) {
                ++finished;
                continue;
            }
            const double t0 = now_s();
            if (b->rc) {
                for (size_t i : b->idx) {
                    items[i].status = b->rc;
                    items[i].message = b->message;
                }
            } else {
                if (!quiet) /* main.c:252,274, once per file */
                    for (size_t u = 0; u < b->idx.size(); ++u) printf("post-processing...\nwriting the doc...\n");
                TraceRange r("mfcc list: write");
                parallel_for(b->idx.size(), nth, [&](size_t u) {
                    ListItem &it = items[b->idx[u]];
                    const size_t n = (size_t)b->frames[u] * c->step;
                    const size_t first = (size_t)b->rowOff[u] * c->step;
                    int wrc;
                    if (el == 8) {
                        wrc = MFCCWriteFeatures(it.dst.c_str(), static_cast<const double *>(b->hOut.p) + first,
                                                (long)b->frames[u], c->step, cfg.saveType);
                    } else {
                        std::vector<double> rows(std::max<size_t>(n, 1));
                        const float *src = static_cast<const float *>(b->hOut.p) + first;
                        for (size_t k = 0; k < n; ++k) rows[k] = (double)src[k];
                        wrc = MFCCWriteFeatures(it.dst.c_str(), rows.data(), (long)b->frames[u], c->step,
                                                cfg.saveType);
                    }
                    if (wrc) {
                        it.status = wrc;
                        it.message = MFCCLastError();
                    }
                });
            }
            tWrite += now_s() - t0;
            freeQ.push(b);
        }
    });

    // stage 1 on the calling thread
    size_t cursor = 0;
    while (cursor < items.size()) {
        ListBatch *b = freeQ.pop();
        b->idx.clear(), b->off.clear(), b->ns.clear();
        b->rc = MFCC_OK, b->message.clear(), b->pcmBytes = 0;
        // candidates: enough files (by size on disk) for one batch
        double t0 = now_s();
        size_t end = cursor, bytes = 0;
        while (end < items.size() && bytes < kBatchPcm && end - cursor < kBatchFiles) {
            ListItem &it = items[end];
            if (it.status == MFCC_OK) {
                if (!quiet && !it.announced) { /* main.c:180,184-187 */
                    printf("Convert %s to %s\n", it.src.c_str(), it.dst.c_str());
                    printf("including : \nMFCCNum\t%d\nenergyFlag %d\nzeroCrossingFlag %d\nbrightFlag %d\nsubBandEFlag %d\n",
                           c->M, cfg.energyFlag, cfg.zeroCrossingFlag, cfg.brightFlag, cfg.subBandEFlag);
                    printf("the frame feature dimension is %d\n", c->step);
                    printf("order %d\n", cfg.regreOrder);
                    printf("start...\n");
                }
                it.announced = true;
                if (!it.file.empty()) {
                    bytes += it.file.size();
                } else {
                    FILE *fp = fopen(it.src.c_str(), "rb");
                    if (fp) {
                        fseek(fp, 0, SEEK_END);
                        const long sz = ftell(fp);
                        fclose(fp);
                        bytes += sz > 0 ? (size_t)sz : 0;
                    }
                }
            }
            ++end;
        }
        parallel_for(end - cursor, nth, [&](size_t k) {
            ListItem &it = items[cursor + k];
            if (it.status != MFCC_OK || !it.file.empty()) return;
            int r = read_file(it.src.c_str(), it.file);
            if (!r) r = wav_parse(it.file.data(), it.file.size(), it.w);
            if (!r && cfg.sampleRate != it.w.sampleRate) { /* main.c:192-196 */
                set_error("Sample rate not right (config %d, file %d)", cfg.sampleRate, it.w.sampleRate);
                r = MFCC_E_SAMPLE_RATE;
            }
            if (r) {
                it.status = r;
                it.message = MFCCLastError();
                std::vector<unsigned char>().swap(it.file);
            }
        });
        tRead += now_s() - t0;
        t0 = now_s();
        // the batch: the leading run of readable files with one PCM layout; files behind a layout
        // change stay loaded for the next batch
        size_t outBytes = 0, i = cursor;
        for (; i < end; ++i) {
            ListItem &it = items[i];
            if (it.status != MFCC_OK) continue;
            if (b->idx.empty()) {
                b->channels = it.w.channels;
                b->bps = it.w.bytesPerSample;
            } else if (b->channels != it.w.channels || b->bps != it.w.bytesPerSample ||
                       outBytes >= kBatchOut) {
                break;
            }
            if (!quiet) /* main.c:212 */
                printf("total coef size: %ld\n", (long)MFCCNumFrames(it.w.numSamples, cfg.wlen, cfg.inc) * c->step);
            b->idx.push_back(i);
            b->off.push_back((int64_t)b->pcmBytes);
            b->ns.push_back(it.w.numSamples);
            b->pcmBytes += ((size_t)it.w.numSamples * b->channels * b->bps + 15) & ~(size_t)15;
            outBytes += (size_t)MFCCNumFrames(it.w.numSamples, cfg.wlen, cfg.inc) * c->step * el;
        }
        cursor = i;
        if (b->idx.empty()) {
            freeQ.push(b);
            continue;
        }
        int prc = b->hPcm.ensure(std::max<size_t>(b->pcmBytes, 16));
        if (prc) {
            b->rc = prc;
            b->message = MFCCLastError();
        } else {
            parallel_for(b->idx.size(), nth, [&](size_t u) {
                ListItem &it = items[b->idx[u]];
                memcpy(static_cast<unsigned char *>(b->hPcm.p) + b->off[u], it.file.data() + it.w.dataOffset,
                       (size_t)it.w.numSamples * b->channels * b->bps);
                std::vector<unsigned char>().swap(it.file);
            });
        }
        tPack += now_s() - t0;
        if (b->rc)
            computedQ.push(b); // straight to the stage that records the failure
        else
            loadedQ.push(b);
    }
    for (int d = 0; d < numDev; ++d) loadedQ.push(nullptr);
    for (auto &th : devThreads) th.join();
    writeThread.join();
    if (timing)
        fprintf(stderr, "MFCCWapperList stages (busy seconds): read %.3f pack %.3f device %.3f (%d GPU%s) write %.3f\n",
                tRead, tPack, tDev, numDev, numDev > 1 ? "s" : "", tWrite);

    int failed = 0;
    for (const ListItem &it : items)
        if (it.status != MFCC_OK) {
            ++failed;
            fprintf(stderr, "mfcc: %s: %s (code %d)\n", it.src.c_str(), it.message.c_str(), it.status);
        }
    g_count.filesFailed += (uint64_t)failed;
    g_count.filesOk += (uint64_t)(items.size() - (size_t)failed);
    g_count.listNanos += (uint64_t)((now_s() - tList0) * 1e9);
    if (failed) set_error("%d of %zu files failed", failed, items.size());
    return failed;
}

// ------------------------------------------------------------------------------------------
// one long recording over several GPUs from one process
// ------------------------------------------------------------------------------------------

namespace {

// All shard threads wait here; reusable.
class ShardBarrier {
public:
    explicit ShardBarrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu_);
        const int gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return gen_ != gen; });
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
};

} // namespace

// The frame-sharded mode for one very long recording, driven from one process: shard g owns frames
// [T g / S, T (g+1) / S), runs on its own GPU from its own host thread, recomputes the halo frames
// its regression deltas need (MFCCShardGeometry), and the shards exchange only the integer channel
// sums (zeroMeanSigFlag) and the z-norm column sums (znormFlag): each shard's vector goes to the
// host, every shard adds all of them in shard order (so all hold bit-identical totals, whatever the
// timing) and goes on.  The per-file semantics are those of main.c:198-266 on the whole recording.
extern "C" int MFCCExtractRecording(const MFCCConfig *cfgIn, const void *pcm, int channels, int bytesPerSample,
                                    int64_t numSamples, int numShards, const int *devices, void *features,
                                    int64_t *numFrames)
{
    if (numFrames) *numFrames = 0;
    if (!cfgIn || numSamples < 0 || (numSamples > 0 && !pcm) || numShards < 0 || numShards > 64)
        return set_error("bad argument to MFCCExtractRecording"), MFCC_E_ARG;
    if (channels < 1 || channels > 2 || bytesPerSample < 1 || bytesPerSample > 4)
        return set_error("unsupported PCM layout: %d channels x %d bytes", channels, bytesPerSample), MFCC_E_WAV;
    const MFCCConfig cfg = *cfgIn;
    const double tRec0 = now_s();
    int rc = validate_config(cfg);
    if (rc) return rc;
    int visible = 0, current = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1 || hipGetDevice(&current) != hipSuccess)
        return set_error("no HIP device available (the library has no CPU path)"), MFCC_E_DEVICE;
    const int S = numShards > 0 ? numShards : visible;
    const int64_t T = MFCCNumFrames((long)numSamples, cfg.wlen, cfg.inc);
    if (numFrames) *numFrames = T;
    if (T == 0) return MFCC_OK;
    if (!features) return set_error("NULL feature buffer"), MFCC_E_ARG;
    const int step = MFCCStep(&cfg);
    const size_t el = cfg.precision == MFCC_PREC_F64 ? 8 : 4;
    const int64_t bpi = (int64_t)channels * bytesPerSample;
    std::vector<std::array<int64_t, 4>> chanSums((size_t)S);
    std::vector<std::vector<double>> colSums((size_t)S, std::vector<double>((size_t)2 * step + 1, 0.0));
    std::vector<int> status((size_t)S, MFCC_OK);
    std::vector<std::string> message((size_t)S);
    std::atomic<bool> failed(false);
    ShardBarrier barrier(S);
    const size_t nSums = (size_t)2 * step + 1;

    // The two exchanges (integer channel sums for the zero-mean option, z-norm column sums) go over
    // RCCL -- all-gather on the GPUs, then a rank-ordered sum on every GPU (MFCCShardExchange) -- when
    // the shards sit on distinct GPUs; shards that share a GPU (RCCL refuses one device twice in a
    // communicator) exchange through host memory, added in the same order.
    std::vector<int> devOf((size_t)S);
    for (int g = 0; g < S; ++g) devOf[(size_t)g] = devices ? devices[g] : (current + g) % visible;
    bool distinct = true;
    for (int g = 0; g < S; ++g)
        for (int k = 0; k < g; ++k) distinct = distinct && devOf[(size_t)g] != devOf[(size_t)k];
    const int rcclMode = tuning().rccl;
    const bool needExchange = cfg.znormFlag || cfg.zeroMeanSigFlag;
    bool useRccl = needExchange && rcclMode >= 0 && distinct && (S > 1 || rcclMode > 0);
    if (rcclMode > 0 && needExchange && !distinct)
        return set_error("the RCCL exchange needs every shard on its own GPU"), MFCC_E_ARG;
    std::vector<void *> comms((size_t)S, nullptr);
    if (useRccl) {
        rc = rccl_load();
        if (!rc) rc = rccl_comm_init_all(comms.data(), S, devOf.data());
        if (rc && rcclMode > 0) return rc;
        if (rc) useRccl = false; // (no RCCL on this machine: host memory carries the sums)
        (void)hipSetDevice(current);
    }

    // Page-locking the caller's buffers for the call makes the shards' copies true DMA transfers that
    // overlap the other shards' kernels (several shards per GPU form a pipeline of their own).
    bool pinnedPcm = false, pinnedOut = false;
    const int pinMode = tuning().pinHost;
    if (pinMode > 0 || (pinMode == 0 && (size_t)(numSamples * bpi) >= ((size_t)64 << 20))) {
        pinnedPcm = hipHostRegister(const_cast<void *>(pcm), (size_t)(numSamples * bpi), hipHostRegisterDefault) == hipSuccess;
        pinnedOut = hipHostRegister(features, (size_t)T * step * el, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }

    auto shard = [&](int g) {
        const int dev = devOf[(size_t)g];
        const int64_t fb = T * g / S, fe = T * (g + 1) / S;
        int64_t sb = 0, sc = 0, rows = 0, ownBegin = 0, ownRows = 0;
        int lead = 0;
        MFCCShardGeometry(&cfg, numSamples, fb, fe, &sb, &sc, &rows, &ownBegin, &ownRows, &lead);
        const bool empty = fe == fb;
        // instants whose channel sums this shard contributes: a partition of the recording
        const int64_t partBegin = g == 0 ? 0 : fb * cfg.inc, partEnd = g == S - 1 ? numSamples : fe * cfg.inc;
        const int64_t upBegin = empty ? partBegin : std::min(sb, partBegin);
        const int64_t upEnd = std::max(empty ? partEnd : sb + sc, partEnd);
        CtxRef ctx;
        DevBuf dPcm, dOut, dSums, dGather;
        MFCCPlan *plan = nullptr;
        hipStream_t st = nullptr;
        auto fail = [&](int code) {
            status[(size_t)g] = code;
            message[(size_t)g] = MFCCLastError();
            failed = true;
        };
        auto hip_ok = [&](hipError_t e) {
            if (e == hipSuccess) return true;
            set_error("HIP error %d (%s) in shard %d on device %d", (int)e, hipGetErrorString(e), g, dev);
            fail(MFCC_E_DEVICE);
            return false;
        };
        // phase 1: device, context, PCM upload, channel sums
        bool ok = hip_ok(hipSetDevice(dev)) && hip_ok(hipStreamCreate(&st));
        if (ok) {
            const int r = cached_context(cfg, &ctx);
            if (r) fail(r), ok = false;
        }
        if (ok) ok = hip_ok(hipMalloc(&dPcm.p, (size_t)std::max<int64_t>((upEnd - upBegin) * bpi, 16) + 16)) &&
                     hip_ok(hipMalloc(&dSums.p, nSums * sizeof(double) + 64)) &&
                     hip_ok(hipMemsetAsync(dSums.p, 0, nSums * sizeof(double) + 64, st)) &&
                     (!useRccl || hip_ok(hipMalloc(&dGather.p, (size_t)S * nSums * sizeof(double)))) &&
                     hip_ok(hipMemcpyAsync(dPcm.p, static_cast<const unsigned char *>(pcm) + upBegin * bpi,
                                           (size_t)((upEnd - upBegin) * bpi), hipMemcpyHostToDevice, st));
        chanSums[(size_t)g] = {0, 0, 0, 0};
        if (ok && cfg.zeroMeanSigFlag) {
            const int r = MFCCChannelSums(ctx.get(), dPcm.p, channels, bytesPerSample, partBegin - upBegin,
                                          partEnd - partBegin, static_cast<int64_t *>(dSums.p), st);
            if (r) fail(r), ok = false;
            else if (!useRccl)
                ok = hip_ok(hipMemcpyAsync(chanSums[(size_t)g].data(), dSums.p, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st)) &&
                     hip_ok(hipStreamSynchronize(st));
        }
        barrier.wait();
        // phase 2: plan, static features and deltas, column sums
        int64_t total2[4] = {0, 0, 0, 0};
        ok = ok && !failed;
        if (cfg.zeroMeanSigFlag && useRccl) {
            if (ok) { // (failed is the same for every shard after the barrier: all of them enter the collective, or none)
                const int r = MFCCShardExchange(comms[(size_t)g], 1, dSums.p, 4, dGather.p, st);
                if (r) fail(r), ok = false;
                else
                    ok = hip_ok(hipMemcpyAsync(total2, dSums.p, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st)) &&
                         hip_ok(hipStreamSynchronize(st));
            }
        } else {
            for (int k = 0; k < S; ++k)
                for (int j = 0; j < 4; ++j) total2[j] += chanSums[(size_t)k][(size_t)j];
        }
        if (ok && useRccl && cfg.znormFlag) ok = hip_ok(hipMemsetAsync(dSums.p, 0, nSums * sizeof(double), st)); // (an empty shard adds zeros)
        if (ok && !empty) {
            int r = MFCCPlanCreateShard(ctx.get(), channels, bytesPerSample, numSamples, fb, fe,
                                        cfg.zeroMeanSigFlag ? total2 : nullptr, &plan);
            if (!r) ok = hip_ok(hipMalloc(&dOut.p, (size_t)rows * step * el));
            const unsigned char *shardPcm = static_cast<const unsigned char *>(dPcm.p) + (sb - upBegin) * bpi;
            if (!r && ok) r = MFCCExtractDevice(ctx.get(), plan, shardPcm, (size_t)(sc * bpi), dOut.p, st);
            if (!r && ok && cfg.znormFlag) {
                r = MFCCShardColumnSums(ctx.get(), plan, dOut.p, static_cast<double *>(dSums.p), st);
                if (!r && !useRccl)
                    ok = hip_ok(hipMemcpyAsync(colSums[(size_t)g].data(), dSums.p, nSums * sizeof(double), hipMemcpyDeviceToHost, st)) &&
                         hip_ok(hipStreamSynchronize(st));
            }
            if (r) fail(r), ok = false;
        }
        barrier.wait();
        // phase 3: normalise with the totals, own rows back to the host
        ok = ok && !failed;
        if (ok && cfg.znormFlag && useRccl) {
            const int r = MFCCShardExchange(comms[(size_t)g], 0, dSums.p, (int)nSums, dGather.p, st);
            if (r) fail(r), ok = false;
        }
        if (ok && !empty) {
            int r = MFCC_OK;
            if (cfg.znormFlag) {
                if (!useRccl) {
                    std::vector<double> tot(nSums, 0.0);
                    for (int k = 0; k < S; ++k)
                        for (size_t j = 0; j < tot.size(); ++j) tot[j] += colSums[(size_t)k][j];
                    ok = hip_ok(hipMemcpyAsync(dSums.p, tot.data(), tot.size() * sizeof(double), hipMemcpyHostToDevice, st)) &&
                         hip_ok(hipStreamSynchronize(st)); // (tot leaves scope)
                }
                if (ok) r = MFCCShardNormalize(ctx.get(), plan, dOut.p, static_cast<const double *>(dSums.p), st);
            }
            if (!r && ok)
                ok = hip_ok(hipMemcpyAsync(static_cast<unsigned char *>(features) + (size_t)fb * step * el,
                                           static_cast<const unsigned char *>(dOut.p) + (size_t)ownBegin * step * el,
                                           (size_t)ownRows * step * el, hipMemcpyDeviceToHost, st)) &&
                     hip_ok(hipStreamSynchronize(st));
            if (r) fail(r);
        }
        if (ok && useRccl) ok = hip_ok(hipStreamSynchronize(st)); // (an empty shard's part of the collectives)
        if (plan) MFCCPlanDestroy(plan);
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < S; ++g) threads.emplace_back(shard, g);
    shard(0);
    for (auto &th : threads) th.join();
    g_count.recordingNanos += (uint64_t)((now_s() - tRec0) * 1e9);
    for (void *cm : comms) rccl_comm_destroy(cm);
    if (pinnedPcm) (void)hipHostUnregister(const_cast<void *>(pcm));
    if (pinnedOut) (void)hipHostUnregister(features);
    (void)hipSetDevice(current);
    for (int g = 0; g < S; ++g)
        if (status[(size_t)g] != MFCC_OK) return set_error("shard %d: %s", g, message[(size_t)g].c_str()), status[(size_t)g];
    return MFCC_OK;
}
