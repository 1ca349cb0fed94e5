// libmfcc_compat.so: the reference's lower-level API (mfcc.h:29-74) under its own names, on top of
// the batched device entry points of libmfcc_amd.so with a batch of one frame.  Kept out of the
// main library so that short global names such as FFT or Mel are only present in processes that ask
// for them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/mfcc_amd.h"
#include "../../include/mfcc_compat.h"

namespace {

thread_local int g_status = 0;

// device tables of one InitFBank result, found again through the address of its loWt vector
// (Wave2FBank receives the record by value)
struct BankEntry {
    const void *key = nullptr;
    int *dLoChan = nullptr;
    double *dLoWt = nullptr;
};
std::mutex g_mu;
std::vector<BankEntry> g_banks;

// Grow-only device scratch and page-locked host staging shared by all calls, one private stream
// (all guarded by g_mu).  They live on the GPU that was current at the library's first call; later
// calls switch to that GPU for their duration, whatever the caller has made current since.
struct Scratch {
    double *p = nullptr;
    size_t cap = 0;
    double *ensure(size_t n)
    {
        if (n <= cap) return p;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        if (hipMalloc((void **)&p, n * sizeof(double)) != hipSuccess) return nullptr;
        cap = n;
        return p;
    }
} g_scratch;
struct Pinned {
    double *p = nullptr;
    size_t cap = 0;
    double *ensure(size_t n)
    {
        if (n <= cap) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr, cap = 0;
        if (hipHostMalloc((void **)&p, n * sizeof(double), hipHostMallocDefault) != hipSuccess) return nullptr;
        cap = n;
        return p;
    }
} g_pinned;
hipStream_t g_stream = nullptr;
int g_device = -1;

// Makes the library's GPU current for one call (g_mu held) and creates the stream on first use.
struct DeviceCall {
    int prev = -1;
    bool ok = false;
    DeviceCall()
    {
        if (hipGetDevice(&prev) != hipSuccess) return;
        if (g_device < 0) g_device = prev;
        if (prev != g_device && hipSetDevice(g_device) != hipSuccess) return;
        if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return;
        ok = true;
    }
    ~DeviceCall()
    {
        if (prev >= 0 && prev != g_device) (void)hipSetDevice(prev);
    }
};

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int fail(int code, const char *what)
{
    g_status = code;
    fprintf(stderr, "%s: %s\n", what, MFCCLastError()[0] ? MFCCLastError() : "device error");
    return code;
}

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_status = MFCC_E_DEVICE;
    fprintf(stderr, "%s: HIP error %d (%s)\n", what, (int)e, hipGetErrorString(e));
    return false;
}

} // namespace

extern "C" Vector MFCCCreateVector(int n)
{
    Vector v = static_cast<Vector>(calloc((size_t)(n > 0 ? n : 0) + 1, sizeof(double)));
    if (v) *reinterpret_cast<int *>(v) = n;
    return v;
}
extern "C" ShortVec MFCCCreateIntVec(int n)
{
    ShortVec v = static_cast<ShortVec>(calloc((size_t)(n > 0 ? n : 0) + 1, sizeof(int)));
    if (v) v[0] = n;
    return v;
}
extern "C" int MFCCVectorSize(const void *v) { return v ? *static_cast<const int *>(v) : 0; }
extern "C" void MFCCFreeVector(void *v) { free(v); }
extern "C" int MFCCCompatStatus(void) { return g_status; }

extern "C" double Mel(int k, double fres) { return MFCCMel(k, fres); }

extern "C" FBankInfo InitFBank(int frameSize, double sampPeriod, int numChans, double lopass, double hipass,
                               int usePower, int takeLogs, int doubleFFT, double alpha, double warpLowCut,
                               double warpUpCut)
{
    FBankInfo fb;
    memset(&fb, 0, sizeof fb);
    g_status = 0;
    int fftN = 0;
    char text[256] = "";
    if (MFCCFBankTables(frameSize, sampPeriod, numChans, lopass, hipass, doubleFFT, alpha, warpLowCut, warpUpCut,
                        &fftN, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0)) {
        fail(MFCC_E_ARG, "InitFBank");
        return fb;
    }
    const int nby2 = fftN / 2;
    fb.frameSize = frameSize, fb.numChans = numChans, fb.sampPeriod = sampPeriod;
    fb.usePower = usePower, fb.takeLogs = takeLogs, fb.fftN = fftN;
    fb.cf = MFCCCreateVector(numChans + 1);
    fb.loChan = MFCCCreateIntVec(nby2);
    fb.loWt = MFCCCreateVector(nby2);
    fb.x = MFCCCreateVector(fftN);
    if (!fb.cf || !fb.loChan || !fb.loWt || !fb.x) {
        g_status = MFCC_E_NOMEM;
        MFCCFreeFBank(&fb);
        return fb;
    }
    MFCCFBankTables(frameSize, sampPeriod, numChans, lopass, hipass, doubleFFT, alpha, warpLowCut, warpUpCut, &fb.fftN,
                    &fb.klo, &fb.khi, &fb.fres, fb.cf + 1, fb.loChan + 1, fb.loWt + 1, text, sizeof text);
    fputs(text, stdout); /* mfcc.c:74-75 */
    BankEntry e;
    e.key = fb.loWt;
    if (!hip_ok(hipMalloc((void **)&e.dLoChan, (size_t)nby2 * sizeof(int)), "InitFBank") ||
        !hip_ok(hipMalloc((void **)&e.dLoWt, (size_t)nby2 * sizeof(double)), "InitFBank") ||
        !hip_ok(hipMemcpy(e.dLoChan, fb.loChan + 1, (size_t)nby2 * sizeof(int), hipMemcpyHostToDevice), "InitFBank") ||
        !hip_ok(hipMemcpy(e.dLoWt, fb.loWt + 1, (size_t)nby2 * sizeof(double), hipMemcpyHostToDevice), "InitFBank")) {
        (void)hipFree(e.dLoChan);
        (void)hipFree(e.dLoWt);
        return fb; // host tables stay valid; Wave2FBank will report the missing device tables
    }
    std::lock_guard<std::mutex> lock(g_mu);
    g_banks.push_back(e);
    return fb;
}

extern "C" void MFCCFreeFBank(FBankInfo *info)
{
    if (!info) return;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (size_t i = 0; i < g_banks.size(); ++i)
            if (g_banks[i].key == info->loWt) {
                (void)hipFree(g_banks[i].dLoChan);
                (void)hipFree(g_banks[i].dLoWt);
                g_banks.erase(g_banks.begin() + (long)i);
                break;
            }
    }
    free(info->cf), free(info->loChan), free(info->loWt), free(info->x);
    info->cf = info->loWt = info->x = nullptr;
    info->loChan = nullptr;
}

extern "C" void Wave2FBank(Vector s, Vector fbank, double *te, double *te2, FBankInfo info)
{
    g_status = 0;
    if (info.frameSize != MFCCVectorSize(s)) printf("Wave2FBank: frame size mismatch");     /* mfcc.c:133-134 */
    if (info.numChans != MFCCVectorSize(fbank)) printf("Wave2FBank: num channels mismatch"); /* mfcc.c:135-136 */
    if (info.frameSize != MFCCVectorSize(s) || info.numChans != MFCCVectorSize(fbank)) {
        g_status = MFCC_E_ARG;
        return;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    const BankEntry *be = nullptr;
    for (const BankEntry &e : g_banks)
        if (e.key == info.loWt) be = &e;
    if (!be) {
        g_status = MFCC_E_ARG;
        fprintf(stderr, "Wave2FBank: the FBankInfo record does not come from InitFBank\n");
        return;
    }
    const int W = info.frameSize, P = info.numChans, N = info.fftN;
    DeviceCall dc;
    // device scratch and host staging, both: frame | fbank | te | te2 | spectrum
    const size_t total = (size_t)W + P + 2 + N;
    double *d = dc.ok ? g_scratch.ensure(total) : nullptr;
    double *h = d ? g_pinned.ensure(total) : nullptr;
    if (!h) {
        hip_ok(hipErrorOutOfMemory, "Wave2FBank");
        return;
    }
    double *dFrame = d, *dFbank = d + W, *dTe = dFbank + P, *dTe2 = dTe + 1, *dSpec = dTe2 + 1;
    memcpy(h, s + 1, (size_t)W * sizeof(double));
    if (!hip_ok(hipMemcpyAsync(dFrame, h, (size_t)W * sizeof(double), hipMemcpyHostToDevice, g_stream), "Wave2FBank")) return;
    if (MFCCWave2FBankTablesDevice(MFCC_PREC_F64, dFrame, 1, W, N, info.klo, info.khi, P, be->dLoChan, be->dLoWt,
                                   info.usePower, info.takeLogs, dFbank, dTe, dTe2, dSpec, g_stream)) {
        fail(MFCC_E_DEVICE, "Wave2FBank");
        return;
    }
    double *out = h + W;
    if (!hip_ok(hipMemcpyAsync(out, dFbank, ((size_t)P + 2 + N) * sizeof(double), hipMemcpyDeviceToHost, g_stream), "Wave2FBank") ||
        !hip_ok(hipStreamSynchronize(g_stream), "Wave2FBank"))
        return;
    memcpy(fbank + 1, out, (size_t)P * sizeof(double));
    if (te) *te = out[P];
    if (te2) *te2 = out[P + 1];
    if (info.x) memcpy(info.x + 1, out + P + 2, (size_t)N * sizeof(double));
}

static void cepstra(Vector fbank, Vector c, int n, double *c0)
{
    g_status = 0;
    const int P = MFCCVectorSize(fbank);
    if (P < 1 || n < 0) {
        g_status = MFCC_E_ARG;
        return;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceCall dc;
    const size_t total = (size_t)P + n + 1;
    double *d = dc.ok ? g_scratch.ensure(total) : nullptr;
    double *h = d ? g_pinned.ensure(total) : nullptr;
    if (!h) {
        hip_ok(hipErrorOutOfMemory, "FBank2MFCC");
        return;
    }
    memcpy(h, fbank + 1, (size_t)P * sizeof(double));
    if (!hip_ok(hipMemcpyAsync(d, h, (size_t)P * sizeof(double), hipMemcpyHostToDevice, g_stream), "FBank2MFCC")) return;
    if (MFCCFBank2MFCCDirectDevice(MFCC_PREC_F64, d, 1, P, n, d + P, c0 ? d + P + n : nullptr, g_stream)) {
        fail(MFCC_E_DEVICE, "FBank2MFCC");
        return;
    }
    if (!hip_ok(hipMemcpyAsync(h + P, d + P, ((size_t)n + 1) * sizeof(double), hipMemcpyDeviceToHost, g_stream), "FBank2MFCC") ||
        !hip_ok(hipStreamSynchronize(g_stream), "FBank2MFCC"))
        return;
    if (c && n > 0) memcpy(c + 1, h + P, (size_t)n * sizeof(double));
    if (c0) *c0 = h[P + n];
}

extern "C" void FBank2MFCC(Vector fbank, Vector c, int n) { cepstra(fbank, c, n, nullptr); }

extern "C" double FBank2C0(Vector fbank)
{
    double c0 = 0.0;
    cepstra(fbank, nullptr, 0, &c0);
    return c0;
}

static void transform(Vector s, int invert, bool real)
{
    g_status = 0;
    const int n = MFCCVectorSize(s);
    const char *what = real ? "Realft" : "FFT";
    if (!is_pow2(n) || n < (real ? 4 : 2)) { // (the reference's loops assume it: mfcc.c:216-229,267)
        g_status = MFCC_E_ARG;
        fprintf(stderr, "%s: the vector size %d is not a power of two >= %d\n", what, n, real ? 4 : 2);
        return;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceCall dc;
    double *d = dc.ok ? g_scratch.ensure((size_t)n) : nullptr;
    double *h = d ? g_pinned.ensure((size_t)n) : nullptr;
    if (!h) {
        hip_ok(hipErrorOutOfMemory, what);
        return;
    }
    memcpy(h, s + 1, (size_t)n * sizeof(double));
    if (!hip_ok(hipMemcpyAsync(d, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_stream), what)) return;
    const int rc = real ? MFCCRealftDevice(MFCC_PREC_F64, d, 1, n, g_stream) : MFCCFFTDevice(MFCC_PREC_F64, d, 1, n, invert, g_stream);
    if (rc) {
        fail(rc, what);
        return;
    }
    if (!hip_ok(hipMemcpyAsync(h, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_stream), what) ||
        !hip_ok(hipStreamSynchronize(g_stream), what))
        return;
    memcpy(s + 1, h, (size_t)n * sizeof(double));
}

extern "C" void FFT(Vector s, int invert) { transform(s, invert, false); }
extern "C" void Realft(Vector s) { transform(s, 0, true); }
