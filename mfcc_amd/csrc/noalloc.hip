// MFCCColumnSumsNoAlloc: the z-norm column sums of MFCCShardColumnSums for callers that cannot
// allocate device memory (code under HIP graph capture, tight serving loops).  It runs the same
// two kernels; their scratch (one utterance descriptor, the per-chunk partial sums) lives in a
// device array of this library instead of in a plan, so the call needs neither a context nor a
// plan and makes no hipMalloc / hipFree.
//
// Every device the library runs on holds its own copy of the array until the process ends.  All
// calls on one device share it: calls that overlap on one device (two streams, two threads) race.
//
// Under stream capture: the first call on a device looks the array's address up
// (hipGetSymbolAddress) and may make the runtime load this library's kernels, neither of which
// belongs inside a capture.  MFCCColumnSumsNoAllocWarmup does both eagerly, once per device, before
// the capture begins; after it, a call only enqueues kernels (and one memset when rows = 0).
#include <hip/hip_runtime.h>

#include <mutex>

#include "mfcc_host.h"

namespace mfcc {

namespace {

constexpr size_t kHeadBytes = 256;                   // the utterance descriptor, padded
constexpr size_t kPartialBytes = 4u << 20;           // per-chunk partial sums: 4 MiB
static __device__ __align__(256) unsigned char scratch[kHeadBytes + kPartialBytes];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
unsigned char *g_scratch[kMaxDevices] = {}; // device address of `scratch` on each device

// One utterance that owns rows 0 .. rows-1 of the matrix, as the statistics kernels expect it.
__global__ void k_single_utt(UttDesc *d, int rows)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        UttDesc u = {};
        u.numFrames = rows;
        u.ownCount = rows;
        *d = u;
    }
}

// Device address of the scratch array on the current device (looked up once per device).
int scratch_of_current_device(unsigned char **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error("hipGetDevice failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
    if (dev < 0 || dev >= kMaxDevices) return set_error("device %d: only devices 0..%d are supported", dev, kMaxDevices - 1), MFCC_E_ARG;
    std::lock_guard<std::mutex> g(g_lock);
    if (!g_scratch[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return set_error("hipGetSymbolAddress failed for the scratch array (HIP error %d)", (int)e), MFCC_E_DEVICE;
        g_scratch[dev] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch[dev];
    return MFCC_OK;
}

// The device the work will run on: the stream's, which must be the current one (the scratch
// address and the kernels' module belong to the current device).
int check_stream_device(hipStream_t st)
{
    if (!st) return MFCC_OK; // the null stream is the current device's
    int cur = -1;
    hipDevice_t sdev = -1;
    if (hipGetDevice(&cur) != hipSuccess || hipStreamGetDevice(st, &sdev) != hipSuccess) return MFCC_OK; // (cannot tell: let the launch report it)
    if ((int)sdev != cur)
        return set_error("MFCCColumnSumsNoAlloc: the stream belongs to device %d, the current device is %d", (int)sdev, cur), MFCC_E_ARG;
    return MFCC_OK;
}

} // namespace

} // namespace mfcc

using namespace mfcc;

extern "C" int MFCCColumnSumsNoAllocWarmup(void)
{
    unsigned char *base = nullptr;
    const int rc = scratch_of_current_device(&base);
    if (rc) return rc;
    // one run of each kernel on a one-row matrix inside the scratch itself: loads the code objects
    float *probe = reinterpret_cast<float *>(base + kHeadBytes + kPartialBytes - 256);
    double *sums = reinterpret_cast<double *>(base + kHeadBytes + kPartialBytes - 128);
    hipError_t e = hipMemsetAsync(probe, 0, 256, nullptr);
    if (e != hipSuccess) return set_error("hipMemsetAsync failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
    for (int prec : {MFCC_PREC_F32, MFCC_PREC_F64}) {
        const int r = MFCCColumnSumsNoAlloc(prec, probe, 1, 1, sums, nullptr);
        if (r) return r;
    }
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return set_error("warm-up kernels failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
    return MFCC_OK;
}

extern "C" int MFCCColumnSumsNoAlloc(int precision, const void *dataDev, int64_t rows, int step, double *sumsDev,
                                     void *stream)
{
    if (rows < 0 || step < 1 || !sumsDev || (rows > 0 && !dataDev) || (precision != MFCC_PREC_F32 && precision != MFCC_PREC_F64))
        return set_error("bad argument to MFCCColumnSumsNoAlloc"), MFCC_E_ARG;
    const int64_t chunks = (rows + kStatRows - 1) / kStatRows;
    if (rows > 0x7fffffffLL || (uint64_t)chunks * 2 * (uint64_t)step * sizeof(double) > kPartialBytes)
        return set_error("MFCCColumnSumsNoAlloc: %lld rows x %d columns need %llu bytes of partial sums, the fixed scratch holds %zu "
                         "(use MFCCShardColumnSums with a plan for larger matrices)",
                         (long long)rows, step, (unsigned long long)chunks * 2 * (unsigned long long)step * sizeof(double), kPartialBytes),
               MFCC_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *base = nullptr;
    int rc = check_stream_device(st);
    if (rc) return rc;
    rc = scratch_of_current_device(&base);
    if (rc) return rc;
    hipError_t e;
    if (rows == 0) {
        e = hipMemsetAsync(sumsDev, 0, (size_t)(2 * step + 1) * sizeof(double), st);
        if (e != hipSuccess) return set_error("hipMemsetAsync failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
        return MFCC_OK;
    }
    UttDesc *utt = reinterpret_cast<UttDesc *>(base);
    double *partial = reinterpret_cast<double *>(base + kHeadBytes);
    hipLaunchKernelGGL(k_single_utt, dim3(1), dim3(64), 0, st, utt, (int)rows);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error("descriptor kernel launch failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
    // every chunk's partial sums are written before they are read: nothing has to start at zero
    int lrc;
    if (precision == MFCC_PREC_F64) {
        StatArgs<double> a = {};
        a.data = static_cast<double *>(const_cast<void *>(dataDev));
        a.utts = utt;
        a.numChunks = (int)chunks;
        a.step = step;
        a.partial = partial;
        lrc = launch_colstats(1, &a, (int)chunks, stream);
    } else {
        StatArgs<float> a = {};
        a.data = static_cast<float *>(const_cast<void *>(dataDev));
        a.utts = utt;
        a.numChunks = (int)chunks;
        a.step = step;
        a.partial = partial;
        lrc = launch_colstats(0, &a, (int)chunks, stream);
    }
    if (lrc) return set_error("z-norm statistics kernel launch failed (HIP error %d)", lrc), MFCC_E_DEVICE;
    StatArgs<float> fa = {};
    fa.utts = utt;
    fa.step = step;
    fa.partial = partial;
    lrc = launch_shardfinal(fa, (int)chunks, 0, sumsDev, stream);
    if (lrc) return set_error("shard statistics kernel launch failed (HIP error %d)", lrc), MFCC_E_DEVICE;
    return MFCC_OK;
}
