// The one exchange step of a frame-sharded recording, over RCCL: every shard contributes a short
// vector of partial sums (the z-norm column sums: sum x, sum x^2 per column and the row count, as
// doubles; or the integer channel sums of the zero-mean option), the vectors are all-gathered on
// the GPUs (xGMI between the GPUs of a node) and every shard adds them in shard order with one
// small kernel, so all shards hold bit-identical totals whatever order the collective works in.
// The reference has no counterpart (it normalises one file in one thread, main.c:263-266); the
// partition is the one README.md:122 implies: statistics "within the single audio".
//
// librccl is loaded on first use (dlopen of librccl.so.1), not linked: the per-file entry points
// and the command-line program never pay for loading a collective library they do not use, and
// the library still loads on a machine without RCCL (the exchange then reports MFCC_E_DEVICE and
// the one-process entry point falls back to host memory).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "mfcc_host.h"

namespace mfcc {

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*commInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*commCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*allGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*groupStart)() = nullptr;
    ncclResult_t (*groupEnd)() = nullptr;
    const char *(*getErrorString)(ncclResult_t) = nullptr;
    bool tried = false, ok = false;
};
RcclApi g_rccl;
std::mutex g_rcclMu;

const RcclApi *rccl_api()
{
    std::lock_guard<std::mutex> g(g_rcclMu);
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            g_rccl.handle = h;
            g_rccl.commInitAll = reinterpret_cast<decltype(g_rccl.commInitAll)>(dlsym(h, "ncclCommInitAll"));
            g_rccl.commDestroy = reinterpret_cast<decltype(g_rccl.commDestroy)>(dlsym(h, "ncclCommDestroy"));
            g_rccl.commCount = reinterpret_cast<decltype(g_rccl.commCount)>(dlsym(h, "ncclCommCount"));
            g_rccl.allGather = reinterpret_cast<decltype(g_rccl.allGather)>(dlsym(h, "ncclAllGather"));
            g_rccl.groupStart = reinterpret_cast<decltype(g_rccl.groupStart)>(dlsym(h, "ncclGroupStart"));
            g_rccl.groupEnd = reinterpret_cast<decltype(g_rccl.groupEnd)>(dlsym(h, "ncclGroupEnd"));
            g_rccl.getErrorString = reinterpret_cast<decltype(g_rccl.getErrorString)>(dlsym(h, "ncclGetErrorString"));
            g_rccl.ok = g_rccl.commInitAll && g_rccl.commDestroy && g_rccl.commCount && g_rccl.allGather && g_rccl.groupStart &&
                        g_rccl.groupEnd && g_rccl.getErrorString;
        }
    }
    return g_rccl.ok ? &g_rccl : nullptr;
}

// out[i] = parts[0][i] + parts[1][i] + ... in that order (parts: [world][count])
template <typename T>
__global__ void k_rank_sum(const T *parts, T *out, int count, int world)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = parts[i];
    for (int r = 1; r < world; ++r) s += parts[(size_t)r * count + i];
    out[i] = s;
}

} // namespace

int rccl_load()
{
    if (rccl_api()) return MFCC_OK;
    const char *why = dlerror();
    return set_error("librccl.so.1 could not be loaded (%s)", why ? why : "a symbol is missing"), MFCC_E_DEVICE;
}

int rccl_comm_init_all(void **comms, int n, const int *devices)
{
    const RcclApi *api = rccl_api();
    if (!api) return rccl_load();
    const ncclResult_t r = api->commInitAll(reinterpret_cast<ncclComm_t *>(comms), n, devices);
    if (r != ncclSuccess) return set_error("ncclCommInitAll over %d devices failed: %s", n, api->getErrorString(r)), MFCC_E_DEVICE;
    return MFCC_OK;
}

void rccl_comm_destroy(void *comm)
{
    const RcclApi *api = rccl_api();
    if (api && comm) (void)api->commDestroy(static_cast<ncclComm_t>(comm));
}

} // namespace mfcc

using namespace mfcc;

extern "C" int MFCCShardExchange(void *comm, int isInt64, void *valuesDev, int count, void *gatherDev, void *stream)
{
    if (!comm || !valuesDev || !gatherDev || count < 1) return set_error("bad argument to MFCCShardExchange"), MFCC_E_ARG;
    const RcclApi *api = rccl_api();
    if (!api) return rccl_load();
    int world = 0;
    ncclResult_t r = api->commCount(static_cast<ncclComm_t>(comm), &world);
    if (r != ncclSuccess || world < 1) return set_error("ncclCommCount failed: %s", api->getErrorString(r)), MFCC_E_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    r = api->allGather(valuesDev, gatherDev, (size_t)count, isInt64 ? ncclInt64 : ncclDouble, static_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) return set_error("ncclAllGather of %d values failed: %s", count, api->getErrorString(r)), MFCC_E_DEVICE;
    const int blocks = (count + 255) / 256;
    if (isInt64)
        hipLaunchKernelGGL(k_rank_sum<long long>, dim3(blocks), dim3(256), 0, st, static_cast<const long long *>(gatherDev),
                           static_cast<long long *>(valuesDev), count, world);
    else
        hipLaunchKernelGGL(k_rank_sum<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double *>(gatherDev),
                           static_cast<double *>(valuesDev), count, world);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("rank-ordered sum kernel launch failed (HIP error %d)", (int)e), MFCC_E_DEVICE;
    return MFCC_OK;
}
