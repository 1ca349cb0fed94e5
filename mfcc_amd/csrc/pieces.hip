// The reference's lower-level API (mfcc.h:29-74: FFT, Realft, Wave2FBank, FBank2MFCC, FBank2C0)
// as batched device entry points.  The transforms are the register Stockham stages of the
// extraction kernels (fft_regs.h / fft_stages): G = nc / 16 lanes hold the nc complex points of a
// frame, 16 (or fewer) per lane, and exchange them through a padded LDS work area between the
// radix-16 stages; the stage twiddles and the real-FFT split factors come from a table built once
// per transform length, device and precision.  These serve callers that use the pieces; the fused
// extraction path does not go through them.
#include <hip/hip_runtime.h>
#include <math.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "front_util.h"
#include "mfcc_host.h"

namespace mfcc {

namespace {

constexpr double kPi = 3.14159265358979; // the reference's `pi` (HTK hmath.h)

template <int LOG2NC>
struct PieceGeo {
    using GEO = Geo<LOG2NC, 4>;
    static constexpr int NC = GEO::NC, G = GEO::G, PPL = GEO::PPL;
    static constexpr int GROUPS = kBlock / G;      // frames a workgroup transforms side by side
    static constexpr int ZEL = NC + NC / 16 + 2;   // complex points of one padded work area
    static constexpr int TWEL = NC + 8;            // stage twiddles in the table, then nc / 2 split factors
};

// Transform with the reference's forward sign e^{+i theta} (mfcc.c:232-247; plus = true) or the
// opposite one (its inverse without the division) of the frame `src` (2 nc interleaved reals, of
// which the first nReal exist and the rest count as zero), by the G lanes of one group.  The stages
// compute sum x e^{-i...}, so the forward sign is had by conjugating the input; the result is left
// in Z, natural order, padded (zpad), still conjugated when plus.
template <typename T, int LOG2NC>
__device__ __forceinline__ void group_transform(const T *src, int nReal, bool plus, cpx<T> *Z, int l, const cpx<T> *tw)
{
    using PG = PieceGeo<LOG2NC>;
    // (the first stage takes, per butterfly b of the lane, the points l + G b + r nc / R, r < R)
    constexpr int R0 = PG::PPL < 16 ? PG::PPL : 16;
    cpx<T> v[PG::PPL];
#pragma unroll
    for (int i = 0; i < PG::PPL; ++i) {
        const int k = l + PG::G * (i / R0) + (i % R0) * (PG::NC / R0);
        const T re = 2 * k < nReal ? src[2 * k] : (T)0, im = 2 * k + 1 < nReal ? src[2 * k + 1] : (T)0;
        v[i] = cpx<T>{re, plus ? -im : im};
    }
    fft_stages<T, LOG2NC, 4, 0, 0, false>(v, Z, l, tw);
}

// Real-FFT split (mfcc.c:274-290) from the conjugated half-size transform in Z to the packed
// spectrum `out` (nc complex values, interleaved): with X = conj(Z), E = (X[a] + conj(X[nc-a])) / 2
// and O = -i (X[a] - conj(X[nc-a])) / 2, out[a] = E + w O and out[nc-a] = conj(E - w O),
// w = e^{i pi a / nc} = sw[a]; out[0] = (Re X[0] + Im X[0], 0); the middle bin stays X[nc/2].
template <typename T>
__device__ __forceinline__ void split_packed(const cpx<T> *Z, T *out, int nc, const cpx<T> *sw, int first, int stride)
{
    cpx<T> *o = reinterpret_cast<cpx<T> *>(out);
    for (int a = 1 + first; a < nc / 2; a += stride) {
        const cpx<T> za = Z[zpad(a)], zb = Z[zpad(nc - a)];
        const cpx<T> xa = {za.x, -za.y}, xbc = {zb.x, zb.y}; // X[a], conj(X[nc-a])
        const cpx<T> E = {(xa.x + xbc.x) / (T)2, (xa.y + xbc.y) / (T)2};
        const cpx<T> D = {(xa.x - xbc.x) / (T)2, (xa.y - xbc.y) / (T)2};
        const cpx<T> O = {D.y, -D.x};
        const cpx<T> w = sw[a];
        const cpx<T> wO = {w.x * O.x - w.y * O.y, w.x * O.y + w.y * O.x};
        o[a] = cpx<T>{E.x + wO.x, E.y + wO.y};
        o[nc - a] = cpx<T>{E.x - wO.x, -(E.y - wO.y)};
    }
    if (first == 0) {
        const cpx<T> z0 = Z[0];
        o[0] = cpx<T>{z0.x + -z0.y, (T)0};
        if (nc >= 2) {
            const cpx<T> zh = Z[zpad(nc / 2)];
            o[nc / 2] = cpx<T>{zh.x, -zh.y};
        }
    }
}

// FFT (mode 0: forward, 1: inverse with the division by nc, mfcc.c:255-257) or Realft (mode 2) of
// numFrames frames of 2 nc reals in place; a workgroup takes GROUPS consecutive frames.
template <typename T, int LOG2NC>
__global__ __launch_bounds__(kBlock) void k_fft_batch(T *data, int numFrames, int mode, const cpx<T> *tab)
{
    using PG = PieceGeo<LOG2NC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int tid = threadIdx.x, g = tid / PG::G, l = tid % PG::G;
    cpx<T> *Z = reinterpret_cast<cpx<T> *>(psm) + (size_t)g * PG::ZEL;
    const long long f = (long long)blockIdx.x * PG::GROUPS + g;
    const bool live = f < numFrames;
    T *frame = data + (live ? f : 0) * 2 * PG::NC;
    group_transform<T, LOG2NC>(frame, live ? 2 * PG::NC : 0, mode != 1, Z, l, tab);
    if (!live) return; // (after the last barrier of the stages)
    if (mode == 2) {
        split_packed<T>(Z, frame, PG::NC, tab + PG::TWEL, l, PG::G);
        return;
    }
    cpx<T> *o = reinterpret_cast<cpx<T> *>(frame);
#pragma unroll
    for (int r = 0; r < PG::PPL; ++r) {
        const int k = l + PG::G * r;
        const cpx<T> z = Z[zpad(k)];
        o[k] = mode == 0 ? cpx<T>{z.x, -z.y} : cpx<T>{z.x / (T)PG::NC, z.y / (T)PG::NC};
    }
}

// mfcc.c:125-177: te, zero padding, Realft, power (or magnitude), two-tap mel weighting summed per
// channel in bin order, log(max(., 1)) if takeLogs.  Optional extras the reference leaves behind for
// its caller: the packed spectrum (info.x, read by main.c:235-236) and te2 (mfcc.c:164-168).
template <typename T, int LOG2NC>
__global__ __launch_bounds__(kBlock) void k_wave2fbank_batch(const T *frames, T *fbank, T *te, T *te2, T *spec,
                                                             int frameSize, int klo, int khi, int P,
                                                             const int *loChan, const T *loWt, int usePower,
                                                             int takeLogs, const cpx<T> *tab)
{
    using PG = PieceGeo<LOG2NC>;
    constexpr int fftN = 2 * PG::NC, nc = PG::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    T *s = reinterpret_cast<T *>(psm);                                  // the frame, then its packed spectrum
    cpx<T> *Zall = reinterpret_cast<cpx<T> *>(psm + (size_t)fftN * sizeof(T)); // work areas of the groups
    __shared__ T red[kBlock];
    const int tid = threadIdx.x;
    const T *in = frames + (size_t)blockIdx.x * frameSize;
    T e = 0;
    for (int i = tid; i < fftN; i += kBlock) {
        const T v = i < frameSize ? in[i] : (T)0;
        s[i] = v;
        e += v * v;
    }
    auto block_sum = [&](T v) { // tree order differs from the reference's running sum only in the last bits
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int o = kBlock / 2; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        return red[0];
    };
    if (te) {
        const T tot = block_sum(e);
        if (tid == 0) te[blockIdx.x] = tot;
    }
    __syncthreads();
    // Realft (mfcc.c:262-291): group 0 transforms the frame.  Groups wider than a wave synchronise
    // through the workgroup barrier, so the other groups then run the same stages on zeros in work
    // areas of their own; groups inside a wave leave the other lanes out.
    {
        const int g = tid / PG::G, l = tid % PG::G;
        if (PG::G > 64 || g == 0) group_transform<T, LOG2NC>(s, g == 0 ? fftN : 0, true, Zall + (size_t)g * PG::ZEL, l, tab);
        __syncthreads();
        split_packed<T>(Zall, s, nc, tab + PG::TWEL, tid, kBlock);
        __syncthreads();
    }
    if (spec)
        for (int i = tid; i < fftN; i += kBlock) spec[(size_t)blockIdx.x * fftN + i] = s[i];
    // power of bin k (1-based) parked at s[k-1]: every value is read before any is written
    constexpr int kPer = (1 << kMaxLog2Nc) / kBlock;
    T ek[kPer];
    T e2 = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int k0 = tid + j * kBlock;
        ek[j] = k0 < nc ? s[2 * k0] * s[2 * k0] + s[2 * k0 + 1] * s[2 * k0 + 1] : (T)0;
        e2 += ek[j];
    }
    if (te2) {
        const T tot = block_sum(e2);
        if (tid == 0) te2[blockIdx.x] = tot;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int k0 = tid + j * kBlock;
        if (k0 < nc) s[k0] = usePower ? ek[j] : sqrt(ek[j]); // mfcc.c:154-157
    }
    __syncthreads();
    for (int ch = tid; ch < P; ch += kBlock) { // channel ch (0-based) = reference channel ch + 1
        T acc = 0;
        for (int k = klo; k <= khi; ++k) {
            const int bin = loChan[k - 1];
            const T p = s[k - 1], t1 = loWt[k - 1] * p;
            if (bin == ch + 1) acc += t1;
            if (bin == ch) acc += p - t1;
        }
        fbank[(size_t)blockIdx.x * P + ch] = takeLogs ? log(acc < (T)1 ? (T)1 : acc) : acc;
    }
}

// FBank2MFCC / FBank2C0 without a table: the cos() per product of mfcc.c:192, for any channel
// count (one lane per coefficient, products in the reference's order).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fbank2mfcc_direct(const T *fbank, T *mfcc, T *c0, int P, int n)
{
    const T *fb = fbank + (size_t)blockIdx.x * P;
    const double mfnorm = sqrt(2.0 / (double)P), piFactor = kPi / (double)P;
    for (int j = threadIdx.x; j < n + (c0 ? 1 : 0); j += kBlock) {
        if (j < n) {
            const double x = (double)(j + 1) * piFactor;
            T acc = 0;
            for (int k = 1; k <= P; ++k) acc += fb[k - 1] * (T)cos(x * (k - 0.5));
            mfcc[(size_t)blockIdx.x * n + j] = acc * (T)mfnorm;
        } else {
            T acc = 0;
            for (int k = 0; k < P; ++k) acc += fb[k];
            c0[blockIdx.x] = acc * (T)mfnorm;
        }
    }
}

// mfcc.c:180-208: c[j] = sqrt(2/P) sum_k fbank[k] cos(pi j (k - 0.5) / P); C0 = sqrt(2/P) sum_k fbank[k]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fbank2mfcc_batch(const T *fbank, T *mfcc, T *c0, int P, int M,
                                                             const T *dctT, int dctPitch, T mfnorm)
{
    const T *fb = fbank + (size_t)blockIdx.x * P;
    for (int j = threadIdx.x; j < M + (c0 ? 1 : 0); j += kBlock) {
        T acc = 0;
        if (j < M) {
            for (int k = 0; k < P; ++k) acc += fb[k] * dctT[j * dctPitch + k];
            mfcc[(size_t)blockIdx.x * M + j] = acc * mfnorm;
        } else {
            for (int k = 0; k < P; ++k) acc += fb[k];
            c0[blockIdx.x] = acc * mfnorm;
        }
    }
}

template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace

// Stage twiddles (laid out as fft_stages reads them, radix 16: for a stage with sub-transform size
// NS and radix R, [r-1][m] = e^{-2 pi i r m / (NS R)}) followed by the split factors
// e^{i pi a / nc}, a < nc / 2, of one transform length, in device memory: built on first use per
// (device, precision, length) and kept for the life of the process.
static const void *piece_table(int precision, int log2nc, int *err)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, void *> cache;
    *err = 0;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return *err = (int)e, nullptr;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, precision, log2nc);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int nc = 1 << log2nc, twel = nc + 8;
    std::vector<double> t;
    for (int logns = 0; logns < log2nc;) {
        const int rb = std::min(4, log2nc - logns), R = 1 << rb, NS = 1 << logns;
        if (logns > 0)
            for (int r = 1; r < R; ++r)
                for (int m = 0; m < NS; ++m) {
                    const double ang = 2.0 * M_PI * (double)r * (double)m / ((double)NS * R);
                    t.push_back(cos(ang));
                    t.push_back(-sin(ang));
                }
        logns += rb;
    }
    t.resize(2 * (size_t)twel, 0.0);
    for (int a = 0; a < std::max(nc / 2, 1); ++a) {
        const double ang = kPi * (double)a / (double)nc; // theta = pi / n of mfcc.c:269
        t.push_back(cos(ang));
        t.push_back(sin(ang));
    }
    void *d = nullptr;
    const size_t el = precision == 1 ? 8 : 4;
    if (hipError_t e = hipMalloc(&d, t.size() * el)) return *err = (int)e, nullptr;
    hipError_t e;
    if (precision == 1) {
        e = hipMemcpy(d, t.data(), t.size() * el, hipMemcpyHostToDevice);
    } else {
        std::vector<float> f(t.begin(), t.end());
        e = hipMemcpy(d, f.data(), f.size() * el, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(d);
        return *err = (int)e, nullptr;
    }
    cache[key] = d;
    return d;
}

template <typename T, int LOG2NC>
static int launch_fft_one(T *data, int numFrames, int mode, const void *tab, hipStream_t st)
{
    using PG = PieceGeo<LOG2NC>;
    const size_t lds = (size_t)PG::GROUPS * PG::ZEL * sizeof(cpx<T>);
    if (int e = allow_lds(k_fft_batch<T, LOG2NC>, lds)) return e;
    const int blocks = (numFrames + PG::GROUPS - 1) / PG::GROUPS;
    hipLaunchKernelGGL((k_fft_batch<T, LOG2NC>), dim3(blocks), dim3(kBlock), lds, st, data, numFrames, mode,
                       static_cast<const cpx<T> *>(tab));
    return (int)hipGetLastError();
}

template <typename T, int LOG2NC>
static int launch_w2fb_one(const T *frames, T *fbank, T *te, T *te2, T *spec, int numFrames, int frameSize, int klo, int khi,
                           int P, const int *loChan, const T *loWt, int usePower, int takeLogs, const void *tab, hipStream_t st)
{
    using PG = PieceGeo<LOG2NC>;
    const size_t lds = (size_t)2 * PG::NC * sizeof(T) + (size_t)(PG::G > 64 ? PG::GROUPS : 1) * PG::ZEL * sizeof(cpx<T>);
    if (int e = allow_lds(k_wave2fbank_batch<T, LOG2NC>, lds)) return e;
    hipLaunchKernelGGL((k_wave2fbank_batch<T, LOG2NC>), dim3(numFrames), dim3(kBlock), lds, st, frames, fbank, te, te2, spec,
                       frameSize, klo, khi, P, loChan, loWt, usePower, takeLogs, static_cast<const cpx<T> *>(tab));
    return (int)hipGetLastError();
}

// One case per transform length: log2 of the complex point count, 0 .. kMaxLog2Nc.
#define MFCC_PIECE_CASES(CALL)                                                                              \
    switch (log2nc) {                                                                                       \
    case 0: return CALL(0);                                                                                 \
    case 1: return CALL(1);                                                                                 \
    case 2: return CALL(2);                                                                                 \
    case 3: return CALL(3);                                                                                 \
    case 4: return CALL(4);                                                                                 \
    case 5: return CALL(5);                                                                                 \
    case 6: return CALL(6);                                                                                 \
    case 7: return CALL(7);                                                                                 \
    case 8: return CALL(8);                                                                                 \
    case 9: return CALL(9);                                                                                 \
    case 10: return CALL(10);                                                                               \
    case 11: return CALL(11);                                                                               \
    case 12: return CALL(12);                                                                               \
    case 13: return CALL(13);                                                                               \
    default: return (int)hipErrorInvalidValue;                                                              \
    }

static int log2_of(int n)
{
    int b = 0;
    while ((1 << b) < n) ++b;
    return b;
}

template <typename T>
static int fft_dispatch(int log2nc, T *data, int numFrames, int mode, const void *tab, hipStream_t st)
{
#define MFCC_CALL(L) launch_fft_one<T, L>(data, numFrames, mode, tab, st)
    MFCC_PIECE_CASES(MFCC_CALL)
#undef MFCC_CALL
}

template <typename T>
static int w2fb_dispatch(int log2nc, const T *frames, T *fbank, T *te, T *te2, T *spec, int numFrames, int frameSize, int klo,
                         int khi, int P, const int *loChan, const T *loWt, int usePower, int takeLogs, const void *tab,
                         hipStream_t st)
{
#define MFCC_CALL(L) \
    launch_w2fb_one<T, L>(frames, fbank, te, te2, spec, numFrames, frameSize, klo, khi, P, loChan, loWt, usePower, takeLogs, tab, st)
    MFCC_PIECE_CASES(MFCC_CALL)
#undef MFCC_CALL
}

int launch_fft_batch(int precision, void *data, int numFrames, int n, int invert, int realft, void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int log2nc = log2_of(n / 2), mode = realft ? 2 : (invert ? 1 : 0);
    int err = 0;
    const void *tab = piece_table(precision, log2nc, &err);
    if (!tab) return err;
    return precision == 1 ? fft_dispatch<double>(log2nc, static_cast<double *>(data), numFrames, mode, tab, st)
                          : fft_dispatch<float>(log2nc, static_cast<float *>(data), numFrames, mode, tab, st);
}

int launch_wave2fbank_batch(int precision, const void *frames, void *fbank, void *te, void *te2, void *spec,
                            int numFrames, int frameSize, int fftN, int klo, int khi, int P, const int *loChan,
                            const void *loWt, int usePower, int takeLogs, void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int log2nc = log2_of(fftN / 2);
    int err = 0;
    const void *tab = piece_table(precision, log2nc, &err);
    if (!tab) return err;
    if (precision == 1)
        return w2fb_dispatch<double>(log2nc, static_cast<const double *>(frames), static_cast<double *>(fbank),
                                     static_cast<double *>(te), static_cast<double *>(te2), static_cast<double *>(spec), numFrames,
                                     frameSize, klo, khi, P, loChan, static_cast<const double *>(loWt), usePower, takeLogs, tab, st);
    return w2fb_dispatch<float>(log2nc, static_cast<const float *>(frames), static_cast<float *>(fbank), static_cast<float *>(te),
                                static_cast<float *>(te2), static_cast<float *>(spec), numFrames, frameSize, klo, khi, P, loChan,
                                static_cast<const float *>(loWt), usePower, takeLogs, tab, st);
}

int launch_fbank2mfcc_direct(int precision, const void *fbank, void *mfcc, void *c0, int numFrames, int P, int n,
                             void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (precision == 1)
        hipLaunchKernelGGL(k_fbank2mfcc_direct<double>, dim3(numFrames), dim3(kBlock), 0, st,
                           static_cast<const double *>(fbank), static_cast<double *>(mfcc),
                           static_cast<double *>(c0), P, n);
    else
        hipLaunchKernelGGL(k_fbank2mfcc_direct<float>, dim3(numFrames), dim3(kBlock), 0, st,
                           static_cast<const float *>(fbank), static_cast<float *>(mfcc),
                           static_cast<float *>(c0), P, n);
    return (int)hipGetLastError();
}

int launch_fbank2mfcc_batch(int precision, const void *fbank, void *mfcc, void *c0, int numFrames, int P,
                            int M, const void *dctT, int dctPitch, double mfnorm, void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (precision == 1)
        hipLaunchKernelGGL(k_fbank2mfcc_batch<double>, dim3(numFrames), dim3(kBlock), 0, st,
                           static_cast<const double *>(fbank), static_cast<double *>(mfcc),
                           static_cast<double *>(c0), P, M, static_cast<const double *>(dctT), dctPitch, mfnorm);
    else
        hipLaunchKernelGGL(k_fbank2mfcc_batch<float>, dim3(numFrames), dim3(kBlock), 0, st,
                           static_cast<const float *>(fbank), static_cast<float *>(mfcc),
                           static_cast<float *>(c0), P, M, static_cast<const float *>(dctT), dctPitch,
                           (float)mfnorm);
    return (int)hipGetLastError();
}

} // namespace mfcc
