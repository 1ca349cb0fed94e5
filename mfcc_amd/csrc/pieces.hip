// The reference's lower-level API (mfcc.h:29-74: FFT, Realft, Wave2FBank, FBank2MFCC, FBank2C0)
// as batched device entry points: one workgroup per frame, the frame held in LDS, arithmetic in
// the order of the reference so that double precision agrees to rounding.  These serve callers
// that use the pieces; the fused extraction path does not go through them.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mfcc_host.h"

namespace mfcc {

namespace {

constexpr double kPi = 3.14159265358979; // the reference's `pi` (HTK hmath.h)

// In place on `s` (n reals = n/2 complex points, interleaved) in LDS: bit reversal, then radix-2
// stages with the forward sign e^{+i theta} of mfcc.c:232-254 (invert: the opposite sign and a
// division by the number of complex points).
template <typename T>
__device__ void fft_lds(T *s, int n, int invert)
{
    const int nc = n / 2, tid = threadIdx.x;
    int bits = 0;
    while ((1 << bits) < nc) ++bits;
    __syncthreads();
    for (int i = tid; i < nc; i += kBlock) {
        const int j = bits > 0 ? (int)(__brev((unsigned)i) >> (32 - bits)) : i;
        if (j > i) {
            const T tr = s[2 * j], ti = s[2 * j + 1];
            s[2 * j] = s[2 * i];
            s[2 * j + 1] = s[2 * i + 1];
            s[2 * i] = tr;
            s[2 * i + 1] = ti;
        }
    }
    __syncthreads();
    for (int half = 1; half < nc; half *= 2) {
        const double theta = (invert ? -kPi : kPi) / (double)half; // 2 pi / limit, limit = 2 half
        for (int i = tid; i < nc / 2; i += kBlock) {
            const int q = i % half, a = (i / half) * 2 * half + q, b = a + half;
            double sn, cs;
            sincos(theta * (double)q, &sn, &cs);
            const T wr = (T)cs, wi = (T)sn;
            const T xre = wr * s[2 * b] - wi * s[2 * b + 1];
            const T xri = wr * s[2 * b + 1] + wi * s[2 * b];
            s[2 * b] = s[2 * a] - xre;
            s[2 * b + 1] = s[2 * a + 1] - xri;
            s[2 * a] = s[2 * a] + xre;
            s[2 * a + 1] = s[2 * a + 1] + xri;
        }
        __syncthreads();
    }
    if (invert) {
        for (int i = tid; i < n; i += kBlock) s[i] = s[i] / (T)nc;
        __syncthreads();
    }
}

// mfcc.c:262-291 on the LDS frame: complex FFT of the packed halves, then the split.
template <typename T>
__device__ void realft_lds(T *s, int n)
{
    const int nc = n / 2, n2 = nc / 2, tid = threadIdx.x;
    fft_lds(s, n, 0);
    const double theta = kPi / (double)nc;
    for (int a = 1 + tid; a < n2; a += kBlock) {
        const int b = nc - a;
        double sn, cs;
        sincos(theta * (double)a, &sn, &cs);
        const T wrs = (T)cs, wis = (T)sn;
        const T xr1 = (s[2 * a] + s[2 * b]) / (T)2, xi1 = (s[2 * a + 1] - s[2 * b + 1]) / (T)2;
        const T xr2 = (s[2 * a + 1] + s[2 * b + 1]) / (T)2, xi2 = (s[2 * b] - s[2 * a]) / (T)2;
        s[2 * a] = xr1 + wrs * xr2 - wis * xi2;
        s[2 * a + 1] = xi1 + wrs * xi2 + wis * xr2;
        s[2 * b] = xr1 - wrs * xr2 + wis * xi2;
        s[2 * b + 1] = -xi1 + wrs * xi2 + wis * xr2;
    }
    __syncthreads();
    if (tid == 0) {
        const T xr1 = s[0];
        s[0] = xr1 + s[1];
        s[1] = 0;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_fft_batch(T *data, int n, int invert, int realft)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    T *s = reinterpret_cast<T *>(psm);
    T *frame = data + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += kBlock) s[i] = frame[i];
    if (realft)
        realft_lds(s, n);
    else
        fft_lds(s, n, invert);
    for (int i = threadIdx.x; i < n; i += kBlock) frame[i] = s[i];
}

// mfcc.c:125-177: te, zero padding, Realft, power (or magnitude), two-tap mel weighting summed per
// channel in bin order, log(max(., 1)) if takeLogs.  Optional extras the reference leaves behind for
// its caller: the packed spectrum (info.x, read by main.c:235-236) and te2 (mfcc.c:164-168).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_wave2fbank_batch(const T *frames, T *fbank, T *te, T *te2, T *spec,
                                                             int frameSize, int fftN, int klo, int khi, int P,
                                                             const int *loChan, const T *loWt, int usePower,
                                                             int takeLogs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    T *s = reinterpret_cast<T *>(psm);
    __shared__ T red[kBlock];
    const int tid = threadIdx.x, nc = fftN / 2;
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
    realft_lds(s, fftN);
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

int launch_fft_batch(int precision, void *data, int numFrames, int n, int invert, int realft, void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)n * (precision == 1 ? 8 : 4);
    if (precision == 1) {
        if (int e = allow_lds(k_fft_batch<double>, lds)) return e;
        hipLaunchKernelGGL(k_fft_batch<double>, dim3(numFrames), dim3(kBlock), lds, st, static_cast<double *>(data), n,
                           invert, realft);
    } else {
        if (int e = allow_lds(k_fft_batch<float>, lds)) return e;
        hipLaunchKernelGGL(k_fft_batch<float>, dim3(numFrames), dim3(kBlock), lds, st, static_cast<float *>(data), n,
                           invert, realft);
    }
    return (int)hipGetLastError();
}

int launch_wave2fbank_batch(int precision, const void *frames, void *fbank, void *te, void *te2, void *spec,
                            int numFrames, int frameSize, int fftN, int klo, int khi, int P, const int *loChan,
                            const void *loWt, int usePower, int takeLogs, void *stream)
{
    if (numFrames <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)fftN * (precision == 1 ? 8 : 4);
    if (precision == 1) {
        if (int e = allow_lds(k_wave2fbank_batch<double>, lds)) return e;
        hipLaunchKernelGGL(k_wave2fbank_batch<double>, dim3(numFrames), dim3(kBlock), lds, st,
                           static_cast<const double *>(frames), static_cast<double *>(fbank),
                           static_cast<double *>(te), static_cast<double *>(te2), static_cast<double *>(spec),
                           frameSize, fftN, klo, khi, P, loChan, static_cast<const double *>(loWt), usePower,
                           takeLogs);
    } else {
        if (int e = allow_lds(k_wave2fbank_batch<float>, lds)) return e;
        hipLaunchKernelGGL(k_wave2fbank_batch<float>, dim3(numFrames), dim3(kBlock), lds, st,
                           static_cast<const float *>(frames), static_cast<float *>(fbank),
                           static_cast<float *>(te), static_cast<float *>(te2), static_cast<float *>(spec),
                           frameSize, fftN, klo, khi, P, loChan, static_cast<const float *>(loWt), usePower,
                           takeLogs);
    }
    return (int)hipGetLastError();
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
