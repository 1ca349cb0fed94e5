// Whole-utterance finishing pass: regression deltas (Regress, call sites main.c:259-260) and the
// per-utterance z-norm (README.md:122) in one kernel.  The static-feature kernel writes its rows
// compactly (numFrames x vSize) into a plan-owned buffer; one workgroup per utterance loads them
// into LDS, takes the column statistics of every order from there, and writes each full output
// row (static | delta | delta-delta, normalised) exactly once, as contiguous lines.  Used when
// every utterance of the plan fits the LDS budget; longer ones take the row-block kernels.
// Arithmetic and rounding of a delta are those of k_regress / k_delta; the statistics are the
// double-precision sums of k_colstats / k_colfinal.
#include <hip/hip_runtime.h>

#include "mfcc_host.h"

namespace mfcc {

namespace {

template <typename T, int ORD>
__global__ __launch_bounds__(kFinishBlock) void k_utt_finish(const FinishArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int W = a.vSize, SR = a.subRows, dw = a.delwin, step = a.step;
    const UttDesc ud = a.utts[blockIdx.x];
    const int n = ud.numFrames, tid = threadIdx.x;
    if (n <= 0) return;
    constexpr int H1 = ORD == 3 ? 1 : 0;     // the first deltas carry a halo of delwin rows if a second order follows
    T *S = reinterpret_cast<T *>(fsm);       // n x W static rows
    T *D1 = S + a.staticElems;               // (SR + 2 delwin) x W first deltas of the sub-tile
    T *stage = D1 + (SR + 2 * dw) * W;       // SR x step finished rows
    double *red = reinterpret_cast<double *>(fsm + a.redOffset); // 2 kFinishBlock sums, then mean | 1/std (2 step)
    double *ms = red + 2 * kFinishBlock;

    const T *src = a.statics + ud.outRow * W;
    for (int i = tid; i < n * W; i += kFinishBlock) S[i] = src[i];
    __syncthreads();

    // Lane (rl, col) handles column col of rows rl, rl + rowLanes, ... of whatever it sweeps: no
    // index divisions in the loops, and a wave touches consecutive LDS words.
    const int ncol = min(kFinishBlock, W), rowLanes = kFinishBlock / ncol;
    const int col = tid % ncol, rl = tid / ncol;
    const bool live = rl < rowLanes;
    const int hrows = H1 * dw;
    T *dst = a.out + ud.outRow * step;
    // (T)((double)sum / sigmaT2): in single precision the correctly rounded float quotient is the
    // same number (sigmaT2 is a small integer and double carries more than 2 x 24 + 2 bits)
    const float sigmaF = (float)a.sigmaT2;
    auto quotient = [&](T sum) -> T {
        if constexpr (sizeof(T) == 4)
            return sum / sigmaF;
        else
            return (T)((double)sum / a.sigmaT2);
    };

    for (int pass = a.znorm ? 0 : 1; pass < 2; ++pass) {
        double s1[ORD], s2[ORD];
#pragma unroll
        for (int o = 0; o < ORD; ++o) s1[o] = s2[o] = 0;
        double mean[ORD], rstd[ORD];
#pragma unroll
        for (int o = 0; o < ORD; ++o) {
            mean[o] = (pass == 1 && a.znorm && live) ? ms[o * W + col] : 0.0;
            rstd[o] = (pass == 1 && a.znorm && live) ? ms[step + o * W + col] : 1.0;
        }
        for (int s = 0; s < n; s += SR) {
            const int rows = min(SR, n - s);
            if constexpr (ORD >= 2) {
                // first deltas of rows s - hrows .. s + rows + hrows - 1, each at its clamped row
                if (live)
                    for (int p = rl; p < rows + 2 * hrows; p += rowLanes) {
                        const int t = min(max(s - hrows + p, 0), n - 1);
                        T sum = 0;
                        for (int tau = 1; tau <= dw; ++tau)
                            sum += (T)tau * (S[min(t + tau, n - 1) * W + col] - S[max(t - tau, 0) * W + col]);
                        D1[p * W + col] = quotient(sum);
                    }
                __syncthreads();
            }
            if (live)
                for (int r = rl; r < rows; r += rowLanes) {
                    T x[ORD];
                    x[0] = S[(s + r) * W + col];
                    if constexpr (ORD >= 2) x[1] = D1[(r + hrows) * W + col];
                    if constexpr (ORD >= 3) { // delta of the first deltas at row s + r
                        T sum = 0;
                        for (int tau = 1; tau <= dw; ++tau)
                            sum += (T)tau * (D1[(r + hrows + tau) * W + col] - D1[(r + hrows - tau) * W + col]);
                        x[2] = quotient(sum);
                    }
                    if (pass == 0) {
#pragma unroll
                        for (int o = 0; o < ORD; ++o) {
                            const double xd = (double)x[o];
                            s1[o] += xd;
                            s2[o] += xd * xd;
                        }
                    } else {
#pragma unroll
                        for (int o = 0; o < ORD; ++o) {
                            T v = x[o];
                            if (a.znorm) v = (T)(((double)v - mean[o]) * rstd[o]);
                            stage[r * step + o * W + col] = v;
                        }
                    }
                }
            if (pass == 1) {
                __syncthreads();
                // rows s .. s + rows - 1 are contiguous in the output
                T *o = dst + (long long)s * step;
                for (int i = tid; i < rows * step; i += kFinishBlock) o[i] = stage[i];
            }
            __syncthreads();
        }
        if (pass == 0) {
#pragma unroll
            for (int o = 0; o < ORD; ++o) {
                red[tid] = s1[o];
                red[kFinishBlock + tid] = s2[o];
                __syncthreads();
                if (tid < ncol) {
                    double t1 = 0, t2 = 0;
                    for (int q = 0; q < rowLanes; ++q) {
                        t1 += red[q * ncol + tid];
                        t2 += red[kFinishBlock + q * ncol + tid];
                    }
                    const double cnt = (double)n;
                    const double m = t1 / cnt;
                    double var = t2 / cnt - m * m;
                    if (var < 0) var = 0;
                    ms[o * W + tid] = m;
                    ms[step + o * W + tid] = 1.0 / sqrt(var);
                }
                __syncthreads();
            }
        }
    }
}

// The same for utterances of at most KMAX * rowLanes rows: a lane keeps the deltas of its KMAX rows
// in registers across the statistics barrier, so every delta is formed once, and writes its
// finished values straight to the output row (the three pieces of a row leave the wave back to
// back and merge in L2).  Sub-tiles are rowLanes rows: one row per lane per round.
template <typename T, int ORD, int KMAX>
__global__ __launch_bounds__(kFinishBlock) void k_utt_finish_reg(const FinishArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int W = a.vSize, dw = a.delwin, step = a.step;
    const UttDesc ud = a.utts[blockIdx.x];
    const int n = ud.numFrames, tid = threadIdx.x;
    if (n <= 0) return;
    constexpr int H1 = ORD == 3 ? 1 : 0;
    constexpr int NX = ORD > 1 ? ORD - 1 : 1;
    T *S = reinterpret_cast<T *>(fsm);
    T *D1base = S + a.staticElems; // two buffers of (rowLanes + 2 delwin) x W, used in turn: one barrier per round
    double *red = reinterpret_cast<double *>(fsm + a.redOffset);
    double *ms = red + 2 * kFinishBlock;

    const T *src = a.statics + ud.outRow * W;
    for (int i = tid; i < n * W; i += kFinishBlock) S[i] = src[i];
    __syncthreads();

    const int ncol = min(kFinishBlock, W), rowLanes = kFinishBlock / ncol;
    const int col = tid % ncol, rl = tid / ncol;
    const bool live = rl < rowLanes;
    const int hrows = H1 * dw;
    const float sigmaF = (float)a.sigmaT2;
    auto quotient = [&](T sum) -> T {
        if constexpr (sizeof(T) == 4)
            return sum / sigmaF;
        else
            return (T)((double)sum / a.sigmaT2);
    };

    T xr[KMAX][NX];
    double s1[ORD], s2[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) s1[o] = s2[o] = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int s = k * rowLanes;
        if (s < n) { // block-uniform
            const int rows = min(rowLanes, n - s);
            T *D1 = D1base + (k & 1) * (rowLanes + 2 * dw) * W;
            if constexpr (ORD >= 2) {
                if (live)
                    for (int p = rl; p < rows + 2 * hrows; p += rowLanes) {
                        const int t = min(max(s - hrows + p, 0), n - 1);
                        T sum = 0;
                        for (int tau = 1; tau <= dw; ++tau)
                            sum += (T)tau * (S[min(t + tau, n - 1) * W + col] - S[max(t - tau, 0) * W + col]);
                        D1[p * W + col] = quotient(sum);
                    }
                __syncthreads();
            }
            if (live && rl < rows) {
                double xd = (double)S[(s + rl) * W + col];
                s1[0] += xd;
                s2[0] += xd * xd;
                if constexpr (ORD >= 2) {
                    xr[k][0] = D1[(rl + hrows) * W + col];
                    xd = (double)xr[k][0];
                    s1[1] += xd;
                    s2[1] += xd * xd;
                }
                if constexpr (ORD >= 3) {
                    T sum = 0;
                    for (int tau = 1; tau <= dw; ++tau)
                        sum += (T)tau * (D1[(rl + hrows + tau) * W + col] - D1[(rl + hrows - tau) * W + col]);
                    xr[k][1] = quotient(sum);
                    xd = (double)xr[k][1];
                    s1[2] += xd;
                    s2[2] += xd * xd;
                }
            }
        }
    }
    double mean[ORD], rstd[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) mean[o] = 0.0, rstd[o] = 1.0;
    if (a.znorm) {
#pragma unroll
        for (int o = 0; o < ORD; ++o) {
            red[tid] = s1[o];
            red[kFinishBlock + tid] = s2[o];
            __syncthreads();
            if (tid < ncol) {
                double t1 = 0, t2 = 0;
                for (int q = 0; q < rowLanes; ++q) {
                    t1 += red[q * ncol + tid];
                    t2 += red[kFinishBlock + q * ncol + tid];
                }
                const double cnt = (double)n;
                const double m = t1 / cnt;
                double var = t2 / cnt - m * m;
                if (var < 0) var = 0;
                ms[o * W + tid] = m;
                ms[step + o * W + tid] = 1.0 / sqrt(var);
            }
            __syncthreads();
        }
        if (live) {
#pragma unroll
            for (int o = 0; o < ORD; ++o) {
                mean[o] = ms[o * W + col];
                rstd[o] = ms[step + o * W + col];
            }
        }
    }
    T *dst = a.out + ud.outRow * step;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int row = k * rowLanes + rl;
        if (live && row < n) {
            T *o = dst + (long long)row * step + col;
            T x[ORD];
            x[0] = S[row * W + col];
#pragma unroll
            for (int q = 1; q < ORD; ++q) x[q] = xr[k][q - 1];
#pragma unroll
            for (int q = 0; q < ORD; ++q) {
                T v = x[q];
                if (a.znorm) v = (T)(((double)v - mean[q]) * rstd[q]);
                o[q * W] = v;
            }
        }
    }
}

template <typename T, int ORD>
int launch_one(const FinishArgs<T> &a, int numUtts, size_t lds, hipStream_t st)
{
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_utt_finish<T, ORD>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (a.regRows > 0) { // every utterance has at most kFinishRegRows rows per lane
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_utt_finish_reg<T, ORD, kFinishRegRows>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((k_utt_finish_reg<T, ORD, kFinishRegRows>), dim3(numUtts), dim3(kFinishBlock), lds, st, a);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((k_utt_finish<T, ORD>), dim3(numUtts), dim3(kFinishBlock), lds, st, a);
    return (int)hipGetLastError();
}

template <typename T>
int launch_t(const FinishArgs<T> &a, int orders, int numUtts, size_t lds, hipStream_t st)
{
    switch (orders) {
    case 1: return launch_one<T, 1>(a, numUtts, lds, st);
    case 2: return launch_one<T, 2>(a, numUtts, lds, st);
    case 3: return launch_one<T, 3>(a, numUtts, lds, st);
    default: return -2;
    }
}

} // namespace

// NormaliseLogEnergy of one column (the commented call main.c:255; HTK's definition, the source
// being absent from the reference's snapshot): per utterance, the column's largest value becomes
// 1, values more than silFloor dB below it are raised to that floor first, differences are scaled
// by escale.  One workgroup per utterance; the arithmetic is double as the reference's is.
template <typename T>
__global__ __launch_bounds__(256) void k_energy_norm(T *data, const UttDesc *utts, int step, int col, double silFloor,
                                                      double escale)
{
    const UttDesc ud = utts[blockIdx.x];
    if (ud.numFrames <= 0) return;
    T *p = data + ud.outRow * step + col;
    __shared__ double red[256];
    double mx = -INFINITY;
    for (int t = threadIdx.x; t < ud.numFrames; t += 256) {
        const double v = (double)p[(long long)t * step];
        mx = v > mx ? v : mx; // (as the reference's `if (*p > max)`: a NaN never becomes the maximum)
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x + o] > red[threadIdx.x] ? red[threadIdx.x + o] : red[threadIdx.x];
        __syncthreads();
    }
    mx = red[0];
    const double mn = mx - (silFloor * log(10.0)) / 10.0;
    for (int t = threadIdx.x; t < ud.numFrames; t += 256) {
        double v = (double)p[(long long)t * step];
        if (v < mn) v = mn;
        p[(long long)t * step] = (T)(1.0 - (mx - v) * escale);
    }
}

int launch_energy_norm(int precision, void *data, const UttDesc *utts, int numUtts, int step, int col, double silFloor,
                       double escale, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (numUtts <= 0) return 0;
    if (precision == 1)
        hipLaunchKernelGGL(k_energy_norm<double>, dim3(numUtts), dim3(256), 0, st, static_cast<double *>(data), utts, step, col,
                           silFloor, escale);
    else
        hipLaunchKernelGGL(k_energy_norm<float>, dim3(numUtts), dim3(256), 0, st, static_cast<float *>(data), utts, step, col,
                           silFloor, escale);
    return (int)hipGetLastError();
}

int launch_utt_finish(int precision, const void *args, int orders, int numUtts, size_t ldsBytes, void *stream)
{
    if (numUtts <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return precision == 1 ? launch_t<double>(*static_cast<const FinishArgs<double> *>(args), orders, numUtts, ldsBytes, st)
                          : launch_t<float>(*static_cast<const FinishArgs<float> *>(args), orders, numUtts, ldsBytes, st);
}

} // namespace mfcc
