// Segment-streaming finishing pass of batch plans: regression deltas (Regress, call sites
// main.c:259-260) and the per-utterance z-norm (README.md:122; commented calls main.c:263-266) for
// utterances of any length, ragged batches included.  The static-feature kernel writes its rows
// compactly (numFrames x vSize) into a plan-owned buffer.  Every utterance is cut into segments of
// 8 chunks of rows, at multiples of the segment length from the utterance's own first row; one
// workgroup of four waves owns one segment (and one block of at most 64 columns: deltas and z-norm
// are per column).  A wave streams its chunks through its own LDS region with the
// (order - 1) x delwin halo rows on either side -- row indices clamped at the utterance's ends
// exactly as Regress clamps its pointers -- and needs wave-level synchronisation only.
//   k_seg_sums    every order's column sums (double) of a segment -> one partial vector per segment
//   k_seg_stats   an utterance's partial vectors added in a fixed order -> mean and 1 / std per column
//   k_seg_write   the same chunks again: deltas formed once more, rows normalised, every output
//                 row written once
// Without z-norm only k_seg_write runs.  The statics are read twice, the output is written once,
// nothing else touches HBM.  The partition of an utterance's rows into segments, chunks and lanes --
// and with it the order in which its sums are added -- depends on the utterance and the option set
// alone, never on its batch neighbours: a file comes out bit for bit the same alone and in any
// batch.  Arithmetic and rounding of a delta and of the normalisation are those of k_regress /
// k_delta / k_normalize.
#include <hip/hip_runtime.h>

#include "front_util.h"
#include "mfcc_host.h"

namespace mfcc {

namespace {

constexpr int kSegWaves = 4;
constexpr int kSegChunks = 2 * kSegWaves; // chunks per segment

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WRITE = false: column sums of the segment; true: finished rows of the segment.
template <typename T, int ORD, bool WRITE>
__global__ __launch_bounds__(kSegWaves * 64) void k_seg_pass(const StreamArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    const int W = a.vSize, dw = a.delwin, step = a.step, RW = a.chunkRows;
    const int2 sg = reinterpret_cast<const int2 *>(a.segs)[blockIdx.x]; // utterance, first row of the segment
    const UttDesc ud = a.utts[sg.x];
    const int n = ud.numFrames;
    const int c0 = blockIdx.y * a.colBlock;
    const int CB = min(a.colBlock, W - c0);
    constexpr int H1 = ORD == 3 ? 1 : 0;        // the first deltas carry a halo of delwin rows if a second order follows
    const int h1 = H1 * dw, h = (ORD - 1) * dw; // halo rows of the first deltas / of the statics
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rowLanes = 64 / CB;               // (CB <= 64)
    const int col = lane % CB, rl = lane / CB;
    const bool live = rl < rowLanes;

    // LDS: per wave [statics (RW + 2 h) x colBlock | first deltas (RW + 2 h1) x colBlock], then the
    // reduction scratch (2 x 256 doubles)
    T *S = reinterpret_cast<T *>(ssm) + (size_t)wave * a.waveElems;
    T *D1 = S + (size_t)(RW + 2 * h) * a.colBlock;
    double *red = reinterpret_cast<double *>(ssm + a.redOffset);

    const T *src = a.statics + ud.outRow * W + c0;
    T *dst = a.out + ud.outRow * step + c0;
    // The delta's quotient sum / sigmaT2, correctly rounded as the other delta kernels form it.  In
    // single precision: the double product with 1 / sigmaT2, rounded to single.  (sigmaT2 is an
    // integer below 1000 and sum has 24 significant bits, so the exact quotient stays 2^-26 / sigmaT2
    // > 1e-11 (relative) away from every rounding boundary of single precision, a thousand times
    // the error of that product: both round to the same number.)
    const double rsigma = 1.0 / a.sigmaT2;
    auto quotient = [&](T sum) -> T {
        if constexpr (sizeof(T) == 4)
            return (T)((double)sum * rsigma);
        else
            return (T)((double)sum / a.sigmaT2);
    };
    const int segEnd = min(n, sg.y + kSegChunks * RW);

    double mean[ORD], rstd[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) mean[o] = 0.0, rstd[o] = 1.0;
    if (WRITE && a.znorm && live) {
        const double *st = a.stats + (size_t)sg.x * 2 * step + c0 + col;
#pragma unroll
        for (int o = 0; o < ORD; ++o) {
            mean[o] = st[o * W];
            rstd[o] = st[step + o * W];
        }
    }
    double s1[ORD], s2[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) s1[o] = s2[o] = 0;

    for (int s = sg.y + wave * RW; s < segEnd; s += kSegWaves * RW) {
        const int rows = min(RW, n - s);
        if (live) {
            // static rows s - h .. s + rows + h - 1, each at its clamped row
            for (int p = rl; p < rows + 2 * h; p += rowLanes) {
                const int g = min(max(s - h + p, 0), n - 1);
                S[p * CB + col] = src[(long long)g * W + col];
            }
        }
        wave_sync_lds();
        if constexpr (ORD >= 2) {
            if (live) {
                // first deltas of rows s - h1 .. s + rows + h1 - 1, each at its clamped row t; the
                // statics around t sit at local rows t - (s - h) +- tau (rows outside the
                // utterance hold the clamped copies)
                for (int p = rl; p < rows + 2 * h1; p += rowLanes) {
                    const int t = min(max(s - h1 + p, 0), n - 1);
                    const T *q = S + (t - (s - h)) * CB + col;
                    T sum = 0;
                    for (int tau = 1; tau <= dw; ++tau) sum += (T)tau * (q[tau * CB] - q[-tau * CB]);
                    D1[p * CB + col] = quotient(sum);
                }
            }
            wave_sync_lds();
        }
        if (live) {
            for (int r = rl; r < rows; r += rowLanes) {
                T x[ORD];
                x[0] = S[(r + h) * CB + col];
                if constexpr (ORD >= 2) x[1] = D1[(r + h1) * CB + col];
                if constexpr (ORD >= 3) { // delta of the first deltas at row s + r
                    const T *q = D1 + (r + h1) * CB + col;
                    T sum = 0;
                    for (int tau = 1; tau <= dw; ++tau) sum += (T)tau * (q[tau * CB] - q[-tau * CB]);
                    x[2] = quotient(sum);
                }
                if constexpr (!WRITE) {
#pragma unroll
                    for (int o = 0; o < ORD; ++o) {
                        const double xd = (double)x[o];
                        s1[o] += xd;
                        s2[o] += xd * xd;
                    }
                } else {
                    T *orow = dst + (long long)(s + r) * step + col;
#pragma unroll
                    for (int o = 0; o < ORD; ++o) {
                        T v = x[o];
                        if (a.znorm) v = (T)(((double)v - mean[o]) * rstd[o]);
                        orow[o * W] = v;
                    }
                }
            }
        }
        wave_sync_lds(); // the chunk's rows are consumed before the next chunk overwrites them
    }
    if constexpr (!WRITE) {
        // the segment's column sums: over the waves and their row lanes, in a fixed order
        double *part = a.partial + (size_t)blockIdx.x * 2 * step + c0;
#pragma unroll
        for (int o = 0; o < ORD; ++o) {
            red[tid] = live ? s1[o] : 0.0;
            red[kSegWaves * 64 + tid] = live ? s2[o] : 0.0;
            __syncthreads();
            if (tid < CB) {
                double t1 = 0, t2 = 0;
                for (int w = 0; w < kSegWaves; ++w)
                    for (int q = 0; q < rowLanes; ++q) {
                        t1 += red[w * 64 + q * CB + tid];
                        t2 += red[kSegWaves * 64 + w * 64 + q * CB + tid];
                    }
                part[o * W + tid] = t1;
                part[step + o * W + tid] = t2;
            }
            __syncthreads();
        }
    }
}

// Per utterance and value (2 x step of them: sums, then sums of squares): the segments' partials
// added by 16 lanes, lane j taking segments j, j + 16, ... in order, the 16 lane sums then added in
// lane order; then mean and 1 / std (population, as k_colfinal forms them).
constexpr int kStatVals = 16, kStatLanes = 16;
__global__ __launch_bounds__(kStatVals *kStatLanes) void k_seg_stats(const double *partial, const UttDesc *utts, const int32_t *segBegin,
                                                                      double *stats, int step)
{
    __shared__ double red[2][kStatLanes][kStatVals];
    const int u = blockIdx.x, v = threadIdx.x % kStatVals, j = threadIdx.x / kStatVals;
    const int c = blockIdx.y * kStatVals + v; // column
    const int b = segBegin[u], e = segBegin[u + 1];
    double t1 = 0, t2 = 0;
    if (c < step)
        for (int sgi = b + j; sgi < e; sgi += kStatLanes) {
            t1 += partial[(size_t)sgi * 2 * step + c];
            t2 += partial[(size_t)sgi * 2 * step + step + c];
        }
    red[0][j][v] = t1;
    red[1][j][v] = t2;
    __syncthreads();
    if (j == 0 && c < step) {
        t1 = 0, t2 = 0;
        for (int q = 0; q < kStatLanes; ++q) {
            t1 += red[0][q][v];
            t2 += red[1][q][v];
        }
        const double cnt = (double)utts[u].numFrames;
        const double m = t1 / cnt;
        double var = t2 / cnt - m * m;
        if (var < 0) var = 0;
        stats[(size_t)u * 2 * step + c] = m;
        stats[(size_t)u * 2 * step + step + c] = 1.0 / sqrt(var);
    }
}

template <typename T, int ORD>
int launch_one(const StreamArgs<T> &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[2][kAttrDevices];
    hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_seg_pass<T, ORD, false>), attrSet[0]);
    if (e == hipSuccess) e = lds_limit_once(reinterpret_cast<const void *>(&k_seg_pass<T, ORD, true>), attrSet[1]);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(a.numSegs, a.numColBlocks);
    if (a.znorm) {
        hipLaunchKernelGGL((k_seg_pass<T, ORD, false>), grid, dim3(kSegWaves * 64), lds, st, a);
        hipLaunchKernelGGL(k_seg_stats, dim3(a.numUtts, (a.step + kStatVals - 1) / kStatVals), dim3(kStatVals * kStatLanes), 0, st,
                           a.partial, a.utts, a.segBegin, a.stats, a.step);
    }
    hipLaunchKernelGGL((k_seg_pass<T, ORD, true>), grid, dim3(kSegWaves * 64), lds, st, a);
    return (int)hipGetLastError();
}

template <typename T>
int launch_t(const StreamArgs<T> &a, int orders, size_t lds, hipStream_t st)
{
    switch (orders) {
    case 1: return launch_one<T, 1>(a, lds, st);
    case 2: return launch_one<T, 2>(a, lds, st);
    case 3: return launch_one<T, 3>(a, lds, st);
    default: return -2;
    }
}

} // namespace

int stream_finish_seg_chunks() { return kSegChunks; }

// Geometry, from the option set alone: column blocks of at most 64 columns (narrower when the halo
// rows of a long regression window would not fit), rows per chunk, per-wave LDS elements, total
// bytes.  false: nothing fits (the caller takes the row-block kernels).
bool stream_finish_geometry(int vSize, int orders, int delwin, int elemBytes, int wantRows, int *colBlock, int *numColBlocks,
                            int *chunkRows, int *waveElems, int *redOffset, size_t *ldsBytes)
{
    const int h = (orders - 1) * delwin, h1 = orders == 3 ? delwin : 0;
    const size_t budget = 48 * 1024;
    for (int nb = (vSize + 63) / 64; nb <= vSize; nb *= 2) {
        const int cb = (vSize + nb - 1) / nb;
        for (int rw = wantRows > 0 ? wantRows : 32; rw >= 16; rw /= 2) {
            if (rw < 2 * h && rw < 32 && cb > 8) continue; // (more halo than rows: rather narrower column blocks)
            size_t elems = (size_t)(rw + 2 * h) * cb + (size_t)(orders >= 2 ? (rw + 2 * h1) * cb : 0);
            elems = (elems + 3) & ~(size_t)3;
            const size_t red = (kSegWaves * elems * elemBytes + 15) & ~(size_t)15;
            const size_t total = red + 2 * kSegWaves * 64 * sizeof(double);
            if (total <= budget) {
                *colBlock = cb, *numColBlocks = (vSize + cb - 1) / cb, *chunkRows = rw, *waveElems = (int)elems, *redOffset = (int)red;
                *ldsBytes = total;
                return true;
            }
            if (wantRows > 0) break;
        }
        if (cb <= 4) break;
    }
    return false;
}

int launch_seg_finish(int precision, const void *args, int orders, size_t ldsBytes, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (precision == 1) {
        const StreamArgs<double> &a = *static_cast<const StreamArgs<double> *>(args);
        return a.numSegs <= 0 ? 0 : launch_t<double>(a, orders, ldsBytes, st);
    }
    const StreamArgs<float> &a = *static_cast<const StreamArgs<float> *>(args);
    return a.numSegs <= 0 ? 0 : launch_t<float>(a, orders, ldsBytes, st);
}

} // namespace mfcc
