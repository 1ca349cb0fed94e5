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

#include <algorithm>

#include "front_util.h"
#include "mfcc_host.h"

namespace mfcc {

namespace {

constexpr int kSegWaves = 4;
constexpr int kSegChunks = 4 * kSegWaves; // chunks per segment

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kFastRows = 8; // rows per lane of a chunk on the register path

// MODE 0: column sums of the segment (over the rows the plan owns: all of them in a batch plan, a
// shard plan's without its halo frames); 1: finished rows of the segment, normalised when a.znorm;
// 2: finished rows as they are and their column sums in one sweep (what a shard plan's extraction
// leaves: the totals come from every shard, the normalisation is a later call).
// DW > 0: delwin = DW is known at compile time and a chunk is kFastRows rows per lane; chunks that
// lie inside the utterance with their halo rows (no clamped row) then take the register path: the
// chunk's statics sit in LDS column by column, a lane reads the kFastRows + 2 halo consecutive rows
// of its column once and forms every delta in registers.  Same arithmetic, same results.
template <typename T, int ORD, int DW, int MODE>
__global__ __launch_bounds__(kSegWaves * 64) void k_seg_pass(const StreamArgs<T> a)
{
    constexpr bool WRITE = MODE >= 1, SUMS = MODE != 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    const int W = a.vSize, dw = DW > 0 ? DW : a.delwin, step = a.step, RW = a.chunkRows;
    const int2 sg = reinterpret_cast<const int2 *>(a.segs)[blockIdx.x]; // utterance, first row of the segment
    const UttDesc ud = a.utts[sg.x];
    const int n = ud.numFrames;
    const int c0 = blockIdx.y * a.colBlock;
    const int CB = min(a.colBlock, W - c0);
    constexpr int H1 = ORD == 3 ? 1 : 0;        // the first deltas carry a halo of delwin rows if a second order follows
    const int h1 = H1 * dw, h = (ORD - 1) * dw; // halo rows of the first deltas / of the statics
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rowLanes = 64 / a.colBlock;       // (colBlock <= 64; the last column block may be narrower: CB)
    const int col = lane % a.colBlock, rl = lane / a.colBlock;
    const bool live = rl < rowLanes && col < CB;

    // LDS: per wave [statics (RW + 2 h) x colBlock | first deltas (RW + 2 h1) x colBlock], then the
    // reduction scratch (2 x 256 doubles)
    T *S = reinterpret_cast<T *>(ssm) + (size_t)wave * a.waveElems;
    T *D1 = S + (((size_t)(RW + 2 * h) * a.colBlock + a.colBlock + 3) & ~(size_t)3); // (the register path's statics: colBlock x (RW + 2 h + 1))
    T *stage = D1;                                               // register path: RW finished rows (stageOut)
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
    if (MODE == 1 && a.znorm && live) {
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

    // what a row's values go through once they are formed
    const int ownLo = ud.ownBegin, ownHi = ud.ownBegin + ud.ownCount;
    auto consume = [&](const T (&x)[ORD], T *orow, int row) {
        if constexpr (SUMS) {
            const bool own = row >= ownLo && row < ownHi;
#pragma unroll
            for (int o = 0; o < ORD; ++o) {
                const double xd = own ? (double)x[o] : 0.0;
                s1[o] += xd;
                s2[o] += xd * xd;
            }
        }
        if constexpr (WRITE) {
#pragma unroll
            for (int o = 0; o < ORD; ++o) {
                T v = x[o];
                if (MODE == 1 && a.znorm) v = (T)(((double)v - mean[o]) * rstd[o]);
                orow[o * W] = v;
            }
        }
    };

    // register path: the rows of a chunk on their way from memory (row rl + i rowLanes in pf[i])
    constexpr int NPF = DW > 0 ? kFastRows + 2 * (ORD - 1) * DW : 1;
    T pf[NPF];
    bool havePf = false;
    auto interior = [&](int s0) { return s0 - h >= 0 && s0 + RW + h <= n; };
    auto fetch = [&](int s0) {
        if (live) {
            const T *g = src + (long long)(s0 - h) * W + col;
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int p = rl + i * rowLanes;
                if (p < RW + 2 * h) pf[i] = g[(long long)p * W];
            }
        }
    };
    // 16-byte vectors of the staged rows: VEC elements; `head` single elements lie before the first
    // aligned output element of a chunk (the same for every chunk: a chunk is a multiple of 8 rows)
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int mis = (int)((ud.outRow * step + c0) % VEC);
    const int head = (VEC - mis) % VEC;
    T *stageP = stage + mis;

    for (int s = sg.y + wave * RW; s < segEnd; s += kSegWaves * RW) {
        const int rows = min(RW, n - s);
        if constexpr (DW > 0 && ORD >= 2) {
            if (interior(s)) { // (wave-uniform) no row of the chunk or its halo is clamped
                constexpr int K = kFastRows, HH = (ORD - 1) * DW, HH1 = ORD == 3 ? DW : 0;
                const int pitch = RW + 2 * HH + 1; // (odd pitches spread the columns over the banks)
                if (!havePf) fetch(s);
                if (live) {
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int p = rl + i * rowLanes;
                        if (p < RW + 2 * HH) S[col * pitch + p] = pf[i];
                    }
                }
                wave_sync_lds();
                // the next chunk's rows travel while this one is worked on
                const int sNext = s + kSegWaves * RW;
                havePf = sNext < segEnd && interior(sNext);
                if (havePf) fetch(sNext);
                if (live) {
                    const T *q = S + col * pitch + rl * K; // the lane's rows: rl K .. rl K + K - 1 of the chunk
                    T sv[K + 2 * HH];
#pragma unroll
                    for (int i = 0; i < K + 2 * HH; ++i) sv[i] = q[i];
                    T dv[K + 2 * HH1]; // first deltas of rows -HH1 .. K + HH1 - 1 of the lane's
#pragma unroll
                    for (int m = 0; m < K + 2 * HH1; ++m) {
                        T sum = 0;
#pragma unroll
                        for (int tau = 1; tau <= DW; ++tau) sum = fma_t((T)tau, sv[HH - HH1 + m + tau] - sv[HH - HH1 + m - tau], sum);
                        dv[m] = quotient(sum);
                    }
                    T *orow = WRITE ? (a.stageOut ? stageP + (rl * K) * step + col : dst + (long long)(s + rl * K) * step + col) : nullptr;
#pragma unroll
                    for (int r = 0; r < K; ++r) {
                        T x[ORD];
                        x[0] = sv[HH + r];
                        x[1] = dv[HH1 + r];
                        if constexpr (ORD >= 3) {
                            T sum = 0;
#pragma unroll
                            for (int tau = 1; tau <= DW; ++tau) sum = fma_t((T)tau, dv[HH1 + r + tau] - dv[HH1 + r - tau], sum);
                            x[2] = quotient(sum);
                        }
                        consume(x, WRITE ? orow + (long long)r * step : nullptr, s + rl * K + r);
                    }
                }
                if constexpr (WRITE) {
                    if (a.stageOut) {
                        // The chunk's RW finished rows are one contiguous piece of the output.  The staging
                        // area is shifted so that an element is 16-byte aligned there exactly when it is
                        // in the output: a few single elements up to the first aligned one, 16-byte
                        // vectors, single elements again.
                        wave_sync_lds();
                        T *o = dst + (long long)s * step;
                        const int total = RW * step;
                        const int nvec = (total - head) / VEC;
                        if (lane < head) o[lane] = stageP[lane];
                        for (int v = lane; v < nvec; v += 64)
                            *reinterpret_cast<vec_t *>(o + head + VEC * v) = *reinterpret_cast<const vec_t *>(stageP + head + VEC * v);
                        const int done = head + VEC * nvec;
                        if (lane < total - done) o[done + lane] = stageP[done + lane];
                    }
                }
                wave_sync_lds();
                continue;
            }
            havePf = false;
        }
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
                    for (int tau = 1; tau <= dw; ++tau) sum = fma_t((T)tau, q[tau * CB] - q[-tau * CB], sum);
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
                    for (int tau = 1; tau <= dw; ++tau) sum = fma_t((T)tau, q[tau * CB] - q[-tau * CB], sum);
                    x[2] = quotient(sum);
                }
                consume(x, WRITE ? dst + (long long)(s + r) * step + col : nullptr, s + r);
            }
        }
        wave_sync_lds(); // the chunk's rows are consumed before the next chunk overwrites them
    }
    if constexpr (SUMS) {
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
                        t1 += red[w * 64 + q * a.colBlock + tid];
                        t2 += red[kSegWaves * 64 + w * 64 + q * a.colBlock + tid];
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

// A shard plan's sums (its one buffer's segments): the same fixed-order addition, the raw sums and
// the number of owned rows behind them (2 step + 1 doubles, the layout the shards exchange).
__global__ __launch_bounds__(kStatVals *kStatLanes) void k_seg_shard_sums(const double *partial, const UttDesc *utts, int numSegs,
                                                                           double *sums, int step)
{
    __shared__ double red[kStatLanes][kStatVals];
    const int v = threadIdx.x % kStatVals, j = threadIdx.x / kStatVals;
    const int c = blockIdx.x * kStatVals + v; // value: sums, then sums of squares
    double t = 0;
    if (c < 2 * step)
        for (int sgi = j; sgi < numSegs; sgi += kStatLanes) t += partial[(size_t)sgi * 2 * step + c];
    red[j][v] = t;
    __syncthreads();
    if (j == 0 && c < 2 * step) {
        t = 0;
        for (int q = 0; q < kStatLanes; ++q) t += red[q][v];
        sums[c] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * step] = (double)utts[0].ownCount;
}

// mean and 1 / std per column from the recording's totals (2 step + 1 doubles: sums, sums of squares, rows)
__global__ void k_totals_to_stats(const double *totals, double *stats, int step)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= step) return;
    const double cnt = totals[2 * step];
    const double m = totals[c] / cnt;
    double var = totals[step + c] / cnt - m * m;
    if (var < 0) var = 0;
    stats[c] = m;
    stats[step + c] = 1.0 / sqrt(var);
}

template <typename T, int ORD, int DW>
int launch_one(const StreamArgs<T> &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[3][kAttrDevices];
    hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_seg_pass<T, ORD, DW, 0>), attrSet[0]);
    if (e == hipSuccess) e = lds_limit_once(reinterpret_cast<const void *>(&k_seg_pass<T, ORD, DW, 1>), attrSet[1]);
    if (e == hipSuccess) e = lds_limit_once(reinterpret_cast<const void *>(&k_seg_pass<T, ORD, DW, 2>), attrSet[2]);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(a.numSegs, a.numColBlocks);
    switch (a.phase) {
    case 0: // batch plan: sums, statistics, finished rows
        if (a.znorm) {
            hipLaunchKernelGGL((k_seg_pass<T, ORD, DW, 0>), grid, dim3(kSegWaves * 64), lds, st, a);
            hipLaunchKernelGGL(k_seg_stats, dim3(a.numUtts, (a.step + kStatVals - 1) / kStatVals), dim3(kStatVals * kStatLanes), 0, st,
                               a.partial, a.utts, a.segBegin, a.stats, a.step);
        }
        hipLaunchKernelGGL((k_seg_pass<T, ORD, DW, 1>), grid, dim3(kSegWaves * 64), lds, st, a);
        break;
    case 1: // shard plan, extraction: rows as they are, with the segments' sums when a z-norm follows
        if (a.znorm) hipLaunchKernelGGL((k_seg_pass<T, ORD, DW, 2>), grid, dim3(kSegWaves * 64), lds, st, a);
        else hipLaunchKernelGGL((k_seg_pass<T, ORD, DW, 1>), grid, dim3(kSegWaves * 64), lds, st, a);
        break;
    default: // shard plan, normalisation with the recording's statistics (a.stats)
        hipLaunchKernelGGL((k_seg_pass<T, ORD, DW, 1>), grid, dim3(kSegWaves * 64), lds, st, a);
        break;
    }
    return (int)hipGetLastError();
}

template <typename T>
int launch_t(const StreamArgs<T> &a, int orders, size_t lds, hipStream_t st)
{
    // register path: delwin 1 .. 3 with the chunk geometry it needs (stream_finish_geometry)
    const int fast = (orders >= 2 && a.fastPath) ? a.delwin : 0;
    switch (orders * 4 + fast) {
    case 4: return launch_one<T, 1, 0>(a, lds, st);
    case 8: return launch_one<T, 2, 0>(a, lds, st);
    case 9: return launch_one<T, 2, 1>(a, lds, st);
    case 10: return launch_one<T, 2, 2>(a, lds, st);
    case 11: return launch_one<T, 2, 3>(a, lds, st);
    case 12: return launch_one<T, 3, 0>(a, lds, st);
    case 13: return launch_one<T, 3, 1>(a, lds, st);
    case 14: return launch_one<T, 3, 2>(a, lds, st);
    case 15: return launch_one<T, 3, 3>(a, lds, st);
    default: return -2;
    }
}

} // namespace

int stream_finish_seg_chunks() { return kSegChunks; }

// Geometry, from the option set alone: column blocks of at most 64 columns (narrower when the halo
// rows of a long regression window would not fit), rows per chunk, per-wave LDS elements, total
// bytes.  *fastPath: delwin is 1 .. 3 and a chunk is kFastRows rows per lane (the register path's
// form); *stageOut: one column block, finished rows leave through LDS as contiguous pieces.
// false: nothing fits (the caller takes the row-block kernels).
bool stream_finish_geometry(int vSize, int orders, int delwin, int elemBytes, int wantRows, int *colBlock, int *numColBlocks,
                            int *chunkRows, int *waveElems, int *redOffset, size_t *ldsBytes, int *fastPath, int *stageOut)
{
    const int h = (orders - 1) * delwin, h1 = orders == 3 ? delwin : 0;
    const size_t budget = 48 * 1024;
    const bool fast = orders >= 2 && delwin >= 1 && delwin <= 3 && wantRows <= 0;
    for (int nb = (vSize + 63) / 64; nb <= vSize; nb *= 2) {
        const int cb = (vSize + nb - 1) / nb, rowLanes = 64 / cb;
        const int nblocks = (vSize + cb - 1) / cb;
        for (int rw = fast ? kFastRows * rowLanes : (wantRows > 0 ? wantRows : 32); rw >= 16 || fast; rw /= 2) {
            if (!fast && rw < 2 * h && rw < 32 && cb > 8) continue; // (more halo than rows: rather narrower column blocks)
            const bool stage = fast && nblocks == 1;
            const size_t statics = ((size_t)(rw + 2 * h) * cb + cb + 3) & ~(size_t)3; // row-major, or column-major with a pitch of rw + 2 h + 1
            const size_t second = std::max<size_t>(orders >= 2 ? (size_t)(rw + 2 * h1) * cb : 0, stage ? (size_t)rw * vSize * orders + 4 : 0);
            const size_t elems = (statics + second + 3) & ~(size_t)3;
            const size_t red = (kSegWaves * elems * elemBytes + 15) & ~(size_t)15;
            const size_t total = red + 2 * kSegWaves * 64 * sizeof(double);
            if (total <= budget) {
                *colBlock = cb, *numColBlocks = nblocks, *chunkRows = rw, *waveElems = (int)elems, *redOffset = (int)red;
                *ldsBytes = total, *fastPath = fast ? 1 : 0, *stageOut = stage ? 1 : 0;
                return true;
            }
            if (wantRows > 0 || fast) break;
        }
        if (cb <= 4) break;
    }
    return false;
}

int launch_seg_shard_sums(const double *partial, const UttDesc *utts, int numSegs, double *sums, int step, void *stream)
{
    hipLaunchKernelGGL(k_seg_shard_sums, dim3((2 * step + kStatVals - 1) / kStatVals), dim3(kStatVals * kStatLanes), 0,
                       static_cast<hipStream_t>(stream), partial, utts, numSegs, sums, step);
    return (int)hipGetLastError();
}

int launch_totals_to_stats(const double *totals, double *stats, int step, void *stream)
{
    hipLaunchKernelGGL(k_totals_to_stats, dim3((step + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), totals, stats, step);
    return (int)hipGetLastError();
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
