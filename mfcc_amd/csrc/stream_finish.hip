// Segment-streaming finishing pass of batch plans: regression deltas (Regress, call sites
// main.c:259-260) and the per-utterance z-norm (README.md:122; commented calls main.c:263-266) for
// utterances of any length, ragged batches included, and of the frame-range shards of one long
// recording.  The static-feature kernel writes its rows compactly (numFrames x vSize) into a
// plan-owned buffer.  Every utterance is cut into segments of 16 chunks of rows, at multiples of the
// segment length from the utterance's own first row; one workgroup of four waves owns one segment
// (and one block of at most 64 columns: deltas and z-norm are per column).  A wave works through
// its chunks with the (order - 1) x delwin halo rows on either side -- row indices clamped at the
// utterance's ends exactly as Regress clamps its pointers -- and needs wave-level synchronisation only.
//   k_seg_pass / k_seg_reg, sums mode   every order's column sums (double) of a segment -> one
//                 partial vector per segment
//   k_seg_stats   an utterance's partial vectors added in a fixed order -> mean and 1 / std per column
//                 (k_seg_shard_sums: a shard's raw sums for the exchange; k_totals_to_stats: mean and
//                 1 / std from the recording's exchanged totals)
//   k_seg_pass / k_seg_reg, write mode  the same chunks again: deltas formed once more, rows
//                 normalised, every output row written once
// Without z-norm only the write mode runs.  The statics are read twice, the output is written once,
// nothing else touches HBM.  The partition of an utterance's rows into segments, chunks and lanes --
// and with it the order in which its sums are added -- depends on the utterance and the option set
// alone, never on its batch neighbours: a file comes out bit for bit the same alone and in any
// batch.  Arithmetic and rounding of a delta and of the normalisation are those of k_regress /
// k_delta / k_normalize.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

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

// Rows per lane of a chunk on the register path: a lane reads rows + 2 halo values of its column
// and forms rows + 2 delwin first deltas from them, so longer windows take more rows at a time.
__host__ __device__ constexpr int reg_rows(int dw) { return dw <= 3 ? 8 : 16; }

template <typename T>
struct SegCommon {
    // The delta's quotient sum / sigmaT2, correctly rounded as the other delta kernels form it.  In
    // single precision: the double product with 1 / sigmaT2, rounded to single.  (sigmaT2 is an
    // integer below 1000 and sum has 24 significant bits, so the exact quotient stays 2^-26 / sigmaT2
    // > 1e-11 (relative) away from every rounding boundary of single precision, a thousand times
    // the error of that product: both round to the same number.)
    double sigmaT2, rsigma;
    __device__ __forceinline__ T quotient(T sum) const
    {
        if constexpr (sizeof(T) == 4)
            return (T)((double)sum * rsigma);
        else
            return (T)((double)sum / sigmaT2);
    }
};

// The segment's column sums: over the waves and their row lanes, in a fixed order.
template <int ORD>
__device__ __forceinline__ void seg_reduce(const double (&s1)[ORD], const double (&s2)[ORD], double *red, double *part, bool live, int tid,
                                           int CB, int colBlock, int rowLanes, int W, int step)
{
#pragma unroll
    for (int o = 0; o < ORD; ++o) {
        red[tid] = live ? s1[o] : 0.0;
        red[kSegWaves * 64 + tid] = live ? s2[o] : 0.0;
        __syncthreads();
        if (tid < CB) {
            double t1 = 0, t2 = 0;
            for (int w = 0; w < kSegWaves; ++w)
                for (int q = 0; q < rowLanes; ++q) {
                    t1 += red[w * 64 + q * colBlock + tid];
                    t2 += red[kSegWaves * 64 + w * 64 + q * colBlock + tid];
                }
            part[o * W + tid] = t1;
            part[step + o * W + tid] = t2;
        }
        __syncthreads();
    }
}

// MODE 0: column sums of the segment (over the rows the plan owns: all of them in a batch plan, a
// shard plan's without its halo frames); 1: finished rows of the segment, normalised when a.znorm.
// Any delwin (read from the arguments): statics and first deltas of a chunk go through LDS.
template <typename T, int ORD, int MODE>
__global__ __launch_bounds__(kSegWaves * 64) void k_seg_pass(const StreamArgs<T> a)
{
    constexpr bool WRITE = MODE == 1, SUMS = MODE == 0;
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
    const int rowLanes = 64 / a.colBlock;       // (colBlock <= 64; the last column block may be narrower: CB)
    const int col = lane % a.colBlock, rl = lane / a.colBlock;
    const bool live = rl < rowLanes && col < CB;

    // LDS: per wave [statics (RW + 2 h) x colBlock | first deltas (RW + 2 h1) x colBlock], then the
    // reduction scratch (2 x 256 doubles)
    T *S = reinterpret_cast<T *>(ssm) + (size_t)wave * a.waveElems;
    T *D1 = S + (size_t)(RW + 2 * h) * a.colBlock;
    double *red = reinterpret_cast<double *>(ssm + a.redOffset);

    const T *src = a.statics + ud.outRow * W + c0;
    T *dst = a.out + ud.outRow * step + c0;
    const SegCommon<T> cm = {a.sigmaT2, 1.0 / a.sigmaT2};
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
    const int ownLo = ud.ownBegin, ownHi = ud.ownBegin + ud.ownCount;

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
                    for (int tau = 1; tau <= dw; ++tau) sum = fma_t((T)tau, q[tau * CB] - q[-tau * CB], sum);
                    D1[p * CB + col] = cm.quotient(sum);
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
                    x[2] = cm.quotient(sum);
                }
                if constexpr (SUMS) {
                    const bool own = s + r >= ownLo && s + r < ownHi;
#pragma unroll
                    for (int o = 0; o < ORD; ++o) {
                        const double xd = own ? (double)x[o] : 0.0;
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
    if constexpr (SUMS)
        seg_reduce<ORD>(s1, s2, red, a.partial + (size_t)blockIdx.x * 2 * step + c0, live, tid, CB, a.colBlock, rowLanes, W, step);
}

// The same with delwin = DW known at compile time: a lane owns reg_rows(DW) consecutive rows of its
// column per chunk, reads them with their halo once -- straight from memory when a column block
// fills the wave (one row lane: a wave's load is one contiguous row piece), else from the chunk's
// statics staged in LDS column by column, the next chunk's rows in flight meanwhile -- and forms
// every delta in registers.  Rows are read at their clamped indices; in a chunk that touches an end
// of the utterance the first deltas of rows outside it are replaced by the end row's, which is what
// Regress's clamped pointers give.  Finished rows of a narrow matrix (one column block) leave
// through LDS as one contiguous piece written in aligned 16-byte vectors.  Same arithmetic, same
// results as k_seg_pass.
// (The single-precision write kernels of windows 1 and 2 fit 128 registers without spills: four
// waves per SIMD for them; the sums kernels and the others spill there and stay at three.)
template <typename T, int ORD, int DW, int MODE, bool DIRECT>
__global__ __launch_bounds__(kSegWaves * 64, (DW <= 2 && sizeof(T) == 4 && MODE == 1) ? 4 : 1) void k_seg_reg(const StreamArgs<T> a)
{
    constexpr bool WRITE = MODE == 1, SUMS = MODE == 0;
    constexpr int K = reg_rows(DW), HH = (ORD - 1) * DW, HH1 = ORD == 3 ? DW : 0;
    constexpr int NS = K + 2 * HH, ND = K + 2 * HH1;
    constexpr bool PREFETCH = !DIRECT && NS <= 24; // the staged form keeps the next chunk's rows in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    const int W = a.vSize, step = a.step, RW = a.chunkRows;
    const int2 sg = reinterpret_cast<const int2 *>(a.segs)[blockIdx.x];
    const UttDesc ud = a.utts[sg.x];
    const int n = ud.numFrames;
    const int c0 = blockIdx.y * a.colBlock;
    const int CB = min(a.colBlock, W - c0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rowLanes = 64 / a.colBlock;
    const int col = lane % a.colBlock, rl = lane / a.colBlock;
    const bool live = rl < rowLanes && col < CB;
    constexpr bool direct = DIRECT; // (one row lane: rowLanes == 1)

    // LDS (staged form): per wave [statics colBlock x (RW + 2 HH + 1) | RW finished rows], then the
    // reduction scratch
    T *S = reinterpret_cast<T *>(ssm) + (size_t)wave * a.waveElems;
    T *stage = S + (direct ? 0 : (((size_t)(RW + 2 * HH + 1) * a.colBlock + 3) & ~(size_t)3));
    double *red = reinterpret_cast<double *>(ssm + a.redOffset);
    const int pitch = RW + 2 * HH + 1; // (odd pitches spread the columns over the banks)

    const T *src = a.statics + ud.outRow * W + c0;
    T *dst = a.out + ud.outRow * step + c0;
    const SegCommon<T> cm = {a.sigmaT2, 1.0 / a.sigmaT2};
    const int segEnd = min(n, sg.y + kSegChunks * RW);

    double mean[ORD], rstd[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) mean[o] = 0.0, rstd[o] = 1.0;
    auto load_stats = [&]() {
        if (WRITE && a.znorm && live) {
            const double *st = a.stats + (size_t)sg.x * 2 * step + c0 + col;
#pragma unroll
            for (int o = 0; o < ORD; ++o) {
                mean[o] = st[o * W];
                rstd[o] = st[step + o * W];
            }
        }
    };
    load_stats();
    double s1[ORD], s2[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) s1[o] = s2[o] = 0;
    const int ownLo = ud.ownBegin, ownHi = ud.ownBegin + ud.ownCount;

    // rows are addressed from the segment's first halo row (a uniform base and a 32-bit offset per lane)
    const int segRow0 = sg.y - HH;
    const T *segBase = src + (long long)segRow0 * W;
    // staged form: the rows of a chunk on their way from memory (row rl + i rowLanes of it in pf[i])
    constexpr int NPF = PREFETCH ? NS : 1;
    T pf[NPF];
    bool havePf = false;
    auto fetch = [&](int s0) {
        if (live) {
            if (s0 - HH >= 0 && s0 + RW + HH <= n) { // (wave-uniform) no row is clamped: one base, constant strides
                const T *g = segBase + (unsigned)((s0 - HH - segRow0 + rl) * W + col);
#pragma unroll
                for (int i = 0; i < NPF; ++i)
                    if (rl + i * rowLanes < RW + 2 * HH) pf[i] = g[(unsigned)(i * rowLanes * W)];
            } else {
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    const int p = rl + i * rowLanes;
                    if (p < RW + 2 * HH) pf[i] = segBase[(unsigned)((min(max(s0 - HH + p, 0), n - 1) - segRow0) * W + col)];
                }
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
        const int r0 = s + rl * K; // the lane's first row
        T sv[NS];                  // statics of rows r0 - HH .. r0 + K + HH - 1, each at its clamped row
        if constexpr (DIRECT) {
            if (live) {
#pragma unroll
                for (int i = 0; i < NS; ++i) sv[i] = segBase[(unsigned)((min(max(r0 - HH + i, 0), n - 1) - segRow0) * W + col)];
            }
        } else {
            if constexpr (PREFETCH) {
                if (!havePf) fetch(s);
                if (live) {
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int p = rl + i * rowLanes;
                        if (p < RW + 2 * HH) S[col * pitch + p] = pf[i];
                    }
                }
            } else {
                if (live)
                    for (int p = rl; p < RW + 2 * HH; p += rowLanes)
                        S[col * pitch + p] = segBase[(unsigned)((min(max(s - HH + p, 0), n - 1) - segRow0) * W + col)];
            }
            wave_sync_lds();
            if constexpr (PREFETCH) {
                // the next chunk's rows travel while this one is worked on
                const int sNext = s + kSegWaves * RW;
                havePf = sNext < segEnd;
                if (havePf) fetch(sNext);
            }
            if (live) {
                const T *q = S + col * pitch + rl * K;
#pragma unroll
                for (int i = 0; i < NS; ++i) sv[i] = q[i];
            }
        }
        if (live) {
            T dv[ND > 0 ? ND : 1]; // first deltas of rows r0 - HH1 .. r0 + K + HH1 - 1
            if constexpr (ORD >= 2) {
#pragma unroll
                for (int m = 0; m < ND; ++m) {
                    T sum = 0;
#pragma unroll
                    for (int tau = 1; tau <= DW; ++tau) sum = fma_t((T)tau, sv[HH - HH1 + m + tau] - sv[HH - HH1 + m - tau], sum);
                    dv[m] = cm.quotient(sum);
                }
                if constexpr (ORD >= 3) {
                    if (s - HH < 0 || s + RW + HH > n) { // (wave-uniform) the chunk touches an end of the utterance
                        // first deltas of the utterance's first and last row, from memory (clamped rows;
                        // addressed from the utterance's first row: row 0 can lie far before the segment)
                        T sumLo = 0, sumHi = 0;
#pragma unroll
                        for (int tau = 1; tau <= DW; ++tau) {
                            const T a0 = src[(long long)min(tau, n - 1) * W + col], b0 = src[col];
                            const T a1 = src[(long long)(n - 1) * W + col], b1 = src[(long long)max(n - 1 - tau, 0) * W + col];
                            sumLo = fma_t((T)tau, a0 - b0, sumLo);
                            sumHi = fma_t((T)tau, a1 - b1, sumHi);
                        }
                        const T dLo = cm.quotient(sumLo), dHi = cm.quotient(sumHi);
#pragma unroll
                        for (int m = 0; m < ND; ++m) {
                            const int g = r0 - HH1 + m; // the row of dv[m]
                            dv[m] = g < 0 ? dLo : (g > n - 1 ? dHi : dv[m]);
                        }
                    }
                }
            }
            T *orow = WRITE ? (a.stageOut ? stageP + (rl * K) * step + col : dst + (long long)r0 * step + col) : nullptr;
            // (ALL: every row of the chunk is one the plan owns and lies inside the utterance -- no per-row test)
            auto rows_out = [&](auto allTag) {
                constexpr bool ALL = decltype(allTag)::value;
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    T x[ORD];
                    x[0] = sv[HH + r];
                    if constexpr (ORD >= 2) x[1] = dv[HH1 + r];
                    if constexpr (ORD >= 3) {
                        T sum = 0;
#pragma unroll
                        for (int tau = 1; tau <= DW; ++tau) sum = fma_t((T)tau, dv[HH1 + r + tau] - dv[HH1 + r - tau], sum);
                        x[2] = cm.quotient(sum);
                    }
                    if constexpr (SUMS) {
                        const bool own = ALL || (r0 + r >= ownLo && r0 + r < ownHi);
#pragma unroll
                        for (int o = 0; o < ORD; ++o) {
                            const double xd = own ? (double)x[o] : 0.0;
                            s1[o] += xd;
                            s2[o] += xd * xd;
                        }
                    } else {
                        if (ALL || a.stageOut || r0 + r < n) { // (staged rows past the last one are not copied out)
#pragma unroll
                            for (int o = 0; o < ORD; ++o) {
                                T v = x[o];
                                if (a.znorm) v = (T)(((double)v - mean[o]) * rstd[o]);
                                orow[(long long)r * step + o * W] = v;
                            }
                        }
                    }
                }
            };
            rows_out(std::false_type{}); // (one copy of the unrolled rows: two made the long-window kernels outgrow the instruction cache)
        }
        if constexpr (WRITE) {
            if (a.stageOut) {
                // The chunk's finished rows are one contiguous piece of the output.  The staging area is
                // shifted so that an element is 16-byte aligned there exactly when it is in the
                // output: a few single elements up to the first aligned one, 16-byte vectors,
                // single elements again.
                wave_sync_lds();
                T *o = dst + (long long)s * step;
                const int total = rows * step;
                const int nvec = total > head ? (total - head) / VEC : 0;
                if (lane < min(head, total)) o[lane] = stageP[lane];
                for (int v = lane; v < nvec; v += 64)
                    *reinterpret_cast<vec_t *>(o + head + VEC * v) = *reinterpret_cast<const vec_t *>(stageP + head + VEC * v);
                const int done = min(head, total) + VEC * nvec;
                if (lane < total - done) o[done + lane] = stageP[done + lane];
            }
        }
        if (!direct) wave_sync_lds(); // the chunk's rows are consumed before the next chunk overwrites them
    }
    if constexpr (SUMS)
        seg_reduce<ORD>(s1, s2, red, a.partial + (size_t)blockIdx.x * 2 * step + c0, live, tid, CB, a.colBlock, rowLanes, W, step);
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
constexpr int kShardLanes = 64; // lanes per value: lane j adds segments j, j + 64, ... in order, then the lanes in lane order
__global__ __launch_bounds__(kStatVals *kShardLanes) void k_seg_shard_sums(const double *partial, const UttDesc *utts, int numSegs,
                                                                            double *sums, int step)
{
    __shared__ double red[kShardLanes][kStatVals];
    const int v = threadIdx.x % kStatVals, j = threadIdx.x / kStatVals;
    const int c = blockIdx.x * kStatVals + v; // value: sums, then sums of squares
    double t = 0;
    if (c < 2 * step)
        for (int sgi = j; sgi < numSegs; sgi += kShardLanes) t += partial[(size_t)sgi * 2 * step + c];
    red[j][v] = t;
    __syncthreads();
    if (j == 0 && c < 2 * step) {
        t = 0;
        for (int q = 0; q < kShardLanes; ++q) t += red[q][v];
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

template <typename T, typename KS, typename KW>
int launch_phases(const StreamArgs<T> &a, KS sums, KW write, std::atomic<bool> (&attrSet)[2][kAttrDevices], size_t lds, hipStream_t st)
{
    hipError_t e = lds_limit_once(reinterpret_cast<const void *>(sums), attrSet[0]);
    if (e == hipSuccess) e = lds_limit_once(reinterpret_cast<const void *>(write), attrSet[1]);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(a.numSegs, a.numColBlocks), block(kSegWaves * 64);
    switch (a.phase) {
    case 0: // batch plan: sums, statistics, finished rows
        if (a.znorm) {
            hipLaunchKernelGGL(sums, grid, block, lds, st, a);
            hipLaunchKernelGGL(k_seg_stats, dim3(a.numUtts, (a.step + kStatVals - 1) / kStatVals), dim3(kStatVals * kStatLanes), 0, st,
                               a.partial, a.utts, a.segBegin, a.stats, a.step);
        }
        hipLaunchKernelGGL(write, grid, block, lds, st, a);
        break;
    case 1: // shard plan, extraction: with a z-norm to follow only the segments' sums (the rows are
            // formed by the normalisation call, once the totals of every shard are known); else the rows
        if (a.znorm) hipLaunchKernelGGL(sums, grid, block, lds, st, a);
        else hipLaunchKernelGGL(write, grid, block, lds, st, a);
        break;
    default: // shard plan, normalisation with the recording's statistics (a.stats): the rows
        hipLaunchKernelGGL(write, grid, block, lds, st, a);
        break;
    }
    return (int)hipGetLastError();
}

template <typename T, int ORD>
int launch_generic(const StreamArgs<T> &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[2][kAttrDevices];
    return launch_phases<T>(a, &k_seg_pass<T, ORD, 0>, &k_seg_pass<T, ORD, 1>, attrSet, lds, st);
}
template <typename T, int ORD, int DW>
int launch_reg(const StreamArgs<T> &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[2][2][kAttrDevices];
    if (a.colBlock > 32) // one row lane: rows straight from memory
        return launch_phases<T>(a, &k_seg_reg<T, ORD, DW, 0, true>, &k_seg_reg<T, ORD, DW, 1, true>, attrSet[1], lds, st);
    return launch_phases<T>(a, &k_seg_reg<T, ORD, DW, 0, false>, &k_seg_reg<T, ORD, DW, 1, false>, attrSet[0], lds, st);
}

template <typename T>
int launch_t(const StreamArgs<T> &a, int orders, size_t lds, hipStream_t st)
{
    const int reg = a.fastPath ? a.delwin : 0; // register path: the window lengths stream_finish_geometry names
    switch (orders * 16 + reg) {
    case 16: return launch_generic<T, 1>(a, lds, st);
    case 32: return launch_generic<T, 2>(a, lds, st);
    case 33: return launch_reg<T, 2, 1>(a, lds, st);
    case 34: return launch_reg<T, 2, 2>(a, lds, st);
    case 35: return launch_reg<T, 2, 3>(a, lds, st);
    case 41: return launch_reg<T, 2, 9>(a, lds, st);
    case 48: return launch_generic<T, 3>(a, lds, st);
    case 49: return launch_reg<T, 3, 1>(a, lds, st);
    case 50: return launch_reg<T, 3, 2>(a, lds, st);
    case 51: return launch_reg<T, 3, 3>(a, lds, st);
    case 57: return launch_reg<T, 3, 9>(a, lds, st);
    default: return -2;
    }
}

} // namespace

int stream_finish_seg_chunks() { return kSegChunks; }

// Geometry, from the option set alone: column blocks of at most 64 columns (narrower when the halo
// rows of a long regression window would not fit), rows per chunk, per-wave LDS elements, total
// bytes.  *fastPath: delwin is one of the register path's (1, 2, 3, 9) and a chunk is reg_rows(delwin)
// rows per lane; *stageOut: one column block, finished rows leave through LDS as contiguous pieces.
// false: nothing fits (the caller takes the row-block kernels).
bool stream_finish_geometry(int vSize, int orders, int delwin, int elemBytes, int wantRows, int *colBlock, int *numColBlocks,
                            int *chunkRows, int *waveElems, int *redOffset, size_t *ldsBytes, int *fastPath, int *stageOut)
{
    const int h = (orders - 1) * delwin, h1 = orders == 3 ? delwin : 0;
    const size_t budget = 48 * 1024;
    const bool fast = orders >= 2 && (delwin == 1 || delwin == 2 || delwin == 3 || delwin == 9) && wantRows <= 0;
    if (fast) {
        // the fewest column blocks whose staging fits; finished rows staged too where that fits
        for (int nb = (vSize + 63) / 64; nb <= vSize; ++nb) {
            const int cb = (vSize + nb - 1) / nb, rowLanes = 64 / cb, nblocks = (vSize + cb - 1) / cb;
            const int rw = reg_rows(delwin) * rowLanes;
            for (int stage = nblocks == 1 ? 1 : 0; stage >= 0; --stage) {
                size_t elems = (rowLanes == 1 ? 0 : (((size_t)(rw + 2 * h + 1) * cb + 3) & ~(size_t)3)) + (stage ? (size_t)rw * vSize * orders + 4 : 0);
                elems = (elems + 3) & ~(size_t)3;
                const size_t red = (kSegWaves * elems * elemBytes + 15) & ~(size_t)15;
                const size_t total = red + 2 * kSegWaves * 64 * sizeof(double);
                if (total <= budget) {
                    *colBlock = cb, *numColBlocks = nblocks, *chunkRows = rw, *waveElems = (int)elems, *redOffset = (int)red;
                    *ldsBytes = total, *fastPath = 1, *stageOut = stage;
                    return true;
                }
            }
            if (cb <= 8) break;
        }
    }
    for (int nb = (vSize + 63) / 64; nb <= vSize; nb *= 2) {
        const int cb = (vSize + nb - 1) / nb;
        const int nblocks = (vSize + cb - 1) / cb;
        for (int rw = wantRows > 0 ? wantRows : 32; rw >= 16; rw /= 2) {
            if (rw < 2 * h && rw < 32 && cb > 8) continue; // (more halo than rows: rather narrower column blocks)
            const size_t elems = ((size_t)(rw + 2 * h) * cb + (orders >= 2 ? (size_t)(rw + 2 * h1) * cb : 0) + 3) & ~(size_t)3;
            const size_t red = (kSegWaves * elems * elemBytes + 15) & ~(size_t)15;
            const size_t total = red + 2 * kSegWaves * 64 * sizeof(double);
            if (total <= budget) {
                *colBlock = cb, *numColBlocks = nblocks, *chunkRows = rw, *waveElems = (int)elems, *redOffset = (int)red;
                *ldsBytes = total, *fastPath = 0, *stageOut = 0;
                return true;
            }
            if (wantRows > 0) break;
        }
        if (cb <= 4) break;
    }
    return false;
}

int launch_seg_shard_sums(const double *partial, const UttDesc *utts, int numSegs, double *sums, int step, void *stream)
{
    hipLaunchKernelGGL(k_seg_shard_sums, dim3((2 * step + kStatVals - 1) / kStatVals), dim3(kStatVals * kShardLanes), 0,
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
