// Static-feature kernel for the 2048-point stereo class: 24-bit stereo PCM, frames of more than
// 1024 and at most 2048 samples (1024 complex points), any of the derived rows {L, R, mid, side},
// log-mel filterbank or cepstra (+ C0), log energy, zero-crossing rate, brightness, sub-band
// energies, single precision.  Same arithmetic as k_front's general path (rows main.c:198-207,
// PreEmphasise main.c:208, zeroCrossingRate main.c:227, Ham main.c:229, Wave2FBank mfcc.c:125-177,
// brightness / sub-bands main.c:235-236, FBank2MFCC / FBank2C0 mfcc.c:180-208, row layout
// main.c:241-249), organised like the 256-point kernel of lean256.hip:
//   - a wave is a team of its own: one (frame, row) at a time, 64 lanes x 16 points, three FFT
//     stages (16 x 16 x 4) with wave-level synchronisation only; 12 waves per workgroup (three per
//     SIMD, 168 registers), one workgroup per CU;
//   - a tile is one frame, and nothing of its raw bytes is kept in LDS: a round asks for the next
//     round's bytes straight from memory (one 12-byte load per sample pair: a wave's load is 768
//     consecutive bytes; the 2.3-fold overlap of consecutive frames is served by L2) once its
//     spectrum has been reduced to per-bin values, turns them into that row's samples just before
//     its own results go out, and carries those samples to the next round -- no round starts by
//     waiting for memory behind the stores of the one before it;
//   - what a round needs at its very start (window pairs) and the lanes' mel part tables sit in
//     LDS, shared by the workgroup; twiddle, split and filterbank factors are read through L1 in
//     the middle of a round;
//   - waves draw chunks of consecutive frames from a shared counter (see lean256.hip).
#include <hip/hip_runtime.h>

#include "front_util.h"
#include "stamps.h"
#include <type_traits>

namespace mfcc {

namespace {

constexpr int kWideLog2Nc = 10, kWideNc = 1024, kWideG = 64, kWidePpl = 16, kWideWaves = 12;

typedef unsigned int u3u __attribute__((ext_vector_type(3), aligned(1)));
typedef unsigned int u2u __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned int u4s __attribute__((ext_vector_type(4)));

// One tile (frame) as the pipeline sees it: the two 64-bit words of its TileDesc (lean encoding:
// byte offset of the frame's first instant, bit 60: it starts the signal, bit 61: a load of the
// frame's bytes or of the 8 bytes in front of them would leave the planned bytes; frames field 1,
// or 0 for "no tile").
struct WideTile {
    unsigned w0, w1;
    int outRow;
    __device__ __forceinline__ bool valid() const { return ((w1 >> 16) & 15u) != 0; }
    __device__ __forceinline__ bool first() const { return (w1 >> 28) & 1u; }
    __device__ __forceinline__ bool edge() const { return (w1 >> 29) & 1u; }
    __device__ __forceinline__ long long src() const { return (long long)(((unsigned long long)(w1 & 0xffffu) << 32) | w0); }
};

__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ WideTile wide_desc(const TileDesc *tab, int idx, bool valid)
{
    WideTile t = {0u, 0u, 0};
    if (valid) {
        typedef const __attribute__((address_space(4))) u4s *cptr;
        const u4s d = reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(tab))[idx];
        t.w0 = d.x;
        t.w1 = d.y;
        t.outRow = (int)d.z;
    }
    return t;
}

// lane i takes x of lane i + 1; lane 63 takes `last`
__device__ __forceinline__ float wave_shl1f(float x, float last)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, x), 0x130,
                                                                 0xf, 0xf, false));
}

// lane i takes x of lane i - 1; lane 0 takes `first`
__device__ __forceinline__ float wave_shr1f(float x, float first)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, x), 0x138,
                                                                 0xf, 0xf, false));
}

// 24-bit little-endian sample at byte O of the lane's window of NW dwords
template <int O, int NW>
__device__ __forceinline__ int sample24(const unsigned (&d)[NW])
{
    return window_sample<3, O, NW>(d);
}


// Sample of derived row ROW (0 L, 1 R, 2 mid = 0.5 (L+R), 3 side = L - R; main.c:198-207) from NW
// dwords that hold the instant's two channels from byte O on.
template <int ROW, int O, int NW>
__device__ __forceinline__ float row_sample(const unsigned (&d)[NW])
{
    if constexpr (ROW == 0) return (float)sample24<O, NW>(d);
    else if constexpr (ROW == 1) return (float)sample24<O + 3, NW>(d);
    else if constexpr (ROW == 2) return 0.5f * (float)(sample24<O, NW>(d) + sample24<O + 3, NW>(d)); // (the integer sum is exact)
    else return (float)(sample24<O, NW>(d) - sample24<O + 3, NW>(d));
}

// The lane's samples of row ROW from the frame's raw bytes: own[r] = the 12 bytes of instants 2n
// and 2n+1, n = l + 64 r (both channels) -> xa[r], xb[r]; pred = the 8 bytes in front of the
// frame's first instant (its predecessor: bytes 2 .. 7) -> pred0.
template <int NZ, int ROW>
__device__ __forceinline__ void wide_samples(float (&xa)[NZ], float (&xb)[NZ], float &pred0, const unsigned (&own)[NZ][3],
                                             const unsigned (&pred)[2])
{
#pragma unroll
    for (int r = 0; r < NZ; ++r) {
        xa[r] = row_sample<ROW, 0, 3>(own[r]);
        xb[r] = row_sample<ROW, 6, 3>(own[r]);
    }
    pred0 = row_sample<ROW, 2, 2>(pred);
}

// The lane's points (2n, 2n+1), n = l + 64 r, of the frame's row from its samples xa[r] = x[2n],
// xb[r] = x[2n+1]; pred0 = the sample in front of the frame.  The predecessor 2n-1 of a pair is
// the second sample of the previous lane's pair (of lane 63's previous pair for lane 0): one lane
// shift.  Pre-emphasis over the whole signal (PreEmphasise main.c:208, first-sample rule for the
// signal's first sample), Hamming window (main.c:229, pairs from winL); with ZC the zero crossings
// of the un-windowed samples (main.c:227), whose signs must be those of the reference's double
// arithmetic with separately rounded product and difference: the single-precision x1 - kf x0 (one
// rounding, inputs exact) has that sign whenever it is not tiny against x0 -- the two differ by
// (kf - k) x0 and the product's rounding, and zcThr is four times that bound.  Returns whether
// some sample of the lane is that close to zero; the caller then counts the frame's crossings
// again with wide_zc_exact.
template <int NZ, bool ZC>
__device__ __forceinline__ bool wide_expand(cpx<float> (&v)[kWidePpl], int &zc, const float (&xa)[NZ], const float (&xb)[NZ], float pred0,
                                            int l, const cpx<float> *winL, double preK, float kf, float zcThr, bool firstTile, int wlen)
{
    constexpr int G = kWideG;
    float nextFirst0 = 0; // sample 2n of lane 0's pair r + 1 (the neighbour of lane 63's pair r)
    float margin = 1.0f;  // least |y| - zcThr |predecessor| over the lane's samples
    cpx<float> wreg[NZ];
#pragma unroll
    for (int r = 0; r < NZ; ++r) wreg[r] = winL[l + G * r];
#pragma unroll
    for (int r = NZ - 1; r >= 0; --r) {
        float x[3];
        float carry = pred0;
        if (r > 0) carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb[r - 1]), 63));
        x[0] = wave_shr1f(xb[r], carry);
        x[1] = xa[r];
        x[2] = xb[r];
        float y1, y2;
        if constexpr (ZC) {
            y1 = __builtin_fmaf(-kf, x[0], x[1]);
            y2 = __builtin_fmaf(-kf, x[1], x[2]);
            // (a zero sample after a zero predecessor is exact: margin 0, not below it)
            margin = __builtin_fminf(margin, __builtin_fminf(__builtin_fmaf(-zcThr, __builtin_fabsf(x[0]), __builtin_fabsf(y1)),
                                                             __builtin_fmaf(-zcThr, __builtin_fabsf(x[1]), __builtin_fabsf(y2))));
            if (firstTile && l == 0 && r == 0) y1 = (float)((double)x[1] * (1.0 - preK)); // s[1] *= 1 - k
            // sample 2n+2 is the first sample of the next lane's pair (lane 63: of lane 0's next pair)
            const float x2 = wave_shl1f(y1, nextFirst0);
            // (opposite signs = negative product; the samples are far from the underflow range)
            bool c01 = y1 * y2 < 0.0f, c12 = y2 * x2 < 0.0f;
            // pairs inside the frame only.  (The instance with 14 inputs per lane runs only for
            // 128 * 13 < wlen <= 128 * 14: every r below the last lies inside the frame.)
            if (r == NZ - 1 || NZ != 14) {
                const int s0 = 2 * (l + G * r);
                c01 = c01 && s0 + 1 < wlen;
                c12 = c12 && s0 + 2 < wlen;
            }
            // counted over the wave at once: zc is the same in every lane
            zc += __builtin_popcountll(__builtin_amdgcn_ballot_w64(c01)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(c12));
            nextFirst0 = __builtin_bit_cast(float, sgpr(__builtin_bit_cast(int, y1)));
        } else {
            y1 = x[1] - kf * x[0];
            y2 = x[2] - kf * x[1];
            if (firstTile && l == 0 && r == 0) y1 = (float)((double)x[1] * (1.0 - preK));
        }
        v[r] = cpx<float>{y1, y2} * wreg[r];
    }
#pragma unroll
    for (int r = NZ; r < kWidePpl; ++r) v[r] = cpx<float>{0.0f, 0.0f};
    return margin < 0.0f;
}

// The frame's zero crossings of row ROW (zeroCrossingRate main.c:227) in the reference's arithmetic:
// samples in double with separately rounded product and difference (PreEmphasise main.c:208).  For
// the rare frame with a sample too close to zero for single precision to know its sign; written
// for size, not speed: one sample pair per step, every byte read on its own and only inside the
// buffer.  The count over the wave, the same in every lane.
template <int ROW>
__device__ __noinline__ int wide_zc_exact(const unsigned char *pcm, long long src, long long planEnd, int l, int nz, double preK,
                                          bool firstTile, int wlen)
{
    int zc = 0;
#pragma unroll 1
    for (int r = 0; r < nz; ++r) {
        const int s0 = 2 * (l + kWideG * r);
        double x[4]; // instants s0 - 1 .. s0 + 2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long at = src + (long long)(s0 - 1 + i) * 6;
            int c[2] = {0, 0};
            if (at >= 0 && at + 6 <= planEnd) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const unsigned b0 = pcm[at + 3 * ch], b1 = pcm[at + 3 * ch + 1], b2 = pcm[at + 3 * ch + 2];
                    c[ch] = ((int)((b0 | (b1 << 8) | (b2 << 16)) << 8)) >> 8;
                }
            }
            x[i] = ROW == 0 ? (double)c[0] : ROW == 1 ? (double)c[1] : ROW == 2 ? 0.5 * (double)(c[0] + c[1]) : (double)(c[0] - c[1]);
        }
        double y1 = sub_mul_exact(x[1], x[0], preK);
        const double y2 = sub_mul_exact(x[2], x[1], preK), y3 = sub_mul_exact(x[3], x[2], preK);
        if (firstTile && s0 == 0) y1 = x[1] * (1.0 - preK); // s[1] *= 1 - k
        const bool c01 = y1 * y2 < 0.0 && s0 + 1 < wlen, c12 = y2 * y3 < 0.0 && s0 + 2 < wlen;
        zc += __builtin_popcountll(__builtin_amdgcn_ballot_w64(c01)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(c12));
    }
    return zc;
}

// 1024-point complex FFT of the lane's 16 points v[r] = z[l + 64 r] into the work area Z (natural
// order, padded): Stockham stages of radix 16, 16 and 4 as fft_stages makes them.  twt: the
// twiddle table ([r-1][l mod 16] for the second stage, [r-1][l + 64 b] behind it for the third);
// a stage's factors are requested while the stage before it runs and exchanges, so that they
// occupy registers for one stage only.
__device__ __forceinline__ void wide_fft(cpx<float> (&v)[kWidePpl], cpx<float> *Z, int l, const cpx<float> *twt, const unsigned lu)
{
    typedef float T;
    constexpr int G = kWideG;
    cpx<T> twB[15];
#pragma unroll
    for (int r = 1; r < 16; ++r) twB[r - 1] = (twt + (r - 1) * 16)[lu & 15u]; // sub-transform size 16: [r-1][l mod 16]
    Radix<T, 16>::run(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) (Z + 17 * l)[r] = v[r]; // zpad(16 l + r) = 17 l + r
    group_sync<G>();
    lds_read_strided<16, (64 + 4) * 8>(v, &Z[zpad(l)]);
#pragma unroll
    for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], twB[r - 1]);
    group_sync<G>(); // every read of Z is done before any lane overwrites it
    cpx<T> twC[12];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 1; r < 4; ++r) twC[b * 3 + r - 1] = (twt + (15 * 16 + (r - 1) * 256 + 64 * b))[lu]; // size 256: [r-1][l + 64 b]
    Radix<T, 16>::run(v);
    {
        const int o = ((l >> 4) << 8) + (l & 15);
        cpx<T> *wp = Z + zpad(o); // zpad(o + 16 r) = zpad(o) + 17 r
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[17 * r] = v[r];
    }
    group_sync<G>();
#pragma unroll
    for (int b = 0; b < 4; ++b) lds_read_strided<4, (256 + 16) * 8>(&v[4 * b], Z + zpad(l) + 68 * b); // zpad(l + 64 b) = zpad(l) + 68 b
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 1; r < 4; ++r) v[4 * b + r] = cmul(v[4 * b + r], twC[b * 3 + r - 1]);
    group_sync<G>();
#pragma unroll
    for (int b = 0; b < 4; ++b) Radix<T, 4>::run(&v[4 * b]);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) (Z + zpad(l))[68 * b + 272 * r] = v[4 * b + r]; // zpad(l + 64 b + 256 r)
    group_sync<G>();
}

// Sub-band sums straight from the lane's registers when a band is a whole number of 64-bin strides
// (S = 2, 4, 8, 16): bin l + 64 q lies in band q / W and bin 1024 - l - 64 q in band (15 - q) / W for
// l >= 1, W = 16 / S; lane 0 holds bins 64 q and 1024 - 64 q (512 for q = 0).  e[b] = sum of the
// band's powers, ek[b] = sum of (bin + 1) x power, over the lane's 16 bins.
template <int S>
__device__ __forceinline__ void wide_bands(const float (&pa)[8], const float (&pb)[8], int l, float (&e)[S], float (&ek)[S])
{
    constexpr int W = 16 / S, NC = kWideNc, G = kWideG;
#pragma unroll
    for (int b = 0; b < S; ++b) e[b] = 0.0f, ek[b] = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = l + G * q;
        const int k2 = k == 0 ? NC / 2 : NC - k;
        e[q / W] += pa[q];
        ek[q / W] += (float)(k + 1) * pa[q];
        // (lane 0's second bin sits one band higher than the other lanes' whenever 64 q starts a band)
        const int bHi = (15 - q) / W, bLane0 = q == 0 ? (S / 2) : (16 - q) / W;
        const float p2 = pb[q], w2 = (float)(k2 + 1) * pb[q];
        if (bHi == bLane0) {
            e[bHi] += p2;
            ek[bHi] += w2;
        } else {
            e[bHi] += l == 0 ? 0.0f : p2;
            ek[bHi] += l == 0 ? 0.0f : w2;
            e[bLane0 < S ? bLane0 : S - 1] += l == 0 ? p2 : 0.0f;
            ek[bLane0 < S ? bLane0 : S - 1] += l == 0 ? w2 : 0.0f;
        }
    }
}

// Totals of the S bands over the frame's 64 lanes, through the (free) work area Pw: rows 0 .. S-1
// hold the lanes' e, rows S .. 2S-1 their ek, [row][lane] with rows of 64 + 4 (the reads then spread
// over the banks).  Lane (i, j) = (l % 2S, l / 2S) adds lanes 2S j .. 2S j + 2S - 1 of row i, lane
// exchanges add the 32 / S column groups: lanes with l % 2S < S end with the band totals of e, the
// others of a row of 2S lanes with those of ek.
template <int S>
__device__ __forceinline__ float wide_band_totals(const float (&pa)[8], const float (&pb)[8], int l, float *Pw)
{
    constexpr int R = 2 * S, NV = R / 4, G = kWideG;
    float e[S], ek[S];
    wide_bands<S>(pa, pb, l, e, ek);
#pragma unroll
    for (int g = 0; g < S; ++g) {
        Pw[g * 68 + l] = e[g];
        Pw[(S + g) * 68 + l] = ek[g];
    }
    group_sync<G>();
    const float4 *q4 = reinterpret_cast<const float4 *>(Pw + (l % R) * 68 + (l / R) * R);
    float4 sv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) sv[i] = q4[i];
    float part[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) part[i] = (sv[i].x + sv[i].y) + (sv[i].z + sv[i].w);
#pragma unroll
    for (int n = NV; n > 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n / 2; ++i) part[i] = part[2 * i] + part[2 * i + 1];
    float tot = part[0];
#pragma unroll
    for (int o = R; o < 64; o <<= 1) tot += __shfl_xor(tot, o, 64);
    group_sync<G>();
    return tot;
}

} // namespace

// NZ: first-stage inputs of a lane that can be non-zero, ceil(wlen / 128).  ZC: zero-crossing rate
// on; the samples are then formed in double with separately rounded product and difference, as
// the reference's code does, so that the sign of a near-cancelling sample is the reference's.
template <int NZ, bool ZC>
__global__ __launch_bounds__(kWideWaves * 64) void k_wide1024(const FrontArgs<float> a)
{
    constexpr int NC = kWideNc, G = kWideG, PPL = kWidePpl, NPAIR = PPL / 2, TEAM = 64, WAVES = kWideWaves;
        typedef float T;
    STAMP_DECL
    STAMP_START;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const int tt0 = tid & 63;
    int tt = tt0, l = tt0;

    // LDS: [window pairs | the lanes' mel part tables | run slots | cepstral table (when there is room;
    // rows of P + 1 elements)] shared by the workgroup, then one region per wave: [work area | log-mel rows
    // of the frame]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWinElems = 2 * G * NZ, kSegElems = 3 * 64 * 4;
    cpx<T> *winL = reinterpret_cast<cpx<T> *>(smem);
    int4 *segL = reinterpret_cast<int4 *>(smem + kWinElems * sizeof(T));
    int *combCtr = reinterpret_cast<int *>(segL + 3 * 64); // the next comb of the workgroup's sequence to be claimed
    T *dctL = reinterpret_cast<T *>(smem) + kWinElems + kSegElems + 4;
    const bool dctInLds = a.rowPitch > 0; // (rowPitch: elements of the table's copy)
    unsigned char *region = smem + ((size_t)a.tabElems + (size_t)wave * a.teamElems) * sizeof(T);
    T *Zt = reinterpret_cast<T *>(region);
    cpx<T> *Z = reinterpret_cast<cpx<T> *>(Zt);
    cpx<T> *AB = Z;                                               // per bin (loWt p, p - loWt p)
    T *bandE = Zt + 2 * NC;                                       // sub-band sums (general band widths)
    T *Pw = Zt;                                                   // plain power values, before AB
    T *fbkSep = Zt + a.rawOrZElems;                               // log-mel values [channel][row], zeros up to a multiple of 4 channels (cepstral mode)

    if (!a.fbankFlag) {
        for (int i = tt0; i < ((a.P + 3) & ~3) * a.V; i += 64) fbkSep[i] = (T)0;
    }
    for (int i = tid; i < G * NZ; i += WAVES * 64) winL[i] = reinterpret_cast<const cpx<T> *>(a.window)[i];
    // mel segment parts of the lanes (round 0 and 1) and the part ranges of the lanes' channels
    for (int i = tid; i < 3 * 64; i += WAVES * 64) segL[i] = reinterpret_cast<const int4 *>(a.laneSeg)[i];
    if (tid == 0) *combCtr = WAVES; // (wave w starts on comb w)
    if (dctInLds) {
        const int rows = a.M + a.c0Flag, pitch = a.tabPitch;
        for (int i = tid; i < rows * pitch; i += WAVES * 64) {
            const int j = i / pitch, k = i - j * pitch;
            dctL[i] = k < a.P ? a.dctT[j * a.dctPitch + k] : (T)0;
        }
    }
    __syncthreads();
    // Tables.  The twiddle factors of the second and third FFT stage, the split factors and the
    // filterbank weights are read through L1 (the same few KB for every wave), each well before its
    // use, long after the previous round's result stores and before this round's: waiting for a
    // table then never means waiting for stores issued just ahead of it.  What is needed at a
    // round's very start -- the window pairs -- and the lanes' mel part tables come from LDS.
    const cpx<T> *twt = reinterpret_cast<const cpx<T> *>(a.tw), *splitTab = reinterpret_cast<const cpx<T> *>(a.splitTw);
    const float kf = (float)a.preK;
    const float zcThr = (float)(4.0 * __builtin_fabs((double)kf - a.preK)) + 1e-12f;
    const int JW = a.M + a.c0Flag;
    const int F_V = a.V, F_S = a.S;

    // The workgroup's stream of tiles: runs of WAVES x C consecutive frames, run r of workgroup b
    // being run b + r gridDim.x of the launch; comb k mod WAVES of such a run is its frames k mod
    // WAVES, k mod WAVES + WAVES, ...  A wave starts on the comb of its own index and claims every
    // further comb from a counter in LDS when it has used up the one before: the waves of a
    // workgroup work on neighbouring frames at about the same time -- the bytes consecutive frames
    // share are fetched once and found in L2 by the others -- and none of them ever waits for
    // another.  A claimed comb whose first frame lies behind the last tile ends the wave's stream
    // (every later comb starts further on).
    const int C = a.tilesPerTeam;
    int chunkBase = wave, chunkPos = 0; // (comb index in the workgroup's sequence, position in it)
    auto comb_first = [&](int k) { return ((long long)blockIdx.x + (long long)(k / WAVES) * gridDim.x) * (WAVES * C) + k % WAVES; };
    auto next_tile = [&]() {
        if (chunkPos == C) {
            int k = 0;
            if (tt == 0) k = atomicAdd(combCtr, 1);
            chunkBase = sgpr(k);
            chunkPos = 0;
        }
        const long long id = comb_first(chunkBase) + WAVES * chunkPos;
        if (id >= a.numTiles) return -1;
        ++chunkPos;
        return (int)id;
    };
    auto next_tile_s = [&]() {
        const int id = sgpr(next_tile());
        chunkBase = sgpr(chunkBase);
        chunkPos = sgpr(chunkPos);
        return id;
    };
    if (comb_first(chunkBase) >= a.numTiles) {
        STAMP_FLUSH;
        return;
    }
    const TileDesc *tab = a.tiles;
    int id = next_tile_s();
    WideTile cur = wide_desc(tab, id, true);
    id = next_tile_s();
    WideTile nx = wide_desc(tab, id < 0 ? 0 : id, id >= 0);
    id = next_tile_s();
    WideTile nn = wide_desc(tab, id < 0 ? 0 : id, id >= 0);

    // The frame's raw bytes as the lanes need them, straight from memory: own[r] = the 12 bytes of
    // the lane's sample pair n = tt + 64 r (a wave's load is 768 consecutive bytes), pred = the 8
    // bytes in front of the frame's first instant.  One path for every frame, so that the loaded
    // registers have a single definition and nothing waits for them here: a frame at the last
    // bytes of the buffer (edge) moves the loads that would pass the end back to it, one at its
    // first bytes reads the predecessor from the buffer's start; fix_raw then shifts those bytes
    // into place.  (Only pairs from kClampFrom on can lie behind the shortest frame's last byte.)
    constexpr int kClampFrom = NZ == 14 ? 13 : 8;
    auto load_raw = [&](const WideTile &t, unsigned (&own)[NZ][3], unsigned (&pred)[2]) {
        const long long src = t.src();
        const unsigned char *g = a.pcm + src;
        const int lastAt = t.edge() ? (int)min(a.planEnd - src, (long long)(1 << 30)) - 12 : 0x7fffffff;
#pragma unroll
        for (int r = 0; r < NZ; ++r) {
            u3u q;
            // (a uniform pointer per pair plus one unsigned lane offset: scalar address arithmetic)
            if (r < kClampFrom) q = *reinterpret_cast<const u3u *>(g + 12 * G * r + 12u * (unsigned)tt);
            else q = *reinterpret_cast<const u3u *>(g + min(12 * (tt + G * r), lastAt));
            own[r][0] = q.x, own[r][1] = q.y, own[r][2] = q.z;
        }
        const u2u pq = *reinterpret_cast<const u2u *>(g - (src >= 8 ? 8 : src));
        pred[0] = pq.x, pred[1] = pq.y;
    };
    auto fix_raw = [&](const WideTile &t, unsigned (&own)[NZ][3], unsigned (&pred)[2]) {
        if (t.edge()) {
            const long long src = t.src();
            const int lastAt = (int)min(a.planEnd - src, (long long)(1 << 30)) - 12;
#pragma unroll
            for (int r = kClampFrom; r < NZ; ++r) {
                const int back = max(12 * (tt + G * r) - lastAt, 0); // bytes the load was moved back by
                const int words = back >> 2;
                const unsigned q0 = own[r][0], q1 = own[r][1], q2 = own[r][2];
                const unsigned e0 = words == 0 ? q0 : words == 1 ? q1 : words == 2 ? q2 : 0u;
                const unsigned e1 = words == 0 ? q1 : words == 1 ? q2 : 0u;
                const unsigned e2 = words == 0 ? q2 : 0u;
                own[r][0] = __builtin_amdgcn_alignbyte(e1, e0, (unsigned)back & 3u);
                own[r][1] = __builtin_amdgcn_alignbyte(e2, e1, (unsigned)back & 3u);
                own[r][2] = __builtin_amdgcn_alignbyte(0u, e2, (unsigned)back & 3u);
            }
            if (src < 8) { // the 8 bytes were read from the buffer's start: its first `src` are the wanted last ones
                const unsigned long long both = ((unsigned long long)pred[1] << 32) | pred[0];
                const unsigned long long moved = src > 0 ? both << (8 * (8 - (int)src)) : 0ull;
                pred[0] = (unsigned)moved, pred[1] = (unsigned)(moved >> 32);
            }
        }
    };

    // The samples of the round to come, carried from round to round: a round asks for the next
    // round's bytes once its spectrum has been reduced to per-bin values, and turns them into that
    // row's samples just before its own results go out, so that no round starts by waiting for
    // memory behind the stores of the one before it.
    float xa[NZ], xb[NZ], pred0;
    auto to_samples = [&](const int row, const unsigned (&own)[NZ][3], const unsigned (&pred)[2]) {
        switch (row) { // (the same for every lane)
        case 0: wide_samples<NZ, 0>(xa, xb, pred0, own, pred); break;
        case 1: wide_samples<NZ, 1>(xa, xb, pred0, own, pred); break;
        case 2: wide_samples<NZ, 2>(xa, xb, pred0, own, pred); break;
        default: wide_samples<NZ, 3>(xa, xb, pred0, own, pred); break;
        }
    };
    {
        unsigned own[NZ][3], pred[2];
        load_raw(cur, own, pred);
        fix_raw(cur, own, pred);
        to_samples(0, own, pred);
    }

    // One (frame, row) round (LAST: the frame's last row, which also forms the cepstra).
    WideTile d3 = {0u, 0u, 0};
    auto round = [&](const int vv, auto lastTag) {
        constexpr bool LAST = decltype(lastTag)::value;
        // (the lane index is made opaque once per round: what derives from it -- addresses, bin
        // weights, masks -- is then recomputed here in a few instructions instead of being kept in
        // registers, or spilled, across the whole loop)
        asm volatile("" : "+v"(tt));
        l = tt & 63; // (says the range: l + 64 q is then known to be zero for q = 0 only)
        float *outRow = a.out + (long long)cur.outRow * a.step;
        const bool firstTile = cur.first();
        {
            STAMP(0);
            float *fo = outRow + vv * a.D;
            // ---- the lane's points (2n, 2n+1), n = l + 64 r, of derived row vv from its samples
            //      (rows main.c:198-207): PreEmphasise main.c:208 (first-sample rule for the signal's
            //      first sample), Ham main.c:229, te mfcc.c:138-142
            cpx<T> v[PPL];
            int zc = 0;
            const bool unsure = wide_expand<NZ, ZC>(v, zc, xa, xb, pred0, l, winL, a.preK, kf, zcThr, firstTile, a.wlen);
            if constexpr (ZC) {
                if (__builtin_amdgcn_ballot_w64(unsure) != 0ull) {
                    const long long src = cur.src();
                    switch (vv) {
                    case 0: zc = wide_zc_exact<0>(a.pcm, src, a.planEnd, l, NZ, a.preK, firstTile, a.wlen); break;
                    case 1: zc = wide_zc_exact<1>(a.pcm, src, a.planEnd, l, NZ, a.preK, firstTile, a.wlen); break;
                    case 2: zc = wide_zc_exact<2>(a.pcm, src, a.planEnd, l, NZ, a.preK, firstTile, a.wlen); break;
                    default: zc = wide_zc_exact<3>(a.pcm, src, a.planEnd, l, NZ, a.preK, firstTile, a.wlen); break;
                    }
                }
            }
            T te = 0;
            if (a.energyFlag) {
                cpx<T> e2 = {0, 0};
#pragma unroll
                for (int r = 0; r < NZ; ++r) e2 += v[r] * v[r];
                te = group_sum<T, G>((T)4 * (e2.x + e2.y), nullptr, tt); // (half-scaled window)
            }
            STAMP(1);
            const unsigned lu = (unsigned)tt; // (one unsigned lane offset under uniform row pointers: scalar address arithmetic)
            STAMP(2);

            // ---- 1024-point complex FFT of z[n] = x[2n] + i x[2n+1] (Realft mfcc.c:262-270)
            wide_fft(v, Z, tt, twt, lu);
            cpx<T> swreg[NPAIR];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) swreg[q] = (splitTab + G * q)[lu];
            STAMP(3);

            // ---- real-FFT split and power (mfcc.c:274-290, ek of mfcc.c:153-155)
            T pa[NPAIR], pb[NPAIR];
            {
                constexpr int QS = (G + G / 16) * 8;
                cpx<T> zas[NPAIR], rev[NPAIR];
                lds_read_strided<NPAIR, QS>(zas, &Z[zpad(tt)]);
                lds_read_strided<NPAIR, QS>(rev, &Z[zpad(NC - tt - G * (NPAIR - 1))]); // (lane 0: one point past the area, unused)
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) split_power<T>(zas[q], rev[NPAIR - 1 - q], swreg[q], pa[q], pb[q]);
                // bin 0 packs Re + Im of Z[0] (mfcc.c:288-290); bin nc/2 is |Z[nc/2]|^2
                const cpx<T> z0 = Z[0], zh = Z[zpad(NC / 2)];
                const T p0 = (T)4 * (z0.x + z0.y) * (z0.x + z0.y);
                const T ph = (T)4 * (zh.x * zh.x + zh.y * zh.y);
                pa[0] = l == 0 ? p0 : pa[0];
                pb[0] = l == 0 ? ph : pb[0];
            }
            // brightness without sub-bands: power-weighted mean bin index straight from registers
            T regTotal = 0, regNum = 0;
            if (a.brightFlag && F_S == 0) {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    const int k2 = k == 0 ? NC / 2 : NC - k;
                    regTotal += pa[q] + pb[q];
                    regNum += (T)(k + 1) * pa[q] + (T)(k2 + 1) * pb[q];
                }
                regTotal = group_sum<T, G>(regTotal, nullptr, tt);
                regNum = group_sum<T, G>(regNum, nullptr, tt);
            }
            group_sync<G>(); // every lane has read its Z points: the work area can be reused
            T lwreg[2 * NPAIR]; // weights of bins l + 64 q and nc - l - 64 q (lane 0, q = 0: bin nc / 2)
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                // (uniform row pointers indexed by one unsigned lane offset: scalar address arithmetic)
                lwreg[2 * q] = (a.loWt + G * q)[lu];
                lwreg[2 * q + 1] = q == 0 ? a.loWt[lu == 0u ? (unsigned)(NC / 2) : (unsigned)NC - lu] : (a.loWt + (NC - G * q - 63))[63u - lu];
            }
            STAMP(4);

            // ---- sub-band energies / brightness (main.c:235-236): from the registers when a band is a
            //      whole number of 64-bin strides, else from the plain power values in the work area
            const bool bandsFromRegs = F_S == 2 || F_S == 4 || F_S == 8 || F_S == 16;
            T bandOut = 0, brightOut = 0; // (stored with the round's other results, at its end)
            if (bandsFromRegs) {
                T tot, totals;
                if (F_S == 8) {
                    tot = wide_band_totals<8>(pa, pb, l, Pw);
                    totals = group_sum<T, 8>(tot, nullptr, tt);
                } else if (F_S == 4) {
                    tot = wide_band_totals<4>(pa, pb, l, Pw);
                    totals = group_sum<T, 4>(tot, nullptr, tt);
                } else if (F_S == 2) {
                    tot = wide_band_totals<2>(pa, pb, l, Pw);
                    totals = group_sum<T, 2>(tot, nullptr, tt);
                } else {
                    tot = wide_band_totals<16>(pa, pb, l, Pw);
                    totals = group_sum<T, 16>(tot, nullptr, tt);
                }
                // total power (lanes 0 .. S-1) and its bin-weighted sum (lanes S .. 2S-1): sums over
                // aligned runs of S lanes
                const T num = __shfl(totals, F_S, 64);
                bandOut = tot / totals;
                brightOut = num / totals / (T)NC;
            } else if (F_S > 0) {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    const int k2 = k == 0 ? NC / 2 : NC - k;
                    Pw[k] = pa[q];
                    Pw[k2] = pb[q];
                }
                group_sync<G>();
                const int nseg = F_S, width = NC / nseg;
                const int subAt = a.M + a.c0Flag + a.energyFlag + a.zcFlag + a.brightFlag;
                int lpb = 1;
                while (lpb * 2 * nseg <= G && lpb < 64) lpb *= 2;
                const int perPass = G / lpb;
                for (int b0 = 0; b0 < nseg; b0 += perPass) {
                    const int b = b0 + l / lpb, sl = l & (lpb - 1);
                    T e = 0, ek = 0;
                    if (b < nseg) {
                        const int lo = b * width, hi = (b == nseg - 1) ? NC : lo + width;
                        for (int k = lo + sl; k < hi; k += lpb) {
                            const T pv = Pw[k];
                            e += pv;
                            ek += (T)(k + 1) * pv;
                        }
                    }
                    for (int o = lpb >> 1; o > 0; o >>= 1) {
                        e += __shfl_xor(e, o, 64);
                        ek += __shfl_xor(ek, o, 64);
                    }
                    if (b < nseg && sl == 0) {
                        bandE[b] = e;
                        bandE[nseg + b] = ek;
                    }
                }
                group_sync<G>();
                if (l == 0) {
                    T total = 0, num = 0;
                    for (int g = 0; g < nseg; ++g) {
                        total += bandE[g];
                        num += bandE[nseg + g];
                    }
                    if (a.brightFlag) fo[a.M + a.c0Flag + a.energyFlag + a.zcFlag] = num / total / (T)NC;
                    for (int g = 0; g < nseg; ++g) fo[subAt + g] = bandE[g] / total;
                }
            }
            if (F_S == 0 && a.brightFlag) brightOut = regNum / regTotal / (T)NC;
            const T teOut = a.energyFlag ? tlog<T>(te) : (T)0;

            STAMP(7);
            // ---- the bytes of the round to come (this frame's next row, or the next frame's first)
            unsigned own[NZ][3], pred[2];
            if constexpr (LAST) {
                load_raw(nx.valid() ? nx : cur, own, pred); // (behind the last frame: this one's again, unused)
            } else {
                load_raw(cur, own, pred);
            }
            // ---- per bin, exactly the two products of mfcc.c:160-162: t1 = loWt ek and ek - t1
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                const int k = l + G * q;
                const int k2 = k == 0 ? NC / 2 : NC - k;
                const T t1 = lwreg[2 * q] * pa[q];
                const T t2 = lwreg[2 * q + 1] * pb[q];
                AB[k] = {t1, pa[q] - t1};
                AB[k2] = {t2, pb[q] - t2};
            }
            group_sync<G>();
            STAMP(5);
            // ---- mel filterbank (mfcc.c:152-163): the bins whose lower channel is c form one
            //      segment, cut by the host into parts of at most 16 bins; a lane sums T = sum t1 and
            //      U = sum (ek - t1) of its part (one per round, two rounds) in bin order, the part sums
            //      go to the work area, and the lane of channel b adds the parts of U[b-1] and of T[b]
            //      in part (= bin) order; then log(max(., 1)) (mfcc.c:171-176)
            cpx<T> ps[2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int4 sg = segL[64 * rd + l];
                cpx<T> acc = {0, 0};
                if (rd < a.maxSeg) {
                    for (int h = 0; h < a.melPieces; ++h) { // melPieces x 8 bins per part at most
                        cpx<T> vq[8];
                        lds_read_strided<8, 8>(vq, &AB[sg.x + 8 * h]);
                        // reads past the part stay inside the work area (finite values of this round)
                        // and enter with the factor clamp(left - i) = 0; the others with 1
                        const float left = (float)(sg.y - sg.x - 8 * h);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float m = __builtin_amdgcn_fmed3f(left - (float)i, 0.0f, 1.0f);
                            acc = cfma_r(vq[i], m, acc);
                        }
                    }
                }
                ps[rd] = acc;
            }
            STAMP(8);
            group_sync<G>(); // every part has been read: the pairs' place takes the part sums
            {
                const int slot0 = segL[l].z, slot1 = segL[64 + l].z;
                if (slot0 >= 0) {
                    Pw[slot0] = ps[0].x;
                    Pw[128 + slot0] = ps[0].y;
                }
                if (a.maxSeg > 1 && slot1 >= 0) {
                    Pw[slot1] = ps[1].x;
                    Pw[128 + slot1] = ps[1].y;
                }
            }
            group_sync<G>();
            T lv = 0;
            if (l < a.P) {
                // (at most 10 parts per segment: all reads go out together, the sums stay in part order)
                const int4 chanParts = segL[128 + l];
                T pu[10], pt[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    pu[j] = Pw[128 + chanParts.x + j];
                    pt[j] = Pw[chanParts.z + j];
                }
                T mu = 0, mt = 0;
                const float nu = (float)chanParts.y, nt = (float)chanParts.w;
#pragma unroll
                for (int j = 0; j < 10; ++j) { // (slots past the segment's parts: finite values, factor 0)
                    mu = __builtin_fmaf(pu[j], __builtin_amdgcn_fmed3f(nu - (float)j, 0.0f, 1.0f), mu);
                    mt = __builtin_fmaf(pt[j], __builtin_amdgcn_fmed3f(nt - (float)j, 0.0f, 1.0f), mt);
                }
                const T m = mu + mt;
                lv = tlog<T>(m < (T)1 ? (T)1 : m);
            }
            fix_raw(LAST && nx.valid() ? nx : cur, own, pred);
            to_samples(LAST ? 0 : vv + 1, own, pred);
            // ---- the round's results (row layout main.c:241-249)
            if (bandsFromRegs) {
                if (l < F_S) fo[a.M + a.c0Flag + a.energyFlag + a.zcFlag + a.brightFlag + l] = bandOut;
            }
            if (l == 0) {
                if (a.brightFlag && (bandsFromRegs || F_S == 0)) fo[a.M + a.c0Flag + a.energyFlag + a.zcFlag] = brightOut;
                if (a.energyFlag) fo[a.M + a.c0Flag] = teOut;                                     // main.c:243
                if (ZC) fo[a.M + a.c0Flag + a.energyFlag] = (T)zc / (T)(a.wlen - 1);              // main.c:244
            }
            if (l < a.P) {
                if (a.fbankFlag) fo[l] = lv;
                else fbkSep[l * F_V + vv] = lv; // [channel][row]
            }
            team_sync<TEAM>(); // the work area is free for the next round
            STAMP(6);
        }
        if (LAST && !a.fbankFlag) {
            // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208) of the frame's
            //      rows: lane j forms coefficient j of every row from its table row (row M is all
            //      ones: C0) and the frame's log-mel values, which all lanes read together.
            //      (One instance per row count and per place of the table: plain LDS or global reads.)
            const int P4 = (a.P + 3) & ~3;
            auto cepstra = [&](auto rowsTag, auto paddedTag, const T *table, const int pitch) {
                constexpr int VV = decltype(rowsTag)::value;
                constexpr bool PADDED = decltype(paddedTag)::value; // table rows hold zeros up to P4
                for (int j0 = 0; j0 < JW; j0 += 64) {
                    const int j = j0 + l;
                    const T *row = table + (j < JW ? j : JW - 1) * pitch;
                    T acc[VV];
#pragma unroll
                    for (int v = 0; v < VV; ++v) acc[v] = 0;
                    for (int k0 = 0; k0 < P4; k0 += 4) {
                        T tv[4], f[4 * VV];
#pragma unroll
                        for (int u = 0; u < 4; ++u) tv[u] = row[PADDED ? k0 + u : min(k0 + u, a.P - 1)];
#pragma unroll
                        for (int v = 0; v < VV; ++v) {
                            const float4 q = *reinterpret_cast<const float4 *>(fbkSep + k0 * VV + 4 * v);
                            f[4 * v] = q.x, f[4 * v + 1] = q.y, f[4 * v + 2] = q.z, f[4 * v + 3] = q.w;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < VV; ++v) acc[v] += tv[u] * f[u * VV + v];
                    }
                    if (j < JW) {
#pragma unroll
                        for (int v = 0; v < VV; ++v) outRow[v * a.D + j] = acc[v] * a.mfnorm;
                    }
                }
            };
            auto cepstra_rows = [&](auto paddedTag, const T *table, const int pitch) {
                switch (F_V) {
                case 1: cepstra(std::integral_constant<int, 1>{}, paddedTag, table, pitch); break;
                case 2: cepstra(std::integral_constant<int, 2>{}, paddedTag, table, pitch); break;
                case 3: cepstra(std::integral_constant<int, 3>{}, paddedTag, table, pitch); break;
                default: cepstra(std::integral_constant<int, 4>{}, paddedTag, table, pitch); break;
                }
            };
            if (dctInLds) cepstra_rows(std::true_type{}, dctL, a.tabPitch);
            else cepstra_rows(std::false_type{}, a.dctT, a.dctPitch);
            team_sync<TEAM>();
        }
        STAMP(9);
    };
    for (;;) {
        // the descriptor three frames ahead: asked for now, moved into place at the frame's end
        const int id3 = next_tile_s();
        d3 = wide_desc(tab, id3 < 0 ? 0 : id3, id3 >= 0);
        for (int vv = 0; vv + 1 < F_V; ++vv) round(vv, std::false_type{});
        round(F_V - 1, std::true_type{});
        if (!nx.valid()) break;
        cur = nx;
        nx = nn;
        nn = d3;
    }
    STAMP_FLUSH;
}

// Geometry the host plans with.
int wide1024_nz(int wlen) { return (wlen + 127) / 128; }
int wide1024_waves() { return kWideWaves; }
// elements of the LDS part the waves share in front of the cepstral table: window pairs, mel part tables, run slots
int wide1024_shared_elems(int wlen) { return 2 * 64 * (wide1024_nz(wlen) == 14 ? 14 : 16) + 3 * 64 * 4 + 4; }
// bytes from a frame's first instant that the kernel's loads reach (12 per sample pair, 64 x NZ pairs)
int wide1024_reach_bytes(int wlen) { return 12 * 64 * (wide1024_nz(wlen) == 14 ? 14 : 16); }

bool wide1024_applies(int log2nc, int channels, int bps, int wlen)
{
    return log2nc == kWideLog2Nc && channels == 2 && bps == 3 && wide1024_nz(wlen) >= 9 && wide1024_nz(wlen) <= 16;
}

template <int NZ, bool ZC>
static int launch_wide_t(const FrontArgs<float> &a, int numBlocks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    const hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_wide1024<NZ, ZC>), attrSet);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_wide1024<NZ, ZC>), dim3(numBlocks), dim3(kWideWaves * 64), lds, st, a);
    return (int)hipGetLastError();
}

int launch_wide1024(const FrontArgs<float> &a, int numBlocks, size_t lds, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the two window lengths that matter get their own count of first-stage inputs; the rest share 16
    const int nz = wide1024_nz(a.wlen);
    if (nz == 14) return a.zcFlag ? launch_wide_t<14, true>(a, numBlocks, lds, st) : launch_wide_t<14, false>(a, numBlocks, lds, st);
    return a.zcFlag ? launch_wide_t<16, true>(a, numBlocks, lds, st) : launch_wide_t<16, false>(a, numBlocks, lds, st);
}

} // namespace mfcc
