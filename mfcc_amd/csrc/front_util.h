// Device helpers shared by the static-feature kernels (mfcc_kernels.hip, lean256.hip): team
// geometry, wave-level synchronisation and sums, explicit LDS reads, the register Stockham stages,
// the real-FFT split, and the PCM tile unpack.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "fft_regs.h"
#include "mfcc_dev.h"

namespace mfcc {

// Raises a kernel's dynamic-LDS limit to the CU's 160 KB, once per device and kernel (the launchers
// run concurrently from the device threads of MFCCWapperList / MFCCExtractRecording, hence the
// atomics; a device index beyond the table sets the attribute on every call).  `done`: one
// function-local static table per kernel instantiation.
constexpr int kAttrDevices = 64;
inline hipError_t lds_limit_once(const void *kernel, std::atomic<bool> (&done)[kAttrDevices])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool known = dev >= 0 && dev < kAttrDevices;
    if (known && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && known) done[dev].store(true, std::memory_order_release);
    return e;
}

// PPLOG: log2 of the complex points a lane keeps in registers (4 -> radix-16 stages, 3 -> radix-8,
// 2 -> radix-4).  Fewer points per lane means more lanes per frame, fewer frames in flight per
// wave, smaller work areas and fewer registers, i.e. more resident waves, at the price of more
// exchange stages through LDS.
template <int LOG2NC, int PPLOG>
struct Geo {
    static constexpr int NC = 1 << LOG2NC;                        // complex FFT length = fftN / 2
    static constexpr int PPL0 = 1 << (PPLOG < LOG2NC ? PPLOG : LOG2NC);
    static constexpr int G = NC / PPL0 > kBlock ? kBlock : NC / PPL0; // lanes per frame
    static constexpr int PPL = NC / G;                            // complex points per lane
    static constexpr int RBMAX = PPLOG < 4 ? PPLOG : 4;           // log2 of the largest stage radix
    static constexpr int GROUPS = kBlock / G;                     // frames in flight per workgroup
    // A team is the set of lanes that shares one tile (its sample rows, work areas and staging
    // area).  Frames that fit a wave make every wave its own team: the waves of a workgroup then
    // run independent tile streams with no workgroup barrier inside the tile loop.
    static constexpr int TEAM = G <= 64 ? 64 : kBlock;
    static constexpr int NTEAMS = kBlock / TEAM;
    static constexpr int GT = TEAM / G;                           // frames in flight per team
};

// Barrier among the lanes of one team.
template <int TEAM>
__device__ __forceinline__ void team_sync()
{
    if constexpr (TEAM <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Index of complex point i inside a work area: one pad point after every 16, so that the
// transposing exchange between two FFT stages stays (nearly) free of LDS bank conflicts.
__device__ __forceinline__ int zpad(int i) { return i + (i >> 4); }

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

// Orders the LDS traffic of the lanes that share one frame.  Up to 64 lanes sit in one wave,
// whose LDS instructions execute in program order, so only the compiler has to be held back.
template <int G>
__device__ __forceinline__ void group_sync()
{
    if constexpr (G <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <typename V, int G>
__device__ __forceinline__ V group_sum(V v, V *scratch, int tid)
{
    if constexpr (G == 1) {
        return v;
    } else if constexpr (G <= 64 && sizeof(V) == 4) {
        // inside a row of 16 lanes through DPP (no LDS traffic), across rows by lane exchange
        auto dpp_add = [](V x, auto ctrl) {
            const int y = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true);
            return x + __builtin_bit_cast(V, y);
        };
        v = dpp_add(v, std::integral_constant<int, 0xB1>{});                        // quad_perm [1,0,3,2]
        if constexpr (G >= 4) v = dpp_add(v, std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
        if constexpr (G >= 8) v = dpp_add(v, std::integral_constant<int, 0x141>{}); // row_half_mirror
        if constexpr (G >= 16) v = dpp_add(v, std::integral_constant<int, 0x140>{}); // row_mirror
        if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
        if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
        return v;
    } else if constexpr (G <= 64) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) scratch[tid >> 6] = v;
        __syncthreads();
        constexpr int WPG = G / 64; // waves per group
        const int w0 = (tid / G) * WPG;
        V s = scratch[w0];
#pragma unroll
        for (int w = 1; w < WPG; ++w) s += scratch[w0 + w];
        return s;
    }
}

// Group sums without barriers of their own.  Up to 64 lanes: group_part already returns the full
// sum.  Wider groups: it returns the wave's sum and parks it in slot[wave]; group_total adds the
// group's waves and may be called once a barrier has passed (every caller has one on its way).
template <typename V, int G>
__device__ __forceinline__ V group_part(V v, V *slot, int tt)
{
    if constexpr (G <= 64) {
        return group_sum<V, G>(v, slot, tt);
    } else {
        v = group_sum<V, 64>(v, slot, tt);
        if ((tt & 63) == 0) slot[tt >> 6] = v;
        return v;
    }
}
template <typename V, int G>
__device__ __forceinline__ V group_total(V part, const V *slot, int tt)
{
    if constexpr (G <= 64) {
        return part;
    } else {
        constexpr int WPG = G / 64;
        const int w0 = (tt / G) * WPG;
        V sum = slot[w0];
#pragma unroll
        for (int w = 1; w < WPG; ++w) sum += slot[w0 + w];
        return sum;
    }
}

__device__ __forceinline__ int lds_sample(const unsigned *raw, int p, int bps)
{
    const int w = p >> 2;
    const unsigned v = __builtin_amdgcn_alignbyte(raw[w + 1], raw[w], (unsigned)(p & 3));
    if (bps == 1) return (int)(v & 0xffu) - 128; // 8-bit WAV PCM is unsigned, offset 128
    if (bps == 2) return (int)(short)(v & 0xffffu);
    if (bps == 3) return ((int)(v << 8)) >> 8;
    return (int)v;
}

// x - k * xp with the product and the difference rounded separately, as the reference's
// un-fused double arithmetic does; keeps the sign of near-cancelling samples exact for the ZCR.
__device__ __forceinline__ double sub_mul_exact(double x, double xp, double k)
{
#pragma clang fp contract(off)
    const double t = xp * k;
    return x - t;
}

// 16 x 16 x 4 matrix-core product-accumulate: lane (i, g) = (lane % 16, lane / 16) supplies
// A[i][g] and B[g][i]; element r of the result is D[4 g + r][i] (float) or D[4 r + g][i] (double).
template <typename T>
using Acc4 = T __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Acc4<float> mfma16(float a, float b, Acc4<float> c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ Acc4<double> mfma16(double a, double b, Acc4<double> c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ T tlog(T x);
// single precision: the hardware log2 instruction times ln 2 (absolute error below 1e-5 for the
// values that occur here, |ln x| < 50, an order of magnitude inside the 1e-4 log-mel tolerance).
// Its arguments are filterbank sums >= 1 and frame energies of integer samples, never denormal
// (0 gives -inf as log does), so the library routine's rescaling of denormals is left out.
template <>
__device__ __forceinline__ float tlog<float>(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
template <>
__device__ __forceinline__ double tlog<double>(double x) { return log(x); }
template <typename T>
__device__ __forceinline__ T tsqrt(T x);
template <>
__device__ __forceinline__ float tsqrt<float>(float x) { return sqrtf(x); }
template <>
__device__ __forceinline__ double tsqrt<double>(double x) { return sqrt(x); }

// Single-precision complex points from LDS as individual ds_read_b64 instructions.  Left to the
// compiler, neighbouring 8-byte reads are fused into ds_read2_b64, which moves 128 B/clk where two
// ds_read_b64 move 256 B/clk.  N points at a constant byte stride are issued back to back and
// drained by one wait, which also ties the results to that wait for the compiler's scheduler.
typedef cpx<float> v2f;

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}

template <int OFF>
__device__ __forceinline__ v2f lds_rd64(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "LDS offset field is 16 bits");
    v2f r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
    return r;
}

__device__ __forceinline__ void lds_drain(v2f &a, v2f &b, v2f &c, v2f &d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_drain(v2f &a, v2f &b, v2f &c)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void lds_drain(v2f &a, v2f &b)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_drain(v2f &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }

// out[i] = *(base + i * STRIDE bytes), i < N (STRIDE >= 0)
template <int N, int STRIDE, int I = 0>
__device__ __forceinline__ void lds_issue(v2f (&t)[N], unsigned addr)
{
    if constexpr (I < N) {
        t[I] = lds_rd64<I * STRIDE>(addr);
        lds_issue<N, STRIDE, I + 1>(t, addr);
    }
}

template <int N, int STRIDE>
__device__ __forceinline__ void lds_read_strided(cpx<float> *out, const void *base)
{
    v2f t[N];
    lds_issue<N, STRIDE>(t, lds_addr(base));
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) lds_drain(t[i], t[i + 1], t[i + 2], t[i + 3]);
    if constexpr (N % 4 == 3) lds_drain(t[N - 3], t[N - 2], t[N - 1]);
    if constexpr (N % 4 == 2) lds_drain(t[N - 2], t[N - 1]);
    if constexpr (N % 4 == 1) lds_drain(t[N - 1]);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = t[i];
}

// ------------------------------------------------------------------------------------------
// Stockham stages through the LDS work area Z (nc complex), radix <= 16 in registers.
// Stage with sub-transform size NS: butterfly j reads points j + r*nc/R, multiplies point r by
// w^{r (j mod NS) nc/(NS R)}, takes the R-point DFT and writes (j/NS) NS R + (j mod NS) + r NS.
// ------------------------------------------------------------------------------------------

template <typename T, int LOG2NC, int PPLOG>
using LanePoints = cpx<T>[Geo<LOG2NC, PPLOG>::PPL]; // the complex points one lane holds

// Twiddles are stored per stage, [r-1][j mod NS] = exp(-2 pi i r (j mod NS) / (NS R)), so that the
// lanes of a wave read consecutive (or identical) 8-byte words: no LDS bank conflicts.  TWOFF is
// the complex offset of this stage's block in that table (the blocks follow each other).
// TWREG: the lane's twiddles of all stages were loaded once into the register array twr (see
// fft_load_twiddles); TWR is this stage's first entry there.

// Twiddle factors one lane uses over all stages: (R - 1) per butterfly of every stage but the first.
template <int LOG2NC, int PPLOG>
constexpr int fft_twiddle_regs()
{
    using GEO = Geo<LOG2NC, PPLOG>;
    int n = 0;
    for (int logns = 0; logns < LOG2NC;) {
        const int rb = (LOG2NC - logns) < GEO::RBMAX ? (LOG2NC - logns) : GEO::RBMAX;
        if (logns > 0) n += (GEO::PPL >> rb) * ((1 << rb) - 1);
        logns += rb;
    }
    return n > 0 ? n : 1;
}

template <typename T, int LOG2NC, int PPLOG, int LOGNS = 0, int TWOFF = 0, int TWR = 0>
__device__ __forceinline__ void fft_load_twiddles(cpx<T> *twr, int l, const cpx<T> *tw)
{
    using GEO = Geo<LOG2NC, PPLOG>;
    constexpr int G = GEO::G, PPL = GEO::PPL;
    constexpr int RB = (LOG2NC - LOGNS) < GEO::RBMAX ? (LOG2NC - LOGNS) : GEO::RBMAX;
    constexpr int R = 1 << RB, NS = 1 << LOGNS, NB = PPL / R;
    if constexpr (LOGNS > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int m = (l + G * b) & (NS - 1);
#pragma unroll
            for (int r = 1; r < R; ++r) twr[TWR + b * (R - 1) + r - 1] = tw[TWOFF + (r - 1) * NS + m];
        }
    }
    if constexpr (LOGNS + RB < LOG2NC)
        fft_load_twiddles<T, LOG2NC, PPLOG, LOGNS + RB, TWOFF + (LOGNS > 0 ? (R - 1) * NS : 0),
                          TWR + (LOGNS > 0 ? NB * (R - 1) : 0)>(twr, l, tw);
}

// KEEPLAST: the last stage's results stay in the lane's registers (v[r] = point l + r G, which
// needs a single butterfly per lane there) instead of going to the work area.
template <typename T, int LOG2NC, int PPLOG, int LOGNS, int TWOFF, bool TWLDS, bool TWREG = false, int TWR = 0,
          bool KEEPLAST = false>
__device__ __forceinline__ void fft_stages(LanePoints<T, LOG2NC, PPLOG> &v, cpx<T> *Z, int l,
                                           const cpx<T> *tw, const cpx<T> *twr = nullptr)
{
    using GEO = Geo<LOG2NC, PPLOG>;
    constexpr int NC = GEO::NC, G = GEO::G, PPL = GEO::PPL;
    constexpr int RB = (LOG2NC - LOGNS) < GEO::RBMAX ? (LOG2NC - LOGNS) : GEO::RBMAX;
    constexpr int R = 1 << RB, NS = 1 << LOGNS, NB = PPL / R, STRIDE = NC / R;
    if constexpr (LOGNS > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = l + G * b;
            // zpad(j + r STRIDE) = zpad(j) + r (STRIDE + STRIDE / 16): STRIDE is a multiple of 16
            if constexpr (sizeof(T) == 4 && STRIDE % 16 == 0) {
                lds_read_strided<R, (STRIDE + STRIDE / 16) * 8>(&v[b * R], &Z[zpad(j)]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) v[b * R + r] = Z[zpad(j + r * STRIDE)];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int m = (l + G * b) & (NS - 1);
            cpx<T> w[R > 1 ? R - 1 : 1];
            if constexpr (TWREG) {
#pragma unroll
                for (int r = 1; r < R; ++r) w[r - 1] = twr[TWR + b * (R - 1) + r - 1];
            } else if constexpr (sizeof(T) == 4 && TWLDS && R > 1) {
                lds_read_strided<R - 1, NS * 8>(w, &tw[TWOFF + m]);
            } else {
#pragma unroll
                for (int r = 1; r < R; ++r) w[r - 1] = tw[TWOFF + (r - 1) * NS + m];
            }
#pragma unroll
            for (int r = 1; r < R; ++r) v[b * R + r] = cmul(v[b * R + r], w[r - 1]);
        }
        group_sync<G>(); // every read of Z is done before any lane overwrites it
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) Radix<T, R>::run(&v[b * R]);
    if constexpr (KEEPLAST && LOGNS + RB == LOG2NC) {
        static_assert(NB == 1 && NS == G, "the last stage must be one butterfly per lane");
        return;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = l + G * b;
        const int o = ((j >> LOGNS) << (LOGNS + RB)) + (j & (NS - 1));
#pragma unroll
        for (int r = 0; r < R; ++r) Z[zpad(o + r * NS)] = v[b * R + r];
    }
    group_sync<G>();
    if constexpr (LOGNS + RB < LOG2NC)
        fft_stages<T, LOG2NC, PPLOG, LOGNS + RB, TWOFF + (LOGNS > 0 ? (R - 1) * NS : 0), TWLDS, TWREG,
                   TWR + (LOGNS > 0 ? NB * (R - 1) : 0), KEEPLAST>(v, Z, l, tw, twr);
}

// Real-FFT split of one bin pair (mfcc.c:274-287) and the two powers (ek of mfcc.c:153-155).
// za = Z[k], zb = Z[nc - k] of the half-size transform of HALF-scaled input (the window table holds
// half the Hamming values), sw = (cos, sin)(pi k / nc).  With A = za + conj(zb), B = za - conj(zb):
// X[k] = A + w' B and conj(X[nc - k]) = A - w' B = 2 A - X[k], w' = -i conj(sw) = (-sin, -cos).
template <typename T>
__device__ __forceinline__ void split_power(cpx<T> za, cpx<T> zb, cpx<T> sw, T &pa, T &pb)
{
    const cpx<T> A = {za.x + zb.x, za.y - zb.y}, B = {za.x - zb.x, za.y + zb.y};
    const cpx<T> P = {A.x - sw.y * B.x + sw.x * B.y, A.y - sw.y * B.y - sw.x * B.x};
    const cpx<T> Q = {(T)2 * A.x - P.x, (T)2 * A.y - P.y};
    pa = P.x * P.x + P.y * P.y;
    pb = Q.x * Q.x + Q.y * Q.y;
}
template <>
__device__ __forceinline__ void split_power<float>(cpx<float> za, cpx<float> zb, cpx<float> sw, float &pa, float &pb)
{
    cpx<float> A, B, t, P;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(A) : "v"(za), "v"(zb));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(B) : "v"(za), "v"(zb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
        : "=v"(t)
        : "v"(B), "v"(sw), "v"(A));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"
        : "=v"(P)
        : "v"(B), "v"(sw), "v"(t));
    const cpx<float> Q = A * cpx<float>{2.0f, 2.0f} - P;
    pa = P.x * P.x + P.y * P.y;
    pb = Q.x * Q.x + Q.y * Q.y;
}

// Value of x in lane (16 - l) mod 16 of the same row of 16 lanes: mirror the row, then rotate it
// right by one (data-parallel primitives, no LDS traffic).
__device__ __forceinline__ float row_negated_lane(float x)
{
    // (every lane has a valid source lane, so no previous value has to be supplied)
    const int m = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true); // row_mirror
    const int r = __builtin_amdgcn_mov_dpp(m, 0x121, 0xf, 0xf, true);                          // row_ror:1
    return __builtin_bit_cast(float, r);
}

// ------------------------------------------------------------------------------------------
// Tile unpack: raw little-endian PCM bytes in LDS -> derived rows {L, R, mid, side} (main.c:198-207)
// minus the row mean, pre-emphasised over the whole signal (PreEmphasise, call site main.c:208).
// One lane produces 4 consecutive samples of every row and stores them as one 16-byte vector.
// EXACT: double arithmetic with separately rounded multiply and subtract, bit-identical to the
// reference's un-fused double code (needed whenever the sign of a sample is observable: ZCR).
// ------------------------------------------------------------------------------------------

template <typename T, bool EXACT, int CH, int STRIDE>
__device__ __forceinline__ void unpack_tile(T *rows, const unsigned *raw, int rowPitch, int V, int bps,
                                            int mis, int lead, int ns,
                                            const double *mean, double preK, int tid)
{
    using A = typename std::conditional<EXACT, double, float>::type; // arithmetic type
    const A k = (A)preK;
    const int bpi = CH * bps;
    const int nq = (ns + 3) >> 2;
    A mu[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) mu[v] = (A)mean[v];
    for (int q = tid; q < nq; q += STRIDE) {
        const int s0 = 4 * q;
        const bool hasPrev = (lead + s0) > 0;            // false only for the signal's first sample
        const int rel = s0 + lead - 1;                    // raw instant of the predecessor of s0
        const int p0 = mis + (rel < 0 ? 0 : rel) * bpi;
        A x[5][4]; // [instant: predecessor, s0 .. s0+3][row]
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int p = (rel < 0 && i > 0) ? p0 + (i - 1) * bpi : p0 + i * bpi;
            const int c0 = lds_sample(raw, p, bps);
            const int c1 = CH == 2 ? lds_sample(raw, p + bps, bps) : 0;
            x[i][0] = (A)c0 - mu[0];
            if (CH == 2) {
                x[i][1] = (A)c1 - mu[1];
                x[i][2] = (A)0.5 * ((A)c0 + (A)c1) - mu[2];
                x[i][3] = ((A)c0 - (A)c1) - mu[3];
            }
        }
#pragma unroll
        for (int v = 0; v < (CH == 2 ? 4 : 1); ++v) {
            if (v < V) {
                T y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A r;
                    if constexpr (EXACT)
                        r = sub_mul_exact(x[i + 1][v], x[i][v], k);
                    else
                        r = x[i + 1][v] - k * x[i][v];
                    y[i] = (T)r;
                }
                if (!hasPrev) y[0] = (T)(x[1][v] * ((A)1 - k)); // s[1] *= 1 - k
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (s0 + i >= ns) y[i] = (T)0; // keeps the zero tail behind the tile intact
                T *dst = rows + v * rowPitch + s0;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    *reinterpret_cast<double2 *>(dst) = make_double2(y[0], y[1]);
                    *reinterpret_cast<double2 *>(dst + 2) = make_double2(y[2], y[3]);
                }
            }
        }
    }
}

// Fast form of the tile unpack for tiles whose first sample has a predecessor: the raw bytes were
// copied so that the predecessor of sample 0 sits at byte 0 of `raw`, hence the 5 instants a lane
// needs (predecessor + 4 samples) start on a dword boundary and every byte offset inside the
// lane's register window is a compile-time constant.
template <int BPS, int O, int NW>
__device__ __forceinline__ int window_sample(const unsigned (&d)[NW])
{
    constexpr int w = O >> 2, sh = O & 3;
    unsigned v;
    if constexpr (sh + BPS <= 4)
        v = d[w] >> (8 * sh);
    else
        v = __builtin_amdgcn_alignbyte(d[w + 1], d[w], (unsigned)sh);
    if constexpr (BPS == 2)
        return (int)(short)(v & 0xffffu);
    else if constexpr (BPS == 3)
        return ((int)(v << 8)) >> 8;
    else
        return (int)v;
}

template <typename T, bool EXACT, int CH, int BPS, int STRIDE>
__device__ __forceinline__ void unpack_tile_fast(T *rows, const unsigned *raw, int rowPitch, int V,
                                                 int ns, const double *mean, double preK, int tid)
{
    using A = typename std::conditional<EXACT, double, float>::type;
    constexpr int BPI = CH * BPS, NW = (5 * BPI + 3) / 4;
    const A k = (A)preK;
    const int nq = (ns + 3) >> 2;
    A mu[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) mu[v] = (A)mean[v];
    for (int q = tid; q < nq; q += STRIDE) {
        const int s0 = 4 * q;
        const unsigned *src = raw + q * BPI; // byte 4 q BPI: predecessor of sample s0
        unsigned d[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) d[i] = src[i];
        A x[5][CH == 2 ? 4 : 1];
        // unrolled by hand over the 5 instants: the offsets must be template arguments
#define MFCC_INSTANT(i)                                                          \
        {                                                                        \
            const int c0 = window_sample<BPS, (i) * BPI, NW>(d);                 \
            x[i][0] = (A)c0 - mu[0];                                             \
            if constexpr (CH == 2) {                                             \
                const int c1 = window_sample<BPS, (i) * BPI + BPS, NW>(d);       \
                x[i][1] = (A)c1 - mu[1];                                         \
                x[i][2] = (A)0.5 * ((A)c0 + (A)c1) - mu[2];                      \
                x[i][3] = ((A)c0 - (A)c1) - mu[3];                               \
            }                                                                    \
        }
        MFCC_INSTANT(0) MFCC_INSTANT(1) MFCC_INSTANT(2) MFCC_INSTANT(3) MFCC_INSTANT(4)
#undef MFCC_INSTANT
#pragma unroll
        for (int v = 0; v < (CH == 2 ? 4 : 1); ++v) {
            if (v < V) {
                T y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A r;
                    if constexpr (EXACT)
                        r = sub_mul_exact(x[i + 1][v], x[i][v], k);
                    else
                        r = x[i + 1][v] - k * x[i][v];
                    y[i] = s0 + i < ns ? (T)r : (T)0; // keeps the zero tail behind the tile intact
                }
                T *dst = rows + v * rowPitch + s0;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    *reinterpret_cast<double2 *>(dst) = make_double2(y[0], y[1]);
                    *reinterpret_cast<double2 *>(dst + 2) = make_double2(y[2], y[3]);
                }
            }
        }
    }
}

template <typename T, int CH, int BPS, int STRIDE>
__device__ __forceinline__ void unpack_tile_fast_sel(bool exact, T *rows, const unsigned *raw,
                                                     int rowPitch, int V, int ns, const double *mean,
                                                     double preK, int tid)
{
    if constexpr (sizeof(T) == 8) {
        unpack_tile_fast<T, true, CH, BPS, STRIDE>(rows, raw, rowPitch, V, ns, mean, preK, tid);
    } else {
        if (exact)
            unpack_tile_fast<T, true, CH, BPS, STRIDE>(rows, raw, rowPitch, V, ns, mean, preK, tid);
        else
            unpack_tile_fast<T, false, CH, BPS, STRIDE>(rows, raw, rowPitch, V, ns, mean, preK, tid);
    }
}


} // namespace mfcc
