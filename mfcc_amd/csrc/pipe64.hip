// Double-precision static-feature kernels for the two input classes the benchmarks name: mono 16-bit
// PCM with frames of at most 512 samples (k_pipe64: 256 complex points, 16 lanes x 16 points per
// frame, four frames per wave) and 24-bit stereo PCM with frames of 1025 .. 2048 samples (k_pipe64w:
// 1024 complex points, 64 lanes x 16 points, one frame per wave).  The reference computes in double
// throughout (rows main.c:198-207, PreEmphasise main.c:208, zeroCrossingRate main.c:227, Ham
// main.c:229, Wave2FBank mfcc.c:125-177, brightness / sub-bands main.c:235-236, FBank2MFCC /
// FBank2C0 mfcc.c:180-208, row layout main.c:241-249); these kernels are its arithmetic at that
// precision, organised as the single-precision kernels of lean256.hip / wide1024.hip are:
//   - a wave is a team of its own (wave-level synchronisation only), twelve waves per workgroup
//     (eight when twelve regions of the 256-point class do not fit the LDS), one workgroup per CU;
//     waves draw chunks of consecutive tiles from a shared counter;
//   - a tile is one round of frames (4 or 1).  Every (frame, row) round forms its samples from the
//     RAW interleaved PCM bytes -- kept in LDS per tile (256-point class) or read straight from
//     memory a round ahead (1024-point class) -- in double with separately rounded product and
//     difference, so every sample (and its sign, for the zero crossings) is the reference's bit
//     for bit;
//   - a work area holds one double per complex point: the transposing exchanges between the FFT
//     stages move real and imaginary parts one after the other (same LDS bandwidth, half the
//     footprint), the last stage stays in registers, and only the upper half of the spectrum
//     goes through the area again for the real-FFT split.
#include <hip/hip_runtime.h>

#include "front_util.h"
#include "stamps.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

namespace mfcc {

namespace {

constexpr int kPipeWaves = 8; // (12 when the plan finds room in LDS, see pipe64_waves)
constexpr int kPipeParts = 128; // entries of the mel part table (the channel table follows it)

typedef unsigned int u4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned int u2u __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned int u4s __attribute__((ext_vector_type(4)));

// One tile as the pipeline sees it (all wave-uniform): the first three dwords of its TileDesc in
// the encoding the pipelined kernels share (byte offset of the tile's first sample, bits 48..59
// frames, bit 60: the tile starts the signal, bit 61: a vector of its fetch would leave the planned bytes).
struct PipeTile {
    unsigned w0, w1;
    int outRow;
    __device__ __forceinline__ int nf() const { return (int)((w1 >> 16) & 0xfffu); }
    __device__ __forceinline__ bool valid() const { return nf() != 0; }
    __device__ __forceinline__ bool first() const { return (w1 >> 28) & 1u; }
    __device__ __forceinline__ bool edge() const { return (w1 >> 29) & 1u; }
    __device__ __forceinline__ long long src() const { return (long long)(((unsigned long long)(w1 & 0xffffu) << 32) | w0); }
};

__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ PipeTile pipe_desc(const TileDesc *tab, int idx, bool valid)
{
    PipeTile t = {0u, 0u, 0};
    if (valid) {
        typedef const __attribute__((address_space(4))) u4s *cptr;
        const u4s d = reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(tab))[idx];
        t.w0 = d.x;
        t.w1 = d.y;
        t.outRow = (int)d.z;
    }
    return t;
}

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ln x for x = 0 (-inf) or a finite normal x > 0 (what the filterbank sums, floored at 1, and the
// frame energies of integer samples are): x = 2^e m with m in [sqrt 1/2, sqrt 2], ln m = 2 atanh s,
// s = (m - 1) / (m + 1), |s| < 0.1716, as its series up to s^21 (remainder < 2e-17); e ln 2 in two
// parts.  Error below 1e-15 relative, a fifth of the library routine's instructions.
__device__ __forceinline__ double fast_log(double x)
{
    const long long bits = __double_as_longlong(x);
    int e = (int)(bits >> 52) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const bool big = m > 1.4142135623730951;
    m = big ? 0.5 * m : m;
    e += big ? 1 : 0;
    const double sq = (m - 1.0) / (m + 1.0), z = sq * sq;
    double pl = 1.0 / 21.0;
    pl = pl * z + 1.0 / 19.0;
    pl = pl * z + 1.0 / 17.0;
    pl = pl * z + 1.0 / 15.0;
    pl = pl * z + 1.0 / 13.0;
    pl = pl * z + 1.0 / 11.0;
    pl = pl * z + 1.0 / 9.0;
    pl = pl * z + 1.0 / 7.0;
    pl = pl * z + 1.0 / 5.0;
    pl = pl * z + 1.0 / 3.0;
    pl = pl * z; // ln m = 2 s (1 + pl)
    const double ed = (double)e, s2 = sq + sq;
    const double ln2hi = 0x1.62e42fee00000p-1, ln2lo = 0x1.a39ef35793c76p-33; // ln2hi: 32 significant bits, e ln2hi is exact
    const double r = ed * ln2hi + (s2 + (s2 * pl + ed * ln2lo));
    return x == 0.0 ? -__builtin_huge_val() : r;
}

// sum over the G lanes of a frame (G = 16: one row of lanes, G = 64: the wave)
// x + x of the lane a DPP control names (both halves of the double by one lane-exchange move each:
// no LDS round trip, as a ds_bpermute-based exchange has)
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// sum over the 16 lanes of a row, in every lane of the row
__device__ __forceinline__ double row_sum_f64(double v)
{
    v = dpp_add_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141>(v); // row_half_mirror
    v = dpp_add_f64<0x140>(v); // row_mirror
    return v;
}
__device__ __forceinline__ double lane_value_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int G>
__device__ __forceinline__ double frame_sum(double v)
{
    static_assert(G == 16 || G == 64, "a frame is a row of lanes or the wave");
    v = row_sum_f64(v);
    if constexpr (G == 64) // the four rows' sums, the same in every lane
        v = (lane_value_f64(v, 0) + lane_value_f64(v, 16)) + (lane_value_f64(v, 32) + lane_value_f64(v, 48));
    return v;
}
template <int G>
__device__ __forceinline__ int frame_sum(int v) { return group_sum<int, G>(v, nullptr, 0); }

} // namespace

// 256-point class (mono 16-bit, frames of at most 512 samples): four frames per wave.  NZ:
// first-stage inputs of a lane that can be non-zero, ceil(wlen / 32).  WAVES: waves per workgroup.
template <int NZ, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_pipe64(const FrontArgs<double> a)
{
    constexpr int NC = 256, G = NC / 16, GT = 64 / G, PPL = 16, NPAIR = 8;
    constexpr int BPI = 2;
    constexpr int PFN = 2;                 // 16-byte vectors of a tile's fetch per lane
    constexpr int ZP = NC + NC / 16;       // doubles per work area
    constexpr int DW = 2;                  // dwords a lane reads per sample pair: instants 2n-1 .. 2n+2
    constexpr int PP = 264;                // doubles between the four frames' power values (8 LDS banks apart)
    typedef double T;
    STAMP_DECL
    STAMP_START;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const int tt = tid & 63;
    const int l = tt & (G - 1), grp = tt / G;

    // LDS: [matrix-core filterbank table (melSteps x 64) | window pairs (NZ G) | twiddles (15 x 16) |
    // split factors (NC / 2) | cepstral table] then one region per wave: [raw PCM bytes | GT work areas]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *lwL = reinterpret_cast<T *>(smem);
    const int lwN = a.melSteps * 64;
    cpx<T> *winL = reinterpret_cast<cpx<T> *>(lwL + lwN);
    cpx<T> *twL = winL + NZ * G;
    cpx<T> *swL = twL + 15 * 16;
    T *dctL = reinterpret_cast<T *>(swL + NC / 2);
    unsigned char *region = smem + ((size_t)a.tabElems + (size_t)wave * a.teamElems) * sizeof(T);
    unsigned *raw = reinterpret_cast<unsigned *>(region);
    T *areasW = reinterpret_cast<T *>(region) + a.rowPitch; // (rowPitch: elements reserved for the raw bytes)
    T *Zt = areasW + grp * ZP;
    cpx<T> *Zc = reinterpret_cast<cpx<T> *>(Zt);
    // power values of the lane's frame, bin order: 8 elements into the wave's areas and PP apart,
    // so that the four frames' values at one bin sit 8 banks apart
    T *Pw = areasW + 8 + grp * PP;

    for (int i = tid; i < lwN; i += WAVES * 64) lwL[i] = a.melW64[i];
    for (int i = tid; i < NZ * G; i += WAVES * 64) winL[i] = reinterpret_cast<const cpx<T> *>(a.window)[i];
    for (int i = tid; i < 15 * 16; i += WAVES * 64) twL[i] = reinterpret_cast<const cpx<T> *>(a.tw)[i];
    for (int i = tid; i < NC / 2; i += WAVES * 64) swL[i] = reinterpret_cast<const cpx<T> *>(a.splitTw)[i];
    for (int i = tid; i < a.dctElems; i += WAVES * 64) dctL[i] = a.dctT[i];
    // every value a masked sum may read from a work area is a finite number from here on
    for (int i = tt; i < GT * ZP; i += 64) areasW[i] = (T)0;
    __syncthreads();
    const cpx<T> *winT = winL, *twT = twL, *swT = swL;
    const T *dctT = dctL;

    // Per slot of the matrix-core filterbank: where the lane reads its power operand (element offset
    // into the wave's areas) and which channel its result belongs to (see the pipe64 tables, api.cpp)
    int melOff[4], melChan[4];
    {
        const int4 m0 = reinterpret_cast<const int4 *>(a.melMeta)[2 * tt], m1 = reinterpret_cast<const int4 *>(a.melMeta)[2 * tt + 1];
        melOff[0] = m0.x, melChan[0] = m0.y, melOff[1] = m0.z, melChan[1] = m0.w;
        melOff[2] = m1.x, melChan[2] = m1.y, melOff[3] = m1.z, melChan[3] = m1.w;
    }
    const T preK = a.preK;
    const int JW = a.M + a.c0Flag;
    const int F_S = a.S;
    const int subAt = a.M + a.c0Flag + a.energyFlag + a.zcFlag + a.brightFlag;

    // The wave's stream of tiles: chunks of a.tilesPerTeam consecutive tiles, the first one by the
    // wave's index, every further one drawn from a counter the waves of the launch share.  a.work[0]
    // is that counter, zero at launch; the last wave to finish (a.work[1] counts them) zeroes it again.
    const int teamId = blockIdx.x * WAVES + wave, numTeams = gridDim.x * WAVES, C = a.tilesPerTeam;
    int chunkBase = teamId * C, chunkPos = 0, nextBase = 0;
    bool haveNext = false, needDraw = true;
    auto draw = [&]() {
        unsigned drawn = 0;
        if (tt == 0) drawn = atomicAdd(a.work, 1u);
        nextBase = (sgpr((int)drawn) + numTeams) * C;
        haveNext = true;
        needDraw = false;
    };
    auto next_tile = [&]() {
        if (chunkPos == C) {
            if (!haveNext) {
                if (!needDraw) return -1;
                draw();
            }
            chunkBase = nextBase;
            chunkPos = 0;
            haveNext = false;
            needDraw = chunkBase < a.numTiles;
        }
        const int id = chunkBase + chunkPos;
        if (id >= a.numTiles) return -1;
        ++chunkPos;
        return id;
    };
    auto draw_step = [&]() {
        if (needDraw && !haveNext) draw();
    };
    auto next_tile_s = [&]() {
        const int id = sgpr(next_tile());
        chunkBase = sgpr(chunkBase);
        chunkPos = sgpr(chunkPos);
        nextBase = sgpr(nextBase);
        return id;
    };
    auto finish = [&]() {
        __threadfence();
        if (tt == 0 && atomicInc(a.work + 1, (unsigned)numTeams - 1u) == (unsigned)numTeams - 1u) {
            __hip_atomic_store(a.work, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (chunkBase >= a.numTiles) {
        finish();
        STAMP_FLUSH;
        return;
    }
    const TileDesc *tab = a.tiles;
    int id = next_tile_s();
    PipeTile cur = pipe_desc(tab, id, true);
    id = next_tile_s();
    PipeTile nx = pipe_desc(tab, id < 0 ? 0 : id, id >= 0);
    id = next_tile_s();
    PipeTile nn = pipe_desc(tab, id < 0 ? 0 : id, id >= 0);

    // Bytes of a tile: its frames' instants and the predecessor of the first.  Vector j of lane tt =
    // bytes 16 (tt + 64 j) .. + 15, byte 0 being the first byte of that predecessor.  Tiles at the
    // first or last bytes of the buffer keep every vector inside it by clamping its offset
    // (store_raw then redoes such a tile byte by byte).
    auto tile_bytes = [&](const PipeTile &t) { return ((t.nf() - 1) * a.inc + a.wlen + 1) * BPI; };
    auto fetch = [&](const PipeTile &t, u4u (&pf)[PFN]) {
        const long long off0 = t.src() - BPI;
        int lo = 0, hi = 16 * (((tile_bytes(t) + 15) >> 4) - 1);
        if (t.edge()) {
            lo = (int)(off0 < 0 ? -off0 : 0);
            hi = (int)min(a.planEnd - 16 - off0, (long long)hi);
        }
        const unsigned char *g = a.pcm + off0;
#pragma unroll
        for (int j = 0; j < PFN; ++j) pf[j] = *reinterpret_cast<const u4u *>(g + min(max(16 * (tt + j * 64), lo), hi));
    };
    auto store_raw = [&](const PipeTile &t, const u4u (&pf)[PFN]) {
        const int nbytes = tile_bytes(t), nvec = (nbytes + 15) >> 4;
#pragma unroll
        for (int j = 0; j < PFN; ++j) {
            const int vj = tt + j * 64;
            if (vj < nvec) reinterpret_cast<uint4 *>(raw)[vj] = make_uint4(pf[j].x, pf[j].y, pf[j].z, pf[j].w);
        }
        if (t.edge()) {
            const long long off0 = t.src() - BPI;
            unsigned char *rb = reinterpret_cast<unsigned char *>(raw);
            for (int b = tt; b < nbytes; b += 64) {
                const long long g = off0 + b;
                rb[b] = (g >= 0 && g < a.planEnd) ? a.pcm[g] : (unsigned char)0;
            }
        }
    };
    {
        u4u pf0[PFN];
        fetch(cur, pf0);
        store_raw(cur, pf0);
        wave_fence();
    }

    T lvr[4] = {0, 0, 0, 0}; // log-mel values of the lane's channels, per channel slot

    // One (frame, row) round.  LAST: the tile's last row.
    auto round = [&](const int vv, auto lastTag) {
        constexpr bool LAST = decltype(lastTag)::value;
        u4u pf[LAST ? PFN : 1];
        const int nf = cur.nf();
        const int fr = grp < nf ? grp : 0; // frame slots past the tile's last frame redo frame 0
        const bool firstTile = cur.first();
        T *fo = a.out + ((long long)cur.outRow + fr) * a.step + vv * a.D;
        const bool live = grp < nf;
        const int lq = l;
        STAMP(0);
        // the next tile's bytes are requested at the start of the round
        if constexpr (LAST) {
            if (nx.valid()) {
                draw_step();
                fetch(nx, pf);
            }
        }

        // ---- the lane's sample pairs (2n, 2n+1), n = l + G r, of derived row vv from the raw bytes
        //      (rows main.c:198-207), pre-emphasised over the whole signal (PreEmphasise main.c:208,
        //      first-sample rule for the signal's first sample), zero crossings of the un-windowed
        //      samples (main.c:227), Hamming window (main.c:229), frame energy (mfcc.c:138-142)
        cpx<T> v[PPL];
        int zcn = 0;
        {
            // instants 2n-1 (predecessor), 2n, 2n+1, 2n+2 start at byte (frame start + 2n) BPI
            const unsigned *src = raw + ((fr * a.inc) >> 1) + l;
            // (one straight-line copy of the loop per setting of the zero-crossing option: a test inside it
            // would cut it into NZ blocks, each waiting for its own two reads)
            auto expand = [&](auto zcTag) {
            constexpr bool ZCON = decltype(zcTag)::value;
#pragma unroll
            for (int r = 0; r < NZ; ++r) {
                unsigned d[DW];
                d[0] = src[G * r];
                d[1] = src[G * r + 1];
                T x[4];
                x[0] = (T)(int)(short)(d[0] & 0xffffu);
                x[1] = (T)((int)d[0] >> 16);
                x[2] = (T)(int)(short)(d[1] & 0xffffu);
                x[3] = (T)((int)d[1] >> 16);
                T y1 = sub_mul_exact(x[1], x[0], preK);
                const T y2 = sub_mul_exact(x[2], x[1], preK);
                if (r == 0 && firstTile && fr == 0 && l == 0) y1 = x[1] * ((T)1 - preK); // s[1] *= 1 - k
                if constexpr (ZCON) {
                    const T y3 = sub_mul_exact(x[3], x[2], preK);
                    const int s0 = 2 * (l + G * r);
                    const bool c01 = (y1 < 0 && y2 > 0) || (y1 > 0 && y2 < 0);
                    const bool c12 = (y2 < 0 && y3 > 0) || (y2 > 0 && y3 < 0);
                    zcn += (c01 && s0 + 1 < a.wlen) ? 1 : 0; // pairs inside the frame only
                    zcn += (c12 && s0 + 2 < a.wlen) ? 1 : 0;
                }
                const cpx<T> w = winT[lq + G * r];
                v[r] = cpx<T>{y1 * w.x, y2 * w.y};
            }
            };
            if (a.zcFlag) expand(std::true_type{});
            else expand(std::false_type{});
#pragma unroll
            for (int r = NZ; r < PPL; ++r) v[r] = cpx<T>{0.0, 0.0};
        }
        T te = 0;
        if (a.energyFlag) {
            cpx<T> e2 = {0, 0};
#pragma unroll
            for (int r = 0; r < NZ; ++r) e2 += v[r] * v[r];
            te = frame_sum<G>((T)4 * (e2.x + e2.y)); // (the window table holds half the Hamming values)
        }
        if (a.zcFlag) zcn = frame_sum<G>(zcn);
        STAMP(1);

        // ---- NC-point complex FFT of z[n] = x[2n] + i x[2n+1] (Realft mfcc.c:262-270): Stockham
        //      stages of radix 16 and 16, real parts and imaginary parts exchanged in turn
        Radix<T, 16>::run(v);
        {
            T *wp = Zt + 17 * l;                                 // zpad(16 l + r) = 17 l + r
            const T *rp = Zt + l + (l >> 4);                     // zpad(l + G r) = zpad(l) + (G + G / 16) r
            constexpr int RS = G + G / 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r] = v[r].x;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r].x = rp[RS * r];
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r] = v[r].y;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r].y = rp[RS * r];
            wave_fence();
        }
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], twT[(r - 1) * 16 + (lq & 15)]); // sub-transform size 16: [r-1][l mod 16]
        Radix<T, 16>::run(v);
        // the lane now holds X[l + G m], m = 0 .. 15, in register xr(m)
        auto xr = [&](int m) -> cpx<T> & { return v[m]; };
        STAMP(3);

        // ---- real-FFT split and power (mfcc.c:274-290, ek of mfcc.c:153-155): the upper half of the
        //      spectrum goes through the work area (complex index = bin - NC / 2) and the lane reads
        //      back X[NC - l - G q], q = 0 .. 7 (lane 0, q = 0: one point past the bins, unused)
        T pa[NPAIR], pb[NPAIR];
        {
#pragma unroll
            for (int m = 8; m < 16; ++m) Zc[l + G * (m - 8)] = xr(m);
            wave_fence();
            cpx<T> rev[NPAIR];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) rev[q] = Zc[NC / 2 - l - G * q];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) split_power<T>(xr(q), rev[q], swT[lq + G * q], pa[q], pb[q]);
            // bin 0 packs Re + Im of X[0] (mfcc.c:288-290); bin NC / 2 is |X[NC / 2]|^2; lane 0 holds both
            const cpx<T> z0 = xr(0), zh = xr(8);
            const T p0 = (T)4 * (z0.x + z0.y) * (z0.x + z0.y);
            const T ph = (T)4 * (zh.x * zh.x + zh.y * zh.y);
            pa[0] = l == 0 ? p0 : pa[0];
            pb[0] = l == 0 ? ph : pb[0];
        }
        wave_fence(); // every lane has read its partner points: the area takes the power values, bin order
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int k = l + G * q;
            const int k2 = k == 0 ? NC / 2 : NC - k;
            Pw[k] = pa[q];
            Pw[k2] = pb[q];
        }
        wave_fence();
        STAMP(4);

        // ---- sub-band energies / brightness (main.c:235-236): every band (the whole spectrum when
        //      only the brightness is asked for) gets a power-of-two share of the frame's lanes
        if (F_S > 0 || a.brightFlag) {
            const int nseg = F_S > 0 ? F_S : 1, width = NC / nseg; // (nseg <= G: one pass)
            int lpb = 1;
            while (lpb * 2 * nseg <= G) lpb *= 2;
            const int b = l / lpb, sl = l & (lpb - 1);
            T e = 0, ek = 0;
            if (b < nseg) {
                const int lo = b * width, hi = (b == nseg - 1) ? NC : lo + width;
                for (int k = lo + sl; k < hi; k += 4 * lpb) { // four reads in flight
                    T pv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pv[i] = Pw[min(k + i * lpb, NC - 1)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool in = k + i * lpb < hi;
                        e += in ? pv[i] : (T)0;
                        ek += in ? (T)(k + i * lpb + 1) * pv[i] : (T)0;
                    }
                }
            }
            for (int o = lpb >> 1; o > 0; o >>= 1) {
                e += __shfl_xor(e, o, 64);
                ek += __shfl_xor(ek, o, 64);
            }
            // the band's sums sit in its first lane; totals over the frame's bands
            const bool head = b < nseg && sl == 0;
            const T total = frame_sum<G>(head ? e : (T)0), num = frame_sum<G>(head ? ek : (T)0);
            if (live) {
                if (F_S > 0 && head) fo[subAt + b] = e / total;
                if (a.brightFlag && l == 0) fo[a.M + a.c0Flag + a.energyFlag + a.zcFlag] = num / total / (T)NC;
            }
        }
        if (live && l == 0) {
            if (a.energyFlag) fo[a.M + a.c0Flag] = fast_log(te);                                  // main.c:243
            if (a.zcFlag) fo[a.M + a.c0Flag + a.energyFlag] = (T)zcn / (T)(a.wlen - 1);           // main.c:244
        }
        STAMP(5);

        // ---- mel filterbank (mfcc.c:152-163) on the matrix cores, then log(max(., 1)) (mfcc.c:171-176)
        {
            // 4 x 4 x 4 products (frame x channel of a quad x bin), four quads at a
            // time (one per block of 16 result lanes ... see the table builder); lane (j, blk, k) =
            // (lane % 4, lane / 4 % 4, lane / 16) feeds the power of frame j at the block's bin
            // r0 + 4 t + k and the weight of that bin for channel j of the block's quad, and receives
            // the sum of frame lane / 16 (its own) for channel lane % 4 of that quad
            int tbase = 0;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int L = a.melSlotSteps[sl];
                if (L > 0) {
                    const T *ap = areasW + melOff[sl];
                    const T *bp = lwL + tbase * 64 + tt;
                    T acc0 = 0, acc1 = 0;
                    int t = 0;
                    for (; t + 4 <= L; t += 4) {
                        T av[4], bv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            av[i] = ap[4 * (t + i)];
                            bv[i] = bp[64 * (t + i)];
                        }
                        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[0], bv[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[1], bv[1], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[2], bv[2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[3], bv[3], acc1, 0, 0, 0);
                    }
                    for (; t < L; ++t) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[4 * t], bp[64 * t], acc0, 0, 0, 0);
                    const T m = acc0 + acc1;
                    const T lv = fast_log(m < (T)1 ? (T)1 : m);
                    if (melChan[sl] >= 0) {
                        if (a.fbankFlag) {
                            if (live) fo[melChan[sl]] = lv;
                        } else {
                            lvr[sl] = lv;
                        }
                    }
                    tbase += L;
                }
            }
        }
        wave_fence(); // the work area is free again
        STAMP(6);
        if constexpr (LAST) {
            // every row of the tile has expanded its samples: the next tile's bytes take their place
            if (nx.valid()) store_raw(nx, pf);
            if (!a.fbankFlag) {
                // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208): the tile's
                //      log-mel values go to the work area as [channel][row]; lane j of a frame forms
                //      coefficient j of every row from its table row (row M is all ones: C0)
                auto cepstra = [&](auto rowsTag) {
                    constexpr int VV = decltype(rowsTag)::value;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl)
                        if (melChan[sl] >= 0 && a.melSlotSteps[sl] > 0) Zt[melChan[sl]] = lvr[sl];
                    wave_fence();
                    T *orow = a.out + ((long long)cur.outRow + fr) * a.step;
                    for (int j0 = 0; j0 < JW; j0 += G) {
                        const int j = j0 + l;
                        const T *row = dctT + (j < JW ? j : JW - 1) * a.dctPitch;
                        T acc[VV];
#pragma unroll
                        for (int vq = 0; vq < VV; ++vq) acc[vq] = 0;
                        for (int k0 = 0; k0 < a.P; k0 += 4) {
                            T tv[4], f[4][VV];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int k = min(k0 + u, a.P - 1);
                                tv[u] = k0 + u < a.P ? row[k] : (T)0;
#pragma unroll
                                for (int vq = 0; vq < VV; ++vq) f[u][vq] = Zt[k * VV + vq];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int vq = 0; vq < VV; ++vq) acc[vq] += tv[u] * f[u][vq];
                        }
                        if (j < JW && live) {
#pragma unroll
                            for (int vq = 0; vq < VV; ++vq) orow[vq * a.D + j] = acc[vq] * a.mfnorm;
                        }
                    }
                };
                cepstra(std::integral_constant<int, 1>{}); // (mono input: one derived row)
            }
            wave_fence(); // the raw bytes are in place, the work area is free
        }
        STAMP(9);
    };
    for (;;) {
        // the descriptor three tiles ahead: asked for now, moved into place at the tile's end
        const int id3 = next_tile_s();
        const PipeTile d3 = pipe_desc(tab, id3 < 0 ? 0 : id3, id3 >= 0);
        round(0, std::true_type{}); // (mono input: one derived row per frame)
        if (!nx.valid()) break;
        cur = nx;
        nx = nn;
        nn = d3;
    }
    finish();
    STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------
// 1024-point class (24-bit stereo, frames of 1025 .. 2048 samples): one wave per frame, 64 lanes x
// 16 points, twelve waves per workgroup (three per SIMD, 168 registers).  Nothing of a frame's raw
// bytes is kept in LDS: a round asks for the next round's bytes straight from memory (one 12-byte
// load per sample pair; a wave's load is 768 consecutive bytes) once its spectrum has been reduced
// to per-bin values, turns them into that row's integer samples just before its own results go
// out, and carries those to the next round -- so no round starts by waiting for memory behind the
// stores of the one before it.  LDS: [filterbank weights | window pairs | the lanes' mel part
// tables | run slots | cepstral table when there is room] shared, then one work area per wave.
// ------------------------------------------------------------------------------------------
namespace {

constexpr int kPipeWideWaves = 12;

typedef unsigned int u3u __attribute__((ext_vector_type(3), aligned(1)));

// lane i takes x of lane i - 1; lane 0 takes `first`
__device__ __forceinline__ int wave_shr1i(int x, int first) { return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xf, 0xf, false); }

// Integer sample of derived row ROW (0 L, 1 R, 2 L + R = twice the mid row, 3 side = L - R;
// main.c:198-207) from NW dwords that hold the instant's two channels from byte O on.
template <int ROW, int O, int NW>
__device__ __forceinline__ int row_int(const unsigned (&d)[NW])
{
    if constexpr (ROW == 0) return window_sample<3, O, NW>(d);
    else if constexpr (ROW == 1) return window_sample<3, O + 3, NW>(d);
    else if constexpr (ROW == 2) return window_sample<3, O, NW>(d) + window_sample<3, O + 3, NW>(d);
    else return window_sample<3, O, NW>(d) - window_sample<3, O + 3, NW>(d);
}

// own[r] = the 12 bytes of instants 2n and 2n+1, n = l + 64 r -> xa[r], xb[r]; pred = the 8 bytes in
// front of the frame's first instant (its predecessor: bytes 2 .. 7) -> pred0
template <int NZ, int ROW>
__device__ __forceinline__ void pipew_samples(int (&xa)[NZ], int (&xb)[NZ], int &pred0, const unsigned (&own)[NZ][3], const unsigned (&pred)[2])
{
#pragma unroll
    for (int r = 0; r < NZ; ++r) {
        xa[r] = row_int<ROW, 0, 3>(own[r]);
        xb[r] = row_int<ROW, 6, 3>(own[r]);
    }
    pred0 = row_int<ROW, 2, 2>(pred);
}

// The lane's points (2n, 2n+1), n = l + 64 r, of the frame's row from its integer samples xa[r] =
// x[2n], xb[r] = x[2n+1] (pred0: the sample in front of the frame; row 2 holds L + R = twice the mid
// row: the factor goes through the whole frame -- scaling by a power of two commutes with every
// rounding here -- and comes off the frame energy and the channel sums as 1/4; the sub-band shares
// and the brightness are ratios).  The predecessor 2n-1 of a pair is the second sample of the previous lane's pair (of lane
// 63's previous pair for lane 0): one lane shift.  Pre-emphasis over the whole signal in double
// with separately rounded product and difference (PreEmphasise main.c:208; first-sample rule for
// the signal's first sample), Hamming window (main.c:229, pairs from winL); with ZC the zero
// crossings of the un-windowed samples (main.c:227), counted over the wave with mask popcounts
// (sample 2n+2 is the next lane's first sample).
template <int NZ, bool ZC>
__device__ __forceinline__ void pipew_expand(cpx<double> (&v)[16], int &zc, const int (&xa)[NZ], const int (&xb)[NZ], int pred0,
                                             int l, const cpx<double> *winL, double preK, bool firstTile, int wlen)
{
    constexpr int G = 64;
    typedef unsigned long long u64;
    u64 nextNeg = 0, nextPos = 0; // sign of sample 2n of lane 0's pair r + 1 (the neighbour of lane 63's pair r), bit 63
#pragma unroll
    for (int r = NZ - 1; r >= 0; --r) {
        // (lane 0's predecessor, lane 63's second sample of the pair before: that register rotated by one
        // lane is what the shift leaves in lane 0 -- two lane-exchange moves, no trip through a scalar)
        int carry = pred0;
        if (r > 0) carry = __builtin_amdgcn_mov_dpp(xb[r - 1], 0x13C, 0xf, 0xf, false);
        const double x0 = (double)wave_shr1i(xb[r], carry), x1 = (double)xa[r], x2 = (double)xb[r];
        double y1 = sub_mul_exact(x1, x0, preK);
        const double y2 = sub_mul_exact(x2, x1, preK);
        if (firstTile && l == 0 && r == 0) y1 = x1 * (1.0 - preK); // s[1] *= 1 - k
        if constexpr (ZC) {
            // A crossing = neighbours of opposite signs (the reference's negative product; the samples
            // are far from the underflow range).  The signs go to lane masks, so the neighbour in the
            // next lane is a shift of the mask and everything behind the compares is scalar work.
            const u64 n1 = __builtin_amdgcn_ballot_w64(y1 < 0.0), p1 = __builtin_amdgcn_ballot_w64(y1 > 0.0);
            const u64 n2 = __builtin_amdgcn_ballot_w64(y2 < 0.0), p2 = __builtin_amdgcn_ballot_w64(y2 > 0.0);
            const u64 n3 = (n1 >> 1) | nextNeg, p3 = (p1 >> 1) | nextPos; // sample 2n + 2: the next lane's first
            u64 c01 = (n1 & p2) | (p1 & n2), c12 = (n2 & p3) | (p2 & n3);
            // pairs inside the frame only (an instance serves every window length from 1025 up to
            // 128 NZ: rows 0 .. 7 lie inside every such frame): lanes with 2 (l + 64 r) + 1 < wlen
            // for c01, + 2 for c12
            if (r >= 8) {
                const int w = wlen - 2 * G * r;
                const int k01 = min(max(w >> 1, 0), 64), k12 = min(max((w - 1) >> 1, 0), 64);
                c01 &= k01 >= 64 ? ~0ull : (1ull << k01) - 1ull;
                c12 &= k12 >= 64 ? ~0ull : (1ull << k12) - 1ull;
            }
            zc += __builtin_popcountll(c01) + __builtin_popcountll(c12);
            nextNeg = n1 << 63;
            nextPos = p1 << 63;
            // (the count is taken here, while the masks are at hand: left to its own schedule the compiler
            // parks every row's masks in spill lanes and counts at the round's end)
            asm volatile("" : "+s"(zc), "+s"(nextNeg), "+s"(nextPos));
        }
        const cpx<double> w = winL[l + G * r];
        v[r] = cpx<double>{y1 * w.x, y2 * w.y};
    }
#pragma unroll
    for (int r = NZ; r < 16; ++r) v[r] = cpx<double>{0.0, 0.0};
}

} // namespace

// NZ: first-stage inputs of a lane that can be non-zero (14 for frames of at most 1792 samples,
// else 16).  ZCT: zero-crossing rate on.
template <int NZ, bool ZCT>
__global__ __launch_bounds__(kPipeWideWaves * 64) void k_pipe64w(const FrontArgs<double> a)
{
    constexpr int NC = 1024, G = 64, PPL = 16, NPAIR = 8, WAVES = kPipeWideWaves;
    constexpr int NPL = 2;          // mel parts per lane
    constexpr int ZP = NC + NC / 16; // doubles per work area
    constexpr int kClampFrom = 8;   // pairs from here on can lie behind the shortest frame's last byte
    typedef double T;
    STAMP_DECL
    STAMP_START;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const int tt0 = tid & 63;
    int tt = tt0, l = tt0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *lwL = reinterpret_cast<T *>(smem);
    cpx<T> *winL = reinterpret_cast<cpx<T> *>(lwL + NC + 8);
    int4 *segL = reinterpret_cast<int4 *>(winL + NZ * G);
    int *combCtr = reinterpret_cast<int *>(segL + kPipeParts + 64); // the next comb of the workgroup's sequence to be claimed
    T *dctL = reinterpret_cast<T *>(segL + kPipeParts + 64 + 1);
    const bool dctInLds = a.rowPitch > 0; // (rowPitch: elements of the table's copy)
    T *Zt = reinterpret_cast<T *>(smem) + a.tabElems + (size_t)wave * a.teamElems;
    cpx<T> *Zc = reinterpret_cast<cpx<T> *>(Zt);
    T *Pw = Zt;

    for (int i = tid; i < NC + 8; i += WAVES * 64) lwL[i] = i < NC ? a.loWt[i] : (T)0;
    for (int i = tid; i < NZ * G; i += WAVES * 64) winL[i] = reinterpret_cast<const cpx<T> *>(a.window)[i];
    for (int i = tid; i < kPipeParts + 64; i += WAVES * 64) segL[i] = reinterpret_cast<const int4 *>(a.laneSeg)[i];
    if (tid == 0) *combCtr = WAVES; // (wave w starts on comb w)
    if (dctInLds) {
        for (int i = tid; i < a.dctElems; i += WAVES * 64) dctL[i] = a.dctT[i];
    }
    // every value a masked sum may read from a work area is a finite number from here on
    for (int i = tt0; i < ZP; i += 64) Zt[i] = (T)0;
    __syncthreads();
    // Twiddle and split factors are read through L1 in the middle of a round: long after the
    // previous round's result stores and before this round's.
    const cpx<T> *twT = reinterpret_cast<const cpx<T> *>(a.tw), *swT = reinterpret_cast<const cpx<T> *>(a.splitTw);
    const T *dctT = dctInLds ? dctL : a.dctT;
    const T preK = a.preK;
    const int JW = a.M + a.c0Flag;
    const int F_V = a.V, F_S = a.S;
    const int subAt = a.M + a.c0Flag + a.energyFlag + a.zcFlag + a.brightFlag;

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
    PipeTile cur = pipe_desc(tab, id, true);
    id = next_tile_s();
    PipeTile nx = pipe_desc(tab, id < 0 ? 0 : id, id >= 0);
    id = next_tile_s();
    PipeTile nn = pipe_desc(tab, id < 0 ? 0 : id, id >= 0);

    // The frame's raw bytes as the lanes need them: one path for every frame, so that the loaded
    // registers have a single definition and nothing waits for them where they are asked for.  A
    // frame at the last bytes of the buffer (edge) moves the loads that would pass the end back
    // to it, one at its first bytes reads the predecessor from the buffer's start; fix_raw then
    // shifts those bytes into place (zeros behind the end: they lie behind the frame's last sample).
    auto load_raw = [&](const PipeTile &t, unsigned (&own)[NZ][3], unsigned (&pred)[2]) {
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
    auto fix_raw = [&](const PipeTile &t, unsigned (&own)[NZ][3], unsigned (&pred)[2]) {
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

    // The integer samples of the round to come, carried from round to round.
    int xa[NZ], xb[NZ], pred0;
    auto to_samples = [&](const int row, const unsigned (&own)[NZ][3], const unsigned (&pred)[2]) {
        switch (row) { // (the same for every lane)
        case 0: pipew_samples<NZ, 0>(xa, xb, pred0, own, pred); break;
        case 1: pipew_samples<NZ, 1>(xa, xb, pred0, own, pred); break;
        case 2: pipew_samples<NZ, 2>(xa, xb, pred0, own, pred); break;
        default: pipew_samples<NZ, 3>(xa, xb, pred0, own, pred); break;
        }
    };
    {
        unsigned own[NZ][3], pred[2];
        load_raw(cur, own, pred);
        fix_raw(cur, own, pred);
        to_samples(0, own, pred);
    }

    T lvr[4] = {0, 0, 0, 0}; // log-mel values of the lane's channel, per row

    // One (frame, row) round.  LAST: the frame's last row, which also forms the cepstra.
    auto round = [&](const int vv, auto lastTag) {
        constexpr bool LAST = decltype(lastTag)::value;
        // (the lane index is made opaque once per round: what derives from it is recomputed here in
        // a few instructions instead of being kept in registers, or spilled, across the whole loop)
        asm volatile("" : "+v"(tt));
        l = tt & 63;
        const bool firstTile = cur.first();
        T *fo = a.out + (long long)cur.outRow * a.step + vv * a.D;
        STAMP(0);

        // ---- the lane's points of derived row vv from its samples (rows main.c:198-207), pre-emphasis
        //      (main.c:208), zero crossings (main.c:227), Hamming window (main.c:229), frame energy
        //      (mfcc.c:138-142)
        cpx<T> v[PPL];
        int zcn = 0;
        const T pscale = vv == 2 ? (T)0.25 : (T)1; // (row 2 runs on L + R: powers come out fourfold)
        pipew_expand<NZ, ZCT>(v, zcn, xa, xb, pred0, l, winL, preK, firstTile, a.wlen);
        T te = 0;
        if (a.energyFlag) {
            cpx<T> e2 = {0, 0};
#pragma unroll
            for (int r = 0; r < NZ; ++r) e2 += v[r] * v[r];
            te = frame_sum<G>((T)4 * (e2.x + e2.y)); // (the window table holds half the Hamming values)
        }
        STAMP(1);

        // ---- 1024-point complex FFT of z[n] = x[2n] + i x[2n+1] (Realft mfcc.c:262-270): Stockham
        //      stages of radix 16, 16 and 4, real parts and imaginary parts exchanged in turn
        Radix<T, 16>::run(v);
        {
            T *wp = Zt + 17 * l;                                 // zpad(16 l + r) = 17 l + r
            const T *rp = Zt + l + (l >> 4);                     // zpad(l + G r) = zpad(l) + (G + G / 16) r
            constexpr int RS = G + G / 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r] = v[r].x;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r].x = rp[RS * r];
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[r] = v[r].y;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r].y = rp[RS * r];
            wave_fence();
        }
        const unsigned lu = (unsigned)tt; // (one unsigned lane offset under uniform row pointers: scalar address arithmetic)
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], (twT + (r - 1) * 16)[lu & 15u]); // sub-transform size 16: [r-1][l mod 16]
        Radix<T, 16>::run(v);
        {
            const int o = ((l >> 4) << 8) + (l & 15);
            T *wp = Zt + o + (o >> 4);                           // zpad(o + 16 r) = zpad(o) + 17 r
            const T *rp = Zt + l + (l >> 4);                     // zpad(l + 64 b + 256 r) = zpad(l) + 68 b + 272 r
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[17 * r] = v[r].x;
            wave_fence();
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * b + r].x = rp[68 * b + 272 * r];
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[17 * r] = v[r].y;
            wave_fence();
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * b + r].y = rp[68 * b + 272 * r];
            wave_fence();
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 1; r < 4; ++r) v[4 * b + r] = cmul(v[4 * b + r], (twT + (15 * 16 + (r - 1) * 256 + 64 * b))[lu]); // size 256: [r-1][l + 64 b]
#pragma unroll
            for (int b = 0; b < 4; ++b) Radix<T, 4>::run(&v[4 * b]);
        }
        // the lane now holds X[l + G m], m = 0 .. 15, in register xr(m)
        auto xr = [&](int m) -> cpx<T> & { return v[4 * (m & 3) + (m >> 2)]; };
        STAMP(3);

        // ---- real-FFT split and power (mfcc.c:274-290, ek of mfcc.c:153-155): the upper half of the
        //      spectrum goes through the work area (complex index = bin - NC / 2) and the lane reads
        //      back X[NC - l - G q], q = 0 .. 7 (lane 0, q = 0: one point past the bins, unused)
        T pa[NPAIR], pb[NPAIR];
        {
#pragma unroll
            for (int m = 8; m < 16; ++m) Zc[l + G * (m - 8)] = xr(m);
            wave_fence();
            cpx<T> rev[NPAIR];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) rev[q] = Zc[NC / 2 - l - G * q];
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) split_power<T>(xr(q), rev[q], (swT + G * q)[lu], pa[q], pb[q]);
            // bin 0 packs Re + Im of X[0] (mfcc.c:288-290); bin NC / 2 is |X[NC / 2]|^2; lane 0 holds both
            const cpx<T> z0 = xr(0), zh = xr(8);
            const T p0 = (T)4 * (z0.x + z0.y) * (z0.x + z0.y);
            const T ph = (T)4 * (zh.x * zh.x + zh.y * zh.y);
            pa[0] = l == 0 ? p0 : pa[0];
            pb[0] = l == 0 ? ph : pb[0];
        }
        wave_fence(); // every lane has read its partner points: the area takes the power values, bin order
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int k = l + G * q;
            const int k2 = k == 0 ? NC / 2 : NC - k;
            Pw[k] = pa[q];
            Pw[k2] = pb[q];
        }
        wave_fence();
        STAMP(4);

        // ---- the bytes of the round to come (this frame's next row, or the next frame's first)
        unsigned own[NZ][3], pred[2];
        if constexpr (LAST) {
            load_raw(nx.valid() ? nx : cur, own, pred); // (behind the last frame: this one's again, unused)
        } else {
            load_raw(cur, own, pred);
        }

        // ---- sub-band energies / brightness (main.c:235-236): every band (the whole spectrum when
        //      only the brightness is asked for) gets a power-of-two share of the frame's lanes
        T bandOut = 0, brightOut = 0; // (stored with the round's other results, at its end)
        int bandAt = -1;
        if (F_S > 0 || a.brightFlag) {
            const int nseg = F_S > 0 ? F_S : 1, width = NC / nseg; // (nseg <= G: one pass)
            int lpb = 1;
            while (lpb * 2 * nseg <= G) lpb *= 2;
            const int b = l / lpb, sl = l & (lpb - 1);
            T e = 0, ek = 0;
            if ((nseg & (nseg - 1)) == 0) {
                // Bands of a power-of-two width lie a multiple of the banks' span apart: the lanes of
                // band b start lpb b bins into it and wrap around, so that the bands' simultaneous
                // reads fall into different banks (the sums only change their order).
                const int lo = b * width, rot = sl + lpb * b; // (16 steps of lpb bins)
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += 8) {
                    T pv[8];
                    int kk[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        kk[i] = lo + ((rot + (i0 + i) * lpb) & (width - 1));
                        pv[i] = Pw[kk[i]];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        e += pv[i];
                        ek += (T)(kk[i] + 1) * pv[i];
                    }
                }
            } else if (b < nseg) {
                const int lo = b * width, hi = (b == nseg - 1) ? NC : lo + width;
                for (int k = lo + sl; k < hi; k += 4 * lpb) { // four reads in flight
                    T pv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pv[i] = Pw[min(k + i * lpb, NC - 1)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool in = k + i * lpb < hi;
                        e += in ? pv[i] : (T)0;
                        ek += in ? (T)(k + i * lpb + 1) * pv[i] : (T)0;
                    }
                }
            }
            // (sums over the band's lpb lanes: lane exchanges inside a row, LDS permutes across rows)
            if (lpb >= 2) e = dpp_add_f64<0xB1>(e), ek = dpp_add_f64<0xB1>(ek);
            if (lpb >= 4) e = dpp_add_f64<0x4E>(e), ek = dpp_add_f64<0x4E>(ek);
            if (lpb >= 8) e = dpp_add_f64<0x141>(e), ek = dpp_add_f64<0x141>(ek);
            if (lpb >= 16) e = dpp_add_f64<0x140>(e), ek = dpp_add_f64<0x140>(ek);
            if (lpb >= 32) e += __shfl_xor(e, 16, 64), ek += __shfl_xor(ek, 16, 64);
            if (lpb >= 64) e += __shfl_xor(e, 32, 64), ek += __shfl_xor(ek, 32, 64);
            // the band's sums sit in its first lane; totals over the frame's bands
            const bool head = b < nseg && sl == 0;
            const T total = frame_sum<G>(head ? e : (T)0), num = frame_sum<G>(head ? ek : (T)0);
            if (F_S > 0 && head) bandOut = e / total, bandAt = subAt + b;
            brightOut = num / total / (T)NC;
        }
        const T teOut = a.energyFlag ? fast_log(te * pscale) : (T)0;
        STAMP(5);

        // ---- mel filterbank (mfcc.c:152-163): the bins whose lower channel is c form one segment,
        //      cut by the host into parts; a lane sums Tc = sum loWt p and Uc = sum (p - loWt p) of its
        //      parts in bin order, the part sums go to the work area, and the lane of channel b adds the
        //      parts of U[b-1] and of T[b] in part (= bin) order; then log(max(., 1)) (mfcc.c:171-176)
        T ps[NPL][2];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int4 sg = segL[l + G * j];
            const int pB = sg.x, pE = sg.z >= 0 ? sg.y : sg.x;
            T accT = 0, accU = 0;
            int k = pB;
            for (; k + 4 <= pE; k += 4) { // whole groups of four bins: no bin of them lies past the part
                T pv[4], wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pv[i] = Zt[k + i];
                    wv[i] = lwL[k + i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const T t1 = wv[i] * pv[i];
                    accT += t1;
                    accU += pv[i] - t1;
                }
            }
            if (k < pE) { // the part's last one to three bins
                T pv[4], wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pv[i] = Zt[k + i]; // (reads past the part stay inside the area / the weight table's zero tail)
                    wv[i] = lwL[k + i];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const bool in = k + i < pE;
                    const T t1 = wv[i] * pv[i];
                    accT += in ? t1 : (T)0;
                    accU += in ? pv[i] - t1 : (T)0;
                }
            }
            ps[j][0] = accT;
            ps[j][1] = accU;
        }
        wave_fence(); // every part has been read: the bins' place takes the part sums
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            Zt[l + G * j] = ps[j][0];
            Zt[kPipeParts + l + G * j] = ps[j][1];
        }
        wave_fence();
        T lv = 0;
        {
            const int4 cp = l < a.P ? segL[kPipeParts + l] : make_int4(0, 0, 0, 0);
            T pu[10], pt[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                pu[i] = Zt[kPipeParts + cp.x + i];
                pt[i] = Zt[cp.z + i];
            }
            T mu = 0, mt = 0;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                mu += i < cp.y ? pu[i] : (T)0;
                mt += i < cp.w ? pt[i] : (T)0;
            }
            const T m = (mu + mt) * pscale;
            lv = fast_log(m < (T)1 ? (T)1 : m);
            if (l < a.P && !a.fbankFlag) {
                lvr[0] = vv == 0 ? lv : lvr[0];
                lvr[1] = vv == 1 ? lv : lvr[1];
                lvr[2] = vv == 2 ? lv : lvr[2];
                lvr[3] = vv == 3 ? lv : lvr[3];
            }
        }
        wave_fence(); // the work area is free again
        STAMP(6);

        // ---- the next round's samples, then this round's results (row layout main.c:241-249)
        fix_raw(LAST && nx.valid() ? nx : cur, own, pred);
        to_samples(LAST ? 0 : vv + 1, own, pred);
        STAMP(7);
        if (bandAt >= 0) fo[bandAt] = bandOut;
        if (l == 0) {
            if (a.brightFlag) fo[a.M + a.c0Flag + a.energyFlag + a.zcFlag] = brightOut;
            if (a.energyFlag) fo[a.M + a.c0Flag] = teOut;                                      // main.c:243
            if (a.zcFlag) fo[a.M + a.c0Flag + a.energyFlag] = (T)zcn / (T)(a.wlen - 1);        // main.c:244
        }
        if (a.fbankFlag && l < a.P) fo[l] = lv;
        STAMP(8);
        if constexpr (LAST) {
            if (!a.fbankFlag) {
                // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208): the frame's
                //      log-mel values go to the work area as [channel][row]; lane j forms coefficient
                //      j of every row from its table row (row M is all ones: C0)
                auto cepstra = [&](auto rowsTag) {
                    constexpr int VV = decltype(rowsTag)::value;
                    if (l < a.P) {
#pragma unroll
                        for (int vq = 0; vq < VV; ++vq) Zt[l * VV + vq] = lvr[vq];
                    }
                    wave_fence();
                    T *orow = a.out + (long long)cur.outRow * a.step;
                    for (int j0 = 0; j0 < JW; j0 += G) {
                        const int j = j0 + l;
                        const T *row = dctT + (j < JW ? j : JW - 1) * a.dctPitch;
                        T acc[VV];
#pragma unroll
                        for (int vq = 0; vq < VV; ++vq) acc[vq] = 0;
                        for (int k0 = 0; k0 < a.P; k0 += 4) {
                            T tv[4], f[4][VV];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int k = min(k0 + u, a.P - 1);
                                tv[u] = k0 + u < a.P ? row[k] : (T)0;
#pragma unroll
                                for (int vq = 0; vq < VV; ++vq) f[u][vq] = Zt[k * VV + vq];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int vq = 0; vq < VV; ++vq) acc[vq] += tv[u] * f[u][vq];
                        }
                        if (j < JW) {
#pragma unroll
                            for (int vq = 0; vq < VV; ++vq) orow[vq * a.D + j] = acc[vq] * a.mfnorm;
                        }
                    }
                };
                switch (F_V) {
                case 1: cepstra(std::integral_constant<int, 1>{}); break;
                case 2: cepstra(std::integral_constant<int, 2>{}); break;
                case 3: cepstra(std::integral_constant<int, 3>{}); break;
                default: cepstra(std::integral_constant<int, 4>{}); break;
                }
                wave_fence(); // the work area is free for the next frame
            }
        }
        STAMP(9);
    };
    for (;;) {
        // the descriptor three tiles ahead: asked for now, moved into place at the tile's end
        const int id3 = next_tile_s();
        const PipeTile d3 = pipe_desc(tab, id3 < 0 ? 0 : id3, id3 >= 0);
        for (int vv = 0; vv + 1 < F_V; ++vv) round(vv, std::false_type{});
        round(F_V - 1, std::true_type{});
        if (!nx.valid()) break;
        cur = nx;
        nx = nn;
        nn = d3;
    }
    STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------
// Geometry the host plans with.
// ------------------------------------------------------------------------------------------

// 0: the kernel does not take this input class; else log2 of the complex points (8 or 10)
int pipe64_class(int log2nc, int channels, int bps, int wlen, int inc)
{
    if (log2nc == 8 && channels == 1 && bps == 2 && inc % 2 == 0 && wlen >= 8) {
        // the fetch of a full tile (4 frames and the predecessor) must fit 2 vectors per lane
        const long bytes = ((long)3 * inc + wlen + 1) * 2;
        return bytes <= 2 * 64 * 16 ? 8 : 0;
    }
    if (log2nc == 10 && channels == 2 && bps == 3 && wlen > 1024 && wlen <= 2048) return 10;
    return 0;
}
// waves per workgroup: the 256-point class takes 12 (three per SIMD) when `fits12` says that 12
// regions fit the LDS beside the tables, else 8
int pipe64_waves(int cls, bool fits12) { return cls == 8 ? (fits12 ? 12 : kPipeWaves) : kPipeWideWaves; }
int pipe64_frames_per_tile(int cls) { return cls == 8 ? 4 : 1; }
int pipe64_lanes(int cls) { return cls == 8 ? 16 : 64; }
int pipe64_nz(int cls, int wlen)
{
    const int per = cls == 8 ? 32 : 128; // samples per first-stage input index
    const int nz = (wlen + per - 1) / per;
    return cls == 8 ? (nz <= 13 ? 13 : 16) : (nz <= 14 ? 14 : 16);
}
// LDS bytes of one wave's raw PCM area: the tile's bytes (256-point class; the 1024-point class
// reads its bytes straight from memory).  (A lane also reads the instants behind the frame's last
// sample up to the end of its first-stage inputs, where the window is zero: those reads may run
// into the wave's own work area, which follows the raw bytes and is larger.)
int pipe64_raw_bytes(int cls, int wlen, int inc)
{
    if (cls != 8) return 0;
    const int gt = pipe64_frames_per_tile(cls), bpi = cls == 8 ? 2 : 6;
    const long tile = ((long)(gt - 1) * inc + wlen + 1) * bpi;
    return (int)((tile + 15 + 16) & ~15L);
}
int pipe64_area_elems(int cls) { return (1 << cls) + (1 << cls) / 16; }
// shared tables in front of the waves' regions, in doubles (1024-point class: filterbank weights,
// window pairs, the lanes' mel part tables; its cepstral table follows them when the plan finds room)
int pipe64_tab_elems(int cls, int wlen, int dctElems, int melSteps)
{
    const int nc = 1 << cls;
    int n = cls == 8 ? melSteps * 64 : nc + 8;
    if (cls == 8) n += 2 * (pipe64_nz(cls, wlen) * 16 + 15 * 16 + nc / 2) + dctElems;
    else n += 2 * pipe64_nz(cls, wlen) * 64 + (kPipeParts + 64) * 2 + 2; // (+ the run slots)
    return (n + 1) & ~1;
}
// bytes from a frame's first instant that the 1024-point class's loads reach (12 per sample pair)
int pipe64_reach_bytes(int wlen) { return 12 * 64 * pipe64_nz(10, wlen); }

template <int NZ, int WAVES>
static int launch_pipe_t(const FrontArgs<double> &a, int numBlocks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    const hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_pipe64<NZ, WAVES>), attrSet);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_pipe64<NZ, WAVES>), dim3(numBlocks), dim3(WAVES * 64), lds, st, a);
    return (int)hipGetLastError();
}

template <int NZ, bool ZCT>
static int launch_pipew_t(const FrontArgs<double> &a, int numBlocks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    const hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_pipe64w<NZ, ZCT>), attrSet);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_pipe64w<NZ, ZCT>), dim3(numBlocks), dim3(kPipeWideWaves * 64), lds, st, a);
    return (int)hipGetLastError();
}

int launch_pipe64(const FrontArgs<double> &a, int cls, int waves, int numBlocks, size_t lds, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nz = pipe64_nz(cls, a.wlen);
    if (cls == 8) {
        if (waves == 12) return nz == 13 ? launch_pipe_t<13, 12>(a, numBlocks, lds, st) : launch_pipe_t<16, 12>(a, numBlocks, lds, st);
        return nz == 13 ? launch_pipe_t<13, 8>(a, numBlocks, lds, st) : launch_pipe_t<16, 8>(a, numBlocks, lds, st);
    }
    if (a.zcFlag) return nz == 14 ? launch_pipew_t<14, true>(a, numBlocks, lds, st) : launch_pipew_t<16, true>(a, numBlocks, lds, st);
    return nz == 14 ? launch_pipew_t<14, false>(a, numBlocks, lds, st) : launch_pipew_t<16, false>(a, numBlocks, lds, st);
}

} // namespace mfcc
