// HIP kernels for gfx950 (MI355X): PCM unpack -> pre-emphasis -> framing -> Hamming -> real FFT
// -> power -> HTK mel filterbank -> log -> DCT / C0 / energy / ZCR / brightness / sub-band energy
// (one fused kernel), regression deltas, per-utterance z-norm, and integer channel sums.
//
// The fused kernel follows the arithmetic of Wave2FBank / FBank2MFCC / FBank2C0 (mfcc.c:125-208)
// and of the frame loop of main.c:217-250, but is organised for the hardware: one 256-thread
// workgroup owns a tile of consecutive frames of one utterance, reads the tile's PCM bytes once
// (16-byte coalesced loads), keeps the pre-emphasised rows in LDS, and gives every (frame, vec)
// pair to a group of G = nc/16 lanes that hold 16 complex points each in registers.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "front_util.h"

namespace mfcc {

#ifdef MFCC_NO_REGSPLIT // A/B switch for measurements: split through the LDS work area everywhere
constexpr bool kNoRegSplit = true;
#else
constexpr bool kNoRegSplit = false;
#endif

#include "stamps.h"

// Registers the compiler may use per lane, as waves per SIMD it has to leave room for.
constexpr int front_waves_rule(bool f64, int log2nc, int pplog)
{
    if (pplog == 4) return f64 ? (log2nc <= 7 ? 2 : 1) : (log2nc <= 11 ? 2 : 1);
    if (pplog == 3) return f64 ? 2 : 3;
    return f64 ? 3 : 5;
}
template <typename T, int LOG2NC, int PPLOG>
constexpr int front_waves_per_simd()
{
    return front_waves_rule(sizeof(T) == 8, LOG2NC, PPLOG);
}

// ------------------------------------------------------------------------------------------
// Fused static-feature kernel.  TLDS: window / twiddle / split tables are staged in LDS.
// ------------------------------------------------------------------------------------------

template <typename T, int LOG2NC, int PPLOG, bool TLDS, int MODE>
__global__ __launch_bounds__(kBlock, (front_waves_per_simd<T, LOG2NC, PPLOG>())) void k_front(
    const FrontArgs<T> a)
{
    using GEO = Geo<LOG2NC, PPLOG>;
    constexpr int NC = GEO::NC, G = GEO::G, PPL = GEO::PPL;
    constexpr int TEAM = GEO::TEAM, NTEAMS = GEO::NTEAMS, GT = GEO::GT;
    constexpr int R0B = LOG2NC < GEO::RBMAX ? LOG2NC : GEO::RBMAX, R0 = 1 << R0B; // first-stage radix
    constexpr int NPAIR = PPL / 2 > 0 ? PPL / 2 : 1;
    constexpr int VEC = 16 / (int)sizeof(T); // elements of one 16-byte store

    // MODE 0: general.  MODE >= 1 (LEAN), MODE 2 (MELM): see below.
    constexpr bool LEAN = MODE >= 1, MELM = MODE == 2;
    // LEAN: the host promises mono 16-bit input, one derived row, even frame shift and none of
    // the zero-crossing / brightness / sub-band / magnitude / zero-mean options, so those paths
    // (and the registers they pin) are compiled out.
    const int F_zc = LEAN ? 0 : a.zcFlag, F_S = LEAN ? 0 : a.S, F_fftLength = LEAN ? 0 : a.fftLength;
    const int F_bright = LEAN ? 0 : a.brightFlag, F_zeroMean = LEAN ? 0 : a.zeroMean;
    const int F_channels = LEAN ? 1 : a.channels, F_bps = LEAN ? 2 : a.bps, F_V = LEAN ? 1 : a.V;
    const int F_pairAligned = LEAN ? 1 : a.pairAligned;

    const int tid = threadIdx.x;
    // wave-uniform when a team is a wave: keeps the tile bookkeeping in scalar registers
    const int team = NTEAMS > 1 ? __builtin_amdgcn_readfirstlane(tid / TEAM) : 0;
    const int tt = tid & (TEAM - 1);                // lane inside the team
    const int l = tt & (G - 1), grp = tt / G;       // lane inside the frame, frame slot inside the team

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ldsTab = reinterpret_cast<T *>(smem);        // table copies if TLDS (see below)
    T *rows = ldsTab + a.tabElems + team * a.teamElems; // V x rowPitch pre-emphasised samples
    T *zbuf = rows + F_V * a.rowPitch;              // GT work areas; raw PCM bytes alias it
    T *featBase = zbuf + a.rawOrZElems;             // framesPerTile x vSize static features (+ VEC)
    T *fbkSep = featBase + a.featElems;             // framesPerTile x V x P log-mel values (cepstral mode)
    T *red = fbkSep + a.fbkElems;                   // 16 elements: 4 slots of per-wave sums
    unsigned *raw = reinterpret_cast<unsigned *>(zbuf);

    const int bpi = F_channels * F_bps;

    // tables: [window (winPad) | twiddles (2 nc) | split factors (nc) | loWt (nc) | laneSeg (4 G maxSeg) | dctT]
    const T *win = a.window, *tw = a.tw, *split = a.splitTw, *lw = a.loWt;
    const int4 *lseg = reinterpret_cast<const int4 *>(a.laneSeg);
    const T *dctT = a.dctT;
    if constexpr (TLDS) {
        T *w = ldsTab, *t = w + a.winPad, *sp = t + 2 * NC, *lo = sp + (NC > 1 ? NC : 2);
        int *c = reinterpret_cast<int *>(lo + NC);
        // (the lean modes read window, split factors and mel weights once from global memory into
        // registers, and the matrix-core mel needs no segment map: those copies are skipped)
        if constexpr (!LEAN) {
            for (int i = tid; i < a.winPad; i += kBlock) w[i] = a.window[i];
            for (int i = tid; i < (NC > 1 ? NC : 2); i += kBlock) sp[i] = a.splitTw[i];
            for (int i = tid; i < NC; i += kBlock) lo[i] = a.loWt[i];
        }
        for (int i = tid; i < 2 * NC; i += kBlock) t[i] = a.tw[i];
        if constexpr (!MELM)
            for (int i = tid; i < 4 * G * a.maxSeg; i += kBlock) c[i] = a.laneSeg[i];
        T *dl = reinterpret_cast<T *>(c + 4 * G * a.maxSeg);
        for (int i = tid; i < a.dctElems; i += kBlock) dl[i] = a.dctT[i];
        dctT = dl;
        tw = t;
        if constexpr (!LEAN) {
            win = w;
            split = sp;
            lw = lo;
        }
        lseg = reinterpret_cast<const int4 *>(c);
        __syncthreads();
    }

    // Lean mode: a lane windows the same sample positions and weights the same bins in every
    // frame, so those table entries live in registers for the whole launch.
    T wreg[LEAN ? 2 * PPL : 1], lwreg[LEAN ? 2 * NPAIR : 1];
    cpx<T> swreg[LEAN ? NPAIR : 1];
    if constexpr (LEAN) {
#pragma unroll
        for (int b = 0; b < PPL / R0; ++b)
#pragma unroll
            for (int r = 0; r < R0; ++r) {
                const int s0 = 2 * (l + G * b + r * (NC / R0));
                wreg[2 * (b * R0 + r)] = win[s0];
                wreg[2 * (b * R0 + r) + 1] = win[s0 + 1];
            }
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int k = l + G * q;
            const int k2 = k == 0 ? NC / 2 : NC - k;
            if constexpr (!MELM) {
                lwreg[2 * q] = lw[k];
                lwreg[2 * q + 1] = lw[NC > 1 ? k2 : 0];
            }
            swreg[q] = reinterpret_cast<const cpx<T> *>(split)[k];
        }
    }
    // MELM: the mel filterbank of the team's frames in flight as 4 x 4 x 1 matrix-core products.
    // The 64 lanes form 16 blocks of 4; block b owns 4 consecutive channels and a contiguous bin
    // range.  Per step t lane (b, i) feeds the power of frame slot i at bin r0_b + t and the weight
    // of that bin for channel 4 q_b + i; element r of the result is frame slot r's sum for the
    // lane's channel.  Wide channel groups are cut over up to 4 neighbouring blocks, added at the end.
    // The lane's weights stay in registers; r0, the role flags and the channel come from melMeta.
    float melW[MELM ? kMelSteps : 1];
    const T *melA = zbuf;
    int melFlags = 0, melCh = -1;
    if constexpr (MELM) {
        const int lane = tid & 63;
#pragma unroll
        for (int t = 0; t < kMelSteps; ++t) melW[t] = t < a.melSteps ? a.melW[t * 64 + lane] : 0.0f;
        const int4 mm = reinterpret_cast<const int4 *>(a.melMeta)[lane];
        melA = zbuf + ((lane & 3) < GT ? (lane & 3) : 0) * a.zPitch + mm.x;
        melFlags = mm.y;
        melCh = mm.z;
    }

    // With 16 points per lane the kernel runs at two waves per SIMD: the 256-register budget also
    // holds the lane's twiddle factors.
    constexpr bool TWREG = LEAN && PPLOG == 4;
    // Matrix-core mode with 16 lanes x 16 points per frame: the last FFT stage stays in registers
    // and the real-FFT split takes its partner points from the neighbouring lane by DPP.
    constexpr bool REGSPLIT = MELM && sizeof(T) == 4 && G == 16 && PPL == 16 && !kNoRegSplit;
    cpx<T> twr[TWREG ? fft_twiddle_regs<LOG2NC, PPLOG>() : 1];
    if constexpr (TWREG) fft_load_twiddles<T, LOG2NC, PPLOG>(twr, l, reinterpret_cast<const cpx<T> *>(tw));

    STAMP_DECL
    STAMP_START;
    // MELM: the next tile's raw samples are fetched into registers during the last round's mel phase
    // (the FFT's point registers are free then): PFN 16-byte vectors per lane, for tiles the host
    // marked as fitting (TileDesc bit 61).
    typedef unsigned int u4u __attribute__((ext_vector_type(4), aligned(1)));
    constexpr int PFN = MELM ? 3 : 1;
    u4u pf[PFN];
    bool pfValid = false; // team-uniform
#pragma unroll
    for (int j = 0; j < PFN; ++j) pf[j] = u4u{0u, 0u, 0u, 0u};

    // Only the first activeTeams teams of a workgroup take tiles (fewer than NTEAMS when the
    // team regions of all of them would not fit the LDS); the others leave here.
    // MELM: the sample pairs of the team's next round (same tile) are requested before the mel
    // phase of the current one, so that their LDS latency is hidden behind it.
    constexpr bool XPRE = false && MELM && sizeof(T) == 4 && PPL == R0;
    cpx<T> xpre[XPRE ? PPL : 1];
    bool xpreValid = false; // team-uniform
    const int tstride = gridDim.x * a.activeTeams;
    int tile = team < a.activeTeams ? blockIdx.x * a.activeTeams + team : a.numTiles;
    TileDesc td = {0, 0};
    if (tile < a.numTiles) td = a.tiles[tile];
    unsigned warm0 = 0, warm1 = 0; // results of the cache-warming loads (see below)
    for (; tile < a.numTiles; tile += tstride) {
        asm volatile("" ::"v"(warm0), "v"(warm1)); // the previous tile's warming loads end here
        const int nf = (int)((td.src >> 48) & 0xfffu);
        const int lead = (int)((td.src >> 60) & 1u);            // the first sample has a predecessor
        const unsigned char *g0 = a.pcm + (td.src & kTileOffMask);
        const int ns = (nf - 1) * a.inc + a.wlen;                 // frame samples in the tile
        const long long outRow = td.outRow;
        const bool direct = MELM && ((td.src >> 61) & 1u) != 0; // see the register path below
        const int tnext = tile + tstride;
        if (tnext < a.numTiles) td = a.tiles[tnext];              // descriptor of the team's next tile
        double mean[4] = {0, 0, 0, 0};
        if (F_zeroMean) {
            const double *m = a.utts[a.tileUtt[tile]].mean;
            mean[0] = m[0], mean[1] = m[1], mean[2] = m[2], mean[3] = m[3];
        }
        // Staging area shifted so that it is 16-byte aligned exactly where the output is.
        T *dst = a.out + outRow * a.step;
        const int mis = a.step == a.vSize ? (int)((reinterpret_cast<unsigned long long>(dst) / sizeof(T)) & (VEC - 1)) : 0;
        T *feat = featBase + mis;

        // A: tile bytes -> LDS -> derived, mean-free, pre-emphasised rows (see unpack_tile*).
        const bool exact = sizeof(T) == 8 || F_zc != 0;
        const bool fastA = lead && (F_bps == 2 || F_bps == 3);
        if (direct) {
            // Register path (16-bit mono, the tile's samples fit PFN vectors per lane, the bytes up
            // to the end of its last vector lie inside the PCM buffer): vector j = raw samples
            // 8j .. 8j+7, sample 0 being the predecessor of the tile's first sample; they arrive
            // in registers (usually fetched during the previous tile), are pre-emphasised there
            // -- row[s] = raw[s+1] - k raw[s], the vector's last output taking raw[8j+8] from the
            // next lane -- and stored as two 16-byte vectors; LDS never sees the raw bytes.
            const int nvec = (ns + 1 + 7) >> 3;
            if (!pfValid) {
#pragma unroll
                for (int j = 0; j < PFN; ++j)
                    if (tt + j * TEAM < nvec) pf[j] = reinterpret_cast<const u4u *>(g0)[tt + j * TEAM];
            }
            const float kf = (float)a.preK;
#pragma unroll
            for (int j = 0; j < PFN; ++j) {
                const int vj = tt + j * TEAM;
                unsigned nx = (unsigned)__shfl_down((int)pf[j].x, 1, 64);
                const unsigned wrap = j + 1 < PFN ? (unsigned)__shfl((int)pf[j + 1 < PFN ? j + 1 : j].x, 0, 64) : 0u;
                nx = (tt & 63) == 63 ? wrap : nx;
                if (vj < nvec) {
                    const unsigned d[5] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w, nx};
                    float sm[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i)
                        sm[i] = (float)(short)((i & 1) ? (d[i >> 1] >> 16) : (d[i >> 1] & 0xffffu));
                    float y[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = sm[i + 1] - kf * sm[i]; // (finite behind the tile's last sample too: zeroed below)
                    float *dstRow = reinterpret_cast<float *>(rows) + 8 * vj;
                    *reinterpret_cast<float4 *>(dstRow) = make_float4(y[0], y[1], y[2], y[3]);
                    *reinterpret_cast<float4 *>(dstRow + 4) = make_float4(y[4], y[5], y[6], y[7]);
                }
            }
            STAMP(0);
        } else if (fastA) {
            // Copy starts exactly at the predecessor of the tile's first sample (16-byte vector
            // loads from a possibly unaligned address), the last partial vector byte by byte so
            // that nothing behind the tile's bytes is touched.
            const int nbytes = (ns + 1) * bpi, nfull = nbytes >> 4;
            if (!SKIP(7)) {
                for (int i = tt; i < nfull; i += 4 * TEAM) { // four loads in flight per lane
                    u4u v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ij = i + j * TEAM < nfull ? i + j * TEAM : i;
                        v4[j] = reinterpret_cast<const u4u *>(g0)[ij];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j * TEAM < nfull)
                            reinterpret_cast<uint4 *>(raw)[i + j * TEAM] = make_uint4(v4[j].x, v4[j].y, v4[j].z, v4[j].w);
                }
                for (int i = (nfull << 4) + tt; i < nbytes; i += TEAM)
                    reinterpret_cast<unsigned char *>(raw)[i] = g0[i];
            }
            team_sync<TEAM>();
            STAMP(0);
            if (!SKIP(0)) {
                if (F_channels == 2) {
                    if (F_bps == 2)
                        unpack_tile_fast_sel<T, 2, 2, TEAM>(exact, rows, raw, a.rowPitch, F_V, ns, mean, a.preK, tt);
                    else
                        unpack_tile_fast_sel<T, 2, 3, TEAM>(exact, rows, raw, a.rowPitch, F_V, ns, mean, a.preK, tt);
                } else {
                    if (F_bps == 2)
                        unpack_tile_fast_sel<T, 1, 2, TEAM>(exact, rows, raw, a.rowPitch, F_V, ns, mean, a.preK, tt);
                    else
                        unpack_tile_fast_sel<T, 1, 3, TEAM>(exact, rows, raw, a.rowPitch, F_V, ns, mean, a.preK, tt);
                }
            }
        } else {
            // General form (first tile of an utterance, 32-bit samples): 16-byte aligned vector
            // loads (each holds >= 1 valid byte, so it stays inside the buffer's pages), samples
            // picked at run-time byte offsets, double arithmetic.
            const unsigned long long gaddr = reinterpret_cast<unsigned long long>(g0);
            const int gmis = (int)(gaddr & 15ull);
            const uint4 *src = reinterpret_cast<const uint4 *>(gaddr - gmis);
            const int nvec = (gmis + (ns + lead) * bpi + 15) >> 4;
            for (int i = tt; i < nvec; i += TEAM) reinterpret_cast<uint4 *>(raw)[i] = src[i];
            team_sync<TEAM>();
            STAMP(0);
            if (F_channels == 2)
                unpack_tile<T, true, 2, TEAM>(rows, raw, a.rowPitch, F_V, F_bps, gmis, lead, ns, mean, a.preK, tt);
            else
                unpack_tile<T, true, 1, TEAM>(rows, raw, a.rowPitch, F_V, F_bps, gmis, lead, ns, mean, a.preK, tt);
        }
        for (int s = tt; s < a.rowTail; s += TEAM)
            for (int v = 0; v < F_V; ++v) rows[v * a.rowPitch + ns + s] = (T)0;
        team_sync<TEAM>();
        STAMP(1);

        // Pull the next tile's cache lines towards L2 while this tile is computed: one 4-byte load
        // per 128-byte line (at most two per lane) whose results are only "used" when the next tile
        // starts, so nothing waits for them; the data itself is loaded when that tile starts.
        // (Loads into registers on purpose: a direct-to-LDS load would make every later LDS read of
        // the tile wait for it.)
        if (tnext < a.numTiles && !SKIP(7)) {
            const int nfn = (int)((td.src >> 48) & 0xfffu);
            const int nbn = ((nfn - 1) * a.inc + a.wlen + 1) * bpi;
            const unsigned char *gn = a.pcm + (td.src & kTileOffMask);
            if (tt * 128 < nbn)
                warm0 = *reinterpret_cast<const unsigned *>(reinterpret_cast<unsigned long long>(gn + tt * 128) & ~3ull);
            if ((tt + TEAM) * 128 < nbn)
                warm1 = *reinterpret_cast<const unsigned *>(reinterpret_cast<unsigned long long>(gn + (tt + TEAM) * 128) & ~3ull);
        }

        const int nItems = nf * F_V;
        for (int base = 0; base < nItems; base += GT) {
            const int item = base + grp;
            const bool active = item < nItems;
            const int it = active ? item : 0; // V is 1..4: no general division needed
            const int fr = F_V == 1 ? it : (F_V == 2 ? it >> 1 : (F_V == 4 ? it >> 2 : it / 3));
            const int vv = it - fr * F_V;
            const T *x = rows + vv * a.rowPitch + fr * a.inc;
            T *Zt = zbuf + grp * a.zPitch;
            cpx<T> *Z = reinterpret_cast<cpx<T> *>(Zt);
            cpx<T> *AB = Z;                               // per bin (loWt*p, p - loWt*p)
            T *Tt = Zt + 2 * NC, *Ut = Tt + a.melPieces * (a.P + 1); // per-segment (and part) sums of the two
            T *Pw = Zt;                                   // plain power values, before AB
            T *fbk = fbkSep + item * a.P;                 // log-mel values of the item (only if fbkElems)
            T *fo = feat + item * a.D;

            // ---- window, frame energy, zero crossings, first FFT stage input (Ham main.c:229,
            //      te mfcc.c:138-142, zero padding mfcc.c:144-147)
            // Loads are unconditional and need no bounds test: rows are zero-filled for
            // fftN - wlen + 2 elements behind the tile's last sample and the window table is
            // zero from index wlen on (a Hamming window is > 0 inside, so w > 0 <=> index < wlen).
            cpx<T> v[PPL];
            T te = 0;
            int zc = 0;
            if (SKIP(1)) {
#pragma unroll
                for (int i = 0; i < PPL; ++i) v[i] = {(T)(l + i), (T)(grp - i)};
            } else if constexpr (XPRE) {
                if (!xpreValid) {
                    const volatile cpx<T> *xp = reinterpret_cast<const volatile cpx<T> *>(x + 2 * l);
#pragma unroll
                    for (int r = 0; r < R0; ++r) xpre[r] = xp[r * (NC / R0)];
                }
#pragma unroll
                for (int r = 0; r < R0; ++r) v[r] = xpre[r] * cpx<T>{wreg[2 * r], wreg[2 * r + 1]};
            } else
#pragma unroll
            for (int b = 0; b < PPL / R0; ++b) {
                // sample pairs (2n, 2n+1) and their window values for n = l + G b + r nc/R0
                cpx<T> xs[R0], ws[R0];
                const int n0 = l + G * b;
                if constexpr (sizeof(T) == 4) {
                    if (F_pairAligned) { // frame starts on an even sample: 8-byte aligned pairs
                        lds_read_strided<R0, 8 * (NC / R0)>(xs, x + 2 * n0);
                    } else {
#pragma unroll
                        for (int r = 0; r < R0; ++r) {
                            const int s0 = 2 * (n0 + r * (NC / R0));
                            xs[r] = {x[s0], x[s0 + 1]};
                        }
                    }
                    if constexpr (LEAN) {
#pragma unroll
                        for (int r = 0; r < R0; ++r) ws[r] = {wreg[2 * (b * R0 + r)], wreg[2 * (b * R0 + r) + 1]};
                    } else if constexpr (TLDS) {
                        lds_read_strided<R0, 8 * (NC / R0)>(ws, win + 2 * n0);
                    } else {
#pragma unroll
                        for (int r = 0; r < R0; ++r)
                            ws[r] = *reinterpret_cast<const cpx<T> *>(win + 2 * (n0 + r * (NC / R0)));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const int s0 = 2 * (n0 + r * (NC / R0));
                        if (F_pairAligned)
                            xs[r] = *reinterpret_cast<const cpx<T> *>(x + s0);
                        else
                            xs[r] = {x[s0], x[s0 + 1]};
                        ws[r] = *reinterpret_cast<const cpx<T> *>(win + s0);
                    }
                }
#pragma unroll
                for (int r = 0; r < R0; ++r) v[b * R0 + r] = {xs[r].x * ws[r].x, xs[r].y * ws[r].y};
            }
            if (a.energyFlag) { // te of mfcc.c:138-142 (the padding adds nothing)
                cpx<T> e2 = {0, 0};
#pragma unroll
                for (int i = 0; i < PPL; ++i) e2 = {e2.x + v[i].x * v[i].x, e2.y + v[i].y * v[i].y};
                te = (T)4 * (e2.x + e2.y); // the window table holds half the Hamming values
            }
            STAMP(2);
            if (a.energyFlag) te = group_part<T, G>(te, red, tt);
            if (F_zc) { // sign changes between neighbours of the un-windowed frame (main.c:227)
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const int s0 = 2 * (l + G * i);
                    const T x0 = x[s0], x1 = x[s0 + 1], x2 = x[s0 + 2];
                    const bool c01 = (x0 < 0 && x1 > 0) || (x0 > 0 && x1 < 0);
                    const bool c12 = (x1 < 0 && x2 > 0) || (x1 > 0 && x2 < 0);
                    zc += (c01 && s0 + 1 < a.wlen) ? 1 : 0; // pairs inside the frame only
                    zc += (c12 && s0 + 2 < a.wlen) ? 1 : 0;
                }
                zc = group_part<int, G>(zc, reinterpret_cast<int *>(red + 4), tt);
            }

            // ---- nc-point complex FFT of z[n] = x[2n] + i x[2n+1]   (Realft mfcc.c:262-270)
            if (!SKIP(2))
                fft_stages<T, LOG2NC, PPLOG, 0, 0, TLDS, TWREG, 0, REGSPLIT>(v, Z, l, reinterpret_cast<const cpx<T> *>(tw),
                                                                          twr);
            STAMP(3);

            // ---- real-FFT split and power  (mfcc.c:274-290, ek of mfcc.c:153-155)
            // X[k] = E + w O, X[nc-k] = conj(E - w O), w = exp(-i pi k / nc); only |X|^2 is kept.
            T pa[NPAIR], pb[NPAIR];
            cpx<T> zas[NPAIR], zbs[NPAIR]; // Z[k] and Z[nc - k] for k = l + G q
            if constexpr (REGSPLIT) {
                // v[r] = Z[l + 16 r].  Z[nc - l - 16 q] is register 15 - q of lane 16 - l; for
                // lane 0 it is the lane's own register 16 - q (q = 0 is fixed up below).
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    zas[q] = v[q];
                    const cpx<T> hi = v[PPL - 1 - q], own = v[(PPL - q) & (PPL - 1)];
                    const float px = row_negated_lane((float)hi.x), py = row_negated_lane((float)hi.y);
                    zbs[q] = {l == 0 ? own.x : (T)px, l == 0 ? own.y : (T)py};
                }
            } else if constexpr (sizeof(T) == 4 && G % 16 == 0 && PPL >= 2) {
                // zpad(l + G q) = zpad(l) + q (G + G/16), zpad(nc - l - G q) = zpad(nc - l) - q (G + G/16)
                constexpr int QS = (G + G / 16) * 8;
                cpx<T> rev[NPAIR];
                lds_read_strided<NPAIR, QS>(zas, &Z[zpad(l)]);
                lds_read_strided<NPAIR, QS>(rev, &Z[zpad(NC - l - G * (NPAIR - 1))]);
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) zbs[q] = rev[NPAIR - 1 - q];
            } else {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    zas[q] = Z[zpad(k)];
                    zbs[q] = Z[zpad(NC - k)];
                }
            }
            if (SKIP(3)) {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    pa[q] = v[q].x * v[q].x + (T)1;
                    pb[q] = v[q].y * v[q].y + (T)1;
                }
            } else
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                const int k = l + G * q; // k == 0 (lane 0, q == 0) reads a don't-care word: fixed up below
                const cpx<T> za = zas[q], zb = zbs[q];
                const cpx<T> sw = LEAN ? swreg[q] : reinterpret_cast<const cpx<T> *>(split)[k];
                split_power<T>(za, zb, sw, pa[q], pb[q]);
            }
            {
                // bin 0 packs Re + Im of Z[0] (mfcc.c:288-290); bin nc/2 is |Z[nc/2]|^2
                // (register split: lane 0, the only user, holds Z[0] and Z[nc/2] = Z[16 * 8] itself)
                const cpx<T> z0 = REGSPLIT ? v[0] : Z[0], zh = REGSPLIT ? v[PPL / 2] : Z[zpad(NC / 2)];
                // (the transform is that of half-scaled input)
                const T p0 = (T)4 * (z0.x + z0.y) * (z0.x + z0.y);
                const T ph = NC > 1 ? (T)4 * (zh.x * zh.x + zh.y * zh.y) : (T)0;
                pa[0] = l == 0 ? p0 : pa[0];
                pb[0] = l == 0 ? ph : pb[0];
            }
            // Brightness without sub-bands: power-weighted mean bin index straight from registers.
            T regTotal = 0, regNum = 0;
            if (F_bright && F_S == 0) {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    const int k2 = k == 0 ? NC / 2 : NC - k;
                    regTotal += pa[q] + pb[q];
                    regNum += (T)(k + 1) * pa[q] + (T)(k2 + 1) * pb[q];
                }
                regTotal = group_part<T, G>(regTotal, red + 8, tt);
                regNum = group_part<T, G>(regNum, red + 12, tt);
            }
            STAMP(4);
            group_sync<G>(); // every lane has read its Z points: the work area can be reused

            // ---- sub-band energies / brightness / debug magnitudes need the plain power values
            //      of the whole frame: park them in the work area first (call sites main.c:235-236)
            if (F_S > 0 || F_fftLength > 0) {
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    const int k2 = k == 0 ? NC / 2 : NC - k;
                    Pw[k] = pa[q];
                    if (NC > 1) Pw[k2] = pb[q];
                }
                group_sync<G>();
                if (F_S > 0) {
                    // Each band gets a power-of-two share of the frame's lanes: a lane sums its
                    // part of the band, the shares are folded by lane exchange, one lane per band
                    // parks the band's two sums behind the power values.
                    const int nseg = F_S;
                    const int width = NC / nseg;
                    const int subAt = a.M + a.c0Flag + a.energyFlag + F_zc + F_bright;
                    int lpb = 1;
                    while (lpb * 2 * nseg <= G && lpb < 64) lpb *= 2;
                    const int perPass = G / lpb;
                    T *bandE = Zt + 2 * NC + 2 * a.melPieces * (a.P + 1); // behind the mel segment sums
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
                    if (active && l == 0) {
                        T total = 0, num = 0;
                        for (int g = 0; g < nseg; ++g) {
                            total += bandE[g];
                            num += bandE[nseg + g];
                        }
                        if (F_bright)
                            fo[a.M + a.c0Flag + a.energyFlag + F_zc] = num / total / (T)NC;
                        for (int g = 0; g < nseg; ++g) fo[subAt + g] = bandE[g] / total;
                    }
                }
                if (F_fftLength > 0) {                                             // main.c:247
                    const int at = a.M + a.c0Flag + a.energyFlag + F_zc + F_bright + F_S;
                    if (active)
                        for (int q = l; q < F_fftLength; q += G) fo[at + q] = tsqrt<T>(Pw[q]);
                    // the per-bin products below overwrite the power values.  (Every frame slot takes
                    // part, the ones past the tile's last frame too: frames wider than a wave meet at
                    // the workgroup's barrier, which must see the same count from all of them.)
                    group_sync<G>();
                }
            }
            if (F_S == 0 && F_bright && active && l == 0)
                fo[a.M + a.c0Flag + a.energyFlag + F_zc] =
                    group_total<T, G>(regNum, red + 12, tt) / group_total<T, G>(regTotal, red + 8, tt) / (T)NC;
            if (active && l == 0) {
                if (a.energyFlag) fo[a.M + a.c0Flag] = tlog<T>(group_total<T, G>(te, red, tt)); // main.c:243
                if (F_zc)                                                           // main.c:244
                    fo[a.M + a.c0Flag + a.energyFlag] =
                        (T)group_total<int, G>(zc, reinterpret_cast<const int *>(red + 4), tt) / (T)(a.wlen - 1);
            }

            if constexpr (MELM) {
                if (base + GT >= nItems) { // last round of the tile: start the next tile's loads
                    pfValid = tnext < a.numTiles && ((td.src >> 61) & 1u) != 0 && !SKIP(7);
                    if (pfValid) {
                        const int nfn = (int)((td.src >> 48) & 0xfffu);
                        const int nvn = ((nfn - 1) * a.inc + a.wlen + 1 + 7) >> 3;
                        const unsigned char *gn = a.pcm + (td.src & kTileOffMask);
#pragma unroll
                        for (int j = 0; j < PFN; ++j)
                            if (tt + j * TEAM < nvn) pf[j] = reinterpret_cast<const u4u *>(gn)[tt + j * TEAM];
                    }
                }
                // plain power values of the frame into its work area, bin order
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    const int k = l + G * q;
                    const int k2 = k == 0 ? NC / 2 : NC - k;
                    Pw[k] = pa[q];
                    Pw[k2] = pb[q];
                }
                team_sync<TEAM>();
                STAMP(5);
                Acc4<float> macc = {0, 0, 0, 0};
                const bool slotLive = (tt & 3) < GT;
                // always kMelSteps steps, straight-line so that the loads run ahead of the products:
                // the weights are zero past the plan's steps, and the extra bins read there are
                // finite values of the same work area
                float av[kMelSteps];
#pragma unroll
                for (int t = 0; t < kMelSteps; ++t) av[t] = melA[t];
                // two accumulators: the products of even and odd steps do not wait for each other
                Acc4<float> macc2 = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < kMelSteps; t += 2) {
                    macc = __builtin_amdgcn_mfma_f32_4x4x1f32(slotLive ? av[t] : 0.0f, melW[t], macc, 0, 0, 0);
                    macc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(slotLive ? av[t + 1] : 0.0f, melW[t + 1], macc2, 0, 0, 0);
                }
                macc += macc2;
                if constexpr (XPRE) {
                    xpreValid = base + GT < nItems && !SKIP(1);
                    if (xpreValid) { // (slots past the tile's last frame read frame 0: finite, unused)
                        const int itn = base + GT + grp < nItems ? base + GT + grp : 0;
                        const volatile cpx<T> *xp = reinterpret_cast<const volatile cpx<T> *>(rows + itn * a.inc + 2 * l);
#pragma unroll
                        for (int r = 0; r < R0; ++r) xpre[r] = xp[r * (NC / R0)];
                    }
                }
                // a channel quad cut over up to 4 neighbouring blocks: the chain's last block adds
                // the sums of the (up to 3) blocks before it, fetched in one batch
                {
                    float o1[4], o2[4], o3[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o1[r] = __shfl_up(macc[r], 4, 64);
                        o2[r] = __shfl_up(macc[r], 8, 64);
                        o3[r] = __shfl_up(macc[r], 12, 64);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sum = macc[r];
                        sum += (melFlags & 1) ? o1[r] : 0.0f;
                        sum += (melFlags & 4) ? o2[r] : 0.0f;
                        sum += (melFlags & 8) ? o3[r] : 0.0f;
                        macc[r] = sum;
                    }
                }
                if ((melFlags & 2) && melCh >= 0) { // owner of channel melCh: log(max(., 1)), mfcc.c:171-176
#pragma unroll
                    for (int r = 0; r < GT; ++r) {
                        const int it2 = base + r;
                        if (it2 < nItems) {
                            const T lv = tlog<T>(macc[r] < 1.0f ? (T)1 : (T)macc[r]);
                            if (a.fbkElems) fbkSep[it2 * a.P + melCh] = lv;
                            if (a.fbankFlag) feat[it2 * a.D + melCh] = lv;
                        }
                    }
                }
                team_sync<TEAM>();
            } else {
            // Per bin, exactly the two products of mfcc.c:160-162: t1 = loWt*ek and ek - t1.
            if (SKIP(4)) {
                asm volatile("" ::"v"(pa[0]), "v"(pb[NPAIR - 1]));
            } else
#pragma unroll
            for (int q = 0; q < NPAIR; ++q) {
                const int k = l + G * q;
                const int k2 = k == 0 ? NC / 2 : NC - k;
                // (magnitude instead of power when usePower is off, mfcc.c:154-157; never in the
                // lean instantiations)
                const T ea = (LEAN || a.usePower) ? pa[q] : tsqrt<T>(pa[q]);
                const T eb = (LEAN || a.usePower) ? pb[q] : tsqrt<T>(pb[q]);
                const T t1 = (LEAN ? lwreg[2 * q] : lw[k]) * ea;
                const T t2 = (LEAN ? lwreg[2 * q + 1] : lw[k2]) * eb;
                AB[k] = {t1, ea - t1};
                if (NC > 1) AB[k2] = {t2, eb - t2};
            }
            group_sync<G>();

            STAMP(5);
            // ---- mel filterbank (mfcc.c:152-163).  The bins whose lower channel is c form one
            //      contiguous segment; its sums T[c] = sum t1 and U[c] = sum (ek - t1) are taken in
            //      bin order, by the lane the host's balancing gave the segment to (long segments are
            //      cut into melPieces parts on different lanes when lanes are spare), and channel b is
            //      U[b-1] + T[b]; then log(max(., 1))  (mfcc.c:171-176).
            for (int sidx = 0; sidx < (SKIP(5) ? 0 : a.maxSeg); ++sidx) {
                const int4 sg = lseg[l * a.maxSeg + sidx];
                cpx<T> acc = {0, 0};
                for (int k = sg.x; k < sg.y; k += 4) { // 4 loads in flight; reads past the segment
                    cpx<T> vq[4];
                    if constexpr (sizeof(T) == 4) {
                        lds_read_strided<4, 8>(vq, &AB[k]);
                    } else {
                        vq[0] = AB[k], vq[1] = AB[k + 1], vq[2] = AB[k + 2], vq[3] = AB[k + 3];
                    }
                    const cpx<T> v0 = vq[0], v1 = vq[1], v2 = vq[2], v3 = vq[3];
                    acc = cadd(acc, v0);                // stay inside the work area and are dropped
                    if (k + 1 < sg.y) acc = cadd(acc, v1);
                    if (k + 2 < sg.y) acc = cadd(acc, v2);
                    if (k + 3 < sg.y) acc = cadd(acc, v3);
                }
                if (sg.z >= 0) {
                    Tt[sg.z] = acc.x;
                    Ut[sg.z] = acc.y;
                }
            }
            group_sync<G>();
            for (int ch = l; ch < (SKIP(6) ? 0 : a.P); ch += G) {
                T mu = Ut[ch], mt = Tt[ch + 1]; // parts of a segment in part (= bin) order
                for (int j = 1; j < a.melPieces; ++j) {
                    mu += Ut[j * (a.P + 1) + ch];
                    mt += Tt[j * (a.P + 1) + ch + 1];
                }
                const T m = mu + mt;
                const T lv = (LEAN || a.takeLogs) ? tlog<T>(m < (T)1 ? (T)1 : m) : m; // mfcc.c:171-176
                if (active && a.fbkElems) fbk[ch] = lv;
                if (active && a.fbankFlag) fo[ch] = lv;
            }
            group_sync<G>();

            }
            STAMP(6); // the barrier that closed the mel phase also frees the work area for the next round
            STAMP(7);
        }
        team_sync<TEAM>();
        // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208) of the whole
        //      tile: one lane per (frame, coefficient), summed over the channels in order
        if (!a.fbankFlag) {
            // out[item][j] = sum_k fbk[item][k] dct[j][k] as 16 x 16 x 4 matrix-core products: per
            // step lane (i, g) = (lane % 16, lane / 16) feeds fbk[item i][k0 + g] and
            // dct[coefficient i][k0 + g] and holds 4 items' results for coefficient i.
            // The table's row M is all ones (C0); its rows are zero from k = P to the pitch.
            const int JW = a.M + a.c0Flag;
            const int wv = tt >> 6, i16 = tt & 15, g4 = (tt >> 4) & 3;
            const int njb = (JW + 15) >> 4, nib = (nItems + 15) >> 4;
            for (int pr = wv; pr < nib * njb; pr += TEAM / 64) { // (item block, coefficient block) per wave
                const int ib = pr / njb, jb = pr - ib * njb;
                {
                    const int itemA = ib * 16 + i16, jB = jb * 16 + i16;
                    const T *fa = fbkSep + (itemA < nItems ? itemA : 0) * a.P;
                    const T *db = dctT + (jB < JW ? jB : 0) * a.dctPitch;
                    Acc4<T> acc = {0, 0, 0, 0};
                    for (int k0 = 0; k0 < a.P; k0 += 4) {
                        const int k = k0 + g4;
                        T av = fa[k < a.P ? k : 0], bv = db[k];
                        av = (itemA < nItems && k < a.P) ? av : (T)0;
                        bv = jB < JW ? bv : (T)0;
                        acc = mfma16(av, bv, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // result rows: 4 g + r in single precision, 4 r + g in double precision
                        const int item = ib * 16 + (sizeof(T) == 4 ? 4 * g4 + r : 4 * r + g4);
                        if (item < nItems && jB < JW) feat[item * a.D + jB] = acc[r] * a.mfnorm;
                    }
                }
            }
            team_sync<TEAM>();
        }
        STAMP(8);

        // ---- tile rows -> HBM (static block of every row; delta slots are written later)
        if (a.step == a.vSize) { // no delta slots: the tile's rows are one contiguous run
            const int n = SKIP(8) ? 0 : nf * a.vSize;
            const int head = mis ? min(n, VEC - mis) : 0; // elements before the first 16-byte boundary
            const int nv = (n - head) / VEC;
            if (tt < head) dst[tt] = feat[tt];
            typedef T vecT __attribute__((ext_vector_type(VEC)));
            const vecT *fv = reinterpret_cast<const vecT *>(feat + head);
            vecT *dv = reinterpret_cast<vecT *>(dst + head);
            for (int i = tt; i < nv; i += TEAM) dv[i] = fv[i];
            for (int i = head + nv * VEC + tt; i < n; i += TEAM) dst[i] = feat[i];
        } else {
            // delta slots follow every row's static block: one element per lane, rows back to back
            for (int i = tt; i < (SKIP(8) ? 0 : nf * a.vSize); i += TEAM) {
                const int r = i / a.vSize, c = i - r * a.vSize;
                dst[(long long)r * a.step + c] = feat[i];
            }
        }
        STAMP(9);
    }
    STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------
// Regression deltas (HTK Regress with head = tail = 0, call sites main.c:259-260)
// ------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(kBlock) void k_regress(const RegressArgs<T> a)
{
    for (int blk = blockIdx.x; blk < a.numBlocks; blk += gridDim.x) {
        const int u = a.blockUtt[blk];
        const UttDesc ud = a.utts[u];
        const int f0 = (blk - ud.regBegin) * a.regRows;
        const int nf = min(a.regRows, ud.numFrames - f0);
        const int n = ud.numFrames;
        T *base = a.data + ud.outRow * a.step;
        for (int i = threadIdx.x; i < nf * a.vSize; i += kBlock) {
            const int r = i / a.vSize, c = i - r * a.vSize;
            const int t0 = f0 + r;
            int back = t0, forw = t0;
            T sum = 0;
            for (int t = 1; t <= a.delwin; ++t) {
                if (t0 - t >= 0) --back;
                if (t0 + t <= n - 1) ++forw;
                sum += (T)t * (base[(long long)forw * a.step + a.srcCol + c] -
                               base[(long long)back * a.step + a.srcCol + c]);
            }
            base[(long long)t0 * a.step + a.dstCol + c] = (T)((double)sum / a.sigmaT2);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused deltas: every regression order of one block of rows in one pass, plus (optionally) the
// block's column sums for the z-norm.  The static rows of a sub-tile and a halo of
// (ORD - 1) * delwin rows per side are loaded once into LDS (row indices clamped to the utterance,
// as Regress clamps them); each order is computed from the previous one inside LDS, where the
// entry for a row outside the utterance is the value at the clamped row.  Arithmetic and
// rounding are those of k_regress.  Statistics cover the rows in [ownBegin, ownBegin + ownCount).
// ------------------------------------------------------------------------------------------

template <typename T, int ORD>
__global__ __launch_bounds__(kBlock) void k_delta(const DeltaArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int H = (ORD - 1) * a.delwin, SR = a.subRows, W = a.vSize;
    T *bufA = reinterpret_cast<T *>(dsm), *bufB = bufA + (size_t)(SR + 2 * H) * W;
    double *red = reinterpret_cast<double *>(dsm + a.redOffset); // 2 * kBlock doubles
    const int blk = blockIdx.x;
    const int u = a.blockUtt[blk];
    const UttDesc ud = a.utts[u];
    const int n = ud.numFrames;
    const int r0 = (blk - ud.regBegin) * a.regRows;
    const int nr = min(a.regRows, n - r0);
    T *base = a.data + ud.outRow * a.step;
    const int tid = threadIdx.x;
    // statistics: lane (rl, col) sums rows rl, rl + rowLanes, ... of every sub-tile
    double s1[ORD], s2[ORD];
#pragma unroll
    for (int o = 0; o < ORD; ++o) s1[o] = s2[o] = 0;
    const bool stats = a.partial != nullptr;
    const int ncol = min(kBlock, W), rowLanes = kBlock / ncol;
    const int col = tid % ncol, rl = tid / ncol;

    for (int s = r0; s < r0 + nr; s += SR) {
        const int rows = min(SR, r0 + nr - s);
        // static rows s - H .. s + rows + H - 1, clamped
        // (eight loads in flight per lane before the first is stored: the loop is bound by the
        // latency of these reads, not by their volume)
        for (int i0 = tid; i0 < (rows + 2 * H) * W; i0 += 8 * kBlock) {
            T tmp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * kBlock;
                if (i < (rows + 2 * H) * W) {
                    const int p = i / W, c = i - p * W;
                    const int t = min(max(s - H + p, 0), n - 1);
                    tmp[j] = base[(long long)t * a.step + c];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * kBlock;
                if (i < (rows + 2 * H) * W) bufA[i] = tmp[j];
            }
        }
        __syncthreads();
        if (stats && rl < rowLanes)
            for (int r = rl; r < rows; r += rowLanes) {
                const int row = s + r;
                if (row >= ud.ownBegin && row < ud.ownBegin + ud.ownCount) {
                    const double x = (double)bufA[(r + H) * W + col];
                    s1[0] += x;
                    s2[0] += x * x;
                }
            }
        T *prev = bufA, *next = bufB;
        int hp = H; // halo of `prev`: its entry p is row s - hp + p
#pragma unroll
        for (int o = 1; o < ORD; ++o) {
            const int hn = hp - a.delwin;
            // (T)((double)sum / sigmaT2): in single precision the correctly rounded float quotient is
            // the same number (sigmaT2 is a small integer, double carries more than 2 x 24 + 2 bits)
            const float sigmaF = (float)a.sigmaT2;
            auto quotient = [&](T sum) -> T {
                if constexpr (sizeof(T) == 4)
                    return sum / sigmaF;
                else
                    return (T)((double)sum / a.sigmaT2);
            };
            if ((W & 1) == 0 && (reinterpret_cast<unsigned long long>(a.data) & (2 * sizeof(T) - 1)) == 0) {
                // even row width (and an output matrix aligned for two elements): a lane forms two neighbouring columns at once (8- or 16-byte LDS
                // accesses, packed arithmetic); lane (rlp, cp) takes column pair cp of the entries
                // rlp, rlp + lanes, ...; the window is walked by pointer steps of one row
                const int WP = W >> 1, npair = min(kBlock, WP), lanes = kBlock / npair;
                const int cp = tid % npair, rlp = tid / npair;
                if (rlp < lanes)
                    for (int c2 = cp; c2 < WP; c2 += npair)
                        for (int p = rlp; p < rows + 2 * hn; p += lanes) {
                            const int t = min(max(s - hn + p, 0), n - 1); // the row whose delta this entry holds
                            const int q = t - (s - hp);                    // its entry in prev
                            const cpx<T> *ph = reinterpret_cast<const cpx<T> *>(prev + q * W) + c2, *pl = ph;
                            cpx<T> sum = {0, 0};
                            T ft = 1;
                            for (int tau = 1; tau <= a.delwin; ++tau) {
                                ph += WP;
                                pl -= WP;
                                sum = cfma_r(*ph - *pl, ft, sum); // sum + tau (x[t + tau] - x[t - tau]) per column
                                ft += (T)1;
                            }
                            const cpx<T> d = {quotient(sum.x), quotient(sum.y)};
                            reinterpret_cast<cpx<T> *>(next + p * W)[c2] = d;
                            const int row = s - hn + p;
                            if (row >= s && row < s + rows)
                                *reinterpret_cast<cpx<T> *>(base + (long long)row * a.step + o * W + 2 * c2) = d;
                        }
            } else if (rl < rowLanes) {
                for (int c = col; c < W; c += ncol)
                    for (int p = rl; p < rows + 2 * hn; p += rowLanes) {
                        const int t = min(max(s - hn + p, 0), n - 1);
                        const int q = t - (s - hp);
                        const T *ph = prev + q * W + c, *pl = ph;
                        T sum = 0, ft = 1;
                        for (int tau = 1; tau <= a.delwin; ++tau) {
                            ph += W;
                            pl -= W;
                            sum += ft * (*ph - *pl);
                            ft += (T)1;
                        }
                        const T d = quotient(sum);
                        next[p * W + c] = d;
                        const int row = s - hn + p;
                        if (row >= s && row < s + rows) base[(long long)row * a.step + o * W + c] = d;
                    }
            }
            __syncthreads();
            T *tmp = prev;
            prev = next;
            next = tmp;
            hp = hn;
            if (stats && rl < rowLanes)
                for (int r = rl; r < rows; r += rowLanes) {
                    const int row = s + r;
                    if (row >= ud.ownBegin && row < ud.ownBegin + ud.ownCount) {
                        const double x = (double)prev[(r + hp) * W + col];
                        s1[o] += x;
                        s2[o] += x * x;
                    }
                }
        }
        __syncthreads();
    }
    if (stats) {
        double *out = a.partial + (long long)blk * 2 * a.step;
#pragma unroll
        for (int o = 0; o < ORD; ++o) {
            red[tid] = s1[o];
            red[kBlock + tid] = s2[o];
            __syncthreads();
            if (tid < ncol) {
                double t1 = 0, t2 = 0;
                for (int q = 0; q < rowLanes; ++q) {
                    t1 += red[q * ncol + tid];
                    t2 += red[kBlock + q * ncol + tid];
                }
                out[o * W + tid] = t1;
                out[a.step + o * W + tid] = t2;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// z-norm: per-chunk column sums (double), per-utterance finalise, normalise
// ------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(kBlock) void k_colstats(const StatArgs<T> a)
{
    __shared__ double sh[2 * kBlock];
    const int chunk = blockIdx.x;
    const int u = a.chunkUtt ? a.chunkUtt[chunk] : 0; // no table: one utterance owns every chunk
    const UttDesc ud = a.utts[u];
    const int r0 = (chunk - ud.statBegin) * kStatRows;
    const int nr = min(kStatRows, ud.ownCount - r0);
    const T *base = a.data + (ud.outRow + ud.ownBegin + r0) * a.step;
    for (int c0 = 0; c0 < a.step; c0 += kBlock) {
        const int ncol = min(kBlock, a.step - c0);
        const int lanesPerRow = ncol;
        const int rowLanes = kBlock / lanesPerRow;
        const int col = threadIdx.x % lanesPerRow, rl = threadIdx.x / lanesPerRow;
        double s = 0, s2 = 0;
        if (rl < rowLanes)
            for (int r = rl; r < nr; r += rowLanes) {
                const double x = (double)base[(long long)r * a.step + c0 + col];
                s += x;
                s2 += x * x;
            }
        sh[threadIdx.x] = s;
        sh[kBlock + threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < ncol) {
            double t = 0, t2 = 0;
            for (int q = 0; q < rowLanes; ++q) {
                t += sh[q * lanesPerRow + threadIdx.x];
                t2 += sh[kBlock + q * lanesPerRow + threadIdx.x];
            }
            a.partial[(long long)chunk * 2 * a.step + c0 + threadIdx.x] = t;
            a.partial[(long long)chunk * 2 * a.step + a.step + c0 + threadIdx.x] = t2;
        }
        __syncthreads();
    }
}

// Sum of the partial rows [first, first + count) of a 2*step-column table, for columns
// [c0, c0 + ncol): the workgroup's lanes form kBlock / ncol row groups, group g adds rows
// g, g + groups, ... in order, and the group sums are added in group order, so the result does not
// depend on timing.  sh holds kBlock doubles.  Returns the total in the lanes tid < ncol.
__device__ __forceinline__ double partial_column_sum(const double *partial, long long first, int count, int width,
                                                      int c0, int ncol, double *sh)
{
    const int tid = threadIdx.x, groups = kBlock / ncol;
    const int col = tid % ncol, g = tid / ncol;
    double s = 0;
    if (g < groups)
        for (int q = g; q < count; q += groups) s += partial[(first + q) * width + c0 + col];
    __syncthreads();
    sh[tid] = s;
    __syncthreads();
    double t = 0;
    if (tid < ncol)
        for (int q = 0; q < groups; ++q) t += sh[q * ncol + tid];
    return t;
}

constexpr int kFinalCols = 16; // columns per workgroup of the finalise kernels

// One workgroup per utterance and group of kFinalCols columns: chunk partials -> mean and 1/std.
// byBlock: the partials are per regression block (k_delta) instead of per statistics chunk.
__global__ __launch_bounds__(kBlock) void k_colfinal(const UttDesc *utts, const double *partial,
                                                     double *stats, int step, int byBlock)
{
    __shared__ double sh[kBlock];
    const int u = blockIdx.x;
    UttDesc ud = utts[u];
    int nchunk = (ud.ownCount + kStatRows - 1) / kStatRows;
    if (byBlock) {
        nchunk = (ud.numFrames + byBlock - 1) / byBlock; // (byBlock: rows per block)
        ud.statBegin = ud.regBegin;
    }
    // blockIdx.y owns kFinalCols columns: kBlock / kFinalCols row groups
    {
        const int c0 = blockIdx.y * kFinalCols;
        const int ncol = min(kFinalCols, step - c0);
        const double s = partial_column_sum(partial, ud.statBegin, nchunk, 2 * step, c0, ncol, sh);
        const double s2 = partial_column_sum(partial, ud.statBegin, nchunk, 2 * step, step + c0, ncol, sh);
        if ((int)threadIdx.x < ncol) {
            const double n = (double)ud.ownCount;
            const double mean = s / n;
            double var = s2 / n - mean * mean;
            if (var < 0) var = 0;
            stats[(long long)u * 2 * step + c0 + threadIdx.x] = mean;
            stats[(long long)u * 2 * step + step + c0 + threadIdx.x] = 1.0 / sqrt(var);
        }
    }
}

// Shard plans: one utterance; raw sums out (sum x, sum x^2 per column, then the row count).
__global__ __launch_bounds__(kBlock) void k_shardfinal(const UttDesc *utts, const double *partial,
                                                       double *sums, int step, int byBlock)
{
    __shared__ double sh[kBlock];
    UttDesc ud = utts[0];
    int nchunk = (ud.ownCount + kStatRows - 1) / kStatRows;
    if (byBlock) {
        nchunk = (ud.numFrames + byBlock - 1) / byBlock; // (byBlock: rows per block)
        ud.statBegin = ud.regBegin;
    }
    {
        const int c0 = blockIdx.y * kFinalCols;
        const int ncol = min(kFinalCols, 2 * step - c0);
        const double s = partial_column_sum(partial, ud.statBegin, nchunk, 2 * step, c0, ncol, sh);
        if ((int)threadIdx.x < ncol) sums[c0 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) sums[2 * step] = (double)ud.ownCount;
}

// SHARD: sums holds all-reduced {sum x, sum x^2 per column, count}; otherwise stats holds the
// per-utterance {mean, 1/std}.  Dynamic LDS: 2*step doubles (mean, 1/std of this chunk's utterance).
template <typename T, bool SHARD>
__global__ __launch_bounds__(kBlock) void k_normalize(const StatArgs<T> a, const double *sums)
{
    extern __shared__ __attribute__((aligned(16))) double shn[];
    const int chunk = blockIdx.x;
    const int u = a.chunkUtt[chunk];
    const UttDesc ud = a.utts[u];
    const int r0 = (chunk - ud.statBegin) * kStatRows;
    const int nr = min(kStatRows, ud.ownCount - r0);
    T *base = a.data + (ud.outRow + ud.ownBegin + r0) * a.step;
    for (int c = threadIdx.x; c < a.step; c += kBlock) {
        if constexpr (SHARD) {
            const double n = sums[2 * a.step];
            const double mean = sums[c] / n;
            double var = sums[a.step + c] / n - mean * mean;
            if (var < 0) var = 0;
            shn[c] = mean;
            shn[a.step + c] = 1.0 / sqrt(var);
        } else {
            shn[c] = a.stats[(long long)u * 2 * a.step + c];
            shn[a.step + c] = a.stats[(long long)u * 2 * a.step + a.step + c];
        }
    }
    __syncthreads();
    for (int c0 = 0; c0 < a.step; c0 += kBlock) {
        const int ncol = min(kBlock, a.step - c0);
        const int rowLanes = kBlock / ncol;
        const int col = threadIdx.x % ncol, rl = threadIdx.x / ncol;
        if (rl >= rowLanes) continue;
        const double mean = shn[c0 + col], rstd = shn[a.step + c0 + col];
        for (int r = rl; r < nr; r += rowLanes) {
            T *p = base + (long long)r * a.step + c0 + col;
            *p = (T)(((double)*p - mean) * rstd);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Integer channel sums for the zero-mean option: 2*sum of {L, R, mid, side}
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int gmem_sample(const unsigned char *p, int bps)
{
    if (bps == 1) return (int)p[0] - 128;
    if (bps == 2) return (int)(short)((unsigned)p[0] | ((unsigned)p[1] << 8));
    if (bps == 3) {
        const unsigned v = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
        return ((int)(v << 8)) >> 8;
    }
    return (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) |
                 ((unsigned)p[3] << 24));
}

__device__ __forceinline__ void block_add4(long long sL, long long sR, unsigned long long *dst)
{
    __shared__ long long shL[kBlock / 64], shR[kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sL += __shfl_xor(sL, o, 64);
        sR += __shfl_xor(sR, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        shL[threadIdx.x >> 6] = sL;
        shR[threadIdx.x >> 6] = sR;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long L = 0, R = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            L += shL[w];
            R += shR[w];
        }
        atomicAdd(dst + 0, (unsigned long long)(2 * L));
        atomicAdd(dst + 1, (unsigned long long)(2 * R));
        atomicAdd(dst + 2, (unsigned long long)(L + R));
        atomicAdd(dst + 3, (unsigned long long)(2 * (L - R)));
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_chansums(const unsigned char *pcm, int channels, int bps,
                                                     long long firstByte, long long count,
                                                     unsigned long long *sums2)
{
    const int bpi = channels * bps;
    long long sL = 0, sR = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (long long)gridDim.x * kBlock) {
        const unsigned char *p = pcm + firstByte + i * bpi;
        sL += gmem_sample(p, bps);
        if (channels == 2) sR += gmem_sample(p + bps, bps);
    }
    block_add4(sL, sR, sums2);
}

// grid = (blocksPerUtt, numUtts): integer sums of one utterance each.
__global__ __launch_bounds__(kBlock) void k_uttsums(const unsigned char *pcm, const UttDesc *utts,
                                                    int channels, int bps,
                                                    unsigned long long *sums2)
{
    const UttDesc ud = utts[blockIdx.x]; // (the utterance in x: y is limited to 65535 workgroups)
    const int bpi = channels * bps;
    long long sL = 0, sR = 0;
    for (long long i = (long long)blockIdx.y * kBlock + threadIdx.x; i < ud.numInstants;
         i += (long long)gridDim.y * kBlock) {
        const unsigned char *p = pcm + ud.byteOff + i * bpi;
        sL += gmem_sample(p, bps);
        if (channels == 2) sR += gmem_sample(p + bps, bps);
    }
    block_add4(sL, sR, sums2 + 4 * blockIdx.x);
}

__global__ void k_uttmeans(UttDesc *utts, int numUtts, const unsigned long long *sums2)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= numUtts) return;
    const double n2 = 2.0 * (double)utts[u].numInstants;
    for (int r = 0; r < 4; ++r)
        utts[u].mean[r] = n2 > 0 ? (double)(long long)sums2[4 * u + r] / n2 : 0.0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

#define MFCC_HIP_OK(expr)                     \
    do {                                      \
        hipError_t e_ = (expr);               \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

template <typename T, int L, int PL, bool TLDS, int MODE>
static int launch_front_t(const FrontArgs<T> &a, int numBlocks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    MFCC_HIP_OK(lds_limit_once(reinterpret_cast<const void *>(&k_front<T, L, PL, TLDS, MODE>), attrSet));
    hipLaunchKernelGGL((k_front<T, L, PL, TLDS, MODE>), dim3(numBlocks), dim3(kBlock), lds, st, a);
    return (int)hipGetLastError();
}

template <typename T, int L, int PL, bool TLDS, int MODE>
static int front_occ_t(size_t lds)
{
    int n = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_front<T, L, PL, TLDS, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_front<T, L, PL, TLDS, MODE>, kBlock, lds) != hipSuccess)
        return 0;
    return n;
}

// The lean instantiation exists for single precision, 8 points per lane, tables in LDS.
template <typename T, int PL>
constexpr bool front_has_lean()
{
    return sizeof(T) == 4 && (PL == 3 || PL == 4);
}

// Dispatch over (precision, log2 nc, points per lane, tables in LDS, mode).  op 0: launch,
// op 1: occupancy.  mode 1 (lean) exists for single precision with 8 or 16 points per lane and
// tables in LDS; mode 2 (lean + matrix-core mel) for 16 points per lane and 256..1024 complex points.
template <typename T, int L, int PL>
static int front_do(int op, bool tl, int mode, const void *args, int numBlocks, size_t lds, hipStream_t st)
{
    if constexpr (front_has_lean<T, PL>()) {
        if constexpr (PL == 4 && L >= 8 && L <= 10) {
            if (mode == 2 && tl) {
                if (op == 1) return front_occ_t<T, L, PL, true, 2>(lds);
                return launch_front_t<T, L, PL, true, 2>(*static_cast<const FrontArgs<T> *>(args), numBlocks, lds, st);
            }
        }
        if (mode >= 1 && tl) {
            if (op == 1) return front_occ_t<T, L, PL, true, 1>(lds);
            return launch_front_t<T, L, PL, true, 1>(*static_cast<const FrontArgs<T> *>(args), numBlocks, lds, st);
        }
    }
    if (op == 1) return tl ? front_occ_t<T, L, PL, true, 0>(lds) : front_occ_t<T, L, PL, false, 0>(lds);
    const FrontArgs<T> &a = *static_cast<const FrontArgs<T> *>(args);
    return tl ? launch_front_t<T, L, PL, true, 0>(a, numBlocks, lds, st)
              : launch_front_t<T, L, PL, false, 0>(a, numBlocks, lds, st);
}

template <typename T>
static int front_dispatch(int op, int log2nc, int pplog, bool tl, int mode, const void *args, int numBlocks,
                          size_t lds, hipStream_t st)
{
    switch (log2nc * 8 + pplog) {
#define MFCC_CASE(L, PL) \
    case L * 8 + PL: return front_do<T, L, PL>(op, tl, mode, args, numBlocks, lds, st);
#ifdef MFCC_DEV_CASES // ISA listings of the benchmark instantiations only (tools/isa_blocks.py); never in the library
        MFCC_CASE(8, 4) MFCC_CASE(10, 3) MFCC_CASE(10, 4)
#else
        MFCC_CASE(1, 4) MFCC_CASE(2, 4) MFCC_CASE(3, 4) MFCC_CASE(4, 4) MFCC_CASE(5, 4)
        MFCC_CASE(6, 4) MFCC_CASE(7, 4) MFCC_CASE(8, 4) MFCC_CASE(9, 4) MFCC_CASE(10, 4)
        MFCC_CASE(11, 4) MFCC_CASE(12, 4) MFCC_CASE(13, 4)
        MFCC_CASE(6, 3) MFCC_CASE(7, 3) MFCC_CASE(8, 3) MFCC_CASE(9, 3) MFCC_CASE(10, 3) MFCC_CASE(11, 3)
        MFCC_CASE(6, 2) MFCC_CASE(7, 2) MFCC_CASE(8, 2) MFCC_CASE(9, 2) MFCC_CASE(10, 2)
#endif
#undef MFCC_CASE
    default: return -1;
    }
}

bool front_variant_exists(int log2nc, int pplog)
{
    if (pplog == 4) return log2nc >= 1 && log2nc <= kMaxLog2Nc;
    if (pplog == 3) return log2nc >= 6 && log2nc <= 11;
    if (pplog == 2) return log2nc >= 6 && log2nc <= 10;
    return false;
}

int launch_front(int precision, int log2nc, int pplog, const void *args, int numBlocks, size_t ldsBytes,
                 void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (numBlocks <= 0) return 0;
    const bool tl = precision == 1 ? static_cast<const FrontArgs<double> *>(args)->tabElems > 0
                                   : static_cast<const FrontArgs<float> *>(args)->tabElems > 0;
    const int mode = precision == 1 ? 0 : static_cast<const FrontArgs<float> *>(args)->lean;
    return precision == 1 ? front_dispatch<double>(0, log2nc, pplog, tl, 0, args, numBlocks, ldsBytes, st)
                          : front_dispatch<float>(0, log2nc, pplog, tl, mode, args, numBlocks, ldsBytes, st);
}

// Workgroups of the static-feature kernel that the runtime says fit one CU (0 = unknown).
int front_occupancy(int precision, int log2nc, int pplog, bool tabLds, int mode, size_t ldsBytes)
{
    const int n = precision == 1 ? front_dispatch<double>(1, log2nc, pplog, tabLds, 0, nullptr, 0, ldsBytes, nullptr)
                                 : front_dispatch<float>(1, log2nc, pplog, tabLds, mode, nullptr, 0, ldsBytes, nullptr);
    return n < 0 ? 0 : n;
}

int front_lanes(int log2nc, int pplog)
{
    const int nc = 1 << log2nc;
    const int ppl0 = 1 << (pplog < log2nc ? pplog : log2nc);
    return nc / ppl0 > kBlock ? kBlock : nc / ppl0;
}

int front_groups(int log2nc, int pplog) { return kBlock / front_lanes(log2nc, pplog); }

int front_teams(int log2nc, int pplog) { return front_lanes(log2nc, pplog) <= 64 ? kBlock / 64 : 1; }

int front_waves_limit(int precision, int log2nc, int pplog)
{
    return front_waves_rule(precision == 1, log2nc, pplog);
}

int launch_regress(int precision, const void *args, int numBlocks, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (numBlocks <= 0) return 0;
    const int numTiles = numBlocks;
    if (precision == 1)
        hipLaunchKernelGGL(k_regress<double>, dim3(numTiles), dim3(kBlock), 0, st,
                           *static_cast<const RegressArgs<double> *>(args));
    else
        hipLaunchKernelGGL(k_regress<float>, dim3(numTiles), dim3(kBlock), 0, st,
                           *static_cast<const RegressArgs<float> *>(args));
    return (int)hipGetLastError();
}

int launch_colstats(int precision, const void *args, int numChunks, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (numChunks <= 0) return 0;
    if (precision == 1)
        hipLaunchKernelGGL(k_colstats<double>, dim3(numChunks), dim3(kBlock), 0, st,
                           *static_cast<const StatArgs<double> *>(args));
    else
        hipLaunchKernelGGL(k_colstats<float>, dim3(numChunks), dim3(kBlock), 0, st,
                           *static_cast<const StatArgs<float> *>(args));
    return (int)hipGetLastError();
}

int launch_colfinal(const StatArgs<float> &a, int numUtts, int blockRows, void *stream)
{
    if (numUtts <= 0) return 0;
    hipLaunchKernelGGL(k_colfinal, dim3(numUtts, (a.step + kFinalCols - 1) / kFinalCols), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       a.utts, a.partial, a.stats, a.step, blockRows);
    return (int)hipGetLastError();
}

// blockRows: 0 = the partials are per statistics chunk (kStatRows rows), else per regression block of
// that many rows (the plan's regRows)
int launch_shardfinal(const StatArgs<float> &a, int numChunks, int blockRows, double *sumsOut, void *stream)
{
    (void)numChunks;
    hipLaunchKernelGGL(k_shardfinal, dim3(1, (2 * a.step + kFinalCols - 1) / kFinalCols), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       a.utts, a.partial, sumsOut, a.step, blockRows);
    return (int)hipGetLastError();
}

// Fused delta (+ statistics) pass; returns -2 when this order count has no instantiation.
template <typename T, int ORD>
static int launch_delta_o(const DeltaArgs<T> &a, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    MFCC_HIP_OK(lds_limit_once(reinterpret_cast<const void *>(&k_delta<T, ORD>), attrSet));
    hipLaunchKernelGGL((k_delta<T, ORD>), dim3(a.numBlocks), dim3(kBlock), lds, st, a);
    return (int)hipGetLastError();
}
template <typename T>
static int launch_delta_t(const DeltaArgs<T> &a, int orders, size_t lds, hipStream_t st)
{
    if (a.numBlocks <= 0) return 0;
    if (orders == 2) return launch_delta_o<T, 2>(a, lds, st);
    if (orders == 3) return launch_delta_o<T, 3>(a, lds, st);
    return -2;
}
int launch_delta(int precision, const void *args, int orders, size_t ldsBytes, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    return precision == 1 ? launch_delta_t<double>(*static_cast<const DeltaArgs<double> *>(args), orders, ldsBytes, st)
                          : launch_delta_t<float>(*static_cast<const DeltaArgs<float> *>(args), orders, ldsBytes, st);
}

int launch_normalize(int precision, const void *args, int numChunks, bool shardSums,
                     const double *sums, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (numChunks <= 0) return 0;
    const size_t lds = 2 * sizeof(double) *
                       (size_t)(precision == 1 ? static_cast<const StatArgs<double> *>(args)->step
                                               : static_cast<const StatArgs<float> *>(args)->step);
    if (lds > 48 * 1024) { // rows of more than 3072 columns (validate_config stops at kMaxZnormStep)
        const void *k = precision == 1 ? (shardSums ? (const void *)&k_normalize<double, true> : (const void *)&k_normalize<double, false>)
                                       : (shardSums ? (const void *)&k_normalize<float, true> : (const void *)&k_normalize<float, false>);
        if (hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return (int)e;
    }
    if (precision == 1) {
        const StatArgs<double> &a = *static_cast<const StatArgs<double> *>(args);
        if (shardSums)
            hipLaunchKernelGGL((k_normalize<double, true>), dim3(numChunks), dim3(kBlock), lds, st, a,
                               sums);
        else
            hipLaunchKernelGGL((k_normalize<double, false>), dim3(numChunks), dim3(kBlock), lds, st,
                               a, sums);
    } else {
        const StatArgs<float> &a = *static_cast<const StatArgs<float> *>(args);
        if (shardSums)
            hipLaunchKernelGGL((k_normalize<float, true>), dim3(numChunks), dim3(kBlock), lds, st, a,
                               sums);
        else
            hipLaunchKernelGGL((k_normalize<float, false>), dim3(numChunks), dim3(kBlock), lds, st, a,
                               sums);
    }
    return (int)hipGetLastError();
}

int launch_chansums(const unsigned char *pcm, int channels, int bps, int64_t firstByte,
                    int64_t count, unsigned long long *sums2, void *stream)
{
    if (count <= 0) return 0;
    long long blocks = (count + kBlock * 16 - 1) / (kBlock * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_chansums, dim3((unsigned)blocks), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), pcm, channels, bps, (long long)firstByte,
                       (long long)count, sums2);
    return (int)hipGetLastError();
}

int launch_uttmeans(const unsigned char *pcm, UttDesc *utts, int numUtts, int channels, int bps,
                    unsigned long long *sums2, void *stream)
{
    if (numUtts <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bx = numUtts >= 2048 ? 1 : (numUtts >= 64 ? 8 : 256);
    hipLaunchKernelGGL(k_uttsums, dim3(numUtts, bx), dim3(kBlock), 0, st, pcm, utts, channels, bps,
                       sums2);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    hipLaunchKernelGGL(k_uttmeans, dim3((numUtts + 255) / 256), dim3(256), 0, st, utts, numUtts,
                       sums2);
    return (int)hipGetLastError();
}

} // namespace mfcc
