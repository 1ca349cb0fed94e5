// Static-feature kernel for the most common speech front end: mono 16-bit PCM, one derived row,
// frames of at most 512 samples (256 complex points), log-mel filterbank or cepstra (+ C0, + log
// energy), single precision.  Same arithmetic as k_front's matrix-core mode (Ham main.c:229,
// Wave2FBank mfcc.c:125-177, FBank2MFCC / FBank2C0 mfcc.c:180-208, row layout main.c:241-249),
// organised as a software pipeline so that a wave never waits for memory it could have asked for
// earlier:
//   - every wave owns a CONTIGUOUS range of tiles and derives each tile's geometry from the
//     utterance table with scalar arithmetic (no per-tile descriptor load);
//   - a tile's raw samples are requested one whole tile ahead (16-byte vectors into registers) and
//     expanded into the LDS sample row during the previous tile's last round, right after that
//     round has fetched its own samples;
//   - a round's sample pairs are requested before the mel phase of the round before it;
//   - results go from registers straight to global memory (no staging, no alignment cases).
#include <hip/hip_runtime.h>

#include "front_util.h"

namespace mfcc {

namespace {

constexpr int kLeanNc = 256, kLeanG = 16, kLeanPpl = 16, kLeanGt = 4, kLeanPfn = 3;

typedef unsigned int u4u __attribute__((ext_vector_type(4), aligned(1)));

// One tile as the pipeline sees it (all wave-uniform).
struct LeanTile {
    unsigned long long src; // byte offset of the predecessor of the first sample (or of that sample)
    long long outRow;       // output row of the tile's first frame
    int nf;                 // frames (0: no tile)
    int hasPrev;            // the first sample has a predecessor inside the buffer
    int direct;             // raw samples may be fetched as kLeanPfn 16-byte vectors per lane
};

// Walks the tiles of the batch in order: utterance u, tile j of it.
struct LeanIter {
    int u, j, nt;           // nt: tiles of utterance u
    long long byteOff, outRow;
    int numFrames, lead;
};

__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long sgpr64(long long x)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void lean_load_utt(LeanIter &it, const FrontArgs<float> &a, int u)
{
    const UttDesc *d = a.utts + u;
    it.u = u;
    it.j = 0;
    it.byteOff = sgpr64(d->byteOff);
    it.outRow = sgpr64(d->outRow);
    it.numFrames = sgpr(d->numFrames);
    it.lead = sgpr(d->lead);
    it.nt = (it.numFrames + a.framesPerTile - 1) / a.framesPerTile;
}

__device__ __forceinline__ LeanTile lean_tile(const LeanIter &it, const FrontArgs<float> &a, bool valid)
{
    LeanTile t;
    const int f0 = it.j * a.framesPerTile;
    t.nf = valid ? min(a.framesPerTile, it.numFrames - f0) : 0;
    const long long first = (long long)f0 * a.inc + it.lead; // first instant of the first frame
    t.hasPrev = first > 0 ? 1 : 0;
    t.src = (unsigned long long)(it.byteOff + (first - t.hasPrev) * 2);
    t.outRow = it.outRow + f0;
    const int vecs = ((t.nf - 1) * a.inc + a.wlen + 1 + 7) >> 3; // 8 samples each, predecessor included
    t.direct = (valid && t.hasPrev && vecs <= kLeanPfn * 64 && (long long)t.src + (long long)vecs * 16 <= a.planEnd) ? 1 : 0;
    return t;
}

// Moves to the next tile of the batch (utterances without frames are skipped).  The caller
// bounds the walk by its tile count, so the utterance index stays inside the table.
__device__ __forceinline__ void lean_advance(LeanIter &it, const FrontArgs<float> &a)
{
    if (++it.j < it.nt) return;
    int u = it.u + 1;
    while (u < a.numUtts) {
        lean_load_utt(it, a, u);
        if (it.nt > 0) return;
        ++u;
    }
    it.u = a.numUtts;
    it.j = 0;
    it.nt = 0;
}

} // namespace

// NZ: number of first-stage inputs of a lane that can be non-zero, ceil(wlen / 32): 13 for
// 400-sample frames in 512 points (the zero padding of mfcc.c:144-147 is then never loaded,
// multiplied or added), 16 otherwise.
template <int NZ>
__global__ __launch_bounds__(kBlock, 2) void k_lean256(const FrontArgs<float> a)
{
    constexpr int NC = kLeanNc, G = kLeanG, PPL = kLeanPpl, GT = kLeanGt, TEAM = 64, PFN = kLeanPfn;
    typedef float T;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const int tt = tid & 63, l = tt & (G - 1), grp = tt >> 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ldsTab = reinterpret_cast<T *>(smem);             // cepstral table (dctElems), mel weights
    T *rows = ldsTab + a.tabElems + wave * a.teamElems;  // pre-emphasised samples of the tile
    T *zbuf = rows + a.rowPitch;                         // GT work areas; raw PCM bytes alias it
    T *fbk = zbuf + a.rawOrZElems;                       // framesPerTile x P log-mel values (cepstral mode)
    unsigned *raw = reinterpret_cast<unsigned *>(zbuf);

    for (int i = tid; i < a.dctElems; i += kBlock) ldsTab[i] = a.dctT[i];
    const T *dctT = ldsTab;

    // Per-lane constants: window pairs (half the Hamming values), split factors, second-stage twiddles.
    cpx<T> wreg[NZ];
    cpx<T> swreg[PPL / 2];
    cpx<T> twr[fft_twiddle_regs<8, 4>()];
#pragma unroll
    for (int r = 0; r < NZ; ++r) wreg[r] = *reinterpret_cast<const cpx<T> *>(a.window + 2 * (l + G * r));
#pragma unroll
    for (int q = 0; q < PPL / 2; ++q) swreg[q] = reinterpret_cast<const cpx<T> *>(a.splitTw)[l + G * q];
    fft_load_twiddles<T, 8, 4>(twr, l, reinterpret_cast<const cpx<T> *>(a.tw));
    // mel weights: [step][lane] behind the cepstral table, shared by the workgroup's waves
    T *melWl = ldsTab + a.dctElems;
    for (int i = tid; i < kMelSteps * 64; i += kBlock) melWl[i] = i < a.melSteps * 64 ? a.melW[i] : 0.0f;
    __syncthreads();
    const T *melB = melWl + tt;
    const int4 mm = reinterpret_cast<const int4 *>(a.melMeta)[tt];
    const T *melA = zbuf + (tt & 3) * a.zPitch + mm.x;
    const int melFlags = mm.y, melCh = mm.z;
    const float kf = (float)a.preK;
    const int JW = a.M + a.c0Flag;

    // The wave's tile range.
    const int teamId = blockIdx.x * (kBlock / TEAM) + wave;
    const int t0 = teamId * a.tilesPerTeam;
    int count = min(a.numTiles, t0 + a.tilesPerTeam) - t0;
    if (count <= 0) return;
    LeanIter it;
    {
        // largest u with tileBegin[u] <= t0 (tileBegin is the running tile count, so among equal
        // values only the last utterance has tiles)
        int lo = 0, hi = a.numUtts - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sgpr(a.utts[mid].tileBegin) <= t0) lo = mid;
            else hi = mid - 1;
        }
        lean_load_utt(it, a, lo);
        it.j = t0 - sgpr(a.utts[lo].tileBegin);
    }
    LeanTile cur = lean_tile(it, a, true);
    lean_advance(it, a);
    LeanTile nx = lean_tile(it, a, count > 1);
    if (count > 1) lean_advance(it, a);
    LeanTile nn = lean_tile(it, a, count > 2);

    u4u pf[PFN];
#pragma unroll
    for (int j = 0; j < PFN; ++j) pf[j] = u4u{0u, 0u, 0u, 0u};

    // Requests the raw samples of tile t (vector j of lane tt = raw samples 8 (tt + 64 j) ..
    // + 7, sample 0 being the predecessor of the tile's first sample).
    auto fetch = [&](const LeanTile &t) {
        const int nvec = ((t.nf - 1) * a.inc + a.wlen + 1 + 7) >> 3;
        const u4u *g = reinterpret_cast<const u4u *>(a.pcm + t.src);
#pragma unroll
        for (int j = 0; j < PFN; ++j)
            if (tt + j * TEAM < nvec) pf[j] = g[tt + j * TEAM];
    };
    // Fetched vectors -> pre-emphasised row: row[s] = raw[s + 1] - k raw[s] (PreEmphasise, call
    // site main.c:208), the vector's last output taking raw[8 j + 8] from the next lane.  Values
    // behind the tile's last sample are finite and zeroed by the tail fill below.
    auto expand = [&](const LeanTile &t) {
        const int nvec = ((t.nf - 1) * a.inc + a.wlen + 1 + 7) >> 3;
#pragma unroll
        for (int j = 0; j < PFN; ++j) {
            const int vj = tt + j * TEAM;
            unsigned nxw = (unsigned)__shfl_down((int)pf[j].x, 1, 64);
            const unsigned wrap = j + 1 < PFN ? (unsigned)__shfl((int)pf[j + 1 < PFN ? j + 1 : j].x, 0, 64) : 0u;
            nxw = tt == 63 ? wrap : nxw;
            if (vj < nvec) {
                const unsigned d[5] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w, nxw};
                float sm[9];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    sm[i] = (float)(short)((i & 1) ? (d[i >> 1] >> 16) : (d[i >> 1] & 0xffffu));
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = sm[i + 1] - kf * sm[i];
                float *dstRow = rows + 8 * vj;
                *reinterpret_cast<float4 *>(dstRow) = make_float4(y[0], y[1], y[2], y[3]);
                *reinterpret_cast<float4 *>(dstRow + 4) = make_float4(y[4], y[5], y[6], y[7]);
            }
        }
    };
    // Tiles that cannot take the vector path (an utterance's first tile, the last bytes of the
    // buffer) go through a copy of their bytes in LDS.  Rare, and synchronous.  First tiles:
    // aligned 16-byte loads, samples picked at run-time byte offsets, double arithmetic.
    auto expand_general = [&](const LeanTile &t) {
        const int ns = (t.nf - 1) * a.inc + a.wlen;
        const double mean[4] = {0, 0, 0, 0};
        if (t.hasPrev) {
            // the last bytes of the buffer: exactly the tile's bytes (16-byte vectors from the
            // possibly unaligned start, the rest byte by byte), then the arithmetic of the vector path
            const unsigned char *g0 = a.pcm + t.src;
            const int nbytes = (ns + 1) * 2, nfull = nbytes >> 4;
            for (int i = tt; i < nfull; i += TEAM) {
                const u4u v4 = reinterpret_cast<const u4u *>(g0)[i];
                reinterpret_cast<uint4 *>(raw)[i] = make_uint4(v4.x, v4.y, v4.z, v4.w);
            }
            for (int i = (nfull << 4) + tt; i < nbytes; i += TEAM) reinterpret_cast<unsigned char *>(raw)[i] = g0[i];
            team_sync<TEAM>();
            unpack_tile_fast<T, false, 1, 2, TEAM>(rows, raw, a.rowPitch, 1, ns, mean, a.preK, tt);
            team_sync<TEAM>();
            return;
        }
        const unsigned long long gaddr = reinterpret_cast<unsigned long long>(a.pcm + t.src);
        const int gmis = (int)(gaddr & 15ull);
        const uint4 *src = reinterpret_cast<const uint4 *>(gaddr - gmis);
        const int nvec = (gmis + (ns + t.hasPrev) * 2 + 15) >> 4;
        for (int i = tt; i < nvec; i += TEAM) reinterpret_cast<uint4 *>(raw)[i] = src[i];
        team_sync<TEAM>();
        unpack_tile<T, true, 1, TEAM>(rows, raw, a.rowPitch, 1, 2, gmis, t.hasPrev, ns, mean, a.preK, tt);
        team_sync<TEAM>();
    };
    auto tail_fill = [&](const LeanTile &t) {
        const int ns = (t.nf - 1) * a.inc + a.wlen;
        for (int s = tt; s < a.rowTail; s += TEAM) rows[ns + s] = (T)0;
    };
    // Sample pairs (2n, 2n+1), n = l + 16 r, of frame `fr` of the tile in the row.
    cpx<T> xpre[NZ];
    auto request = [&](int fr) {
        const cpx<T> *xp = reinterpret_cast<const cpx<T> *>(rows + fr * a.inc + 2 * l);
#pragma unroll
        for (int r = 0; r < NZ; ++r) xpre[r] = xp[r * (NC / 16)];
    };

    // Prologue: the first tile's row, the second tile's fetch, the first round's samples.
    if (cur.direct) {
        fetch(cur);
        expand(cur);
    } else {
        expand_general(cur);
    }
    tail_fill(cur);
    if (nx.direct) fetch(nx);
    team_sync<TEAM>();
    request(grp < cur.nf ? grp : 0);

    // Results of a round are written one round later, behind the point where the next tile's fetched
    // vectors are expanded: that point waits for every older memory operation, and the newest store
    // it can then meet is a whole round old.
    float dlv[GT] = {0, 0, 0, 0}, dte = 0;
    long long dOff = 0; // element offset of the pending round's first row
    int dCount = 0;     // its frames (0: nothing pending)
    auto flush = [&]() {
        if (dCount > 0) {
            float *ob = a.out + dOff;
            if (a.fbankFlag && (melFlags & 2) && melCh >= 0) {
#pragma unroll
                for (int r = 0; r < GT; ++r)
                    if (r < dCount) ob[(long long)r * a.step + melCh] = dlv[r];
            }
            if (a.energyFlag && l == 0 && grp < dCount) ob[(long long)grp * a.step + JW] = dte; // main.c:243
            dCount = 0;
        }
    };

    for (;;) {
        const int nf = cur.nf;
        float *outBase = a.out + cur.outRow * a.step;
        for (int base = 0; base < nf; base += GT) {
            const int item = base + grp;
            const bool active = item < nf;
            const bool last = base + GT >= nf;
            T *Zt = zbuf + grp * a.zPitch;
            cpx<T> *Z = reinterpret_cast<cpx<T> *>(Zt);

            // ---- window (Ham main.c:229), frame energy (te, mfcc.c:138-142)
            cpx<T> v[PPL];
#pragma unroll
            for (int r = 0; r < NZ; ++r) v[r] = xpre[r] * wreg[r];
#pragma unroll
            for (int r = NZ; r < PPL; ++r) v[r] = cpx<T>{0.0f, 0.0f};
            T te = 0;
            if (a.energyFlag) {
                cpx<T> e2 = {0, 0};
#pragma unroll
                for (int r = 0; r < NZ; ++r) e2 += v[r] * v[r];
                te = group_sum<T, G>((T)4 * (e2.x + e2.y), nullptr, tt); // (half-scaled window)
            }

            // ---- the row is free once every lane holds its samples: next tile's row
            if (last && nx.nf > 0) {
                team_sync<TEAM>();
                if (nx.direct) expand(nx);
                else expand_general(nx);
                tail_fill(nx);
                if (nn.direct) fetch(nn);
            }
            flush();

            // ---- 256-point complex FFT of z[n] = x[2n] + i x[2n+1] (Realft mfcc.c:262-270), last
            //      stage in registers: v[r] = Z[l + 16 r]
            fft_stages<T, 8, 4, 0, 0, false, true, 0, true>(v, Z, l, nullptr, twr);

            // ---- real-FFT split and power (mfcc.c:274-290, ek of mfcc.c:153-155): Z[256 - k] for
            //      k = l + 16 q is register 15 - q of lane 16 - l (own register 16 - q in lane 0)
            T pa[PPL / 2], pb[PPL / 2];
#pragma unroll
            for (int q = 0; q < PPL / 2; ++q) {
                const cpx<T> hi = v[PPL - 1 - q], own = v[(PPL - q) & (PPL - 1)];
                const float px = row_negated_lane(hi.x), py = row_negated_lane(hi.y);
                const cpx<T> zb = {l == 0 ? own.x : px, l == 0 ? own.y : py};
                split_power<T>(v[q], zb, swreg[q], pa[q], pb[q]);
            }
            {
                // bin 0 packs Re + Im of Z[0] (mfcc.c:288-290); bin 128 is |Z[128]|^2; lane 0 holds both
                const T p0 = (T)4 * (v[0].x + v[0].y) * (v[0].x + v[0].y);
                const T ph = (T)4 * (v[PPL / 2].x * v[PPL / 2].x + v[PPL / 2].y * v[PPL / 2].y);
                pa[0] = l == 0 ? p0 : pa[0];
                pb[0] = l == 0 ? ph : pb[0];
            }
            // power values of the frame into its work area, bin order
#pragma unroll
            for (int q = 0; q < PPL / 2; ++q) {
                const int k = l + G * q;
                const int k2 = k == 0 ? NC / 2 : NC - k;
                Zt[k] = pa[q];
                Zt[k2] = pb[q];
            }
            team_sync<TEAM>();

            // ---- mel filterbank (mfcc.c:152-163) as 4 x 4 x 1 matrix-core products: block b = 4
            //      lanes owns 4 consecutive channels and a run of bins; lane (b, i) feeds the power
            //      of frame slot i at bin r0_b + t and the weight of that bin for channel 4 q_b + i
            // (loads of 8 steps run two groups ahead of the products that use them)
            static_assert(kMelSteps == 32, "four groups of 8 steps");
            float av[4][8], bw[4][8];
            Acc4<float> macc = {0, 0, 0, 0}, macc2 = {0, 0, 0, 0};
            auto mel_load = [&](int c) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    av[c][t] = melA[8 * c + t];
                    bw[c][t] = melB[(8 * c + t) * 64];
                }
            };
            auto mel_mac = [&](int c) {
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    macc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][t], bw[c][t], macc, 0, 0, 0);
                    macc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][t + 1], bw[c][t + 1], macc2, 0, 0, 0);
                }
            };
            mel_load(0);
            mel_load(1);
            __builtin_amdgcn_sched_barrier(0);
            mel_mac(0);
            mel_load(2);
            __builtin_amdgcn_sched_barrier(0);
            mel_mac(1);
            mel_load(3);
            // the next round's sample pairs (its row is ready: same tile, or expanded above)
            if (!last) request(base + GT + grp < nf ? base + GT + grp : 0);
            else if (nx.nf > 0) request(grp < nx.nf ? grp : 0);
            __builtin_amdgcn_sched_barrier(0);
            mel_mac(2);
            mel_mac(3);
            macc += macc2;
            {
                // a channel quad cut over up to 4 neighbouring blocks: the chain's last block adds the others
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
                    if (it2 < nf) {
                        const T lv = __logf(macc[r] < 1.0f ? 1.0f : macc[r]);
                        dlv[r] = lv;
                        if (!a.fbankFlag) fbk[it2 * a.P + melCh] = lv;
                    }
                }
            }
            if (a.energyFlag) dte = __logf(te);
            dOff = (cur.outRow + base) * (long long)a.step;
            dCount = min(GT, nf - base);
            team_sync<TEAM>(); // the work areas are free for the next round
        }
        if (!a.fbankFlag) {
            // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208) of the tile
            //      as 16 x 16 x 4 matrix-core products: lane (i, g) = (lane % 16, lane / 16) feeds
            //      fbk[item i][k0 + g] and dct[coefficient i][k0 + g] and holds the results of items
            //      4 g .. 4 g + 3 for coefficient i.  The table's row M is all ones (C0).
            const int i16 = tt & 15, g4 = tt >> 4;
            for (int ib = 0; ib * 16 < nf; ++ib)
                for (int jb = 0; jb * 16 < JW; ++jb) {
                    const int itemA = ib * 16 + i16, jB = jb * 16 + i16;
                    const T *fa = fbk + (itemA < nf ? itemA : 0) * a.P;
                    const T *db = dctT + (jB < JW ? jB : 0) * a.dctPitch;
                    Acc4<T> acc = {0, 0, 0, 0};
                    for (int k0 = 0; k0 < a.P; k0 += 4) {
                        const int k = k0 + g4;
                        T fv = fa[k < a.P ? k : 0], bv = db[k];
                        fv = (itemA < nf && k < a.P) ? fv : (T)0;
                        bv = jB < JW ? bv : (T)0;
                        acc = mfma16(fv, bv, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int item = ib * 16 + 4 * g4 + r;
                        if (item < nf && jB < JW) outBase[(long long)item * a.step + jB] = acc[r] * a.mfnorm;
                    }
                }
            team_sync<TEAM>();
        }
        if (--count <= 0) break;
        cur = nx;
        nx = nn;
        if (count > 2) lean_advance(it, a);
        nn = lean_tile(it, a, count > 2);
    }
    flush();
}

bool lean256_applies(int log2nc, int mode, int framesPerTile, int inc, int wlen)
{
    // the fetch of a full tile must fit kLeanPfn vectors per lane
    const long vecs = ((long)(framesPerTile - 1) * inc + wlen + 1 + 7) >> 3;
    return log2nc == 8 && mode == 2 && vecs <= kLeanPfn * 64;
}

int launch_lean256(const FrontArgs<float> &a, int numBlocks, size_t lds, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attrSet[16][2] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const bool nz13 = a.wlen <= 2 * 16 * 13;
    if (dev < 16 && !attrSet[dev][nz13]) {
        const void *fn = nz13 ? reinterpret_cast<const void *>(&k_lean256<13>) : reinterpret_cast<const void *>(&k_lean256<16>);
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attrSet[dev][nz13] = true;
    }
    if (nz13) hipLaunchKernelGGL((k_lean256<13>), dim3(numBlocks), dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((k_lean256<16>), dim3(numBlocks), dim3(kBlock), lds, st, a);
    return (int)hipGetLastError();
}

} // namespace mfcc
