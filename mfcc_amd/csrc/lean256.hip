// Static-feature kernel for the most common speech front end: mono 16-bit PCM, one derived row,
// frames of at most 512 samples (256 complex points), log-mel filterbank or cepstra (+ C0, + log
// energy), single precision.  Same arithmetic as k_front's matrix-core mode (Ham main.c:229,
// Wave2FBank mfcc.c:125-177, FBank2MFCC / FBank2C0 mfcc.c:180-208, row layout main.c:241-249),
// organised for occupancy and as a software pipeline:
//   - a wave is a team of its own (4 frames in flight, 16 lanes x 16 points each) and a workgroup
//     is up to 12 of them, three per SIMD, so that the vector units stay busy while other waves
//     wait for LDS round trips.  To fit that many, the per-lane tables a wave re-reads every round
//     (window, mel weights) sit in LDS, shared by the workgroup, and only twiddles and split
//     factors stay in registers;
//   - a tile is one round (4 consecutive frames of one utterance); every wave owns a CONTIGUOUS
//     range of tiles and reads their host-resolved 16-byte descriptors through the scalar data
//     path, three tiles ahead, so the tile loop carries no index arithmetic;
//   - a tile's raw samples are requested one whole round ahead (16-byte vectors into registers)
//     and expanded into the LDS sample row right after the round before it has read its own
//     samples; tiles at the first or last bytes of the buffer fetch sample by sample instead
//     (the only special case: the arithmetic is the same for every tile);
//   - results go from registers straight to global memory, one round late (see flush below).
#include <hip/hip_runtime.h>

#include "front_util.h"
#include "stamps.h"

namespace mfcc {

namespace {

constexpr int kLeanNc = 256, kLeanG = 16, kLeanPpl = 16, kLeanGt = 4, kLeanPfn = 2;

typedef unsigned int u4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned int u4s __attribute__((ext_vector_type(4)));

// One tile as the pipeline sees it (all wave-uniform): the two 64-bit words of its TileDesc.
//   w0, w1 & 0xffff: byte offset of the tile's first sample inside the PCM buffer
//   w1 bits 16..19:  frames (0: no tile)
//   w1 bit 28:       the tile starts the signal (first-sample rule of PreEmphasise)
//   w1 bit 29:       a 16-byte vector of its fetch would leave the planned PCM bytes
struct LeanTile {
    unsigned w0, w1;
    int outRow;
    __device__ __forceinline__ int nf() const { return (int)((w1 >> 16) & 15u); }
    __device__ __forceinline__ bool first() const { return (w1 >> 28) & 1u; }
    __device__ __forceinline__ bool edge() const { return (w1 >> 29) & 1u; }
    __device__ __forceinline__ long long src() const { return (long long)(((unsigned long long)(w1 & 0xffffu) << 32) | w0); }
};

__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Descriptor `idx` of the wave's range through the scalar data path (the table is read-only for
// the whole launch and the index is wave-uniform).
__device__ __forceinline__ LeanTile lean_desc(const TileDesc *tab, int idx, bool valid)
{
    LeanTile t = {0u, 0u, 0};
    if (valid) {
        typedef const __attribute__((address_space(4))) u4s *cptr;
        const u4s d = reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(tab))[idx];
        t.w0 = d.x;
        t.w1 = d.y;
        t.outRow = (int)d.z;
    }
    return t;
}

// Work areas are 48 elements apart modulo 64 (slots at banks 0, 48, 32, 16 of the 64 that 8-byte
// reads see), which keeps the mel phase's 8 blocks x 4 slots out of each other's banks (the host
// places the blocks' first bins for that); the frame of lane group g uses area slot_of(g) =
// {0, 2, 1, 3}, so that the two groups of a 32-lane pass sit 32 banks apart and their 128-byte
// runs never meet.  Result r of the mel products therefore belongs to the tile's frame slot_of(r).
__device__ __forceinline__ constexpr int slot_of(int g) { return ((g & 1) << 1) | (g >> 1); }

// lane i takes x of lane i + 1; lane 63 takes `last`
__device__ __forceinline__ unsigned wave_shl1(unsigned x, unsigned last)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)last, (int)x, 0x130, 0xf, 0xf, false);
}

// Operands of the mel products: the power values of two consecutive bins and the weights of two
// consecutive steps, one ds_read_b64 each (256 B/clk; fused ds_read2 forms move half of that).
// Group G8 covers steps 8 G8 .. 8 G8 + 7 (the last group of MS steps may be half a group).
template <int N, int STRIDE, int BASE, int I = 0>
__device__ __forceinline__ void lean_issue(v2f *t, unsigned addr)
{
    if constexpr (I < N) {
        t[I] = lds_rd64<BASE + I * STRIDE>(addr);
        lean_issue<N, STRIDE, BASE, I + 1>(t, addr);
    }
}
template <int MS, int G8>
constexpr int lean_mel_pairs() { return MS - 8 * G8 >= 8 ? 4 : (MS - 8 * G8) / 2; }
template <int MS, bool WL, int G8>
__device__ __forceinline__ void lean_mel_issue(v2f *av, v2f *bw, unsigned aAddr, unsigned wAddr)
{
    constexpr int N = lean_mel_pairs<MS, G8>();
    lean_issue<N, 8, 32 * G8>(av, aAddr);
    if constexpr (WL) lean_issue<N, 512, 2048 * G8>(bw, wAddr);
}
// waits until at most K newer LDS operations are outstanding; ties N results to the wait
template <int K, int N>
__device__ __forceinline__ void lean_wait(v2f *t)
{
    static_assert(N == 2 || N == 4, "pairs per group");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "i"(K));
    else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t[0]), "+v"(t[1]) : "i"(K));
}
// Products of group G8 while the operands of group G8 + 1 are on their way.
template <int MS, bool TREG, int G8>
__device__ __forceinline__ void lean_mel_groups(v2f (&av)[2][4], v2f (&bw)[2][4], unsigned aAddr, unsigned wAddr,
                                                const float *mwreg, Acc4<float> &m1, Acc4<float> &m2)
{
    constexpr int MSG = (MS + 7) / 8;
    if constexpr (G8 < MSG) {
        constexpr int c = G8 & 1, N = lean_mel_pairs<MS, G8>();
        constexpr int NEXT = G8 + 1 < MSG ? lean_mel_pairs<MS, G8 + 1>() * (TREG ? 1 : 2) : 0;
        if constexpr (G8 + 1 < MSG) lean_mel_issue<MS, !TREG, G8 + 1>(av[c ^ 1], bw[c ^ 1], aAddr, wAddr);
        lean_wait<NEXT, N>(av[c]);
        if constexpr (!TREG) lean_wait<NEXT, N>(bw[c]);
#pragma unroll
        for (int t = 0; t < N; ++t) {
            m1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][t].x, TREG ? mwreg[8 * G8 + 2 * t] : bw[c][t].x, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][t].y, TREG ? mwreg[8 * G8 + 2 * t + 1] : bw[c][t].y, m2, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        lean_mel_groups<MS, TREG, G8 + 1>(av, bw, aAddr, wAddr, mwreg, m1, m2);
    }
}

} // namespace

// NZ: number of first-stage inputs of a lane that can be non-zero, ceil(wlen / 32): 13 for
// 400-sample frames in 512 points (the zero padding of mfcc.c:144-147 is then never loaded,
// multiplied or added), 16 otherwise.  WAVES: waves (teams) per workgroup.  MSQ: groups of 4 steps
// of the mel filterbank products.
template <int NZ, int WAVES, int MSQ>
__global__ __launch_bounds__(WAVES * 64) void k_lean256(const FrontArgs<float> a)
{
    constexpr int NC = kLeanNc, G = kLeanG, PPL = kLeanPpl, GT = kLeanGt, TEAM = 64, NT = WAVES * 64;
    constexpr int PFN = kLeanPfn, MS = MSQ * 4;
    constexpr bool TREG = WAVES <= 8; // two waves per SIMD: window pairs and mel weights in registers
    typedef float T;
    STAMP_DECL
    STAMP_START;
    const int tid = threadIdx.x;
    const int wave = sgpr(tid >> 6);
    const int tt = tid & 63, l = tt & (G - 1), grp = tt >> 4;

    // LDS: [cepstral table | mel weights (MS x 64) | window pairs (NZ x 16)] then one region per
    // wave: [sample row | GT work areas (the tile's log-mel rows alias the first in cepstral mode)]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ldsTab = reinterpret_cast<T *>(smem);
    T *melWl = ldsTab + ((a.dctElems + 1) & ~1);
    cpx<T> *winl = reinterpret_cast<cpx<T> *>(melWl + MS * 64);
    T *rows = ldsTab + a.tabElems + wave * a.teamElems;
    T *zbuf = rows + a.rowPitch;
    T *fbk = zbuf;

    for (int i = tid; i < a.dctElems; i += NT) ldsTab[i] = a.dctT[i];
    // (mel weights of two consecutive steps side by side: one 8-byte read per pair of steps)
    for (int i = tid; i < MS * 64; i += NT) melWl[(i >> 7) * 128 + 2 * (i & 63) + ((i >> 6) & 1)] = i < a.melSteps * 64 ? a.melW[i] : 0.0f;
    for (int i = tid; i < NZ * 16; i += NT) winl[i] = reinterpret_cast<const cpx<T> *>(a.window)[i]; // half the Hamming values
    // Everything a frame reads from the row is finite from here on: a tile only overwrites it with
    // converted samples, and what lies behind a frame's last sample meets a zero window value.
    for (int s = tt; s < a.rowPitch; s += TEAM) rows[s] = (T)0;
    __syncthreads();
    const T *dctT = ldsTab;
    const T *melB = melWl + 2 * tt;
    const cpx<T> *winp = winl + l; // pair r of the lane: winp[16 r]

    // Per-lane constants in registers: split factors, second-stage twiddles.
    cpx<T> swreg[PPL / 2];
    cpx<T> twr[fft_twiddle_regs<8, 4>()];
#pragma unroll
    for (int q = 0; q < PPL / 2; ++q) swreg[q] = reinterpret_cast<const cpx<T> *>(a.splitTw)[l + G * q];
    fft_load_twiddles<T, 8, 4>(twr, l, reinterpret_cast<const cpx<T> *>(a.tw));
    cpx<T> wreg[TREG ? NZ : 1];
    float mwreg[TREG ? MS : 1];
    if constexpr (TREG) {
#pragma unroll
        for (int r = 0; r < NZ; ++r) wreg[r] = reinterpret_cast<const cpx<T> *>(a.window)[l + G * r];
#pragma unroll
        for (int t = 0; t < MS; ++t) mwreg[t] = t < a.melSteps ? a.melW[t * 64 + tt] : 0.0f;
    }
    const int4 mm = reinterpret_cast<const int4 *>(a.melMeta)[tt];
    const T *melA = zbuf + (tt & 3) * a.zPitch + mm.x;
    const int melFlags = mm.y, melCh = mm.z;
    const float chain1 = (melFlags & 1) ? 1.0f : 0.0f, chain2 = (melFlags & 4) ? 1.0f : 0.0f, chain3 = (melFlags & 8) ? 1.0f : 0.0f;
    const float kf = (float)a.preK;
    const int JW = a.M + a.c0Flag;

    // The wave's stream of tiles: chunks of a.tilesPerTeam consecutive tiles, the first one by the
    // wave's index, every further one drawn from a counter the waves of the launch share (waves run
    // at uneven speeds; fixed ranges left a fifth of the wave time idle at the end).  A chunk is
    // drawn while the one before it is worked on.  a.work[0] is that counter, zero at launch; the
    // last wave to finish (a.work[1] counts them) zeroes it again for the next launch.
    const int teamId = blockIdx.x * WAVES + wave, numTeams = gridDim.x * WAVES, C = a.tilesPerTeam;
    int chunkBase = teamId * C, chunkPos = 0, nextBase = 0;
    bool haveNext = false, needDraw = true;
    // A draw waits for its result, so it is made where the wave has just waited for all its memory
    // operations anyway and before the next ones are issued (draw_step below, once per chunk).
    auto draw = [&]() {
        unsigned drawn = 0;
        if (tt == 0) drawn = atomicAdd(a.work, 1u);
        nextBase = (sgpr((int)drawn) + numTeams) * C;
        haveNext = true;
        needDraw = false;
    };
    // next tile of the stream, or -1 behind its end
    auto next_tile = [&]() {
        if (chunkPos == C) {
            if (!haveNext) { // (only chunks of a single tile come here with the draw still to do)
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
    // (the compiler does not see that the stream state is the same in every lane: say so)
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
    LeanTile cur = lean_desc(tab, id, true);
    id = next_tile_s();
    LeanTile nx = lean_desc(tab, id < 0 ? 0 : id, id >= 0);
    id = next_tile_s();
    LeanTile nn = lean_desc(tab, id < 0 ? 0 : id, id >= 0);

    u4u pf[PFN];
#pragma unroll
    for (int j = 0; j < PFN; ++j) pf[j] = u4u{0u, 0u, 0u, 0u};

    // 16-byte vectors a tile's fetch covers: 8 samples each, the predecessor of the first included
    auto tile_vecs = [&](const LeanTile &t) { return ((t.nf() - 1) * a.inc + a.wlen + 1 + 7) >> 3; };
    // Requests the raw samples of tile t (vector j of lane tt = raw samples 8 (tt + 64 j) ..
    // + 7, sample 0 being the predecessor of the tile's first sample).
    auto fetch = [&](const LeanTile &t) {
        const int nvec = tile_vecs(t);
        const long long off0 = t.src() - 2;
        // Tiles at the first or last bytes of the buffer keep every vector inside it by clamping
        // its offset (expand() then redoes such a tile's samples one by one).
        int lo = 0, hi = 0x7ffffff0;
        if (t.edge()) {
            lo = (int)(off0 < 0 ? -off0 : 0);
            hi = (int)min(a.planEnd - 16 - off0, (long long)hi);
        }
        const unsigned char *g = a.pcm + off0;
#pragma unroll
        for (int j = 0; j < PFN; ++j) {
            const int vj = tt + j * TEAM;
            if (vj < nvec) pf[j] = *reinterpret_cast<const u4u *>(g + min(max(16 * vj, lo), hi));
        }
    };
    // Raw sample i of tile t (0: the predecessor of its first sample) for the one-by-one path:
    // zero outside the planned bytes.
    auto raw_checked = [&](const LeanTile &t, int i) {
        const long long b = t.src() - 2 + 2ll * i;
        int sv = 0;
        if (b >= 0 && b + 2 <= a.planEnd) sv = (int)(short)((unsigned)a.pcm[b] | ((unsigned)a.pcm[b + 1] << 8));
        return (float)sv;
    };
    // Fetched vectors -> pre-emphasised row: row[s] = raw[s + 1] - k raw[s] (PreEmphasise, call
    // site main.c:208), the vector's last output taking raw[8 j + 8] from the next lane; the
    // signal's first sample becomes (1 - k) raw[1].
    auto expand = [&](const LeanTile &t) {
        const int nvec = tile_vecs(t);
#pragma unroll
        for (int j = 0; j < PFN; ++j) {
            const int vj = tt + j * TEAM;
            const unsigned wrap = j + 1 < PFN ? (unsigned)sgpr((int)pf[j + 1 < PFN ? j + 1 : j].x) : 0u;
            const unsigned nxw = wave_shl1(pf[j].x, wrap);
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
        if (t.edge()) {
            const int ns = (t.nf() - 1) * a.inc + a.wlen;
            for (int sIdx = tt; sIdx < ns; sIdx += TEAM) rows[sIdx] = raw_checked(t, sIdx + 1) - kf * raw_checked(t, sIdx);
        }
        if (t.first() && tt == 0) rows[0] = (float)((double)raw_checked(t, 1) * (1.0 - a.preK));
    };

    // Prologue: the first tile's row, the second tile's fetch.
    fetch(cur);
    expand(cur);
    if (nx.nf() > 0) fetch(nx);
    team_sync<TEAM>();

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
                    if (slot_of(r) < dCount) ob[slot_of(r) * a.step + melCh] = dlv[r];
            }
            if (a.energyFlag && l == 0 && grp < dCount) ob[grp * a.step + JW] = dte; // main.c:243
            dCount = 0;
        }
    };

    for (;;) {
        const int id3 = next_tile_s();
        const LeanTile d3 = lean_desc(tab, id3 < 0 ? 0 : id3, id3 >= 0);
        const int nf = cur.nf();
        {
            STAMP(0);
            T *Zt = zbuf + slot_of(grp) * a.zPitch;
            cpx<T> *Z = reinterpret_cast<cpx<T> *>(Zt);

            // ---- sample pairs (2n, 2n+1), n = l + 16 r, times the window (Ham main.c:229); frame
            //      energy (te, mfcc.c:138-142).  Slots past the tile's last frame redo frame 0.
            cpx<T> v[PPL];
            {
                const cpx<T> *xp = reinterpret_cast<const cpx<T> *>(rows + (grp < nf ? grp : 0) * a.inc + 2 * l);
                cpx<T> xs[NZ], ws[NZ];
                lds_read_strided<NZ, 128>(xs, xp);
                if constexpr (!TREG) lds_read_strided<NZ, 128>(ws, winp);
#pragma unroll
                for (int r = 0; r < NZ; ++r) v[r] = xs[r] * (TREG ? wreg[r] : ws[r]);
#pragma unroll
                for (int r = NZ; r < PPL; ++r) v[r] = cpx<T>{0.0f, 0.0f};
            }
            T te = 0;
            if (a.energyFlag) {
                cpx<T> e2 = {0, 0};
#pragma unroll
                for (int r = 0; r < NZ; ++r) e2 += v[r] * v[r];
                te = group_sum<T, G>((T)4 * (e2.x + e2.y), nullptr, tt); // (half-scaled window)
            }

            STAMP(1);
            // ---- the row is free once every lane holds its samples: next tile's row
            if (nx.nf() > 0) {
                team_sync<TEAM>();
                expand(nx);
                draw_step();
                if (nn.nf() > 0) fetch(nn);
            }
            flush();
            STAMP(2);

            // ---- 256-point complex FFT of z[n] = x[2n] + i x[2n+1] (Realft mfcc.c:262-270), last
            //      stage in registers: v[r] = Z[l + 16 r]
            fft_stages<T, 8, 4, 0, 0, false, true, 0, true>(v, Z, l, nullptr, twr);
            STAMP(3);

            // ---- real-FFT split and power (mfcc.c:274-290, ek of mfcc.c:153-155): Z[256 - k] for
            //      k = l + 16 q is register 15 - q of lane 16 - l (register 16 - q of lane 0 itself)
            T pa[PPL / 2], pb[PPL / 2];
            {
                // the upper half of the lane's points goes through the (free) work area, complex
                // index = bin; the lane reads back Z[256 - l - 16 q], q = 7 .. 0 (lane 0, q = 0: one
                // point past the bins, unused).  The power values below use the area's first half.
#pragma unroll
                for (int r = PPL / 2; r < PPL; ++r) Z[l + G * r] = v[r];
                team_sync<TEAM>();
                cpx<T> rev[PPL / 2];
                lds_read_strided<PPL / 2, G * 8>(rev, &Z[NC - l - G * (PPL / 2 - 1)]);
#pragma unroll
                for (int q = 0; q < PPL / 2; ++q) split_power<T>(v[q], rev[PPL / 2 - 1 - q], swreg[q], pa[q], pb[q]);
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
            STAMP(4);

            // ---- mel filterbank (mfcc.c:152-163) as 4 x 4 x 1 matrix-core products: block b = 4
            //      lanes owns 4 consecutive channels and a run of bins; lane (b, i) feeds the power
            //      of frame slot i at bin r0_b + t and the weight of that bin for channel 4 q_b + i
            //      (operands of 8 steps run a group ahead of the products that use them)
            v2f av[2][4], bw[2][4];
            Acc4<float> macc = {0, 0, 0, 0}, macc2 = {0, 0, 0, 0};
            lean_mel_issue<MS, !TREG, 0>(av[0], bw[0], lds_addr(melA), lds_addr(melB));
            lean_mel_groups<MS, TREG, 0>(av, bw, lds_addr(melA), lds_addr(melB), mwreg, macc, macc2);
            macc += macc2;
            STAMP(5);
            {
                // a channel quad cut over up to 4 neighbouring blocks of one row of 16 lanes: the
                // chain's last block adds the others (row-wise lane shifts by 4, 8, 12)
                float ms[4] = {macc[0], macc[1], macc[2], macc[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = __builtin_bit_cast(int, ms[r]);
                    const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(m, 0x114, 0xf, 0xf, true));
                    const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(m, 0x118, 0xf, 0xf, true));
                    const float o3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(m, 0x11C, 0xf, 0xf, true));
                    macc[r] = fmaf(o3, chain3, fmaf(o2, chain2, fmaf(o1, chain1, ms[r])));
                }
            }
            if (!a.fbankFlag) team_sync<TEAM>(); // the log-mel rows alias the first work area
            if ((melFlags & 2) && melCh >= 0) { // owner of channel melCh: log(max(., 1)), mfcc.c:171-176
#pragma unroll
                for (int r = 0; r < GT; ++r) {
                    const T lv = tlog<T>(fmaxf(macc[r], 1.0f));
                    dlv[r] = lv;
                    if (!a.fbankFlag && slot_of(r) < nf) fbk[slot_of(r) * a.P + melCh] = lv;
                }
            }
            if (a.energyFlag) dte = tlog<T>(te);
            dOff = (long long)cur.outRow * a.step;
            dCount = nf;
            team_sync<TEAM>(); // the work areas are free for the next round
            STAMP(6);
        }
        if (!a.fbankFlag) {
            // ---- cepstra (FBank2MFCC mfcc.c:180-195) and C0 (FBank2C0 mfcc.c:197-208) of the tile
            //      as 16 x 16 x 4 matrix-core products: lane (i, g) = (lane % 16, lane / 16) feeds
            //      fbk[item i][k0 + g] and dct[coefficient i][k0 + g] and holds the results of items
            //      4 g .. 4 g + 3 for coefficient i.  The table's row M is all ones (C0).
            float *outBase = a.out + (long long)cur.outRow * a.step;
            const int i16 = tt & 15, g4 = tt >> 4;
            for (int jb = 0; jb * 16 < JW; ++jb) {
                const int jB = jb * 16 + i16;
                const T *fa = fbk + (i16 < nf ? i16 : 0) * a.P;
                const T *db = dctT + (jB < JW ? jB : 0) * a.dctPitch;
                Acc4<T> acc = {0, 0, 0, 0};
                for (int k0 = 0; k0 < a.P; k0 += 4) {
                    const int k = k0 + g4;
                    T fv = fa[k < a.P ? k : 0], bv = db[k];
                    fv = (i16 < nf && k < a.P) ? fv : (T)0;
                    bv = jB < JW ? bv : (T)0;
                    acc = mfma16(fv, bv, acc);
                }
                // items 4 g4 .. 4 g4 + 3: only g4 = 0 holds frames of a one-round tile
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (g4 == 0 && r < nf && jB < JW) outBase[r * a.step + jB] = acc[r] * a.mfnorm;
            }
            team_sync<TEAM>();
        }
        if (nx.nf() == 0) break;
        cur = nx;
        nx = nn;
        nn = d3;
    }
    flush();
    finish();
    STAMP_FLUSH;
}

// Geometry the host plans with: tiles of one round (4 frames), 12 waves per workgroup (8 when the
// frame fills more than 13 of a lane's 16 first-stage inputs, or when 12 regions do not fit the
// LDS), one workgroup per CU.
int lean256_nz(int wlen) { return wlen <= 2 * 16 * 13 ? 13 : 16; }
int lean256_waves(int wlen, bool fits12) { return (lean256_nz(wlen) == 13 && fits12) ? 12 : 8; }
// (instantiated for 8, 16, 24, 28 and 32 steps)
int lean256_mel_steps(int melSteps) { return melSteps <= 24 ? ((melSteps + 7) / 8) * 8 : ((melSteps + 3) / 4) * 4; }

bool lean256_applies(int log2nc, int mode, int inc, int wlen, int melSteps)
{
    // the fetch of a full tile must fit the vectors a lane holds
    const long vecs = ((long)(kLeanGt - 1) * inc + wlen + 1 + 7) >> 3;
    return log2nc == 8 && mode == 2 && vecs <= kLeanPfn * 64 && melSteps >= 1 && melSteps <= kMelSteps;
}

template <int NZ, int WAVES, int MSQ>
static int launch_lean_t(const FrontArgs<float> &a, int numBlocks, size_t lds, hipStream_t st)
{
    static std::atomic<bool> attrSet[kAttrDevices];
    const hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&k_lean256<NZ, WAVES, MSQ>), attrSet);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_lean256<NZ, WAVES, MSQ>), dim3(numBlocks), dim3(WAVES * 64), lds, st, a);
    return (int)hipGetLastError();
}

template <int NZ, int WAVES>
static int launch_lean_w(const FrontArgs<float> &a, int numBlocks, size_t lds, hipStream_t st)
{
    switch (lean256_mel_steps(a.melSteps) / 4) {
    case 2: return launch_lean_t<NZ, WAVES, 2>(a, numBlocks, lds, st);
    case 4: return launch_lean_t<NZ, WAVES, 4>(a, numBlocks, lds, st);
    case 6: return launch_lean_t<NZ, WAVES, 6>(a, numBlocks, lds, st);
    case 7: return launch_lean_t<NZ, WAVES, 7>(a, numBlocks, lds, st);
    default: return launch_lean_t<NZ, WAVES, 8>(a, numBlocks, lds, st);
    }
}

int launch_lean256(const FrontArgs<float> &a, int numBlocks, size_t lds, int waves, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (lean256_nz(a.wlen) == 13) return waves == 12 ? launch_lean_w<13, 12>(a, numBlocks, lds, st) : launch_lean_w<13, 8>(a, numBlocks, lds, st);
    return launch_lean_w<16, 8>(a, numBlocks, lds, st);
}

} // namespace mfcc
