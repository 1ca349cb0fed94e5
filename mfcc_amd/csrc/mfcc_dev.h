// Internal structures shared by the host code and the HIP kernels (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mfcc {

constexpr int kBlock = 256;       // threads per workgroup of every kernel (4 waves of 64)
constexpr int kMaxLog2Nc = 13;    // fftN <= 16384
constexpr int kMaxZnormStep = 8192; // columns per row the z-norm kernel holds statistics for (16 bytes each in LDS)
constexpr int kStatRows = 256;    // rows per workgroup of the z-norm statistics kernels
constexpr int kRegRows = 256;     // rows per workgroup of the regression / fused delta kernels (fewer in small batches: MFCCPlan::regRows)
constexpr int kMelSteps = 32;     // most bins one block of the matrix-core mel filterbank walks

// One utterance (or one rank's shard of a long recording) as the kernels see it.
struct UttDesc {
    int64_t byteOff;    // first byte of instant 0 of this utterance's buffer inside the PCM buffer
    int64_t outRow;     // output row of local frame 0
    int32_t numInstants;// sample instants in the buffer (lead instant included)
    int32_t numFrames;  // frames computed from the buffer (local frames 0 .. numFrames-1)
    int32_t lead;       // 1: instant 0 only serves as the pre-emphasis predecessor of instant 1
                        //    (the buffer starts inside a longer recording); 0: instant 0 starts the signal
    int32_t regBegin;   // index of this utterance's first row block in the regression table
    int32_t ownBegin;   // first local frame that enters the z-norm statistics / is normalised
    int32_t ownCount;   // number of such frames
    int32_t statBegin;  // index of this utterance's first statistics chunk
    int32_t tileBegin;  // tiles of all utterances before this one (its first tile's index)
    double mean[4];     // value subtracted from the derived rows {L, R, mid, side} (0 = option off)
};

// One tile of the static-feature kernel: up to framesPerTile consecutive frames of one utterance,
// resolved on the host so that a team needs a single 16-byte load to start on it.
struct TileDesc {
    uint64_t src;   // bits 0..47: byte offset (inside the PCM buffer) of the pre-emphasis predecessor
                    // of the tile's first sample, or of that sample itself when it starts the signal;
                    // bits 48..59: frames in the tile; bit 60: the predecessor exists; bit 61: register path
                    // allowed (kernel mode 2: predecessor exists, at most 3 x 64 vectors of 8 samples, and
                    // the bytes up to the end of the last vector lie inside the planned PCM extent)
    int64_t outRow; // output row of the tile's first frame
};
constexpr uint64_t kTileOffMask = (1ull << 48) - 1;

// Arguments of the static-feature kernel.
template <typename T>
struct FrontArgs {
    const unsigned char *pcm;
    const TileDesc *tiles;    // (k_lean256: its own encoding, see LeanTile in lean256.hip)
    const UttDesc *utts;      // read only for the row means (zeroMean != 0)
    const int32_t *tileUtt;   // utterance of each tile (read only if zeroMean != 0)
    int zeroMean;
    int lean;                 // kernel mode: 0 general, 1 lean, 2 lean with the matrix-core mel filterbank
    const float *melW;        // mode 2: melSteps x 64 weights, [step][lane]
    const int32_t *melMeta;   // mode 2: 64 x {first bin of the lane's block, flags (1 / 4 / 8: add the block
                              // one / two / three before, 2: writes results), channel or -1, 0}
    int melSteps;
    const double *melW64;     // k_pipe64, 256-point class: melSteps x 64 weights of the 4 x 4 x 4 matrix-core filterbank
    int melSlotSteps[4];      // ... steps of each of its four slots (a slot: four quads of channels at a time)
    T *out;
    int numTiles;
    int numUtts;              // entries of utts (k_lean256 walks them)
    int tilesPerTeam;         // k_lean256: tiles per chunk of a wave's work
    unsigned *work;           // k_lean256: {chunks drawn, waves finished}, both zero at launch (the kernel
                              // leaves them zero)
    long long planEnd;        // k_lean256: one past the last PCM byte the plan owns
    int wlen, inc, framesPerTile;
    int V, channels, bps;
    int P, M, D, vSize, step;
    int c0Flag, energyFlag, zcFlag, brightFlag, S, fftLength, fbankFlag;
    int usePower, takeLogs; // mfcc.c:154-157,171-176 (both 1 at the reference's call site)
    double preK;
    T mfnorm;
    const T *window;       // wlen values, zero-padded to fftN + 8
    const T *tw;           // nc complex (cos, -sin)(2 pi m / nc), interleaved
    const T *splitTw;      // nc/2 complex (cos, sin)(pi a / nc), interleaved
    const T *loWt;         // nc
    const int32_t *laneSeg;   // lanes-per-frame x maxSeg x 4: {first bin, end bin, segment index or -1, 0}
    int maxSeg;               // mel segments (bins sharing one lower channel) a lane sums at most
    int melPieces;            // parts a segment is cut into (different lanes), summed in part order
    int pairAligned;          // 1: every frame starts on an even sample of its row (inc is even)
    int zPitch;               // elements per frame work area
    const T *dctT;         // (M + c0) x dctPitch, j-major: cos(pi (j+1) (k + 0.5) / P), then a row of ones
                           // for C0; rows are zero from k = P to the (odd, >= P + 3) pitch
    int dctPitch, dctElems; // dctElems = (M + c0) * dctPitch (0 in filterbank mode)
    int rowPitch;          // elements per derived row in LDS
    int rowTail;           // elements zero-filled behind the last sample of a tile (>= fftN - wlen + 1)
    int rawOrZElems;       // elements reserved for max(raw PCM bytes, a team's work areas)
    int teamElems;         // elements of one team's LDS region (rows, work areas, staging, sums)
    int activeTeams;       // teams of a workgroup that take tiles (1 .. teams per workgroup)
    int tabElems;          // elements of the LDS copy of the tables (0: tables are read through L1/L2)
    int tabPitch;          // k_wide1024: row pitch of the cepstral table's LDS copy
    int winPad;            // elements reserved for the window inside that copy
    int featElems;         // elements of the static-feature staging area (framesPerTile x vSize, padded)
    int fbkElems;          // elements of the log-mel area, framesPerTile x V x P (0 in filterbank mode)
    unsigned long long *stamps; // diagnostic builds only: {interval cycles, intervals, wave lifetimes} (else NULL)
    int skip;                   // diagnostic builds only: bit mask of phases to leave out
    int stampBegin, stampEnd;   // diagnostic builds only: stamp sites that start / stop the clock
};

template <typename T>
struct RegressArgs {
    T *data;
    const UttDesc *utts;
    const int32_t *blockUtt; // utterance of each block of regRows rows
    int numBlocks;
    int vSize, step, srcCol, dstCol, delwin;
    int regRows;             // rows per block
    T invSigmaT2;
    double sigmaT2;
};

// Fused delta kernel: all regression orders of a block of rows, plus its column sums when partial != 0.
template <typename T>
struct DeltaArgs {
    T *data;
    const UttDesc *utts;
    const int32_t *blockUtt; // utterance of each block of regRows rows
    int numBlocks;
    int vSize, step, delwin;
    int subRows;             // rows per LDS sub-tile
    int regRows;             // rows per block (a multiple of subRows)
    int redOffset;           // byte offset of the 2 * kBlock doubles of reduction scratch in LDS
    double sigmaT2;
    double *partial;         // numBlocks x 2*step: per-block column sums and sums of squares, or NULL
};

// Whole-utterance finishing pass (finish.hip)
constexpr int kFinishBlock = 512; // threads per workgroup (one workgroup per utterance)
constexpr int kFinishRegRows = 32; // rows per lane the register variant holds
template <typename T>
struct FinishArgs {
    const T *statics;   // totalRows x vSize, compact
    T *out;             // totalRows x step
    const UttDesc *utts;
    int vSize, step, delwin;
    int subRows;        // rows finished per round
    int staticElems;    // elements reserved in LDS for the longest utterance's static rows
    int redOffset;      // byte offset of the reduction scratch (2 kFinishBlock + 2 step doubles) in LDS
    int znorm;
    int regRows;        // > 0: no utterance has more than kFinishRegRows rows per lane (register variant)
    double sigmaT2;
};

// Segment-streaming finishing pass of batch plans (stream_finish.hip)
template <typename T>
struct StreamArgs {
    const T *statics;       // totalRows x vSize, compact
    T *out;                 // totalRows x step
    const UttDesc *utts;
    const int32_t *segs;    // numSegs x {utterance, first row of the segment inside it}
    const int32_t *segBegin;// numUtts + 1: first segment of every utterance
    int numSegs, numUtts;
    int vSize, step, delwin, znorm;
    int colBlock, numColBlocks; // columns per workgroup (<= 64), workgroups per segment
    int chunkRows;          // rows a wave finishes per chunk (a segment: stream_finish_seg_chunks() chunks)
    int waveElems;          // LDS elements of one wave's region
    int redOffset;          // byte offset of the reduction scratch in LDS
    int fastPath;           // register path: delwin is one of its compile-time values, a chunk is reg_rows(delwin) rows per lane
    int stageOut;           // ... and finished rows leave through LDS as one contiguous piece
    int phase;              // 0: batch plan (sums, statistics, rows); 1: shard plan's extraction (the segments' sums when a
                            // z-norm follows, else the rows); 2: shard plan's normalisation (the rows, normalised with `stats`)
    double sigmaT2;
    double *partial;        // numSegs x 2*step: column sums and sums of squares of every segment
    double *stats;          // numUtts x 2*step: mean | 1 / std
};
int stream_finish_seg_chunks();
bool stream_finish_geometry(int vSize, int orders, int delwin, int elemBytes, int wantRows, int *colBlock, int *numColBlocks,
                            int *chunkRows, int *waveElems, int *redOffset, size_t *ldsBytes, int *fastPath, int *stageOut);
int launch_seg_finish(int precision, const void *args, int orders, size_t ldsBytes, void *stream);
int launch_seg_shard_sums(const double *partial, const UttDesc *utts, int numSegs, double *sums, int step, void *stream);
int launch_totals_to_stats(const double *totals, double *stats, int step, void *stream);

template <typename T>
struct StatArgs {
    T *data;
    const UttDesc *utts;
    const int32_t *chunkUtt;
    int numChunks;
    int step;
    double *partial; // numChunks x 2*step
    double *stats;   // numUtts x 2*step : mean, 1/std  (batch plans)  or  shard sums (2*step+1)
};

int launch_front(int precision, int log2nc, int pplog, const void *args, int numBlocks,
                 size_t ldsBytes, void *stream);
// lean256.hip: pipelined kernel for mono 16-bit, 256 complex points, matrix-core mel (mode 2 plans)
bool lean256_applies(int log2nc, int mode, int inc, int wlen, int melSteps);
int lean256_nz(int wlen);                 // first-stage inputs per lane that can be non-zero (13 or 16)
int lean256_waves(int wlen, bool fits12); // waves (teams) per workgroup
int lean256_mel_steps(int melSteps);      // rows of the mel weight table the kernel keeps in LDS
int launch_lean256(const FrontArgs<float> &a, int numBlocks, size_t ldsBytes, int waves, void *stream);
// wide1024.hip: 24-bit stereo, 1024 complex points, samples expanded straight from the PCM bytes in memory
bool wide1024_applies(int log2nc, int channels, int bps, int wlen);
int wide1024_waves();
int wide1024_shared_elems(int wlen);      // LDS elements the waves share in front of the cepstral table
int wide1024_reach_bytes(int wlen);       // bytes from a frame's first instant that its loads reach
int launch_wide1024(const FrontArgs<float> &a, int numBlocks, size_t ldsBytes, void *stream);
// pipe64.hip: double precision, mono 16-bit with 256 complex points or 24-bit stereo with 1024
int pipe64_class(int log2nc, int channels, int bps, int wlen, int inc); // 0: not taken, else 8 or 10
int pipe64_waves(int cls, bool fits12);   // waves per workgroup (256-point class: 12 when that many regions fit the LDS, else 8)
int pipe64_frames_per_tile(int cls);
int pipe64_lanes(int cls);                // lanes that share one frame
int pipe64_raw_bytes(int cls, int wlen, int inc); // LDS bytes a wave reserves for a tile's raw PCM
int pipe64_area_elems(int cls);           // doubles per frame work area
int pipe64_tab_elems(int cls, int wlen, int dctElems, int melSteps); // doubles of the shared tables in LDS
int pipe64_reach_bytes(int wlen);          // 1024-point class: bytes from a frame's first instant that its loads reach
int launch_pipe64(const FrontArgs<double> &a, int cls, int waves, int numBlocks, size_t ldsBytes, void *stream);
bool front_variant_exists(int log2nc, int pplog); // pplog: log2 of complex points per lane (4, 3, 2)
int front_groups(int log2nc, int pplog);          // frames in flight per workgroup
int front_teams(int log2nc, int pplog);           // independent teams (tile streams) per workgroup
int front_lanes(int log2nc, int pplog);           // lanes that share one frame
int front_waves_limit(int precision, int log2nc, int pplog); // waves per SIMD the kernel is compiled for
int front_occupancy(int precision, int log2nc, int pplog, bool tabLds, int mode, size_t ldsBytes);
int launch_regress(int precision, const void *args, int numBlocks, void *stream);
int launch_colstats(int precision, const void *args, int numChunks, void *stream);
int launch_colfinal(const StatArgs<float> &a, int numUtts, int blockRows, void *stream); // blockRows > 0: partial sums per block of that many rows
int launch_shardfinal(const StatArgs<float> &a, int numChunks, int blockRows, double *sumsOut, void *stream);
int launch_delta(int precision, const void *args, int orders, size_t ldsBytes, void *stream); // -2: no such variant
int launch_utt_finish(int precision, const void *args, int orders, int numUtts, size_t ldsBytes, void *stream);
int launch_energy_norm(int precision, void *data, const UttDesc *utts, int numUtts, int step, int col, double silFloor,
                       double escale, void *stream);
int launch_normalize(int precision, const void *args, int numChunks, bool shardSums,
                     const double *sums, void *stream);
// batched pieces of the reference's lower-level API (pieces.hip)
int launch_fft_batch(int precision, void *data, int numFrames, int n, int invert, int realft, void *stream);
int launch_wave2fbank_batch(int precision, const void *frames, void *fbank, void *te, void *te2, void *spec,
                            int numFrames, int frameSize, int fftN, int klo, int khi, int P, const int *loChan,
                            const void *loWt, int usePower, int takeLogs, void *stream);
int launch_fbank2mfcc_direct(int precision, const void *fbank, void *mfcc, void *c0, int numFrames, int P, int n,
                             void *stream);
int launch_fbank2mfcc_batch(int precision, const void *fbank, void *mfcc, void *c0, int numFrames, int P,
                            int M, const void *dctT, int dctPitch, double mfnorm, void *stream);
int launch_chansums(const unsigned char *pcm, int channels, int bps, int64_t firstByte,
                    int64_t count, unsigned long long *sums2 /*4, pre-zeroed*/, void *stream);
int launch_uttmeans(const unsigned char *pcm, UttDesc *utts, int numUtts, int channels, int bps,
                    unsigned long long *sums2 /*numUtts x 4, pre-zeroed*/, void *stream);

} // namespace mfcc
