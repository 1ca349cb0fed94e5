// Diagnostic builds only (-DMFCC_STAMPS): per-phase cycle shares of the static-feature kernels,
// summed over waves into FrontArgs::stamps.  Never compiled into the shipped library.
#pragma once

#if defined(MFCC_STAMPS) && defined(MFCC_SKIP_ONLY)
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#define SKIP(b) ((a.skip >> (b)) & 1)
#elif defined(MFCC_STAMPS)
// One interval per run, chosen at launch: a.stampBegin is a bit mask of the stamp sites that start
// the clock, a.stampEnd the site that stops it (site 15 = kernel start / STAMP_START).  Only the
// chosen sites read the clock (and drain the LDS queue to do so), so the rest of the kernel runs
// as in the shipped build.  stamps[0] = summed interval cycles, stamps[1] = intervals,
// stamps[2] = summed wave lifetimes.
#define STAMP_DECL unsigned long long st_last = 0, st_acc = 0, st_t0 = 0, st_r0 = 0; unsigned st_n = 0; bool st_on = false;
#define STAMP_READ(var)                                         \
    do {                                                        \
        __builtin_amdgcn_sched_barrier(0);                      \
        var = __builtin_amdgcn_s_memtime();                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                     \
        __builtin_amdgcn_sched_barrier(0);                      \
    } while (0)
#define STAMP_START                                             \
    do {                                                        \
        STAMP_READ(st_t0);                                      \
        st_r0 = __builtin_amdgcn_s_memrealtime(); \
        if ((a.stampBegin >> 15) & 1) { st_last = st_t0; st_on = true; } \
    } while (0)
#define STAMP(i)                                                \
    do {                                                        \
        if (a.stampEnd == (i) && st_on) {                       \
            unsigned long long t_;                              \
            STAMP_READ(t_);                                     \
            st_acc += t_ - st_last;                             \
            ++st_n;                                             \
            st_on = false;                                      \
        }                                                       \
        if ((a.stampBegin >> (i)) & 1) {                        \
            STAMP_READ(st_last);                                \
            st_on = true;                                       \
        }                                                       \
    } while (0)
#define STAMP_FLUSH                                                                        \
    do {                                                                                   \
        unsigned long long t_;                                                             \
        STAMP_READ(t_);                                                                    \
        if ((threadIdx.x & 63) == 0 && a.stamps) {                                         \
            atomicAdd(a.stamps + 0, st_acc);                                               \
            atomicAdd(a.stamps + 1, (unsigned long long)st_n);                             \
            atomicAdd(a.stamps + 2, t_ - st_t0);                                           \
            const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();               \
            atomicMin(a.stamps + 3, st_r0);                                                \
            atomicMax(a.stamps + 4, r1_);                                                  \
            atomicAdd(a.stamps + 5, r1_ - st_r0);                                          \
            atomicAdd(a.stamps + 6, 1ull);                                                 \
        }                                                                                  \
    } while (0)
#define SKIP(b) ((a.skip >> (b)) & 1) // ablation: results are wrong, only counters / time matter
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#define STAMP_FLUSH
#define SKIP(b) false
#endif
