// In-register radix-2/4/8/16 DFT butterflies (forward transform, e^{-i...} sign) for gfx950.
// A lane keeps up to 16 complex points in VGPRs; every index below is a compile-time constant so
// nothing is spilled to scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace mfcc {

// A complex value is a two-element vector (.x real, .y imaginary), aligned to its full size: LDS
// accesses are single ds_read_b64 / ds_write_b64 (b128 for double) instructions, and a
// single-precision value sits in an even-aligned register pair, the operand form of the packed
// instructions (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) that process both halves at once.
template <typename T>
using cpx = T __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ cpx<T> cadd(cpx<T> a, cpx<T> b) { return a + b; }
template <typename T>
__device__ __forceinline__ cpx<T> csub(cpx<T> a, cpx<T> b) { return a - b; }
template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b)
{
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i
template <typename T>
__device__ __forceinline__ cpx<T> cmul_mi(cpx<T> a) { return {a.y, -a.x}; }
// a + (-i) b, a - (-i) b
template <typename T>
__device__ __forceinline__ cpx<T> cadd_mi(cpx<T> a, cpx<T> b) { return {a.x + b.y, a.y - b.x}; }
template <typename T>
__device__ __forceinline__ cpx<T> csub_mi(cpx<T> a, cpx<T> b) { return {a.x - b.y, a.y + b.x}; }
// a + (-i) h b, a - (-i) h b for a real factor h
template <typename T>
__device__ __forceinline__ cpx<T> cfma_mi(cpx<T> b, T h, cpx<T> a) { return {a.x + h * b.y, a.y - h * b.x}; }
template <typename T>
__device__ __forceinline__ cpx<T> cfms_mi(cpx<T> b, T h, cpx<T> a) { return {a.x - h * b.y, a.y + h * b.x}; }
// a + h b, a - h b
template <typename T>
__device__ __forceinline__ cpx<T> cfma_r(cpx<T> b, T h, cpx<T> a) { return {a.x + h * b.x, a.y + h * b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> cfms_r(cpx<T> b, T h, cpx<T> a) { return {a.x - h * b.x, a.y - h * b.y}; }

// Single precision: every helper above is one packed instruction (the complex product two).  The
// operand halves are picked by op_sel (low result) / op_sel_hi (high result) and negated per
// result half by neg_lo / neg_hi, so swapping real and imaginary parts or conjugating costs
// nothing.
#define MFCC_PK_ADD(name, mods)                                                            \
    template <>                                                                            \
    __device__ __forceinline__ cpx<float> name<float>(cpx<float> a, cpx<float> b)          \
    {                                                                                      \
        cpx<float> r;                                                                      \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));                   \
        return r;                                                                          \
    }
MFCC_PK_ADD(cadd_mi, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
MFCC_PK_ADD(csub_mi, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
#undef MFCC_PK_ADD
// (hv holds the factor in both halves)
#define MFCC_PK_FMA(name, mods)                                                                      \
    __device__ __forceinline__ cpx<float> name(cpx<float> b, cpx<float> hv, cpx<float> a)            \
    {                                                                                                \
        cpx<float> r;                                                                                \
        asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(r) : "v"(b), "v"(hv), "v"(a));                \
        return r;                                                                                    \
    }
MFCC_PK_FMA(pk_fma_mi, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")
MFCC_PK_FMA(pk_fms_mi, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")
#undef MFCC_PK_FMA
template <>
__device__ __forceinline__ cpx<float> cfma_mi<float>(cpx<float> b, float h, cpx<float> a)
{
    return pk_fma_mi(b, cpx<float>{h, h}, a);
}
template <>
__device__ __forceinline__ cpx<float> cfms_mi<float>(cpx<float> b, float h, cpx<float> a)
{
    return pk_fms_mi(b, cpx<float>{h, h}, a);
}
template <>
__device__ __forceinline__ cpx<float> cfma_r<float>(cpx<float> b, float h, cpx<float> a)
{
    return b * cpx<float>{h, h} + a;
}
template <>
__device__ __forceinline__ cpx<float> cfms_r<float>(cpx<float> b, float h, cpx<float> a)
{
    return a - b * cpx<float>{h, h};
}
// t = (a.x b.x, a.y b.x), then (-a.y b.y + t.x, a.x b.y + t.y)
template <>
__device__ __forceinline__ cpx<float> cmul<float>(cpx<float> a, cpx<float> b)
{
    cpx<float> t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}

template <typename T>
__device__ __forceinline__ void fft2(cpx<T> &a, cpx<T> &b)
{
    cpx<T> t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// Natural-order 4-point DFT of (a, b, c, d) in place: 8 complex additions.
template <typename T>
__device__ __forceinline__ void fft4(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d)
{
    const cpx<T> s0 = cadd(a, c), d0 = csub(a, c);
    const cpx<T> s1 = cadd(b, d), t = csub(b, d);
    a = cadd(s0, s1);
    c = csub(s0, s1);
    b = cadd_mi(d0, t);
    d = csub_mi(d0, t);
}

template <typename T, int R>
struct Radix;

template <typename T>
struct Radix<T, 1> {
    static __device__ __forceinline__ void run(cpx<T> *) {}
};

template <typename T>
struct Radix<T, 2> {
    static __device__ __forceinline__ void run(cpx<T> *v) { fft2(v[0], v[1]); }
};

template <typename T>
struct Radix<T, 4> {
    static __device__ __forceinline__ void run(cpx<T> *v) { fft4(v[0], v[1], v[2], v[3]); }
};

// 8 = 2 x 4: 4-point DFTs over the even inputs (A) and the odd inputs (B), then
// X[k] = A[k] + w8^k B[k], X[k + 4] = A[k] - w8^k B[k].  With u = (1 - i) b = b + (-i) b and
// u' = (1 + i) b = b - (-i) b:  w8 b = h u,  w8^2 b = -i b,  w8^3 b = -h u'  (h = 1/sqrt 2), so
// every twiddle folds into the additions around it.
template <typename T>
struct Radix<T, 8> {
    static __device__ __forceinline__ void run(cpx<T> *v)
    {
        const T h = (T)0.70710678118654752440;
        fft4(v[0], v[2], v[4], v[6]);
        fft4(v[1], v[3], v[5], v[7]);
        const cpx<T> a0 = v[0], a1 = v[2], a2 = v[4], a3 = v[6], b0 = v[1], b2 = v[5];
        const cpx<T> u1 = cadd_mi(v[3], v[3]), u3 = csub_mi(v[7], v[7]);
        v[0] = cadd(a0, b0);
        v[4] = csub(a0, b0);
        v[1] = cfma_r(u1, h, a1);
        v[5] = cfms_r(u1, h, a1);
        v[2] = cadd_mi(a2, b2);
        v[6] = csub_mi(a2, b2);
        v[3] = cfms_r(u3, h, a3);
        v[7] = cfma_r(u3, h, a3);
    }
};

// 16 = 4 x 4: 4-point DFTs over inputs c, c+4, c+8, c+12 (result k of column c in v[c + 4 k]),
// twiddle by w16^{c k}, 4-point DFTs across c.  Output X[k + 4 m] is result m of the second DFT for
// inner index k.  The twiddles w16^2 = h (1 - i), w16^4 = -i, w16^6 = -i h (1 - i) fold into the
// additions of the second pass; w16^1, w16^3 are complex products with constants, w16^9 = -w16^1.
template <typename T>
struct Radix<T, 16> {
    static __device__ __forceinline__ void run(cpx<T> *v)
    {
        const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
        const T h = (T)0.70710678118654752440;
        const cpx<T> w1 = {c1, -s1}, w3 = {s1, -c1};
#pragma unroll
        for (int c = 0; c < 4; ++c) fft4(v[c], v[c + 4], v[c + 8], v[c + 12]);
        fft4(v[0], v[1], v[2], v[3]); // k = 0: no twiddles
        { // k = 1: b = w^1 v5, c = h u6, d = w^3 v7
            const cpx<T> b = cmul(v[5], w1), d = cmul(v[7], w3), u = cadd_mi(v[6], v[6]);
            const cpx<T> s0 = cfma_r(u, h, v[4]), d0 = cfms_r(u, h, v[4]);
            const cpx<T> s1v = cadd(b, d), t = csub(b, d);
            v[4] = cadd(s0, s1v);
            v[6] = csub(s0, s1v);
            v[5] = cadd_mi(d0, t);
            v[7] = csub_mi(d0, t);
        }
        { // k = 2: b = h u9, c = -i v10, d = -i h u11
            const cpx<T> u9 = cadd_mi(v[9], v[9]), u11 = cadd_mi(v[11], v[11]);
            const cpx<T> s0 = cadd_mi(v[8], v[10]), d0 = csub_mi(v[8], v[10]);
            const cpx<T> S = cadd_mi(u9, u11), Tm = csub_mi(u9, u11); // s1 = h S, t = h Tm
            v[8] = cfma_r(S, h, s0);
            v[10] = cfms_r(S, h, s0);
            v[9] = cfma_mi(Tm, h, d0);
            v[11] = cfms_mi(Tm, h, d0);
        }
        { // k = 3: b = w^3 v13, c = -i h u14, d = -w^1 v15
            const cpx<T> b = cmul(v[13], w3), e = cmul(v[15], w1), u = cadd_mi(v[14], v[14]);
            const cpx<T> s0 = cfma_mi(u, h, v[12]), d0 = cfms_mi(u, h, v[12]);
            const cpx<T> s1v = csub(b, e), t = cadd(b, e);
            v[12] = cadd(s0, s1v);
            v[14] = csub(s0, s1v);
            v[13] = cadd_mi(d0, t);
            v[15] = csub_mi(d0, t);
        }
        // now v[4k + m] = X[k + 4m]; transpose the 4x4 index grid into natural order
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = k + 1; m < 4; ++m) {
                cpx<T> t = v[4 * k + m];
                v[4 * k + m] = v[4 * m + k];
                v[4 * m + k] = t;
            }
    }
};

} // namespace mfcc
