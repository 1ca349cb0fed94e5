// In-register radix-2/4/8/16 DFT butterflies (forward transform, e^{-i...} sign) for gfx950.
// A lane keeps up to 16 complex points in VGPRs; every index below is a compile-time constant so
// nothing is spilled to scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace mfcc {

// Aligned to its full size so that LDS accesses are single ds_read_b64 / ds_write_b64 (b128 for
// double) instructions; as two 4-byte halves they would hit the 32-bank path at stride 2 and
// take a 2-way bank conflict on every access.
template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};

template <typename T>
__device__ __forceinline__ cpx<T> cadd(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> csub(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b)
{
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// Single precision: two packed instructions.  t = (a.x b.x, a.y b.x), then
// (-a.y b.y + t.x, a.x b.y + t.y) with the operand halves picked by op_sel / op_sel_hi.
template <>
__device__ __forceinline__ cpx<float> cmul<float>(cpx<float> a, cpx<float> b)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    const v2 av = {a.x, a.y}, bv = {b.x, b.y};
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(av), "v"(bv), "v"(t));
    return {r.x, r.y};
}
// multiply by -i
template <typename T>
__device__ __forceinline__ cpx<T> cmul_mi(cpx<T> a) { return {a.y, -a.x}; }

template <typename T>
__device__ __forceinline__ void fft2(cpx<T> &a, cpx<T> &b)
{
    cpx<T> t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// Natural-order 4-point DFT of (a, b, c, d) in place.
template <typename T>
__device__ __forceinline__ void fft4(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d)
{
    cpx<T> s0 = cadd(a, c), d0 = csub(a, c);
    cpx<T> s1 = cadd(b, d), d1 = cmul_mi(csub(b, d));
    a = cadd(s0, s1);
    c = csub(s0, s1);
    b = cadd(d0, d1);
    d = csub(d0, d1);
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

// 8 = 2 x 4: 4-point DFTs over the even and the odd inputs, twiddle by w8^k, then 2-point DFTs.
template <typename T>
struct Radix<T, 8> {
    static __device__ __forceinline__ void run(cpx<T> *v)
    {
        const T h = (T)0.70710678118654752440;
        fft4(v[0], v[2], v[4], v[6]); // A[k] in v[0], v[2], v[4], v[6]
        fft4(v[1], v[3], v[5], v[7]); // B[k] in v[1], v[3], v[5], v[7]
        cpx<T> b1 = {h * (v[3].x + v[3].y), h * (v[3].y - v[3].x)};  // * (1 - i)/sqrt2
        cpx<T> b2 = cmul_mi(v[5]);                                    // * -i
        cpx<T> b3 = {h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y)}; // * (-1 - i)/sqrt2
        cpx<T> a0 = v[0], a1 = v[2], a2 = v[4], a3 = v[6], b0 = v[1];
        v[0] = cadd(a0, b0);
        v[4] = csub(a0, b0);
        v[1] = cadd(a1, b1);
        v[5] = csub(a1, b1);
        v[2] = cadd(a2, b2);
        v[6] = csub(a2, b2);
        v[3] = cadd(a3, b3);
        v[7] = csub(a3, b3);
    }
};

// 16 = 4 x 4: 4-point DFTs over inputs c, c+4, c+8, c+12; twiddle by w16^{c k}; 4-point DFTs
// across c.  Output X[k + 4 m] is result m of the second DFT for inner index k.
template <typename T>
struct Radix<T, 16> {
    static __device__ __forceinline__ void run(cpx<T> *v)
    {
        const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
        const T h = (T)0.70710678118654752440;
#pragma unroll
        for (int c = 0; c < 4; ++c) fft4(v[c], v[c + 4], v[c + 8], v[c + 12]);
        // after the inner pass v[c + 4k] = U_c[k]; twiddle w16^{c k} = (cos, -sin)(2 pi c k / 16)
        v[5] = cmul(v[5], cpx<T>{c1, -s1});                            // c=1,k=1 : w^1
        v[9] = cpx<T>{h * (v[9].x + v[9].y), h * (v[9].y - v[9].x)};   // c=1,k=2 : w^2
        v[13] = cmul(v[13], cpx<T>{s1, -c1});                          // c=1,k=3 : w^3
        v[6] = cpx<T>{h * (v[6].x + v[6].y), h * (v[6].y - v[6].x)};   // c=2,k=1 : w^2
        v[10] = cmul_mi(v[10]);                                        // c=2,k=2 : w^4
        v[14] = cpx<T>{h * (v[14].y - v[14].x), -h * (v[14].x + v[14].y)}; // c=2,k=3 : w^6
        v[7] = cmul(v[7], cpx<T>{s1, -c1});                            // c=3,k=1 : w^3
        v[11] = cpx<T>{h * (v[11].y - v[11].x), -h * (v[11].x + v[11].y)}; // c=3,k=2 : w^6
        v[15] = cmul(v[15], cpx<T>{-c1, s1});                          // c=3,k=3 : w^9
#pragma unroll
        for (int k = 0; k < 4; ++k) fft4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
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
