// Exact, allocation-free "%f" and "%e" formatting of doubles (6 digits after the point, the C
// default), byte-identical to printf: the text writers of main.c:276-285 spend almost all their
// time in fprintf once extraction runs on the GPU.  A double is m * 2^e exactly, so the decimal
// digits follow from 128-bit integer arithmetic with round-half-even on the exact remainder;
// anything outside the covered range (non-finite, denormal, huge, tiny) falls back to snprintf.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mfcc_host.h"

namespace mfcc {

namespace {

typedef unsigned __int128 u128;

const uint64_t kPow10[20] = {1ull,
                             10ull,
                             100ull,
                             1000ull,
                             10000ull,
                             100000ull,
                             1000000ull,
                             10000000ull,
                             100000000ull,
                             1000000000ull,
                             10000000000ull,
                             100000000000ull,
                             1000000000000ull,
                             10000000000000ull,
                             100000000000000ull,
                             1000000000000000ull,
                             10000000000000000ull,
                             100000000000000000ull,
                             1000000000000000000ull,
                             10000000000000000000ull};

inline u128 pow10_128(int p) // p <= 38
{
    u128 r = 1;
    while (p >= 19) {
        r *= kPow10[19];
        p -= 19;
    }
    return r * kPow10[p];
}

// |x| = m * 2^e with m < 2^53; false for zero, denormal and non-finite values
inline bool decompose(double x, uint64_t &m, int &e, bool &neg)
{
    uint64_t bits;
    memcpy(&bits, &x, sizeof bits);
    neg = (bits >> 63) != 0;
    const int be = (int)((bits >> 52) & 0x7ff);
    if (be == 0 || be == 0x7ff) return false;
    m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
    e = be - 1075;
    return true;
}

// round_half_even(n / 2^s), s >= 1
inline u128 shift_round(u128 n, int s)
{
    if (s >= 128) return 0; // n < 2^127 here, so the quotient is below one half
    const u128 q = n >> s, rem = n & ((((u128)1) << s) - 1), half = ((u128)1) << (s - 1);
    return (rem > half || (rem == half && (q & 1))) ? q + 1 : q;
}

// round_half_even(n / d)
inline u128 div_round(u128 n, u128 d)
{
    const u128 q = n / d, rem = n % d;
    const u128 twice = rem * 2; // d < 2^126 in every use
    return (twice > d || (twice == d && (q & 1))) ? q + 1 : q;
}

inline int put_u64(char *out, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (int i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}

} // namespace

// "%f": returns the number of characters written (no terminator); out needs 32 bytes.
int fmt_f6(double x, char *out)
{
    uint64_t m;
    int e;
    bool neg;
    if (x == 0.0) {
        int n = 0;
        if (signbit(x)) out[n++] = '-';
        memcpy(out + n, "0.000000", 8);
        return n + 8;
    }
    if (!decompose(x, m, e, neg) || fabs(x) >= 1e15) return snprintf(out, 400, "%f", x);
    u128 r; // round(|x| * 10^6)
    if (e >= 0)
        r = ((u128)m << e) * 1000000u; // |x| < 1e15 < 2^50
    else
        r = shift_round((u128)m * 1000000u, -e);
    const uint64_t ip = (uint64_t)(r / 1000000u), fp = (uint64_t)(r % 1000000u);
    int n = 0;
    if (neg) out[n++] = '-';
    n += put_u64(out + n, ip);
    out[n++] = '.';
    uint64_t f = fp;
    for (int i = 5; i >= 0; --i) {
        out[n + i] = (char)('0' + f % 10);
        f /= 10;
    }
    return n + 6;
}

// "%e": returns the number of characters written (no terminator); out needs 32 bytes.
int fmt_e6(double x, char *out)
{
    uint64_t m;
    int e;
    bool neg;
    if (x == 0.0) {
        int n = 0;
        if (signbit(x)) out[n++] = '-';
        memcpy(out + n, "0.000000e+00", 12);
        return n + 12;
    }
    if (!decompose(x, m, e, neg)) return snprintf(out, 400, "%e", x);
    const double ax = fabs(x);
    if (ax < 1e-15 || ax >= 1e22) return snprintf(out, 400, "%e", x);
    // decimal exponent estimate from the binary one, corrected below
    int d = (int)floor((double)(e + 52) * 0.30102999566398120);
    u128 digits = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int p = 6 - d; // digits = round(|x| * 10^p)
        if (p >= 0) {
            const u128 n = (u128)m * pow10_128(p); // p <= 22: below 2^53 * 2^74
            digits = e >= 0 ? (n << e) : shift_round(n, -e);
        } else {
            const u128 t = pow10_128(-p);
            digits = e >= 0 ? div_round((u128)m << e, t) : div_round((u128)m, t << (-e));
        }
        if (digits < 1000000u) {
            --d;
        } else if (digits >= 10000000u) {
            // exactly 10^7: the estimate was one short with |x| within rounding of 10^(d+1), or
            // rounding carried 9.9999995... up; both print as 1.000000 with the next exponent
            if (digits == 10000000u) {
                digits = 1000000u;
                ++d;
                break;
            }
            ++d;
        } else {
            break;
        }
    }
    if (digits < 1000000u || digits >= 10000000u) return snprintf(out, 400, "%e", x);
    int n = 0;
    if (neg) out[n++] = '-';
    uint64_t v = (uint64_t)digits;
    char dg[7];
    for (int i = 6; i >= 0; --i) {
        dg[i] = (char)('0' + v % 10);
        v /= 10;
    }
    out[n++] = dg[0];
    out[n++] = '.';
    memcpy(out + n, dg + 1, 6);
    n += 6;
    out[n++] = 'e';
    out[n++] = d < 0 ? '-' : '+';
    const int ad = d < 0 ? -d : d;
    out[n++] = (char)('0' + ad / 10);
    out[n++] = (char)('0' + ad % 10);
    return n;
}

} // namespace mfcc

extern "C" int MFCCFormatValue(double x, int saveType, char *buf, size_t cap)
{
    if (!buf || cap < 400) return mfcc::set_error("MFCCFormatValue needs a 400-byte buffer"), MFCC_E_ARG;
    const int n = saveType == MFCC_SAVE_F ? mfcc::fmt_f6(x, buf) : mfcc::fmt_e6(x, buf);
    buf[n] = 0;
    return n;
}
