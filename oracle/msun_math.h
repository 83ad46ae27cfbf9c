/*
 * msun_math.h -- TEST INFRASTRUCTURE (part of the CPU oracle).
 *
 * The reference is built for wasm32-unknown-unknown, where Rust's f32::cos,
 * f32::sin, f32::exp and f32::log10 (used at lib.rs:713-752,281,1114) resolve to
 * the pure-Rust `libm` crate, a port of the FreeBSD msun / musl single-precision
 * routines.  glibc's cosf/expf are different (equally accurate) implementations
 * and differ from them by 1 ulp on roughly 1 % of arguments, which is enough to
 * change window-table entries.  These are restatements of the published msun
 * algorithms (k_cosf.c, k_sinf.c, e_rem_pio2f.c, s_cosf.c, s_sinf.c, e_expf.c,
 * e_log10f.c) so that the oracle's window tables are bit-identical to the
 * reference's.  cosf/sinf/expf are pinned by the golden window tables in
 * tests/golden; log10f is only pinned indirectly (u8 frames).
 */
#ifndef SO_MSUN_MATH_H
#define SO_MSUN_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t so_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float so_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* k_cosf.c: |cos(x) - c(x)| < 2**-34.1 on [-pi/4, pi/4] */
static inline float so_k_cosdf(double x)
{
    static const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57,
                        C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    double z = x * x, w = z * z, r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}

/* k_sinf.c: |sin(x)/x - s(x)| < 2**-37.5 on [-pi/4, pi/4] */
static inline float so_k_sindf(double x)
{
    static const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59,
                        S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}

/* e_rem_pio2f.c, medium-size branch only (|x| < 2^28 * pi/2); window arguments
 * never exceed 4*pi.  Larger arguments fall back to the C library. */
static inline int so_rem_pio2f(float x, double *y)
{
    static const double toint = 1.5 / 2.220446049250313e-16,
                        invpio2 = 6.36619772367581382433e-01,
                        pio2_1 = 1.57079631090164184570e+00,
                        pio2_1t = 1.58932547735281966916e-08;
    double x64 = (double)x;
    volatile double tmp = x64 * invpio2 + toint;
    double fn = tmp - toint;
    *y = x64 - fn * pio2_1 - fn * pio2_1t;
    return (int)fn;
}

static inline float so_cosf(float x)
{
    static const double c1pio2 = 1 * 1.57079632679489661923, c2pio2 = 2 * 1.57079632679489661923,
                        c3pio2 = 3 * 1.57079632679489661923, c4pio2 = 4 * 1.57079632679489661923;
    uint32_t ix = so_f2u(x);
    unsigned sign = ix >> 31;
    double y;
    int n;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {                    /* |x| ~<= pi/4 */
        if (ix < 0x39800000) return 1.0f;      /* |x| < 2**-12 */
        return so_k_cosdf(x);
    }
    if (ix <= 0x407b53d1) {                    /* |x| ~<= 5*pi/4 */
        if (ix > 0x4016cbe3)                   /* |x| ~> 3*pi/4 */
            return -so_k_cosdf(sign ? x + c2pio2 : x - c2pio2);
        return sign ? so_k_sindf(x + c1pio2) : so_k_sindf(c1pio2 - x);
    }
    if (ix <= 0x40e231d5) {                    /* |x| ~<= 9*pi/4 */
        if (ix > 0x40afeddf)                   /* |x| ~> 7*pi/4 */
            return so_k_cosdf(sign ? x + c4pio2 : x - c4pio2);
        return sign ? so_k_sindf(-x - c3pio2) : so_k_sindf(x - c3pio2);
    }
    if (ix >= 0x7f800000) return x - x;
    if (ix >= 0x4dc90fdb) return cosf(x);
    n = so_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return so_k_cosdf(y);
    case 1: return so_k_sindf(-y);
    case 2: return -so_k_cosdf(y);
    default: return so_k_sindf(y);
    }
}

static inline float so_sinf(float x)
{
    static const double s1pio2 = 1 * 1.57079632679489661923, s2pio2 = 2 * 1.57079632679489661923,
                        s3pio2 = 3 * 1.57079632679489661923, s4pio2 = 4 * 1.57079632679489661923;
    uint32_t ix = so_f2u(x);
    int sign = (int)(ix >> 31), n;
    double y;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {
        if (ix < 0x39800000) return x;
        return so_k_sindf(x);
    }
    if (ix <= 0x407b53d1) {
        if (ix <= 0x4016cbe3)
            return sign ? -so_k_cosdf(x + s1pio2) : so_k_cosdf(x - s1pio2);
        return so_k_sindf(sign ? -(x + s2pio2) : -(x - s2pio2));
    }
    if (ix <= 0x40e231d5) {
        if (ix <= 0x40afeddf)
            return sign ? so_k_cosdf(x + s3pio2) : -so_k_cosdf(x - s3pio2);
        return so_k_sindf(sign ? x + s4pio2 : x - s4pio2);
    }
    if (ix >= 0x7f800000) return x - x;
    if (ix >= 0x4dc90fdb) return sinf(x);
    n = so_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return so_k_sindf(y);
    case 1: return so_k_cosdf(y);
    case 2: return so_k_sindf(-y);
    default: return -so_k_cosdf(y);
    }
}

/* e_expf.c (the pre-2017 msun form the Rust libm crate carries) */
static inline float so_expf(float x)
{
    static const float half[2] = {0.5f, -0.5f};
    static const float ln2hi = 6.9314575195e-01f, ln2lo = 1.4286067653e-06f,
                       invln2 = 1.4426950216e+00f, P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    float hi, lo, c, xx, y;
    int k, sign;
    uint32_t hx = so_f2u(x);
    sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x42aeac50) {                    /* |x| >= 87.33655 or NaN */
        if (hx > 0x7f800000) return x;
        if (hx >= 0x42b17218 && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5) return 0.0f;
    }
    if (hx > 0x3eb17218) {                     /* |x| > 0.5 ln2 */
        if (hx > 0x3f851592) k = (int)(invln2 * x + half[sign]);
        else k = 1 - sign - sign;
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000) {              /* |x| > 2**-14 */
        k = 0; hi = x; lo = 0.0f;
    } else {
        return 1.0f + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * P2);
    y = 1.0f + (x * c / (2.0f - c) - lo + hi);
    return k == 0 ? y : scalbnf(y, k);
}

/* e_log10f.c */
static inline float so_log10f(float x)
{
    static const float ivln10hi = 4.3432617188e-01f, ivln10lo = -3.1689971365e-05f,
                       log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f,
                       Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f,
                       Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    float hfsq, f, s, z, R, w, t1, t2, dk, hi, lo;
    uint32_t ix = so_f2u(x);
    int k = 0;
    if (ix < 0x00800000 || (ix >> 31)) {
        if ((ix << 1) == 0) return -1.0f / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25; x *= 0x1p25f; ix = so_f2u(x);
    } else if (ix >= 0x7f800000) {
        return x;
    } else if (ix == 0x3f800000) {
        return 0.0f;
    }
    ix += 0x3f800000 - 0x3f3504f3;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffff) + 0x3f3504f3;
    x = so_u2f(ix);
    f = x - 1.0f;
    s = f / (2.0f + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * Lg4);
    t2 = z * (Lg1 + w * Lg3);
    R = t2 + t1;
    hfsq = 0.5f * f * f;
    hi = f - hfsq;
    hi = so_u2f(so_f2u(hi) & 0xfffff000);
    lo = (f - hi) - hfsq + s * (hfsq + R);
    dk = (float)k;
    return dk * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi + hi * ivln10hi + dk * log10_2hi;
}

#endif
