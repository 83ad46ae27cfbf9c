// sono_hostmath.h -- single-precision cos / sin / exp with the exact rounding
// behaviour of the FreeBSD msun routines (k_cosf, k_sinf, e_rem_pio2f, s_cosf,
// s_sinf, e_expf), which is what Rust's f32::cos / sin / exp evaluate to on
// wasm32-unknown-unknown.  The window tables (lib.rs:698-776) are defined
// through these functions, and a window entry that is 1 ulp off changes every
// frame, so the host-side table builder uses them instead of the C library's
// (glibc rounds about 1 % of arguments differently).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace sono {
namespace msun {

inline uint32_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// kernel cosine on [-pi/4, pi/4], evaluated in double
inline float kcos(double x)
{
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57;
    const double C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double z = x * x, w = z * z, r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}

// kernel sine on [-pi/4, pi/4], evaluated in double
inline float ksin(double x)
{
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59;
    const double S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}

// x = n*pi/2 + y, |y| <= pi/4 (medium arguments, |x| < 2^28 * pi/2)
inline int rem_pio2(float x, double &y)
{
    const double toint = 1.5 / 2.220446049250313e-16;
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08;
    const double xd = (double)x;
    volatile double shifted = xd * invpio2 + toint;     // rint() by adding 1.5 * 2^52
    const double fn = shifted - toint;
    y = xd - fn * pio2_1 - fn * pio2_1t;
    return (int)fn;
}

constexpr double kPio2 = 1.57079632679489661923;

inline float cosf(float x)
{
    uint32_t ix = bits_of(x);
    const bool neg = (ix >> 31) != 0;
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? 1.0f : kcos(x);
    if (ix <= 0x407b53d1u) {
        if (ix > 0x4016cbe3u) return -kcos(neg ? x + 2 * kPio2 : x - 2 * kPio2);
        return neg ? ksin(x + kPio2) : ksin(kPio2 - x);
    }
    if (ix <= 0x40e231d5u) {
        if (ix > 0x40afeddfu) return kcos(neg ? x + 4 * kPio2 : x - 4 * kPio2);
        return neg ? ksin(-x - 3 * kPio2) : ksin(x - 3 * kPio2);
    }
    if (ix >= 0x7f800000u) return x - x;
    if (ix >= 0x4dc90fdbu) return std::cos(x);
    double y;
    switch (rem_pio2(x, y) & 3) {
    case 0: return kcos(y);
    case 1: return ksin(-y);
    case 2: return -kcos(y);
    default: return ksin(y);
    }
}

inline float sinf(float x)
{
    uint32_t ix = bits_of(x);
    const bool neg = (ix >> 31) != 0;
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : ksin(x);
    if (ix <= 0x407b53d1u) {
        if (ix <= 0x4016cbe3u) return neg ? -kcos(x + kPio2) : kcos(x - kPio2);
        return ksin(neg ? -(x + 2 * kPio2) : -(x - 2 * kPio2));
    }
    if (ix <= 0x40e231d5u) {
        if (ix <= 0x40afeddfu) return neg ? kcos(x + 3 * kPio2) : -kcos(x - 3 * kPio2);
        return ksin(neg ? x + 4 * kPio2 : x - 4 * kPio2);
    }
    if (ix >= 0x7f800000u) return x - x;
    if (ix >= 0x4dc90fdbu) return std::sin(x);
    double y;
    switch (rem_pio2(x, y) & 3) {
    case 0: return ksin(y);
    case 1: return kcos(y);
    case 2: return ksin(-y);
    default: return -kcos(y);
    }
}

inline float expf(float x)
{
    const float ln2hi = 6.9314575195e-01f, ln2lo = 1.4286067653e-06f, invln2 = 1.4426950216e+00f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = bits_of(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    if (hx >= 0x42aeac50u) {
        if (hx > 0x7f800000u) return x;
        if (hx >= 0x42b17218u && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5u) return 0.0f;
    }
    int k;
    float hi, lo;
    if (hx > 0x3eb17218u) {
        if (hx > 0x3f851592u) k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        else k = 1 - sign - sign;
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000u) {
        k = 0; hi = x; lo = 0.0f;
    } else {
        return 1.0f + x;
    }
    const float xx = x * x;
    const float c = x - xx * (P1 + xx * P2);
    const float y = 1.0f + (x * c / (2.0f - c) - lo + hi);
    return k == 0 ? y : std::scalbn(y, k);
}

}  // namespace msun

// Window table of lib.rs:698-776, every expression in f32 and in the
// reference's evaluation order.  Unknown names select Hann (lib.rs:767-772).
inline void make_window(const char *name, unsigned int size, float alpha, float *w)
{
    const float PI = 3.14159274101257324219f;
    const float fs = (float)size;
    const char *n = name ? name : "";
    auto is = [&](const char *s) { return std::strcmp(n, s) == 0; };
    for (unsigned int i = 0; i < size; ++i) {
        const float fi = (float)i;
        float v;
        if (is("bartlett")) {
            v = 2.0f / (fs - 1.0f) * ((fs - 1.0f) / 2.0f - std::fabs(fi - (fs - 1.0f) / 2.0f));
        } else if (is("bartlettHann")) {
            const float ni = fi / (fs - 1.0f);
            v = 0.62f - 0.48f * std::fabs(ni - 0.5f) - 0.38f * msun::cosf(2.0f * PI * ni);
        } else if (is("blackman")) {
            v = (1.0f - alpha) / 2.0f - 0.5f * msun::cosf(2.0f * PI * fi / (fs - 1.0f))
                + alpha / 2.0f * msun::cosf(4.0f * PI * fi / (fs - 1.0f));
        } else if (is("cosine")) {
            v = msun::cosf(PI * fi / (fs - 1.0f) - PI / 2.0f);
        } else if (is("gauss")) {
            const float sigma = 0.25f * (fs - 1.0f) / 2.0f;
            const float x = (fi - (fs - 1.0f) / 2.0f) / sigma;
            v = msun::expf(-0.5f * x * x);
        } else if (is("hamming")) {
            v = 0.54f - 0.46f * msun::cosf(2.0f * PI * fi / (fs - 1.0f));
        } else if (is("lanczos")) {
            const float x = 2.0f * fi / (fs - 1.0f) - 1.0f;
            const float pix = PI * x;
            v = std::fabs(pix) < 1e-6f ? 1.0f : msun::sinf(pix) / pix;
        } else if (is("rectangular")) {
            v = 1.0f;
        } else if (is("triangular")) {
            v = 2.0f / fs * (fs / 2.0f - std::fabs(fi - (fs - 1.0f) / 2.0f));
        } else {
            v = 0.5f * (1.0f - msun::cosf(2.0f * PI * fi / (fs - 1.0f)));
        }
        w[i] = v;
    }
}

}  // namespace sono
