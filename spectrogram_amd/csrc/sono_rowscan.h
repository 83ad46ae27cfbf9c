// sono_rowscan.h -- scans along one row of a dB matrix (host, x86-64 SSE2): the first / last entry above a threshold,
// the maximum, the first entry equal to a value.  The measurement stage asks these questions a few thousand times per
// call; each answers exactly what the reference's scalar loops answer (a NaN is never above anything, never a
// maximum and never equal to anything), four entries per instruction.
#pragma once
#include <emmintrin.h>
#include <stddef.h>
#include <stdint.h>

#include <cmath>

namespace sono {

// The largest float <= thr.  For a float x:  (double)x > thr  exactly when  x > float_below(thr), so a row of floats can
// be compared against a double threshold without widening every entry.  (thr itself when it is a float; NaN stays NaN
// and compares false either way.)
inline float float_below(double thr)
{
    float t = (float)thr;
    if ((double)t > thr) t = std::nextafterf(t, -INFINITY);
    return t;
}

// lowest index in [lo, hi] with p[i] > t, or -1
inline int64_t first_gt(const float *p, int64_t lo, int64_t hi, float t)
{
    const __m128 vt = _mm_set1_ps(t);
    int64_t i = lo;
    for (; i + 3 <= hi; i += 4) {
        const int mask = _mm_movemask_ps(_mm_cmpgt_ps(_mm_loadu_ps(p + i), vt));
        if (mask) return i + __builtin_ctz((unsigned int)mask);
    }
    for (; i <= hi; ++i) if (p[i] > t) return i;
    return -1;
}

// highest index in [lo, hi] with p[i] > t, or -1
inline int64_t last_gt(const float *p, int64_t lo, int64_t hi, float t)
{
    const __m128 vt = _mm_set1_ps(t);
    int64_t i = hi;
    for (; i - 3 >= lo; i -= 4) {
        const int mask = _mm_movemask_ps(_mm_cmpgt_ps(_mm_loadu_ps(p + i - 3), vt));
        if (mask) return i - 3 + (31 - __builtin_clz((unsigned int)mask));
    }
    for (; i >= lo; --i) if (p[i] > t) return i;
    return -1;
}

// maximum of p[0..n), NaNs skipped; -INFINITY when no entry is above that
inline float row_max(const float *p, size_t n)
{
    // _mm_max_ps(x, m) hands back its SECOND operand when either is a NaN: the running maximum survives a NaN entry.
    __m128 m0 = _mm_set1_ps(-INFINITY), m1 = m0, m2 = m0, m3 = m0;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        m0 = _mm_max_ps(_mm_loadu_ps(p + i), m0);
        m1 = _mm_max_ps(_mm_loadu_ps(p + i + 4), m1);
        m2 = _mm_max_ps(_mm_loadu_ps(p + i + 8), m2);
        m3 = _mm_max_ps(_mm_loadu_ps(p + i + 12), m3);
    }
    for (; i + 4 <= n; i += 4) m0 = _mm_max_ps(_mm_loadu_ps(p + i), m0);
    m0 = _mm_max_ps(_mm_max_ps(m0, m1), _mm_max_ps(m2, m3));
    m0 = _mm_max_ps(m0, _mm_movehl_ps(m0, m0));
    m0 = _mm_max_ps(m0, _mm_shuffle_ps(m0, m0, 1));
    float m = _mm_cvtss_f32(m0);
    for (; i < n; ++i) if (p[i] > m) m = p[i];
    return m;
}

// lowest index in [0, n) with p[i] == v, or -1
inline int64_t first_eq(const float *p, size_t n, float v)
{
    const __m128 vv = _mm_set1_ps(v);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const int mask = _mm_movemask_ps(_mm_cmpeq_ps(_mm_loadu_ps(p + i), vv));
        if (mask) return (int64_t)i + __builtin_ctz((unsigned int)mask);
    }
    for (; i < n; ++i) if (p[i] == v) return (int64_t)i;
    return -1;
}

// What "for b in 0..n: if p[b] > best { best = p[b]; at = b }" leaves behind, started from best = -INFINITY: the first
// maximum.  -> its index and, in *value, the entry itself; -1 (and -INFINITY) when no entry is above -INFINITY.
inline int64_t first_max(const float *p, size_t n, float *value)
{
    const float m = row_max(p, n);
    *value = -INFINITY;
    if (!(m > -INFINITY)) return -1;
    const int64_t at = first_eq(p, n, m);
    *value = p[at];                                          // the entry met first (a -0 before a +0 stays a -0)
    return at;
}

}  // namespace sono
