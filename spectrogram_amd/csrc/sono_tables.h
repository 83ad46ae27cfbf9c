// sono_tables.h -- host-side builders of the constant tables the kernels read.
#pragma once
#include <cmath>
#include <vector>

#include "sono_core.h"

namespace sono {

// exp(-2*pi*i*k/len) for k in [0, count), computed in f64 and rounded once.
inline std::vector<float2> make_twiddles(unsigned int len, unsigned int count)
{
    std::vector<float2> t(count);
    for (unsigned int k = 0; k < count; ++k) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)len;
        t[k].x = (float)std::cos(a);
        t[k].y = (float)std::sin(a);
    }
    return t;
}

// Inter-pass twiddles of the tiled kernel for N = R1*R2 (R2 = 32 or 64): one row of
// R2 entries W_N^(n2*k1) per k1: rows k1 = 1..R1/2.  R1 = R2 = 64: rows
// k1 = 1..R1/4, then one row with k1 = R1/2 (the kernel derives the rows in
// between from these).
inline std::vector<float2> make_tile_twiddles(unsigned int N, unsigned int r2)
{
    const unsigned int r1 = N / r2, half = r1 / 2;
    std::vector<unsigned int> ks;
    if (r1 == 64 && r2 == 64) {
        for (unsigned int k1 = 1; k1 <= half / 2; ++k1) ks.push_back(k1);
        ks.push_back(half);
    } else {
        for (unsigned int k1 = 1; k1 <= half; ++k1) ks.push_back(k1);
    }
    std::vector<float2> t(ks.size() * r2);
    for (size_t r = 0; r < ks.size(); ++r)
        for (unsigned int n2 = 0; n2 < r2; ++n2) {
            // Rows k1 < R1/2 carry a factor 2: pass 1 hands Y (not 2Y) of a frame's own rows to
            // pass 2, which ends with |2X|^2 in every bin.  The row k1 = R1/2 is shared by the two
            // frames of a pair and doubles in the separation after pass 2.
            const double a = -2.0 * 3.14159265358979323846 * (double)(ks[r] * n2) / (double)N;
            const double g = ks[r] < half ? 2.0 : 1.0;
            t[r * r2 + n2].x = (float)(g * std::cos(a));
            t[r * r2 + n2].y = (float)(g * std::sin(a));
        }
    return t;
}

inline bool supported_fft_size(unsigned int n)
{
    return n >= 256 && n <= 4096 && (n & (n - 1)) == 0;
}

}  // namespace sono
