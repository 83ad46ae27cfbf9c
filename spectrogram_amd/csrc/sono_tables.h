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

// Inter-pass twiddles of the tiled kernel for N = 32*R1: row k1-1 (k1 = 1..R1/2)
// holds W_N^(n2*k1) for n2 in [0, 32).
inline std::vector<float2> make_tile_twiddles(unsigned int N)
{
    const unsigned int half = N / 64;
    std::vector<float2> t((size_t)half * 32);
    for (unsigned int k1 = 1; k1 <= half; ++k1)
        for (unsigned int n2 = 0; n2 < 32; ++n2) {
            const double a = -2.0 * 3.14159265358979323846 * (double)(k1 * n2) / (double)N;
            t[(size_t)(k1 - 1) * 32 + n2].x = (float)std::cos(a);
            t[(size_t)(k1 - 1) * 32 + n2].y = (float)std::sin(a);
        }
    return t;
}

inline bool supported_fft_size(unsigned int n)
{
    return n >= 256 && n <= 4096 && (n & (n - 1)) == 0;
}

}  // namespace sono
