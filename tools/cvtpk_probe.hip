// cvtpk_probe.hip -- prints what v_cvt_pk_u8_f32 does with fractions, negatives,
// values above 255, infinities and NaN (is it truncating?  saturating?).
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/cvtpk_probe tools/cvtpk_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

__global__ void probe(const float *in, unsigned int *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}

int main()
{
    const float v[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 127.999f, 254.5f, 254.999f, 255.0f,
                       255.5f, 256.0f, 300.0f, 1e10f, INFINITY, -0.3f, -0.5f, -1.0f, -300.0f, -INFINITY, NAN};
    const int n = sizeof(v) / sizeof(v[0]);
    float *din; unsigned int *dout, h[64];
    if (hipMalloc(&din, sizeof(v)) != hipSuccess || hipMalloc(&dout, 4 * n) != hipSuccess) return 1;
    hipMemcpy(din, v, sizeof(v), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, dout, n);
    if (hipMemcpy(h, dout, 4 * n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) printf("%12g -> %08x (byte %u)\n", v[i], h[i], (h[i] >> 8) & 255u);
    return 0;
}
