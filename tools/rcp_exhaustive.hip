// Development check (not part of the product): is  y1 = fma(y0, fma(-x, y0, 1), y0),  y0 = v_rcp_f32(x),
// equal to the IEEE quotient 1.0f/x for EVERY binary32 x in the range the triangle test feeds it?
// Enumerates all 2^32 bit patterns on the device and counts mismatches per variant.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rcp_exhaustive.hip -o /tmp/rcp_exhaustive && /tmp/rcp_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float refine1(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}
__device__ __forceinline__ float refine2(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}

__global__ void check(float lo, float hi, unsigned long long *out /* [0] in range, [1] mism1, [2] mism2, [3] rcp raw mism, [4..] examples */)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n = 0, m1 = 0, m2 = 0, m0 = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float ax = fabsf(x);
        if (!(ax >= lo && ax <= hi)) continue;
        ++n;
        const float ref = 1.0f / x;
        const float a = refine1(x), c = refine2(x), raw = __builtin_amdgcn_rcpf(x);
        if (__float_as_uint(raw) != __float_as_uint(ref)) ++m0;
        if (__float_as_uint(a) != __float_as_uint(ref)) {
            ++m1;
            const unsigned long long slot = atomicAdd(&out[4], 1ull);
            if (slot < 8) out[5 + slot] = b;
        }
        if (__float_as_uint(c) != __float_as_uint(ref)) ++m2;
    }
    atomicAdd(&out[0], n); atomicAdd(&out[1], m1); atomicAdd(&out[2], m2); atomicAdd(&out[3], m0);
}

int main()
{
    unsigned long long *d, h[16];
    hipMalloc(&d, sizeof h);
    const float ranges[][2] = {{0.00001f, 0x1p126f}, {0x1p-126f, 0x1p126f}, {0x1p-60f, 2.0f}};
    for (auto &rg : ranges) {
        hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, rg[0], rg[1], d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("|x| in [%g, %g]: %llu inputs; raw v_rcp_f32 != IEEE: %llu; one step != IEEE: %llu; two steps != IEEE: %llu\n",
               rg[0], rg[1], h[0], h[3], h[1], h[2]);
        for (unsigned long long k = 0; k < (h[4] < 8 ? h[4] : 8); ++k) {
            uint32_t bits = (uint32_t)h[5 + k];
            float x; memcpy(&x, &bits, 4);
            printf("   mismatch at x = %a (0x%08x)\n", x, bits);
        }
    }
    hipFree(d);
    return 0;
}
