// Development check (not part of the product): x / d for a divisor d that a whole launch shares (rtx_kernel.hip:
// div_denom — the sample's contribution (colour * |n.l|) / (NB_RAY * NB_LIGHT_SAMPLE), main.rs:211-215), computed as
//   y = 1/d (v_rcp_f32 + one Newton step), q0 = x*y; r0 = fma(-d,q0,x); q1 = fma(r0,y,q0); r1 = fma(-d,q1,x); q = fma(r1,y,q1)
// — the compiler's IEEE division without its range scaling and fix-up — against x / d for EVERY binary32 x in
// [2^-60, 2^60] and a list of divisors in [1, 2^30] (the range div_denom admits; 2.0e9 inputs per divisor).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_denom_check.hip -o tools/div_denom_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float recip(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(y0, __builtin_fmaf(-x, y0, 1.0f), y0);
}
__device__ __forceinline__ float quot(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

__global__ void check(float d, uint32_t first, uint32_t count, unsigned long long *mism)
{
    const float y = recip(d);
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + (uint32_t)i);
        volatile float dd = d;
        const float ref = x / dd;
        if (__float_as_uint(quot(x, d, y)) != __float_as_uint(ref)) ++m;
    }
    if (m) atomicAdd(mism, m);
}

int main()
{
    const float divisors[] = {1.0f, 2.0f, 3.0f, 7.0f, 10.0f, 64.0f, 100.0f, 200.0f, 255.0f, 400.0f, 1000.0f, 4096.0f, 10000.0f, 12345.0f,
                              65535.0f, 1000000.0f, 16777215.0f, 1073741824.0f};
    unsigned long long *d_m;
    hipMalloc(&d_m, 8);
    const uint32_t first = (127u - 60u) << 23, last = ((127u + 60u) << 23) + 1u;     // [2^-60, 2^60]
    unsigned long long total = 0, bad = 0;
    for (float d : divisors) {
        hipMemset(d_m, 0, 8);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, first, last - first, d_m);
        unsigned long long m = 0;
        hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost);
        printf("d = %.1f: %u inputs, %llu differences\n", d, last - first, m);
        total += last - first;
        bad += m;
    }
    printf("%llu inputs, %llu differences\n", total, bad);
    return bad != 0;
}
