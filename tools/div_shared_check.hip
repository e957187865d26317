// Development check (not part of the product): three quotients a_k / b that share the divisor, computed as
//   y = 1/b (v_rcp_f32 + one Newton step, exhaustively verified in rcp_exhaustive.hip)
//   q0 = a*y; r0 = fma(-b,q0,a); q1 = fma(r0,y,q0); r1 = fma(-b,q1,a); q = fma(r1,y,q1)
// which is the instruction sequence of the compiler's IEEE division with its range scaling (v_div_scale) and
// special-case fix-up (v_div_fixup) left out; inside 2^-60 <= |a| <= |b'|, 2^-60 <= b <= 2^60 neither does anything.
// Compares against a / b on 2^37 pseudo-random pairs plus significand edge patterns.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_shared_check.hip -o tools/div_shared_check
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
__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a float with a chosen exponent window and random or edge significand
__device__ __forceinline__ float make(uint64_t bits, int emin, int emax, uint32_t mode)
{
    const uint32_t e = 127 + emin + (uint32_t)((bits >> 40) % (uint32_t)(emax - emin + 1));
    uint32_t m = (uint32_t)bits & 0x7FFFFFu;
    switch (mode & 7u) {
    case 1: m = 0; break;                       // power of two
    case 2: m = 0x7FFFFFu; break;               // all ones
    case 3: m &= 0x7u; break;                   // just above a power of two
    case 4: m |= 0x7FFFF8u; break;              // just below
    default: break;
    }
    return __uint_as_float((uint32_t)((bits >> 63) << 31) | (e << 23) | m);
}

__global__ void check(uint64_t seed, uint64_t per_thread, unsigned long long *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mism = 0, n = 0;
    uint64_t s = mix(seed ^ tid);
    for (uint64_t k = 0; k < per_thread; ++k) {
        s = mix(s);
        const uint64_t s2 = mix(s ^ 0xABCDEFull);
        const float b = fabsf(make(s, -60, 60, (uint32_t)(s >> 56)));
        // the numerators of interest satisfy |a| <= b (components of a vector divided by its length) — also test beyond
        const int eb = (int)((__float_as_uint(b) >> 23) & 0xFF) - 127;
        int lo = eb - 40, hi = eb + ((s2 >> 50) & 1 ? 0 : 3);
        if (lo < -60) lo = -60;
        if (hi > 60) hi = 60;
        if (hi < lo) hi = lo;
        const float a = make(s2, lo, hi, (uint32_t)(s2 >> 56));
        const float y = recip(b);
        const float q = quot(a, b, y);
        const float ref = a / b;
        ++n;
        if (__float_as_uint(q) != __float_as_uint(ref)) {
            ++mism;
            const unsigned long long slot = atomicAdd(&out[2], 1ull);
            if (slot < 6) { out[3 + 2 * slot] = __float_as_uint(a); out[4 + 2 * slot] = __float_as_uint(b); }
        }
    }
    atomicAdd(&out[0], n);
    atomicAdd(&out[1], mism);
}

int main()
{
    unsigned long long *d, h[16] = {0};
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    const uint64_t threads = 4096ull * 256ull, per_thread = (1ull << 37) / threads;
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, 20261004ull, per_thread, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("pairs %llu, quotients differing from a / b: %llu\n", h[0], h[1]);
    for (unsigned long long k = 0; k < (h[2] < 6 ? h[2] : 6); ++k) {
        uint32_t ab = (uint32_t)h[3 + 2 * k], bb = (uint32_t)h[4 + 2 * k];
        printf("   a = 0x%08x  b = 0x%08x\n", ab, bb);
    }
    hipFree(d);
    return h[1] != 0;
}
