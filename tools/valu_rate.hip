// Development check (not part of the product): how many cycles a SIMD of gfx950 spends per wave64 vector instruction of the
// kinds the walk is made of, at the walk's occupancy (8 wavefronts per SIMD) and at 1 and 2 per SIMD — v_fma_f32 on vector
// registers only, with one scalar-register operand (how the walk's records arrive), v_min/v_max (VOP2), v_max3 (VOP3),
// v_cmp to VCC and to a scalar pair — measured with s_memtime around 8192 instructions per wavefront, median over wavefronts.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                                                     \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
    for (int i = 0; i < 128; ++i) { asm volatile(REP8(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(k), "v"(b) : "vcc", "s20", "s21"); } \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                               \
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;

#define EIGHT(op, tail) op " %0, " tail "\n\t" op " %1, " tail "\n\t" op " %2, " tail "\n\t" op " %3, " tail "\n\t" op " %4, " tail "\n\t" op " %5, " tail "\n\t" op " %6, " tail "\n\t" op " %7, " tail "\n\t"

__global__ void k_fma_vvv(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_fma_f32 %0, %0, %9, %9\n\tv_fma_f32 %1, %1, %9, %9\n\tv_fma_f32 %2, %2, %9, %9\n\tv_fma_f32 %3, %3, %9, %9\n\tv_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t") }
__global__ void k_fma_svv(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %1, %8, %1, %9\n\tv_fma_f32 %2, %8, %2, %9\n\tv_fma_f32 %3, %8, %3, %9\n\tv_fma_f32 %4, %8, %4, %9\n\tv_fma_f32 %5, %8, %5, %9\n\tv_fma_f32 %6, %8, %6, %9\n\tv_fma_f32 %7, %8, %7, %9\n\t") }
__global__ void k_fma_svv_neg(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_fma_f32 %0, %8, %0, -%9\n\tv_fma_f32 %1, %8, %1, -%9\n\tv_fma_f32 %2, %8, %2, -%9\n\tv_fma_f32 %3, %8, %3, -%9\n\tv_fma_f32 %4, %8, %4, -%9\n\tv_fma_f32 %5, %8, %5, -%9\n\tv_fma_f32 %6, %8, %6, -%9\n\tv_fma_f32 %7, %8, %7, -%9\n\t") }
__global__ void k_mul_vv(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_mul_f32 %0, %0, %9\n\tv_mul_f32 %1, %1, %9\n\tv_mul_f32 %2, %2, %9\n\tv_mul_f32 %3, %3, %9\n\tv_mul_f32 %4, %4, %9\n\tv_mul_f32 %5, %5, %9\n\tv_mul_f32 %6, %6, %9\n\tv_mul_f32 %7, %7, %9\n\t") }
__global__ void k_mul_sv(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_mul_f32 %0, %8, %0\n\tv_mul_f32 %1, %8, %1\n\tv_mul_f32 %2, %8, %2\n\tv_mul_f32 %3, %8, %3\n\tv_mul_f32 %4, %8, %4\n\tv_mul_f32 %5, %8, %5\n\tv_mul_f32 %6, %8, %6\n\tv_mul_f32 %7, %8, %7\n\t") }
__global__ void k_max_vv(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_max_f32 %0, %0, %9\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %9\n\tv_max_f32 %3, %3, %9\n\tv_max_f32 %4, %4, %9\n\tv_max_f32 %5, %5, %9\n\tv_max_f32 %6, %6, %9\n\tv_max_f32 %7, %7, %9\n\t") }
__global__ void k_max3(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_max3_f32 %0, %0, %9, %1\n\tv_max3_f32 %1, %1, %9, %2\n\tv_max3_f32 %2, %2, %9, %3\n\tv_max3_f32 %3, %3, %9, %4\n\tv_max3_f32 %4, %4, %9, %5\n\tv_max3_f32 %5, %5, %9, %6\n\tv_max3_f32 %6, %6, %9, %7\n\tv_max3_f32 %7, %7, %9, %0\n\t") }
__global__ void k_cmp_vcc(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_cmp_ngt_f32 vcc, %0, %9\n\tv_cmp_ngt_f32 vcc, %1, %9\n\tv_cmp_ngt_f32 vcc, %2, %9\n\tv_cmp_ngt_f32 vcc, %3, %9\n\tv_cmp_ngt_f32 vcc, %4, %9\n\tv_cmp_ngt_f32 vcc, %5, %9\n\tv_cmp_ngt_f32 vcc, %6, %9\n\tv_cmp_ngt_f32 vcc, %7, %9\n\t") }
__global__ void k_cmp_sgpr(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_cmp_ngt_f32 s[20:21], %0, %9\n\tv_cmp_ngt_f32 s[20:21], %1, %9\n\tv_cmp_ngt_f32 s[20:21], %2, %9\n\tv_cmp_ngt_f32 s[20:21], %3, %9\n\tv_cmp_ngt_f32 s[20:21], %4, %9\n\tv_cmp_ngt_f32 s[20:21], %5, %9\n\tv_cmp_ngt_f32 s[20:21], %6, %9\n\tv_cmp_ngt_f32 s[20:21], %7, %9\n\t") }
__global__ void k_fma_mix(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %8, %1, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %8, %3, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %8, %5, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %6, %8, %6, %9 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %8, %7, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t") }
// a dependent chain (each instruction needs the previous one's result): what a lone wavefront pays per instruction
__global__ void k_fma_chain(float *out, unsigned long long *ticks, float seed, float k, float b) { BODY("v_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\tv_fma_f32 %0, %8, %0, %9\n\t") }

typedef void (*kern)(float *, unsigned long long *, float, float, float);

int main()
{
    struct { const char *name; kern k; } ks[] = {{"v_fma_f32 v, v, v, v", k_fma_vvv}, {"v_fma_f32 v, s, v, v", k_fma_svv}, {"v_fma_f32 v, s, v, -v", k_fma_svv_neg},
                                                 {"v_mul_f32 v, v, v", k_mul_vv}, {"v_mul_f32 v, s, v", k_mul_sv}, {"v_max_f32 v, v, v", k_max_vv},
                                                 {"v_max3_f32 v, v, v, v", k_max3}, {"v_cmp_ngt_f32 vcc, v, v", k_cmp_vcc}, {"v_cmp_ngt_f32 s[..], v, v", k_cmp_sgpr},
                                                 {"v_fma_mix_f32 v, s(f16 half), v, v", k_fma_mix}, {"v_fma_f32 v, s, v, v (dependent chain)", k_fma_chain}};
    float *out;
    unsigned long long *ticks;
    const int blocks = 256;                                     // one workgroup per compute unit
    hipMalloc(&out, sizeof(float) * blocks * 2048);
    hipMalloc(&ticks, 8 * blocks * 32);
    for (auto &e : ks) {
        printf("%-42s", e.name);
        for (int waves_per_simd : {1, 2, 4, 8}) {
            const int threads = 64 * 4 * waves_per_simd;        // 4 SIMDs per compute unit
            if (threads > 1024) {                               // 8 per SIMD = two workgroups of 1024 per compute unit
                hipLaunchKernelGGL(e.k, dim3(blocks * 2), dim3(1024), 0, 0, out, ticks, 1.0f, 1.0000001f, 0.5f);
            } else {
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, ticks, 1.0f, 1.0000001f, 0.5f);
            }
            hipDeviceSynchronize();
            const int n = (threads > 1024 ? blocks * 2 * 16 : blocks * threads / 64);
            std::vector<unsigned long long> t(n);
            hipMemcpy(t.data(), ticks, 8 * n, hipMemcpyDeviceToHost);
            std::sort(t.begin(), t.end());
            // s_memtime ticks are shader cycles (MI355X_MICROARCH.md); 8192 instructions per wavefront; a SIMD ran waves_per_simd of them
            const double per_wave = double(t[n / 2]) / 8192.0;
            printf("  %d/SIMD: %5.2f cyc per instr per wave = %5.2f per SIMD", waves_per_simd, per_wave, per_wave / waves_per_simd);
        }
        printf("\n");
    }
    return 0;
}
