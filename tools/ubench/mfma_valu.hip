// tools/ubench/mfma_valu.hip — does an f32 MFMA stream (v_mfma_f32_16x16x4_f32) run beside packed-f32 VALU work of
// OTHER waves on the same SIMD?  Blocks of 256 threads (4 waves, one per SIMD), `bpc` blocks per CU; blocks with
// (blockIdx.x / 256) < n_mfma run the MFMA loop, the others the VALU loop.  Prints the wall time of each mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k(float *out, int rep_valu, int rep_mfma, int n_mfma_blocks_per_cu, float seed)
{
    const bool mf = (int)(blockIdx.x / 256) < n_mfma_blocks_per_cu;
    float s = 0.f;
    if (mf) {
        f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        float a = seed + threadIdx.x, b = seed - threadIdx.x;
        for (int i = 0; i < rep_mfma; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            }
        }
        s = c0.x + c1.y;
    } else {
        f2 p[16];
        for (int i = 0; i < 16; ++i) p[i] = (f2){seed + i, seed - i};
        f2 pb = {seed, 0.5f}, pc = {0.25f, seed};
        for (int i = 0; i < rep_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u]) : "v"(pb), "v"(pc));
        }
        for (int i = 0; i < 16; ++i) s += p[i].x;
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static float run(int bpc, int n_mfma, int rep_valu, int rep_mfma)
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * bpc), dim3(256), 0, 0, out, rep_valu, rep_mfma, n_mfma, 1.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipFree(out);
    return ms;
}

int main()
{
    const int RV = 8192, RM = 2048;  // per wave: 131072 pk_fma / 32768 MFMAs
    printf("3 VALU waves/SIMD alone:            %.3f ms\n", run(3, 0, RV, RM));
    printf("2 VALU waves/SIMD alone:            %.3f ms\n", run(2, 0, RV, RM));
    printf("1 MFMA wave/SIMD alone:             %.3f ms  (%.1f cycles per MFMA at 2.4 GHz)\n", run(1, 1, RV, RM), run(1, 1, RV, RM) * 2.4e6 / (RM * 16.0));
    printf("1 MFMA + 2 VALU waves/SIMD:         %.3f ms\n", run(3, 1, RV, RM));
    printf("1 MFMA + 3 VALU waves/SIMD:         %.3f ms\n", run(4, 1, RV, RM));
    printf("2 MFMA + 2 VALU waves/SIMD:         %.3f ms\n", run(4, 2, RV, RM));
    // the stage-4 proportion: MFMA work ~1/3 of the VALU time
    printf("1 MFMA (1/3 as long) + 2 VALU:      %.3f ms\n", run(3, 1, RV, RM / 3));
    return 0;
}
