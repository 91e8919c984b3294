// tools/ubench/rates_power.hip — librates_power.so: the issue-rate loops of rates2.hip as LONG launches, so that
// tools/rates_power.py can read the SMU's gfx clock beside them (noaa_apt_amd/testing/smu.py) and price an instruction
// in cycles of the clock the chip actually ran at, not of an assumed 2.4 GHz.  Not part of the product.
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define X64(S) X16(S) X16(S) X16(S) X16(S)
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_ADD(k) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb[k & 3]), "v"(pc[k & 3]));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb[k & 3]));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb[k & 3]));
#define OP_PKMULS(k) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[k]) : "s"(sb));
#define OP_MOV(k) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[k]) : "v"(b[k & 3]));

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int rep, float seed)
{
    float a[16], b[4], c[4];
    f2 p[16], pb[4], pc[4];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = (f2){seed + i, seed - i}; }
    for (int i = 0; i < 4; ++i) { b[i] = seed * 0.5f + i; c[i] = seed * 0.25f - i; pb[i] = (f2){b[i], c[i]}; pc[i] = (f2){c[i], b[i]}; }
    f2 sb = (f2){seed, 2.f};
    asm volatile("" : "+s"(sb));
    for (int i = 0; i < rep; ++i) {
        if constexpr (OP == 0) { X64(OP_FMA) }
        if constexpr (OP == 1) { X64(OP_ADD) }
        if constexpr (OP == 2) { X64(OP_PKFMA) }
        if constexpr (OP == 3) { X64(OP_PKMUL) }
        if constexpr (OP == 4) { X64(OP_PKADD) }
        if constexpr (OP == 5) { X64(OP_PKMULS) }
        if constexpr (OP == 6) { X64(OP_MOV) }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// one launch of `rep` iterations of 64 instructions of kind `op` in every wave, `wps` waves per SIMD on every CU;
// returns its duration in ms by HIP events (< 0: error)
extern "C" float rates_power_launch(int op, int wps, int rep)
{
    static float *out = nullptr;
    if (!out && hipMalloc(&out, 4096) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(256 * wps), block(256);
    hipEventRecord(e0);
    switch (op) {
    case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, rep, 1.5f); break;
    case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, rep, 1.5f); break;
    case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, rep, 1.5f); break;
    case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, rep, 1.5f); break;
    case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, rep, 1.5f); break;
    case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, out, rep, 1.5f); break;
    default: hipLaunchKernelGGL(k<6>, grid, block, 0, 0, out, rep, 1.5f); break;
    }
    hipEventRecord(e1);
    float ms = -1.f;
    if (hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms;
}
