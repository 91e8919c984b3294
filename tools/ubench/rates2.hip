// tools/ubench/rates2.hip — wall-clock VALU issue rates by instruction kind and waves per SIMD (64 instructions per
// loop iteration, 16 independent accumulators, distinct source registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define X64(S) X16(S) X16(S) X16(S) X16(S)
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_FMAC(k) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_FMAS(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "s"(sb.x), "v"(c[k & 3]));
#define OP_MULFMA(k) asm volatile("v_fma_f32 %0, %1, %0, -0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_ADDFMA(k) asm volatile("v_fma_f32 %0, %1, 1.0, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_ADD(k) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_MUL(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb[k & 3]), "v"(pc[k & 3]));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb[k & 3]));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb[k & 3]));
#define OP_MAX(k) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_ADDU(k) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_CNDMASK(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_RSQ(k) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(a[k]));
#define OP_SQRT(k) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[k]));

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
        if constexpr (OP == 1) { X64(OP_FMAC) }
        if constexpr (OP == 2) { X64(OP_FMAS) }
        if constexpr (OP == 3) { X64(OP_MULFMA) }
        if constexpr (OP == 4) { X64(OP_ADDFMA) }
        if constexpr (OP == 5) { X64(OP_ADD) }
        if constexpr (OP == 6) { X64(OP_MUL) }
        if constexpr (OP == 7) { X64(OP_PKFMA) }
        if constexpr (OP == 8) { X64(OP_PKMUL) }
        if constexpr (OP == 9) { X64(OP_PKADD) }
        if constexpr (OP == 10) { X64(OP_MAX) }
        if constexpr (OP == 11) { X64(OP_ADDU) }
        if constexpr (OP == 12) { X64(OP_LSHLADD) }
        if constexpr (OP == 13) { X64(OP_CNDMASK) }
        if constexpr (OP == 14) { X64(OP_RSQ) }
        if constexpr (OP == 15) { X64(OP_SQRT) }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char *name)
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-26s", name);
    const int rep = 1024;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, rep, 1.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  w%d %.2f", wps, ms * 1e6 / (double(rep) * 64 * wps));  // ns per instruction per SIMD
    }
    printf("   ns/instr/SIMD\n");
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32 (vvv)");
    run<1>("v_fmac_f32_e32");
    run<2>("v_fma_f32 (svv)");
    run<3>("v_fma_f32 a,b,-0 (mul)");
    run<4>("v_fma_f32 a,1.0,b (add)");
    run<5>("v_add_f32_e32");
    run<6>("v_mul_f32_e32");
    run<7>("v_pk_fma_f32");
    run<8>("v_pk_mul_f32");
    run<9>("v_pk_add_f32");
    run<10>("v_max_f32_e32");
    run<11>("v_add_u32_e32");
    run<12>("v_lshl_add_u32");
    run<13>("v_cndmask_b32_e32");
    run<14>("v_rsq_f32");
    run<15>("v_sqrt_f32");
    return 0;
}
