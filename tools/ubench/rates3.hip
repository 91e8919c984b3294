// tools/ubench/rates2.hip — wall-clock VALU issue rates by instruction kind and waves per SIMD (64 instructions per
// loop iteration, 16 independent accumulators, distinct source registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define X64(S) X16(S) X16(S) X16(S) X16(S)
#define OP_0(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_1(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k & 3]), "s"(msk));
#define OP_2(k) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]) : "vcc");
#define OP_3(k) asm volatile("v_cmp_lt_f32_e64 %1, %2, %3\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[k]), "+s"(msk) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_4(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[k]) : "s"(sb.x));
#define OP_5(k) asm volatile("v_add_f32_e64 %0, |%1|, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_6(k) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_7(k) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_8(k) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[k]), "v"(b[k & 3]) : "vcc");
#define OP_9(k) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_10(k) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_11(k) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_12(k) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(b[k & 3]));
#define OP_13(k) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b[k & 3]), "v"(c[k & 3]));
#define OP_14(k) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[k]) : "v"(b[k & 3]));
#define OP_15(k) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[k]) : "s"(sb));

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int rep, float seed)
{
    float a[16], b[4], c[4];
    f2 p[16], pb[4], pc[4];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = (f2){seed + i, seed - i}; }
    for (int i = 0; i < 4; ++i) { b[i] = seed * 0.5f + i; c[i] = seed * 0.25f - i; pb[i] = (f2){b[i], c[i]}; pc[i] = (f2){c[i], b[i]}; }
    f2 sb = (f2){seed, 2.f};
    asm volatile("" : "+s"(sb));
    unsigned long long msk = 0x5555555555555555ull;
    asm volatile("" : "+s"(msk));
    for (int i = 0; i < rep; ++i) {
        if constexpr (OP == 0) { X64(OP_0) }
        if constexpr (OP == 1) { X64(OP_1) }
        if constexpr (OP == 2) { X64(OP_2) }
        if constexpr (OP == 3) { X64(OP_3) }
        if constexpr (OP == 4) { X64(OP_4) }
        if constexpr (OP == 5) { X64(OP_5) }
        if constexpr (OP == 6) { X64(OP_6) }
        if constexpr (OP == 7) { X64(OP_7) }
        if constexpr (OP == 8) { X64(OP_8) }
        if constexpr (OP == 9) { X64(OP_9) }
        if constexpr (OP == 10) { X64(OP_10) }
        if constexpr (OP == 11) { X64(OP_11) }
        if constexpr (OP == 12) { X64(OP_12) }
        if constexpr (OP == 13) { X64(OP_13) }
        if constexpr (OP == 14) { X64(OP_14) }
        if constexpr (OP == 15) { X64(OP_15) }
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
    run<0>("v_cndmask e32 vcc");
    run<1>("v_cndmask e64 sgpr mask");
    run<2>("v_cmp e32 + cndmask e32 (x2 instr)");
    run<3>("v_cmp e64 + cndmask e64 (x2 instr)");
    run<4>("v_mul_f32_e32 v,s,v");
    run<5>("v_add_f32_e64 |v|");
    run<6>("v_max3_f32");
    run<7>("v_and_b32_e32");
    run<8>("v_cmp_lt_f32_e32 vcc");
    run<9>("v_sub_f32_e32");
    run<10>("v_fma_f32 -v,v,v");
    run<11>("v_mul_f32_e64");
    run<12>("v_min_f32_e32");
    run<13>("v_med3_f32");
    run<14>("v_mov_b32_e32");
    run<15>("v_pk_mul_f32 sgpr op_sel");
    return 0;
}
