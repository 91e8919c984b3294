// tools/ubench/cycles.hip — cycles per VALU instruction on gfx950 from s_memtime, by waves per SIMD, plus the
// effective clock (s_memtime ticks over wall_clock64 ticks, the latter at 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));
#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_ADD(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_ADD3(k) asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_MUL(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb), "v"(pc));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
#define OP_PKMULS(k) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[k]) : "s"(sb));
#define OP_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define OP_MAX3(k) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_MAX(k) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_ADDU(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_MIX(k) asm volatile("v_pk_mul_f32 %0, %2, %0\n v_add_f32 %1, %3, %1" : "+v"(p[k]), "+v"(a[k]) : "v"(pb), "v"(b));
#define OP_MIXF(k) asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %3, %5, %1" : "+v"(p[k]), "+v"(a[k]) : "v"(pb), "v"(b), "v"(pc), "v"(c));
// the strict FIR pair: products then dependent adds (8 MAC pairs per 16 instructions)
#define OP_MACP(k) asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %1, %0" : "=&v"(q[k]), "+v"(p[k]) : "s"(sb), "v"(pb));

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, int rep, float seed)
{
    float a[16];
    f2 p[16], q[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = (f2){seed + i, seed - i}; q[i] = p[i]; }
    float b = seed * 0.5f, c = seed * 0.25f;
    f2 pb = {b, c}, pc = {c, b};
    f2 sb = (f2){seed, 2.f};
    asm volatile("" : "+s"(sb));
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t w0 = wall_clock64();
    for (int i = 0; i < rep; ++i) {
        if constexpr (OP == 0) { X16(OP_FMA) }
        if constexpr (OP == 1) { X16(OP_ADD) }
        if constexpr (OP == 2) { X16(OP_MUL) }
        if constexpr (OP == 3) { X16(OP_PKFMA) }
        if constexpr (OP == 4) { X16(OP_PKMUL) }
        if constexpr (OP == 5) { X16(OP_PKADD) }
        if constexpr (OP == 6) { X16(OP_PKMULS) }
        if constexpr (OP == 7) { X16(OP_SQRT) }
        if constexpr (OP == 8) { X16(OP_MAX3) }
        if constexpr (OP == 9) { X16(OP_MAX) }
        if constexpr (OP == 10) { X16(OP_ADDU) }
        if constexpr (OP == 11) { X16(OP_MIX) }     // 32 instructions
        if constexpr (OP == 12) { X16(OP_MIXF) }    // 32 instructions
        if constexpr (OP == 13) { X16(OP_MACP) }    // 32 instructions
        if constexpr (OP == 14) { X16(OP_ADD3) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + q[i].x;
    if (s == 12345.678f) out[1000] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}

template <int OP>
static void run(const char *name, int per_instr)
{
    uint64_t *out;
    hipMalloc(&out, 16384);
    const int rep = 2048;
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4, 8}) {  // waves per SIMD = 256-thread blocks per CU
        hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, rep, 1.5f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, rep, 1.5f);
        hipDeviceSynchronize();
        uint64_t h[2];
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double instr_per_simd = double(rep) * per_instr * wps;  // the SIMD's waves together
        printf("  w%d: %5.2f cyc/instr (%.2f GHz)", wps, double(h[0]) / instr_per_simd, double(h[0]) / double(h[1]) * 0.1);
    }
    printf("\n");
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32", 16);
    run<1>("v_add_f32_e32", 16);
    run<14>("v_add_f32_e64", 16);
    run<2>("v_mul_f32", 16);
    run<3>("v_pk_fma_f32", 16);
    run<4>("v_pk_mul_f32", 16);
    run<5>("v_pk_add_f32", 16);
    run<6>("v_pk_mul_f32 sgpr op_sel", 16);
    run<7>("v_sqrt_f32", 16);
    run<8>("v_max3_f32", 16);
    run<9>("v_max_f32", 16);
    run<10>("v_add_u32", 16);
    run<11>("pk_mul + add_f32 alternating", 32);
    run<12>("pk_fma + fma_f32 alternating", 32);
    run<13>("pk_mul(sgpr) -> dependent pk_add", 32);
    return 0;
}
