// tools/ubench/rates.hip — issue rates of the instructions the fused front end is built from, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rates tools/ubench/rates.hip && /tmp/rates
// Every kernel runs REP iterations of 16 independent instructions of one kind in every wave; the figure printed is
// wave-instructions per SIMD and nanosecond, and the cycles per instruction at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int REP = 4096;

#define KERNEL(name, DECL, BODY)                                                         \
    __global__ void __launch_bounds__(256) name(float *out, int rep, float seed)         \
    {                                                                                     \
        DECL;                                                                             \
        for (int i = 0; i < rep; ++i) {                                                   \
            BODY;                                                                         \
        }                                                                                 \
        float s = 0.f;                                                                    \
        for (int k = 0; k < 16; ++k) s += sink(a[k]);                                     \
        if (s == 12345.678f) out[threadIdx.x] = s;                                        \
    }

__device__ inline float sink(float v) { return v; }
__device__ inline float sink(f2 v) { return v.x + v.y; }

#define DECL_F float a[16]; for (int k = 0; k < 16; ++k) a[k] = seed + k; float b = seed * 0.5f, c = seed * 0.25f
#define DECL_F2 f2 a[16]; for (int k = 0; k < 16; ++k) a[k] = (f2){seed + k, seed - k}; f2 b = {seed * 0.5f, seed}, c = {seed * 0.25f, 1.f}

#define X16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define OP_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_ADD(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_MUL(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define OP_RSQ(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
#define OP_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define OP_MAX3(k) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define OP_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[k]) : "v"(b));
#define OP_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define OP_ADDU(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
#define OP_PKMOV(k) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(a[k]) : "v"(b));
#define OP_PKMULS(k) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a[k]) : "s"(sb));
#define OP_MIXED(k) asm volatile("v_pk_mul_f32 %0, %1, %0\n v_add_f32 %2, %3, %2" : "+v"(a[k]), "+v"(e[k]) : "v"(b), "v"(c));

KERNEL(k_fma, DECL_F, X16(OP_FMA))
KERNEL(k_add, DECL_F, X16(OP_ADD))
KERNEL(k_mul, DECL_F, X16(OP_MUL))
KERNEL(k_pkfma, DECL_F2, X16(OP_PKFMA))
KERNEL(k_pkmul, DECL_F2, X16(OP_PKMUL))
KERNEL(k_pkadd, DECL_F2, X16(OP_PKADD))
KERNEL(k_sqrt, DECL_F, X16(OP_SQRT))
KERNEL(k_rsq, DECL_F, X16(OP_RSQ))
KERNEL(k_rcp, DECL_F, X16(OP_RCP))
KERNEL(k_max3, DECL_F, X16(OP_MAX3))
KERNEL(k_cndmask, DECL_F, X16(OP_CNDMASK))
KERNEL(k_cmp, DECL_F, X16(OP_CMP))
KERNEL(k_addu, DECL_F, X16(OP_ADDU))
KERNEL(k_mov, DECL_F, X16(OP_MOV))
KERNEL(k_pkmov, DECL_F2, X16(OP_PKMOV))
__global__ void __launch_bounds__(256) k_pkmul_sgpr(float *out, int rep, float seed)
{
    DECL_F2;
    f2 sb = (f2){seed, 2.f};
    asm volatile("" : "+s"(sb));
    for (int i = 0; i < rep; ++i) {
        X16(OP_PKMULS)
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += sink(a[k]);
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// 8 transcendental + 8x4 plain: do transcendentals overlap with plain VALU?
__global__ void __launch_bounds__(256) k_sqrt_plus_fma(float *out, int rep, float seed)
{
    float a[16];
    for (int k = 0; k < 16; ++k) a[k] = seed + k;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < rep; ++i) {
        OP_SQRT(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_SQRT(4) OP_FMA(5) OP_FMA(6) OP_FMA(7)
        OP_SQRT(8) OP_FMA(9) OP_FMA(10) OP_FMA(11) OP_SQRT(12) OP_FMA(13) OP_FMA(14) OP_FMA(15)
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += a[k];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// LDS reads: 16 reads per iteration, lane stride `stride` words, `kind` 0: ds_read_b32 x2 (two instr), 1: ds_read2_b32,
// 2: ds_read_b64, 3: ds_read_b128, 4: ds_read_b32
template <int KIND>
__global__ void __launch_bounds__(256) k_lds(float *out, int rep, int stride)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = i;
    __syncthreads();
    const uint32_t base = (threadIdx.x * stride * 4u) % 60000u;
    float acc = 0.f;
    for (int i = 0; i < rep; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (KIND == 1) {
                f2 v;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(base), "n"(2 * k), "n"(2 * k + 1));
                asm volatile("s_waitcnt lgkmcnt(8)");
                acc += 0.f * 0.f;
                asm volatile("" : : "v"(v));
            } else if constexpr (KIND == 2) {
                f2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(8 * k));
                asm volatile("s_waitcnt lgkmcnt(8)");
                asm volatile("" : : "v"(v));
            } else if constexpr (KIND == 3) {
                f4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(16 * k));
                asm volatile("s_waitcnt lgkmcnt(8)");
                asm volatile("" : : "v"(v));
            } else {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(4 * k));
                asm volatile("s_waitcnt lgkmcnt(8)");
                asm volatile("" : : "v"(v));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <typename K>
static double run(K kern, int blocks_per_cu, size_t lds, int arg_i, float arg_f, bool is_lds, int rep)
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        if (is_lds) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, rep, arg_i);
        else hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, rep, arg_f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    // wave-instructions per SIMD: blocks * 4 waves * rep * 16 / 1024 SIMDs
    const double per_simd = double(blocks) * 4 * rep * 16 / 1024.0;
    return ms * 1e6 / per_simd;  // ns per wave-instruction per SIMD
}

int main()
{
#define V(name, k) { double ns = run(k, 8, 0, 0, 1.5f, false, REP); printf("%-22s %.3f ns/instr/SIMD = %.2f cycles @2.4GHz\n", name, ns, ns * 2.4); }
    V("v_fma_f32", k_fma) V("v_add_f32", k_add) V("v_mul_f32", k_mul)
    V("v_pk_fma_f32", k_pkfma) V("v_pk_mul_f32", k_pkmul) V("v_pk_add_f32", k_pkadd) V("v_pk_mul_f32 sgpr", k_pkmul_sgpr)
    V("v_sqrt_f32", k_sqrt) V("v_rsq_f32", k_rsq) V("v_rcp_f32", k_rcp) V("1 sqrt + 3 fma (x4)", k_sqrt_plus_fma)
    V("v_max3_f32", k_max3) V("v_cndmask_b32", k_cndmask) V("v_cmp_lt_f32", k_cmp) V("v_add_u32", k_addu)
    V("v_mov_b32", k_mov) V("v_pk_mov_b32", k_pkmov)
    for (int stride : {50, 51, 100, 101, 1, 2, 4}) {
        for (int bpc : {2}) {
#define L(name, KIND) { hipFuncSetAttribute((const void *)k_lds<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
            double ns = run(k_lds<KIND>, bpc, 65536, stride, 0.f, true, 1024); \
            printf("%-14s stride %3d words, %d blocks/CU: %.3f ns/instr/SIMD = %.2f cycles/instr/SIMD -> %.2f LDS cycles per wave-instr (4 SIMDs share the LDS)\n", name, stride, bpc, ns, ns * 2.4, ns * 2.4 / 4); }
            L("ds_read_b32", 4) L("ds_read2_b32", 1) L("ds_read_b64", 2) L("ds_read_b128", 3)
        }
    }
    return 0;
}
