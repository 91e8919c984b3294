// tools/ubench/mfma_valu2.hip — does a 16-bit MFMA stream (v_mfma_f32_16x16x32_f16) run beside VALU work on the same
// SIMD?  (mfma_valu.hip asked that of the f32 MFMA and packed-f32 work and found the times add up.)
//   across waves: blocks of 256 threads (one wave per SIMD), `bpc` blocks per CU, the first n of every CU's blocks run
//                 the MFMA loop, the others a VALU loop of kind K (0 v_pk_fma_f32, 1 plain v_fma_f32, 2 v_add_u32,
//                 3 v_cvt_pkrtz_f16_f32)
//   inside a wave: one MFMA followed by `fill` VALU instructions of kind K, repeated
// MK: 0 v_mfma_f32_16x16x4_f32, 1 v_mfma_f32_16x16x32_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int K>
__device__ __forceinline__ void valu_op(f2 &p, f2 pb, f2 pc)
{
    if constexpr (K == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(pb), "v"(pc));
    else if constexpr (K == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p.x) : "v"(pb.x), "v"(pc.x));
    else if constexpr (K == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(p.x) : "v"(pb.x));
    else asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "+v"(p.x) : "v"(pb.x), "v"(pc.x));
}

template <int MK, int K>
__global__ void __launch_bounds__(256) k_mix(float *out, int rep_valu, int rep_mfma, int n_mfma_blocks_per_cu, int bpc, float seed)
{
    // (blocks are dealt to CUs round-robin in practice: block b of every group of `bpc` consecutive ones is not
    // guaranteed to share a CU with the others — the split is by blockIdx / 256, as in mfma_valu.hip)
    const bool mf = (int)(blockIdx.x / 256) < n_mfma_blocks_per_cu;
    float s = 0.f;
    if (mf) {
        f4 c[4] = {};
        if constexpr (MK == 0) {
            float a = seed + threadIdx.x, b = seed - threadIdx.x;
            for (int i = 0; i < rep_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u & 3], 0, 0, 0);
            }
        } else {
            h8 a, b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = static_cast<_Float16>(seed + j);
                b[j] = static_cast<_Float16>(seed - j * 0.25f);
            }
            for (int i = 0; i < rep_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u & 3], 0, 0, 0);
            }
        }
        s = c[0].x + c[1].y + c[2].z + c[3].w;
    } else {
        f2 p[16];
        for (int i = 0; i < 16; ++i) p[i] = (f2){seed + i, seed - i};
        f2 pb = {seed, 0.5f}, pc = {0.25f, seed};
        for (int i = 0; i < rep_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) valu_op<K>(p[u], pb, pc);
        }
        for (int i = 0; i < 16; ++i) s += p[i].x;
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// one wave: an MFMA, then FILL VALU instructions, repeated
template <int MK, int K, int FILL>
__global__ void __launch_bounds__(256) k_inwave(float *out, int rep, float seed)
{
    f4 c[4] = {};
    f2 p[16];
    for (int i = 0; i < 16; ++i) p[i] = (f2){seed + i, seed - i};
    f2 pb = {seed, 0.5f}, pc = {0.25f, seed};
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = static_cast<_Float16>(seed + j);
        b[j] = static_cast<_Float16>(seed - j * 0.25f);
    }
    float af = seed + threadIdx.x, bf = seed - threadIdx.x;
    for (int i = 0; i < rep; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (MK == 0) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c[u & 3], 0, 0, 0);
            else if constexpr (MK == 1) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < FILL; ++v) valu_op<K>(p[(u * FILL + v) & 15], pb, pc);
        }
    }
    float s = c[0].x + c[1].y + c[2].z + c[3].w;
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static float *g_out;
template <typename F>
static float timed(F &&launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

template <int MK, int K>
static float mix(int bpc, int n_mfma, int rv, int rm)
{
    return timed([&] { hipLaunchKernelGGL((k_mix<MK, K>), dim3(256 * bpc), dim3(256), 0, 0, g_out, rv, rm, n_mfma, bpc, 1.5f); });
}
template <int MK, int K, int FILL>
static float inwave(int bpc, int rep)
{
    return timed([&] { hipLaunchKernelGGL((k_inwave<MK, K, FILL>), dim3(256 * bpc), dim3(256), 0, 0, g_out, rep, 1.5f); });
}

template <int MK, int K>
static void report(const char *mname, const char *kname, int rv, int rm)
{
    const float v2 = mix<MK, K>(2, 0, rv, rm), m1 = mix<MK, K>(1, 1, rv, rm), both = mix<MK, K>(3, 1, rv, rm);
    printf("%-26s beside 2 waves of %-20s: VALU alone %.3f  MFMA alone %.3f  together %.3f ms  (sum %.3f, max %.3f)\n", mname, kname, v2, m1, both,
           v2 + m1, v2 > m1 ? v2 : m1);
}

template <int MK, int K>
static void report_inwave(const char *mname, const char *kname, int rep)
{
    const float f0 = inwave<MK, K, 0>(1, rep), f1 = inwave<MK, K, 1>(1, rep), f2_ = inwave<MK, K, 2>(1, rep), f4_ = inwave<MK, K, 4>(1, rep), f8_ = inwave<MK, K, 8>(1, rep);
    const double cyc = 2.4e6 / (rep * 16.0);
    printf("%-26s + n x %-20s per MFMA, one wave per SIMD: cycles per MFMA at 2.4 GHz: n=0 %.1f  n=1 %.1f  n=2 %.1f  n=4 %.1f  n=8 %.1f\n", mname, kname, f0 * cyc,
           f1 * cyc, f2_ * cyc, f4_ * cyc, f8_ * cyc);
    const float g0 = inwave<MK, K, 0>(2, rep), g4 = inwave<MK, K, 4>(2, rep), g8 = inwave<MK, K, 8>(2, rep);
    printf("%-26s   ... two such waves per SIMD: n=0 %.1f  n=4 %.1f  n=8 %.1f cycles per MFMA and wave\n", "", g0 * cyc, g4 * cyc, g8 * cyc);
}

int main()
{
    hipMalloc(&g_out, 4096);
    const int RV = 4096, RM = 2048;
    report<1, 0>("v_mfma_f32_16x16x32_f16", "v_pk_fma_f32", RV, RM * 2);
    report<1, 1>("v_mfma_f32_16x16x32_f16", "v_fma_f32 (VGPR)", RV * 2, RM * 2);
    report<1, 2>("v_mfma_f32_16x16x32_f16", "v_add_u32", RV * 2, RM * 2);
    report<1, 3>("v_mfma_f32_16x16x32_f16", "v_cvt_pkrtz_f16_f32", RV * 2, RM * 2);
    report<0, 0>("v_mfma_f32_16x16x4_f32", "v_pk_fma_f32", RV, RM);
    report<0, 1>("v_mfma_f32_16x16x4_f32", "v_fma_f32 (VGPR)", RV * 2, RM);
    report<0, 2>("v_mfma_f32_16x16x4_f32", "v_add_u32", RV * 2, RM);
    report_inwave<1, 0>("v_mfma_f32_16x16x32_f16", "v_pk_fma_f32", RM);
    report_inwave<1, 1>("v_mfma_f32_16x16x32_f16", "v_fma_f32 (VGPR)", RM);
    report_inwave<1, 2>("v_mfma_f32_16x16x32_f16", "v_add_u32", RM);
    report_inwave<0, 1>("v_mfma_f32_16x16x4_f32", "v_fma_f32 (VGPR)", RM / 2);
    return 0;
}
