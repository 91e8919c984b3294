// tools/ubench/mfma_fir.hip — go / no-go for "the two FIRs on the matrix pipe" (Next 1).
//
// Stage 1 of the 48 kHz standard-profile front end (l = 13, m = 50, 959 taps: dsp.rs:252-263), one workgroup of 256
// threads per tile of 256 windows = 3328 work samples, input tile through LDS, result R through LDS to HBM — the
// frame the product's k_fused gives its stage 1 — computed five ways:
//   0  nothing (tile in, zeros out): the frame's own cost
//   1  VALU: one window per thread, 13 accumulators, wave-uniform taps, fused multiply-adds in tap order
//      (what APTGPU_MODE_FAST defines; the product's kernel does the same arithmetic with 481 v_pk_fma_f32)
//   2  v_mfma_f32_16x16x4_f32 on the banded Toeplitz matrix H[16 branches][124] x X[124][16 windows]
//      (f32 in, f32 accumulate: a k-ordered fmaf chain — should be bit-identical to 1)
//   3  v_mfma_f32_16x16x32_f16, three terms: taps h = h0 + h1, samples x = x0 + x1 (f16 pieces of the f32 values,
//      taps prescaled by a power of two), h0 x0 + h0 x1 + h1 x0 accumulated in f32
//   4  as 3 with two terms (h0 x0 + h0 x1): fp16 taps, exact samples — BASELINE config 5's arithmetic
// and each of them again with `extra` dependent plain-VALU instructions per thread behind stage 1 (a stand-in for
// the envelope / bounds / bookkeeping of the real kernel): does the matrix pipe run beside them?
// Prints ms per launch of 16 x 2361 tiles (config 2's launch), bit / error statistics against variant 1 and against an
// f64 evaluation on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int L = 13, M = 50, T1 = 959, TP = (T1 + L - 1) / L;       // 74 taps per branch
constexpr int NW = 256;                                              // windows per tile
constexpr int CLAST = ((L - 1) * M + L - 1) / L, WIN = CLAST + TP;   // 47, 121
constexpr int KF16 = 128, KF32 = 124;                                // Toeplitz K, padded to the MFMA's K
constexpr int XT = (NW - 1) * M + KF16, XT_PAD = (XT + 7) & ~7;      // 12878 -> 12880 input samples per tile
constexpr int TILE_K = NW * L;                                       // 3328 outputs per tile

__host__ __device__ constexpr int c_of(int b) { return (b * M + L - 1) / L; }
__host__ __device__ constexpr int p_of(int b) { return c_of(b) * L - b * M; }
__host__ __device__ constexpr bool uses(int b, int q) { return q >= c_of(b) && p_of(b) + (q - c_of(b)) * L < T1; }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define CONST_AS __attribute__((address_space(4)))

struct Params {
    const float *x;        // [nrec][n_in]
    float *r;              // [nrec][tiles * TILE_K]
    const float *hq;       // variant 1: [WIN][16] phase-major taps (zero where a branch does not use the sample)
    const float *a32;      // variant 2: [31][64] A fragments
    const u4 *a16_0;       // variants 3, 4: [4][64] A fragments of h0 (8 halves each)
    const u4 *a16_1;       // variant 3: ... of h1
    float unscale;         // 2^-s of the f16 tap prescale
    uint64_t n_in, w_out;
    int tiles, extra;
};

template <int VAR>
__global__ void __launch_bounds__(256, 3) k_stage1(const Params p)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x;
    const float *x = p.x + static_cast<uint64_t>(blockIdx.y) * p.n_in + tile * (NW * M);
    float *rout = p.r + static_cast<uint64_t>(blockIdx.y) * p.w_out + tile * TILE_K;
    constexpr bool HALF = VAR >= 3;
    uint32_t *l32 = reinterpret_cast<uint32_t *>(lds);
    constexpr int PLANE = XT_PAD / 2;  // dwords per f16 plane
    // ---- tile -> LDS (f32, or two f16 planes x0 / x1 with x = x0 + x1 exactly for 16-bit data)
#pragma unroll
    for (int e = 0; e < (XT_PAD / 4 + 255) / 256; ++e) {
        const int q = (tid + e * 256) * 4;
        if (q < XT_PAD) {
            const f4 v = *reinterpret_cast<const f4 *>(x + q);
            if constexpr (!HALF) {
                *reinterpret_cast<f4 *>(lds + q) = v;
            } else {
                typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
                const hh2 a0 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), b0 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
                const float r0 = v.x - static_cast<float>(a0.x), r1 = v.y - static_cast<float>(a0.y);
                const float r2 = v.z - static_cast<float>(b0.x), r3 = v.w - static_cast<float>(b0.y);
                const hh2 a1 = __builtin_amdgcn_cvt_pkrtz(r0, r1), b1 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
                uint32_t wa0, wb0, wa1, wb1;
                __builtin_memcpy(&wa0, &a0, 4);
                __builtin_memcpy(&wb0, &b0, 4);
                __builtin_memcpy(&wa1, &a1, 4);
                __builtin_memcpy(&wb1, &b1, 4);
                l32[q / 2] = wa0;
                l32[q / 2 + 1] = wb0;
                l32[PLANE + q / 2] = wa1;
                l32[PLANE + q / 2 + 1] = wb1;
            }
        }
    }
    __syncthreads();
    float rr[L];
#pragma unroll
    for (int b = 0; b < L; ++b) rr[b] = 0.f;
    [[maybe_unused]] f4 acc[4];
    if constexpr (VAR == 1) {
        typedef const float CONST_AS *cf;
        const cf hq = (cf)(p.hq);
        f2 a2[6] = {};
        float al = 0.f;
        static_for<0, WIN>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float xq = lds[tid * M + q];
            static_for<0, 6>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                constexpr bool va = uses(2 * k, q), vb = uses(2 * k + 1, q);
                if constexpr (va && vb) a2[k] = __builtin_elementwise_fma((f2){hq[q * 16 + 2 * k], hq[q * 16 + 2 * k + 1]}, (f2){xq, xq}, a2[k]);
                else if constexpr (va) a2[k].x = __builtin_fmaf(hq[q * 16 + 2 * k], xq, a2[k].x);
                else if constexpr (vb) a2[k].y = __builtin_fmaf(hq[q * 16 + 2 * k + 1], xq, a2[k].y);
            });
            if constexpr (uses(12, q)) al = __builtin_fmaf(hq[q * 16 + 12], xq, al);
        });
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            rr[2 * k] = a2[k].x;
            rr[2 * k + 1] = a2[k].y;
        }
        rr[12] = al;
    } else if constexpr (VAR == 2) {
        // lane (a = lane & 15, kk = lane >> 4): B[k = 4 s + kk][a] = x[50 (a0 + a) + 4 s + kk]; A[b = lane & 15][4 s + kk]
        float af[KF32 / 4];
#pragma unroll
        for (int s = 0; s < KF32 / 4; ++s) af[s] = p.a32[s * 64 + lane];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f4){0.f, 0.f, 0.f, 0.f};
        const int a = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int s = 0; s < KF32 / 4; ++s) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int a0 = (wave * 4 + g) * 16;
                const float bv = lds[(a0 + a) * M + 4 * s + kk];
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bv, acc[g], 0, 0, 0);
            }
        }
    } else if constexpr (VAR >= 3) {
        // lane (a = lane & 15, g8 = lane >> 4): B[k = 32 s + 8 g8 + j][a] = x[50 (a0 + a) + 32 s + 8 g8 + j], j < 8
        u4 a0f[KF16 / 32], a1f[KF16 / 32];
#pragma unroll
        for (int s = 0; s < KF16 / 32; ++s) {
            a0f[s] = p.a16_0[s * 64 + lane];
            if constexpr (VAR == 3) a1f[s] = p.a16_1[s * 64 + lane];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f4){0.f, 0.f, 0.f, 0.f};
        const int a = lane & 15, g8 = lane >> 4;
        auto frag = [&](u4 v) -> h8 {
            h8 r;
            __builtin_memcpy(&r, &v, 16);
            return r;
        };
#pragma unroll
        for (int s = 0; s < KF16 / 32; ++s) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int a0 = (wave * 4 + g) * 16;
                const int d = (a0 + a) * (M / 2) + 16 * s + 4 * g8;
                const u4 b0 = (u4){l32[d], l32[d + 1], l32[d + 2], l32[d + 3]};
                const u4 b1 = (u4){l32[PLANE + d], l32[PLANE + d + 1], l32[PLANE + d + 2], l32[PLANE + d + 3]};
                // small terms first
                if constexpr (VAR == 3) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(a1f[s]), frag(b0), acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(a0f[s]), frag(b1), acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(a0f[s]), frag(b0), acc[g], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // everyone is done with the input tile: R lands on it
    if constexpr (VAR <= 1) {
#pragma unroll
        for (int b = 0; b < L; ++b) lds[tid * L + b] = rr[b];
    } else {
        // D: col = lane & 15 (window), row = 4 (lane >> 4) + reg (branch)
        const float us = VAR >= 3 ? p.unscale : 1.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int wdw = (wave * 4 + g) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 4 * (lane >> 4) + r;
                if (b < L) lds[wdw * L + b] = acc[g][r] * us;
            }
        }
    }
    __syncthreads();
    // stand-in for the later stages: `extra` dependent plain-VALU instructions on the thread's 13 values
    float v[L];
#pragma unroll
    for (int b = 0; b < L; ++b) v[b] = lds[tid * L + b];
    if (p.extra > 0) {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = v[i];
        for (int it = 0; it < p.extra; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(v[8 + (i & 3)]), "v"(v[i]));
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += w[i];
        if (s == 1.2345e-30f) lds[tid] = s;  // (never: keeps the loop)
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < (TILE_K / 4 + 255) / 256; ++e) {
        const int q = (tid + e * 256) * 4;
        if (q < TILE_K) *reinterpret_cast<f4 *>(rout + q) = *reinterpret_cast<const f4 *>(lds + q);
    }
}

static void design(std::vector<float> &h)
{
    // a band-pass of the reference's shape (filters.rs:98-132: sinc(cutout) - sinc(dw / 2), Kaiser window), rates as
    // resample_with_filter sets them (cutout 4800 Hz, dw 1000 Hz at 48 kHz x 13); not bit-identical to the product's
    // design and it need not be
    const double fs = 48000.0 * L, cut = 2.0 * 4800.0 / fs, dw = 2.0 * 1000.0 / fs, beta = 0.1102 * (30.0 - 8.7);
    auto i0 = [](double z) { double s = 1, t = 1; for (int k = 1; k < 40; ++k) { t *= (z / (2 * k)) * (z / (2 * k)); s += t; } return s; };
    h.resize(T1);
    const int mid = (T1 - 1) / 2;
    for (int n = 0; n < T1; ++n) {
        const int k = n - mid;
        double v;
        if (k == 0) v = cut - dw / 2;
        else v = (std::sin(M_PI * cut * k) - std::sin(M_PI * dw / 2 * k)) / (M_PI * k);
        const double r = static_cast<double>(k) / (T1 / 2);
        const double wdw = i0(beta * std::sqrt(std::max(0.0, 1 - r * r))) / i0(beta);
        h[n] = static_cast<float>(v * wdw);
    }
}

static float f16_round(float v, bool *sub = nullptr)
{
    const _Float16 hv = static_cast<_Float16>(v);
    if (sub) *sub = hv != 0 && std::fabs(static_cast<float>(hv)) < 6.1035156e-5f;
    return static_cast<float>(hv);
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int nrec = argc > 1 ? atoi(argv[1]) : 16;
    const int tiles = argc > 2 ? atoi(argv[2]) : 2361;
    const int reps = 10;
    const uint64_t n_in = static_cast<uint64_t>(tiles) * NW * M + XT_PAD, w_out = static_cast<uint64_t>(tiles) * TILE_K;
    std::vector<float> h;
    design(h);
    // ---- host tables
    std::vector<float> hq(WIN * 16, 0.f), a32((KF32 / 4) * 64, 0.f);
    std::vector<uint16_t> a16_0(4 * 64 * 8, 0), a16_1(4 * 64 * 8, 0);
    float hmax = 0;
    for (float v : h) hmax = std::max(hmax, std::fabs(v));
    int sexp = 0;
    while (std::ldexp(hmax, sexp + 1) < 16384.f) ++sexp;   // max |h| 2^s in [2^13, 2^14)
    auto H = [&](int b, int q) -> float { return (b < L && q < WIN && uses(b, q)) ? h[p_of(b) + (q - c_of(b)) * L] : 0.f; };
    int n_sub1 = 0;
    for (int q = 0; q < WIN; ++q)
        for (int b = 0; b < L; ++b) hq[q * 16 + b] = H(b, q);
    for (int s = 0; s < KF32 / 4; ++s)
        for (int l = 0; l < 64; ++l) a32[s * 64 + l] = H(l & 15, 4 * s + (l >> 4));
    for (int s = 0; s < 4; ++s)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const float hv = std::ldexp(H(l & 15, 32 * s + 8 * (l >> 4) + j), sexp);
                const float h0 = f16_round(hv);
                bool sub = false;
                const float h1 = f16_round(hv - h0, &sub);
                n_sub1 += sub;
                const _Float16 q0 = static_cast<_Float16>(h0), q1 = static_cast<_Float16>(h1);
                memcpy(&a16_0[(s * 64 + l) * 8 + j], &q0, 2);
                memcpy(&a16_1[(s * 64 + l) * 8 + j], &q1, 2);
            }
    // ---- input: a 2400 Hz carrier, amplitude-modulated, plus noise, rounded to 16-bit integers (wav.rs:37: unscaled)
    std::vector<float> x(n_in);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return static_cast<double>(st >> 11) / 9007199254740992.0; };
    for (uint64_t i = 0; i < n_in; ++i) {
        const double env = 0.3 + 0.7 * (0.5 + 0.5 * std::sin(i * 2e-4));
        const double v = 20000.0 * env * std::cos(2 * M_PI * 2400.0 / 48000.0 * i) + 400.0 * (rnd() + rnd() + rnd() - 1.5) * 2;
        x[i] = static_cast<float>(std::lrint(std::fmin(32767.0, std::fmax(-32768.0, v))));
    }
    float *dx, *dr, *dhq, *da32;
    u4 *da0, *da1;
    CK(hipMalloc(&dx, nrec * n_in * 4));
    CK(hipMalloc(&dr, nrec * w_out * 4));
    for (int r = 0; r < nrec; ++r) CK(hipMemcpy(dx + r * n_in, x.data(), n_in * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dhq, hq.size() * 4));
    CK(hipMalloc(&da32, a32.size() * 4));
    CK(hipMalloc(&da0, a16_0.size() * 2));
    CK(hipMalloc(&da1, a16_1.size() * 2));
    CK(hipMemcpy(dhq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(da32, a32.data(), a32.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(da0, a16_0.data(), a16_0.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(da1, a16_1.data(), a16_1.size() * 2, hipMemcpyHostToDevice));
    Params p{dx, dr, dhq, da32, da0, da1, std::ldexp(1.f, -sexp), n_in, w_out, tiles, 0};
    const size_t lds_bytes = XT_PAD * 4;
    auto launch = [&](int var) {
        const dim3 grid(tiles, nrec), block(256);
        switch (var) {
        case 0: hipLaunchKernelGGL(k_stage1<0>, grid, block, lds_bytes, 0, p); break;
        case 1: hipLaunchKernelGGL(k_stage1<1>, grid, block, lds_bytes, 0, p); break;
        case 2: hipLaunchKernelGGL(k_stage1<2>, grid, block, lds_bytes, 0, p); break;
        case 3: hipLaunchKernelGGL(k_stage1<3>, grid, block, lds_bytes, 0, p); break;
        case 4: hipLaunchKernelGGL(k_stage1<4>, grid, block, lds_bytes, 0, p); break;
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_ms = [&](int var) {
        launch(var);
        launch(var);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(var);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    // ---- results of one recording's first tiles, per variant
    const uint64_t ncmp = std::min<uint64_t>(w_out, 64 * TILE_K);
    std::vector<std::vector<float>> res(5, std::vector<float>(ncmp));
    for (int var = 1; var <= 4; ++var) {
        CK(hipMemset(dr, 0, ncmp * 4));
        launch(var);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(res[var].data(), dr, ncmp * 4, hipMemcpyDeviceToHost));
    }
    std::vector<double> ref(ncmp);
    std::vector<float> chain(ncmp);
    double rmax = 0;
    for (uint64_t k = 0; k < ncmp; ++k) {
        const uint64_t a = k / L;
        const int b = static_cast<int>(k % L);
        double s = 0;
        float c = 0.f;
        for (int i = 0; p_of(b) + i * L < T1; ++i) {
            const float hv = h[p_of(b) + i * L], xv = x[a * M + c_of(b) + i];
            s += static_cast<double>(hv) * xv;
            c = std::fmaf(hv, xv, c);
        }
        ref[k] = s;
        chain[k] = c;
        rmax = std::max(rmax, std::fabs(s));
    }
    printf("stage 1, l = %d, m = %d, %d taps (%d per branch), window %d; tap prescale 2^%d, %d subnormal h1 pieces\n", L, M, T1, TP, WIN, sexp, n_sub1);
    printf("max |R| = %.3f\n", rmax);
    const char *names[5] = {"frame only", "VALU fma chain", "MFMA f32 16x16x4 Toeplitz", "MFMA f16 16x16x32, 3 terms", "MFMA f16 16x16x32, 2 terms"};
    for (int var = 1; var <= 4; ++var) {
        uint64_t same1 = 0, samec = 0;
        double emax = 0;
        for (uint64_t k = 0; k < ncmp; ++k) {
            uint32_t u, v1, vc;
            memcpy(&u, &res[var][k], 4);
            memcpy(&v1, &res[1][k], 4);
            memcpy(&vc, &chain[k], 4);
            same1 += u == v1;
            samec += u == vc;
            emax = std::max(emax, std::fabs(res[var][k] - ref[k]));
        }
        printf("variant %d (%s): bits equal to variant 1 on %llu / %llu, to the host fmaf chain on %llu; max |err vs f64| = %.3e = %.3e of max |R|\n",
               var, names[var], (unsigned long long)same1, (unsigned long long)ncmp, (unsigned long long)samec, emax, emax / rmax);
    }
    for (int extra : {0, 256, 448, 640}) {
        p.extra = extra;
        printf("extra %4d plain VALU instructions per thread:", extra);
        for (int var = 0; var <= 4; ++var) printf("  v%d %.4f ms", var, time_ms(var));
        printf("\n");
    }
    const double balg = nrec * (n_in * 4.0);
    printf("(input bytes per launch %.1f MB; 8 TB/s -> %.4f ms)\n", balg / 1e6, balg / 8e12 * 1e3);
    return 0;
}
