// apt_sync_corr.hpp — the two ways the sync cross-correlation of find_sync() (src/decode.rs:225-233)
// is evaluated on the device, shared by the fused front ends (which only emit per-group maxima of
// it) and by k_sync_words (which evaluates it for the few candidate groups the picker looks at).
// Both sides must produce the SAME bits for the same position, so the arithmetic lives here once.
//
//   strict : corr = 0; for j in 0..38*pw: corr +-= F[i+j]   — the reference's sequential chain,
//            bit-identical to the Rust loop.
//   fast   : the template is piecewise constant over pulses of P = 2*pw samples (19 pulses with
//            signs - | -+ x7 | ----), so with pulse sums
//                B2[i] = F[i] + F[i+1]
//                B[i]  = ((B2[i] + B2[i+2]) + B2[i+4]) + ...      (pw terms)
//            corr[i] = -B[i] - B[i+P] + B[i+2P] - ... (19 terms, in the fixed association of
//            sync_corr_from_pulses below).  22 operations per position instead of 114 at pw = 3 — 14 where a
//            thread evaluates positions one pulse apart and shares the partial sums (sync_corr_pulse_stride);
//            differs from `strict` by reassociation only (a few ulp of the largest partial sum).  APTGPU_MODE_FAST.
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

// sign of the sync template at index j, pulse = 2*pw (decode.rs:188-198): + inside the seven high pulses
__host__ __device__ constexpr bool sync_template_plus(int j, int pw)
{
    const int pulse = 2 * pw;
    if (j < pulse || j >= 15 * pulse) return false;
    return (((j - pulse) / pulse) & 1) == 1;
}
// sign of pulse k (0..18) of the template
__host__ __device__ constexpr bool sync_pulse_plus(int k) { return k >= 2 && k <= 14 && (k & 1) == 0; }

// strict: at(j) returns F[i + j]
template <typename At>
__device__ __forceinline__ float sync_corr_strict(uint32_t pw, At &&at)
{
#pragma clang fp contract(off)
    // (with a compile-time pw the loops unroll completely and the reads pipeline; run-time pw loops)
    const uint32_t pulse = 2 * pw;
    float c = 0.f;
    uint32_t j = 0;
#pragma unroll
    for (uint32_t e = 0; e < pulse; ++e, ++j) c = c - at(j);
#pragma unroll
    for (int rep = 0; rep < 7; ++rep) {
#pragma unroll
        for (uint32_t e = 0; e < pulse; ++e, ++j) c = c - at(j);
#pragma unroll
        for (uint32_t e = 0; e < pulse; ++e, ++j) c = c + at(j);
    }
#pragma unroll
    for (uint32_t e = 0; e < 8 * pw; ++e, ++j) c = c - at(j);
    return c;
}

// the same chain with the seven (-, +) pulse pairs and the four tail pulses as loops that stay loops: with the pixel
// width at compile time the form above unrolls to 38*pw additions whose LDS reads the compiler issues as early as its
// register budget allows — 120 VGPRs and scratch in k_sync_words at pw = 4 / 5 (68 at pw = 3), four waves per SIMD
// instead of seven for a latency-bound kernel.  Same additions in the same order: same bits.
template <typename At>
__device__ __forceinline__ float sync_corr_strict_rolled(uint32_t pw, At &&at)
{
#pragma clang fp contract(off)
    const uint32_t pulse = 2 * pw;
    float c = 0.f;
    uint32_t j = 0;
#pragma unroll
    for (uint32_t e = 0; e < pulse; ++e, ++j) c = c - at(j);
#pragma unroll 1
    for (int rep = 0; rep < 7; ++rep) {
#pragma unroll
        for (uint32_t e = 0; e < pulse; ++e, ++j) c = c - at(j);
#pragma unroll
        for (uint32_t e = 0; e < pulse; ++e, ++j) c = c + at(j);
    }
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (uint32_t e = 0; e < pulse; ++e, ++j) c = c - at(j);
    }
    return c;
}

// fast, step 1: pulse sum B[i]; at(j) returns F[i + j], j < 2*pw
template <typename At>
__device__ __forceinline__ float sync_pulse_sum(uint32_t pw, At &&at)
{
#pragma clang fp contract(off)
    float b = at(0) + at(1);
#pragma unroll
    for (uint32_t e = 1; e < pw; ++e) b = b + (at(2 * e) + at(2 * e + 1));
    return b;
}

// fast, step 2: corr[i] from the pulse sums; bat(k) returns B[i + k*2*pw], k < 19.
// The signs are - | (-+) x7 | ----: with E_k = B_{k+1} - B_k the alternating part is E_1 + E_3 + ... + E_13, summed as
//   ((E_1 + E_3) + (E_5 + E_7)) + (E_9 + E_11)) + E_13,
// the tail as (B_15 + B_16) + (B_17 + B_18), and corr = (alternating - B_0) - tail.  A fixed association, so a front end
// that evaluates many positions one pulse apart and shares the partial sums between them (sync_corr_pulse_stride)
// produces the same bits as this per-position form.
template <typename Bat>
__device__ __forceinline__ float sync_corr_from_pulses(Bat &&bat)
{
#pragma clang fp contract(off)
    float e[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) e[m] = bat(2 * m + 2) - bat(2 * m + 1);
    const float s4 = (e[0] + e[1]) + (e[2] + e[3]);
    const float alt = (s4 + (e[4] + e[5])) + e[6];
    const float tail = (bat(15) + bat(16)) + (bat(17) + bat(18));
    return (alt - bat(0)) - tail;
}

// The same values for N positions ONE PULSE APART: V[n] = B[i + n*2*pw]; c[j] = corr[i + j*2*pw].
// E[n] = V[n+1] - V[n], S2[n] = E[n] + E[n+2], S4[n] = S2[n] + S2[n+4]:
//   c[j] = (((S4[j+1] + S2[j+9]) + E[j+13]) - V[j]) - ((V[j+15] + V[j+16]) + (V[j+17] + V[j+18])).
// 10.7 additions per position at N = 13 instead of 18 — and every one of them in packed form (a packed f32
// instruction takes the issue slot of a plain one): the caller hands V over twice, as even-aligned pairs
// A[p] = (V[2p], V[2p+1]) and odd-aligned pairs S[p] = (V[2p+1], V[2p+2]), and gets cp[q] = (c[2q], c[2q+1]).
// Then Eo[p] = A[p+1] - S[p] = (E[2p+1], E[2p+2]), S2o[p] = Eo[p] + Eo[p+1] = (S2[2p+1], S2[2p+2]),
// T2o[p] = S[p] + A[p+1] = (T2[2p+1], T2[2p+2]), and for the output pair q (j = 2q):
//   cp[q] = ((((S2o[q] + S2o[q+2]) + S2o[q+4]) + Eo[q+6]) - A[q]) - (T2o[q+7] + T2o[q+8]).
// NP2 = ceil(N / 2) output pairs; needs A[0 .. NP2+8], S[0 .. NP2+7].
typedef float sync_f2 __attribute__((ext_vector_type(2)));
template <int NP2>
__device__ __forceinline__ void sync_corr_pulse_stride(const sync_f2 (&A)[NP2 + 9], const sync_f2 (&S)[NP2 + 8], sync_f2 (&cp)[NP2])
{
#pragma clang fp contract(off)
    sync_f2 Eo[NP2 + 6];
#pragma unroll
    for (int p = 0; p < NP2 + 6; ++p) Eo[p] = A[p + 1] - S[p];
    sync_f2 S2o[NP2 + 4];
#pragma unroll
    for (int p = 0; p < NP2 + 4; ++p) S2o[p] = Eo[p] + Eo[p + 1];
    sync_f2 T2o[NP2 + 8];  // T2o[7 .. NP2+7] used
#pragma unroll
    for (int p = 7; p < NP2 + 8; ++p) T2o[p] = S[p] + A[p + 1];
#pragma unroll
    for (int q = 0; q < NP2; ++q) {
        const sync_f2 s4 = S2o[q] + S2o[q + 2];
        const sync_f2 alt = (s4 + S2o[q + 4]) + Eo[q + 6];
        const sync_f2 tail = T2o[q + 7] + T2o[q + 8];
        cp[q] = (alt - A[q]) - tail;
    }
}

}  // namespace apt::gpu
