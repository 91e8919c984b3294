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
//            corr[i] = -B[i] - B[i+P] + B[i+2P] - ... (19 terms, left to right).
//            22 operations per position instead of 114 at pw = 3; differs from `strict` by
//            reassociation only (a few ulp of the largest partial sum).  APTGPU_MODE_FAST.
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

// fast, step 2: corr[i] from the pulse sums; bat(k) returns B[i + k*2*pw], k < 19
template <typename Bat>
__device__ __forceinline__ float sync_corr_from_pulses(Bat &&bat)
{
#pragma clang fp contract(off)
    float c = -bat(0);
#pragma unroll
    for (int k = 1; k < 19; ++k) c = sync_pulse_plus(k) ? c + bat(k) : c - bat(k);
    return c;
}

}  // namespace apt::gpu
