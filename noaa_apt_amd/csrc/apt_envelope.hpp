// apt_envelope.hpp — the AM envelope of demodulate() (dsp.rs:369-377),
//     y = sqrt(prev^2 + curr^2 - prev*curr*(2 cos phi)) / sin(phi),
// with IEEE-correct sqrt and divide but without the compiler's fully general sequences.
//
// hipcc lowers a correctly rounded `__builtin_sqrtf(x) / c` to ~35 VALU instructions + hazard
// nops per value: range scaling for tiny inputs, class checks for 0 / inf, and a divide that
// starts from an approximate reciprocal.  Here, for x in [2^-96, 2^100] (every real recording):
//   sqrt: s = v_sqrt_f32(x) is within 1 ulp; the residuals fma(-(s-1ulp), s, x) and
//         fma(-(s+1ulp), s, x) pick the correctly rounded neighbour — the same core the
//         compiler emits, minus its scaling and class handling;
//   x/c:  with r = RN(1/c) from the host, q0 = RN(x*r), rem = fma(-c, q0, x) (exact),
//         q = fma(rem, r, q0).  Whether that is the correctly rounded quotient for EVERY x
//         depends on c; it is scale-invariant in x, so apt::gpu::verify_fast_divide() checks
//         all 2^24 significands in two binades against `/` on the device when a plan is
//         created, and the fast path is only enabled for a divisor that passes.
// Values outside the range (digital silence gives x == 0) take the general code: the caller
// branches wave-uniformly on envelope_in_range().
#pragma once

#include <hip/hip_runtime.h>

namespace apt::gpu {

// the radicand, in the reference's evaluation order (dsp.rs:373)
__device__ __forceinline__ float envelope_radicand(float prev, float curr, float cosphi2)
{
#pragma clang fp contract(off)
    const float s = (prev * prev) + (curr * curr);
    const float c = (prev * curr) * cosphi2;
    return s - c;
}

// general path: correctly rounded for every input (needs -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ float envelope_general(float x, float sinphi)
{
    return __builtin_sqrtf(x) / sinphi;
}

// x in [2^-96, 2^100]: positive floats order like their bit patterns
__device__ __forceinline__ bool envelope_in_range(float x)
{
    return (__float_as_uint(x) - 0x0F800000u) <= (0x71800000u - 0x0F800000u);
}

__device__ __forceinline__ float fast_divide(float x, float c, float rc)
{
    const float q0 = x * rc;
    const float rem = __builtin_fmaf(-c, q0, x);
    return __builtin_fmaf(rem, rc, q0);
}

// fast path: caller guarantees envelope_in_range(x) and a verified (c, rc) pair
__device__ __forceinline__ float envelope_fast(float x, float sinphi, float inv_sinphi)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float root = (0.f >= r_dn) ? s_dn : s;
    root = (0.f < r_up) ? s_up : root;
    return fast_divide(root, sinphi, inv_sinphi);
}

// Host: true when fast_divide(x, c, rc) == x / c bit for bit for every x whose result stays in
// the normal range (exhaustive over the significands; once per device and divisor).
bool verify_fast_divide(int device, float c, float rc);

}  // namespace apt::gpu
