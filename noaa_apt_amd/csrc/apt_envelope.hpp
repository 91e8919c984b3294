// apt_envelope.hpp — the AM envelope of demodulate() (dsp.rs:369-377),
//     y = sqrt(prev^2 + curr^2 - prev*curr*(2 cos phi)) / sin(phi),
// with IEEE-correct sqrt and divide but without the compiler's fully general sequences.
//
// hipcc lowers a correctly rounded `__builtin_sqrtf(x) / c` to ~35 VALU instructions + hazard
// nops per value: range scaling for tiny inputs, class checks for 0 / inf, and a divide that
// starts from an approximate reciprocal.  Here, for x in [2^-96, 2^100] (every real recording):
//   sqrt: y = v_rsq_f32(x); g = x*y; h = y/2; root = fma(fma(-g, g, x), h, g) — one Newton-Markstein
//         correction of g ~ sqrt(x) by its exact residual.  Whether that is the correctly rounded root for
//         EVERY x of the range is a property of the hardware's v_rsq_f32; apt::gpu::verify_fast_divide()
//         compares it with sqrtf() for all 1 644 167 169 floats of the range (2 ms on an MI355X; 0 mismatches
//         on gfx950, tools/ubench/sqrt_check.hip) once per device and process.  No compares, no selects: a
//         v_cmp + v_cndmask pair costs as much as five of these instructions (tools/ubench/rates3.hip), which
//         is what the previous form (v_sqrt_f32, then picking among s-1ulp, s, s+1ulp by two residual tests) paid
//         twice per value;
//   x/c:  with r = RN(1/c) from the host, q0 = RN(x*r), rem = fma(-c, q0, x) (exact),
//         q = fma(rem, r, q0).  Whether that is the correctly rounded quotient for EVERY x
//         depends on c; it is scale-invariant in x, so verify_fast_divide() checks
//         all 2^24 significands in two binades against `/` on the device when a plan is
//         created, and the fast path is only enabled for a divisor that passes (and hardware whose
//         root passes).
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

// x in [2^-96, 2^100]: the correctly rounded square root (on hardware that passed verify_fast_divide())
__device__ __forceinline__ float exact_sqrt_inrange(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// fast path: caller guarantees envelope_in_range(x) and a verified (c, rc) pair
__device__ __forceinline__ float envelope_fast(float x, float sinphi, float inv_sinphi)
{
    return fast_divide(exact_sqrt_inrange(x), sinphi, inv_sinphi);
}

// Host: true when fast_divide(x, c, rc) == x / c bit for bit for every x whose result stays in
// the normal range (exhaustive over the significands; once per device and divisor) AND
// exact_sqrt_inrange(x) == sqrtf(x) for every float x of [2^-96, 2^100] (once per device).
bool verify_fast_divide(int device, float c, float rc);

}  // namespace apt::gpu
