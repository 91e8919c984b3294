// tools/ubench/sqrt_check.hip — exhaustive check of cheap exactly-rounded sqrt sequences on gfx950 against the
// correctly rounded sqrtf (build with -fhip-fp32-correctly-rounded-divide-sqrt), every float in [2^-96, 2^100].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float v1(float x)  // rsq + one Newton-Markstein correction
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float v2(float x)  // LLVM's refinement: two corrections
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float v3(float x)  // sqrt + one correction with h = 0.5 * rsq
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float v4(float x)  // v1 applied twice (second residual with the same h)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    const float h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

__global__ void __launch_bounds__(256) k(uint32_t lo, uint32_t count, unsigned long long *bad)
{
    const uint64_t base = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 64;
    uint32_t b1 = 0, b2 = 0, b3 = 0, b4 = 0;
    for (int i = 0; i < 64; ++i) {
        const uint64_t idx = base + i;
        if (idx >= count) break;
        const float x = __uint_as_float(lo + uint32_t(idx));
        const uint32_t want = __float_as_uint(__builtin_sqrtf(x));
        b1 += __float_as_uint(v1(x)) != want;
        b2 += __float_as_uint(v2(x)) != want;
        b3 += __float_as_uint(v3(x)) != want;
        b4 += __float_as_uint(__builtin_amdgcn_sqrtf(x)) != want;  // control: the raw 1-ulp instruction
    }
    if (b1) atomicAdd(bad + 0, (unsigned long long)b1);
    if (b2) atomicAdd(bad + 1, (unsigned long long)b2);
    if (b3) atomicAdd(bad + 2, (unsigned long long)b3);
    if (b4) atomicAdd(bad + 3, (unsigned long long)b4);
}

int main()
{
    unsigned long long *d, h[4];
    hipMalloc(&d, 32);
    hipMemset(d, 0, 32);
    const uint32_t lo = 0x0F800000u, hi = 0x71800000u;
    const uint32_t count = hi - lo + 1;
    const uint32_t threads = (count + 63) / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3((threads + 255) / 256), dim3(256), 0, 0, lo, count, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%u values in %.2f ms; mismatches: rsq+1 correction %llu, rsq+2 (LLVM) %llu, sqrt+1 correction %llu, raw v_sqrt_f32 (control) %llu\n",
           count, ms, h[0], h[1], h[2], h[3]);
    return 0;
}
