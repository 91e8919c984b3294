// tools/ubench/clocks.hip — shader-clock sampler (libclockprobe.so; not part of the product).
//
// k_clock_sampler: a few single-wave workgroups that stay resident while other work runs and record, every few
// microseconds, the pair (s_memtime, s_memrealtime): the second ticks at a constant 100 MHz and the first was expected
// to tick at the shader clock of the wave's XCD.  It does not (measured: constant 2.40 GHz while the SMU reports 2.0 GHz
// under load, and the s_sleep period below is constant too) — see tools/power_regimes.py for the real clocks.  One wave
// per workgroup, ~10 VALU instructions per sample and an s_sleep in between: it takes a wave slot (and nothing else
// worth mentioning) on the CUs it lands on.  tools/clock_regimes.py runs it beside the decode pipeline.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void __launch_bounds__(64) k_clock_sampler(uint64_t *__restrict__ out, int n_samples, int sleeps, uint64_t *__restrict__ meta)
{
    // out[(wg * n_samples + i) * 2 + {0, 1}] = (shader ticks, 100 MHz ticks) of sample i of workgroup wg
    uint64_t *o = out + static_cast<size_t>(blockIdx.x) * n_samples * 2;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        meta[blockIdx.x] = (static_cast<uint64_t>(xcc) << 32) | hw;
    }
    for (int i = 0; i < n_samples; ++i) {
        const uint64_t t = __builtin_readcyclecounter();   // s_memtime
        const uint64_t r = wall_clock64();                 // s_memrealtime, 100 MHz
        if (threadIdx.x == 0) {
            o[2 * i] = t;
            o[2 * i + 1] = r;
        }
        for (int k = 0; k < sleeps; ++k) __builtin_amdgcn_s_sleep(127);  // 127 * 64 clocks each
    }
}

extern "C" int clockprobe_launch(void *stream, uint64_t *d_out, int n_wgs, int n_samples, int sleeps, uint64_t *d_meta)
{
    hipLaunchKernelGGL(k_clock_sampler, dim3(n_wgs), dim3(64), 0, static_cast<hipStream_t>(stream), d_out, n_samples, sleeps, d_meta);
    return static_cast<int>(hipGetLastError());
}
