// apt_kernels_fused_probe.hip — the timing probes of the 48 kHz f32 front end, ONE source compiled once per probe with
// -DAPT_PROBE_N=n (make -C noaa_apt_amd/csrc PROBES=1; selected at run time with APTGPU_PROBE_STOP=n):
//   1 .. 5    the FAST kernel cut off after stage n (APT_FUSED_STOP: 1 tile in LDS, 2 resampler, 3 envelope, 4 low-pass,
//             5 F stored): the output is meaningless, only the duration counts
//   11 .. 15  the same cuts of the STRICT kernel
//   8, 16     the complete fast / strict kernel without its HBM reads (synthetic tile contents)
//   9, 17     the fast / strict kernel as a persistent kernel that requests the next tile's input right after this
//             tile's has gone to LDS (complete, valid output)
// (Until round 5: fourteen files that differed in these two lines.)
#ifndef APT_PROBE_N
#error "compile with -DAPT_PROBE_N=<probe number>"
#endif
#if APT_PROBE_N >= 1 && APT_PROBE_N <= 5
#define APT_FUSED_STOP APT_PROBE_N
#define APT_PROBE_MODE kModeFast
#elif APT_PROBE_N >= 11 && APT_PROBE_N <= 15
#define APT_FUSED_STOP (APT_PROBE_N - 10)
#define APT_PROBE_MODE kModeStrict
#elif APT_PROBE_N == 8 || APT_PROBE_N == 16
#define APT_FUSED_NOLOAD 1
#define APT_PROBE_MODE (APT_PROBE_N == 8 ? kModeFast : kModeStrict)
#elif APT_PROBE_N == 9 || APT_PROBE_N == 17
#define APT_FUSED_PERSIST 1
#define APT_PROBE_MODE (APT_PROBE_N == 9 ? kModeFast : kModeStrict)
#else
#error "no such probe"
#endif
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

#define APT_PROBE_CAT_(a, b) a##b
#define APT_PROBE_CAT(a, b) APT_PROBE_CAT_(a, b)

namespace apt::gpu {

void APT_PROBE_CAT(fused_launch_probe, APT_PROBE_N)(const FusedLaunch &a)
{
    launch_fused_args<13, 50, 959, 37, 3, 256, APT_PROBE_MODE, float>(a);
}

}  // namespace apt::gpu
