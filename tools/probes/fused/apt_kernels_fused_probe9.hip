// apt_kernels_fused_probe9.hip — experiment: the FAST 48 kHz f32 front end as a persistent kernel that requests the
// next tile's input right after this tile's has gone to LDS (APTGPU_PROBE_STOP=9).  Complete kernel, valid output.
#define APT_FUSED_PERSIST 1
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe9(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeFast, float>(a); }

}  // namespace apt::gpu
