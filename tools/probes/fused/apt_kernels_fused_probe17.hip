// apt_kernels_fused_probe17.hip — experiment: the STRICT 48 kHz f32 front end as a persistent kernel that requests the
// next tile's input right after this tile's has gone to LDS (APTGPU_PROBE_STOP=17).  Complete kernel, valid output.
#define APT_FUSED_PERSIST 1
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe17(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeStrict, float>(a); }

}  // namespace apt::gpu
