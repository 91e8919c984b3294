// apt_kernels_fused_probe15.hip — timing probe: the STRICT 48 kHz f32 front end cut off after stage 5
// (see APT_FUSED_STOP in apt_kernels_fused_impl.hpp).  Selected with APTGPU_PROBE_STOP=15; its output is
// meaningless, only its duration is.
#define APT_FUSED_STOP 5
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe15(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeStrict, float>(a); }

}  // namespace apt::gpu
