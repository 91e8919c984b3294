// apt_kernels_fused_probe7.hip — experiment: the fast 48 kHz f32 front end with 192-thread workgroups
// (APTGPU_PROBE_STOP=7).  Complete kernel, valid output.
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe7(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 192, kModeFast, float>(a); }

}  // namespace apt::gpu
