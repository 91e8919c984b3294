// apt_kernels_fused_probe8.hip — timing probe: the complete FAST 48 kHz f32 front end WITHOUT its HBM reads
// (interior tiles get synthetic contents).  APTGPU_PROBE_STOP=8; output meaningless.
#define APT_FUSED_NOLOAD 1
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe8(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeFast, float>(a); }

}  // namespace apt::gpu
