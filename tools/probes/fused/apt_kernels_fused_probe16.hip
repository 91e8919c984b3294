// apt_kernels_fused_probe16.hip — timing probe: the complete STRICT 48 kHz f32 front end WITHOUT its HBM reads
// (interior tiles get synthetic contents): what the kernel costs when no load has to be waited for.
// APTGPU_PROBE_STOP=16; output meaningless.
#define APT_FUSED_NOLOAD 1
#include "../../../noaa_apt_amd/csrc/apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_probe16(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeStrict, float>(a); }

}  // namespace apt::gpu
