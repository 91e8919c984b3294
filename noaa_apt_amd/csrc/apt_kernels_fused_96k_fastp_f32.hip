// apt_kernels_fused_96k_fastp_f32.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_96k_fastp_f32(const FusedLaunch &a) { launch_fused_args<13, 75, 639, 43, 4, 256, kModeStrict, float>(a); }

}  // namespace apt::gpu
