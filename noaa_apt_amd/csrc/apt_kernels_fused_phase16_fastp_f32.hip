// apt_kernels_fused_phase16_fastp_f32.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_phase16_fastp_f32(const FusedLaunch &a) { launch_fused_args<13, -16, 0, 43, 4, 256, kModeStrict, float>(a); }

}  // namespace apt::gpu
