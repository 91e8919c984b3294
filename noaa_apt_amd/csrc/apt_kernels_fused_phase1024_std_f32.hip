// apt_kernels_fused_phase1024_std_f32.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_phase1024_std_f32(const FusedLaunch &a) { launch_fused_args<13, -1, 0, 37, 3, 1024, kModeStrict, float>(a); }

}  // namespace apt::gpu
