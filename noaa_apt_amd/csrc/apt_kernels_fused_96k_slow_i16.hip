// apt_kernels_fused_96k_slow_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_96k_slow_i16(const FusedLaunch &a) { launch_fused_args<13, 60, 5565, 61, 5, 256, kModeStrict, int16_t>(a); }

}  // namespace apt::gpu
