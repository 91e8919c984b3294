// apt_kernels_fused_48k_f16taps_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_48k_f16taps_i16(const FusedLaunch &a) { launch_fused_args<13, 50, 959, 37, 3, 256, kModeF16Taps, int16_t>(a); }

}  // namespace apt::gpu
