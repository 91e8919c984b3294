// apt_kernels_fused_96k_pad2_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): the strict SPLIT kernel
// compiled for a bound on the resampler's tap count AND on the low-pass length (kModeStrictPad2: zero-padded tables).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_96k_pad2_i16(const FusedLaunch &a) { launch_fused_args<13, 100, kPadT1Max96k, kPadT2Max, 3, 256, kModeStrictPad2, int16_t>(a); }

}  // namespace apt::gpu
