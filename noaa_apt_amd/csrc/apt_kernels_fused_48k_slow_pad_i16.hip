// apt_kernels_fused_48k_slow_pad_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): the strict SPLIT kernel
// compiled for a tap-count bound (kModeStrictPad: any tap count up to kPadT1Max48kSlow, zero-padded table).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_48k_slow_pad_i16(const FusedLaunch &a) { launch_fused_args<13, 30, kPadT1Max48kSlow, 61, 5, 256, kModeStrictPad, int16_t>(a); }

}  // namespace apt::gpu
