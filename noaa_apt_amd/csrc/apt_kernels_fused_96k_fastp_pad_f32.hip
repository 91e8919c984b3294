// apt_kernels_fused_96k_fastp_pad_f32.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): the strict SPLIT kernel
// compiled for a tap-count bound (kModeStrictPad: any tap count up to kPadT1Max96kFastp, zero-padded table).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_96k_fastp_pad_f32(const FusedLaunch &a) { launch_fused_args<13, 75, kPadT1Max96kFastp, 43, 4, 256, kModeStrictPad, float>(a); }

}  // namespace apt::gpu
