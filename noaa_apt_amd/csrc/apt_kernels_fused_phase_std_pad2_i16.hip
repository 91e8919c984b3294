// apt_kernels_fused_phase_std_pad2_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): the phase-resident stage 1 with
// 1 branch per thread and the standard profile's work-rate stages compiled for a BOUND on the low-pass length
// (kModeStrictPad2: a tuned demodulation_atten at the rates a sound card records at).
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_phase_std_pad2_i16(const FusedLaunch &a) { launch_fused_args<13, -1, 0, kPadT2Max, 3, 256, kModeStrictPad2, int16_t>(a); }

}  // namespace apt::gpu
