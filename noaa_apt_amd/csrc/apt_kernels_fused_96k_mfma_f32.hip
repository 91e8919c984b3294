// apt_kernels_fused_96k_mfma_f32.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): APTGPU_MODE_FAST with the
// FIRs on the matrix cores (kModeMfma), any tap count up to kMfmaT1Max96k.
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_96k_mfma_f32(const FusedLaunch &a) { launch_fused_args<13, 100, kMfmaT1Max96k, 37, 3, 256, kModeMfma, float>(a); }

}  // namespace apt::gpu
