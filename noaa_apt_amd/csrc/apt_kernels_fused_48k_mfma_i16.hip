// apt_kernels_fused_48k_mfma_i16.hip — one instantiation of k_fused (see apt_kernels_fused_impl.hpp): APTGPU_MODE_FAST with the
// FIRs on the matrix cores (kModeMfma), any tap count up to kMfmaT1Max48k.
#include "apt_kernels_fused_impl.hpp"

namespace apt::gpu {

void fused_launch_48k_mfma_i16(const FusedLaunch &a) { launch_fused_args<13, 50, kMfmaT1Max48k, 37, 3, 256, kModeMfma, int16_t>(a); }

}  // namespace apt::gpu
