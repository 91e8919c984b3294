// apt_kernels_fused_launch.hpp — launch interface between apt_kernels_fused.hip (host-side tables and
// dispatch) and the apt_kernels_fused_*.hip translation units (one k_fused instantiation each).
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

// arguments of one launch (single recording, or a batch described by d_batch)
struct FusedLaunch {
    hipStream_t s;
    const void *x;
    uint64_t n;
    const float *hb, *h2, *h2p;
    float cosphi2, sinphi, inv_sinphi, f16_unscale;
    float *f_out, *c_out, *gm_out;
    uint64_t w, n_corr;
    const FusedRec *d_batch;
    int count;
};

// one function per instantiation, each in its own translation unit
void fused_launch_48k_f32(const FusedLaunch &a);
void fused_launch_48k_i16(const FusedLaunch &a);
void fused_launch_96k_f32(const FusedLaunch &a);
void fused_launch_96k_i16(const FusedLaunch &a);
void fused_launch_48k_f16taps_f32(const FusedLaunch &a);
void fused_launch_48k_f16taps_i16(const FusedLaunch &a);
void fused_launch_48k_batch_f32(const FusedLaunch &a);
void fused_launch_48k_batch_i16(const FusedLaunch &a);

}  // namespace apt::gpu
