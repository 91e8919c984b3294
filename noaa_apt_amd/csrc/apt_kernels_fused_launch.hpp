// apt_kernels_fused_launch.hpp — launch interface between apt_kernels_fused.hip (host-side tables and
// dispatch) and the apt_kernels_fused_*.hip translation units (one k_fused instantiation each).
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

// k_fused's MODE template argument
constexpr int kModeStrict = 0;
constexpr int kModeF16Taps = 1;
constexpr int kModeFast = 2;

// arguments of one launch: the recordings of one call (see CallArgs / SlotPtrs in apt_kernels.hpp)
struct FusedLaunch {
    hipStream_t s;
    const CallArgs *call;      // host copy; passed to the kernel by value
    const FusedParams *prm;    // device-resident parameters of the plan
    uint64_t max_w;            // longest recording of the call, work samples (sizes grid.x)
};

// one function per instantiation, each in its own translation unit
void fused_launch_48k_f32(const FusedLaunch &a);
void fused_launch_48k_i16(const FusedLaunch &a);
void fused_launch_96k_f32(const FusedLaunch &a);
void fused_launch_96k_i16(const FusedLaunch &a);
void fused_launch_48k_f16taps_f32(const FusedLaunch &a);
void fused_launch_48k_f16taps_i16(const FusedLaunch &a);
void fused_launch_48k_fast_f32(const FusedLaunch &a);
void fused_launch_48k_fast_i16(const FusedLaunch &a);
void fused_launch_96k_fast_f32(const FusedLaunch &a);
void fused_launch_96k_fast_i16(const FusedLaunch &a);

}  // namespace apt::gpu
