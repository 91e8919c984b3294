// apt_kernels_fused_launch.hpp — launch interface between apt_kernels_fused.hip (host-side tables and
// dispatch) and the apt_kernels_fused_*.hip translation units (one k_fused instantiation each).
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

// k_fused's MODE template argument
constexpr int kModeStrict = 0;
constexpr int kModeF16Taps = 1;
constexpr int kModeFast = 2;

// Window samples per stage-1 chunk (one scalar-load wait per chunk) of the specialised kernels, and the dwords of a
// chunk's run in the tap table: CH x (l/2) branch pairs, then the odd branch's taps of the chunk — those of its
// aligned sample pair (q even, q + 1) first, then (CH == 3) that of the sample left over — padded so that three scalar
// loads fetch it (26 = 16 + 8 + 2 dwords; 40 = 16 + 16 + 8).  Host (table builder) and device agree through these.
// Three-sample chunks (80 pinned SGPRs) give the 96 kHz kernels — 1.5 waves per SIMD, bound by the scalar cache's
// latency — half as much cover again per wait: config 3 in fast mode 0.709 -> 0.631 ms per recording, strict 0.660 -> 0.654.
constexpr int fused_chunk(int m, int mode)
{
#ifdef APT_FUSED_CH_ALL
    return APT_FUSED_CH_ALL;
#else
    return m >= 100 ? 3 : 2;  // (48 kHz fast mode: measured 3 % slower with three)
#endif
}
constexpr int fused_chunk_dwords(int l, int ch) { return ch == 2 ? 4 * (l / 2) + 2 : 6 * (l / 2) + 4; }

// arguments of one launch: the recordings of one call (see CallArgs / SlotPtrs in apt_kernels.hpp)
struct FusedLaunch {
    hipStream_t s;
    const CallArgs *call;      // host copy; passed to the kernel by value
    const FusedParams *prm;    // device-resident parameters of the plan
    uint64_t max_w;            // longest recording of the call, work samples (sizes grid.x)
    size_t table_lds_floats;   // TABLE mode: floats of LDS the table and the input tile take
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
// table-driven stage 1 + standard-profile work-rate stages, 512-thread workgroups
void fused_launch_tab_std_f32(const FusedLaunch &a);
void fused_launch_tab_std_i16(const FusedLaunch &a);
void fused_launch_tab_std_fast_f32(const FusedLaunch &a);
void fused_launch_tab_std_fast_i16(const FusedLaunch &a);
// phase-resident taps (stage 1 of rates like 44 100 Hz) + standard-profile work-rate stages, 256-thread workgroups
void fused_launch_phase_std_f32(const FusedLaunch &a);
void fused_launch_phase_std_i16(const FusedLaunch &a);
void fused_launch_phase_std_fast_f32(const FusedLaunch &a);
void fused_launch_phase_std_fast_i16(const FusedLaunch &a);
// ... 512-thread workgroups (256 < l <= 512: 22 050 Hz)
void fused_launch_phase512_std_f32(const FusedLaunch &a);
void fused_launch_phase512_std_i16(const FusedLaunch &a);
void fused_launch_phase512_std_fast_f32(const FusedLaunch &a);
void fused_launch_phase512_std_fast_i16(const FusedLaunch &a);
#ifdef APT_WITH_PROBES
// timing probes (make PROBES=1; APTGPU_PROBE_STOP=1..7; sources under tools/probes/): the fast 48 kHz f32
// kernel cut off after a stage (1..5), or complete with 128 / 192-thread workgroups (6, 7)
void fused_launch_probe1(const FusedLaunch &a);
void fused_launch_probe2(const FusedLaunch &a);
void fused_launch_probe3(const FusedLaunch &a);
void fused_launch_probe4(const FusedLaunch &a);
void fused_launch_probe5(const FusedLaunch &a);
void fused_launch_probe6(const FusedLaunch &a);
void fused_launch_probe7(const FusedLaunch &a);
// ... and the strict kernel cut off after a stage (11..15)
void fused_launch_probe11(const FusedLaunch &a);
void fused_launch_probe12(const FusedLaunch &a);
void fused_launch_probe13(const FusedLaunch &a);
void fused_launch_probe14(const FusedLaunch &a);
void fused_launch_probe15(const FusedLaunch &a);
void fused_launch_probe16(const FusedLaunch &a);  // strict, complete, no HBM reads
void fused_launch_probe8(const FusedLaunch &a);   // fast, complete, no HBM reads
void fused_launch_probe9(const FusedLaunch &a);   // fast, persistent with register prefetch
void fused_launch_probe17(const FusedLaunch &a);  // strict, persistent with register prefetch
#endif

}  // namespace apt::gpu
