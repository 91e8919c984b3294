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
// m >= 100 (the 96 kHz kernels): 4 = the SPLIT layout below.  (Until round 3 those kernels were 128-thread
// workgroups — 1.5 waves per SIMD under their 52 KB input tile — with three-sample chunks of all 13 branches.)
constexpr int fused_chunk(int m, int mode)
{
    return 4;  // the SPLIT layout below (round 3: every specialised kernel with f32 taps)
}
constexpr int fused_chunk_dwords(int l, int ch) { return ch == 2 ? 4 * (l / 2) + 2 : 6 * (l / 2) + 4; }

// SPLIT stage 1 (m >= 100: an input tile of 256 windows would be 102 KB of LDS).  A 256-thread workgroup runs stage 1
// over TWO sub-tiles of 128 windows, one after the other through the same LDS; in each, thread (half h, window a)
// computes the branches [b0, b0 + nbr) of window a only — h = 0: the first (l + 1) / 2 branches, h = 1: the rest —
// so that all 256 threads (four waves, three workgroups per CU: 3 waves per SIMD) share a 52 KB tile.  The stages
// behind it see 256 threads x l outputs as in the 48 kHz kernels.  A half's taps: its own chunk-major table over
// the window samples [w0, w0 + 4 nch) its branches use, w0 a multiple of 4 (16-byte LDS reads); a chunk is 4
// samples x 3 branch pairs (24 dwords) followed by the odd branch's 4 taps (half 0 only): 28 = 16 + 8 + 4 dwords.
constexpr int kSplitChunk = 4, kSplitChunkDwords = 28;
constexpr int fused_branch_first(int l, int m, int b) { return (b * m + l - 1) / l; }
constexpr int fused_branch_end(int l, int m, int t1, int b)  // one past the last window sample branch b has a tap for
{
    const int cb = fused_branch_first(l, m, b), pb = cb * l - b * m;
    return cb + (t1 - pb + l - 1) / l;
}
constexpr int fused_split_b0(int l, int h) { return h == 0 ? 0 : (l + 1) / 2; }
constexpr int fused_split_nbr(int l, int h) { return h == 0 ? (l + 1) / 2 : l / 2; }
constexpr int fused_split_w0(int l, int m, int h) { return fused_branch_first(l, m, fused_split_b0(l, h)) / 4 * 4; }
constexpr int fused_split_nch(int l, int m, int t1, int h)
{
    int end = 0;
    for (int b = fused_split_b0(l, h); b < fused_split_b0(l, h) + fused_split_nbr(l, h); ++b)
        end = fused_branch_end(l, m, t1, b) > end ? fused_branch_end(l, m, t1, b) : end;
    return (end - fused_split_w0(l, m, h) + kSplitChunk - 1) / kSplitChunk;
}
// floats in front of half h's table (one zero row behind each half)
constexpr int fused_split_table_offset(int l, int m, int t1, int h)
{
    return h == 0 ? 0 : (fused_split_nch(l, m, t1, 0) + 1) * kSplitChunkDwords;
}

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
