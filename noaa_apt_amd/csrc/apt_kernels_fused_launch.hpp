// apt_kernels_fused_launch.hpp — launch interface between apt_kernels_fused.hip (host-side tables and
// dispatch) and the apt_kernels_fused_*.hip translation units (one k_fused instantiation each).
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

// k_fused's MODE template argument
constexpr int kModeStrict = 0;
constexpr int kModeF16Taps = 1;
constexpr int kModeFast = 2;
// kModeMfma (round 6): APTGPU_MODE_FAST with the resampler on the matrix cores.  Stage 1 is the banded Toeplitz product
// R[16 branches][16 windows] = H[16][K] X[K][16] through v_mfma_f32_16x16x32_bf16, f32 accumulation, on bf16 PIECES of the
// f32 operands: taps h = h0 + h1 + h2 exactly (three 8-bit pieces, split on the host), samples x = x0 + x1 (+ a remainder
// below 2^-16 |x|: none for 16-bit samples), split while the tile goes to LDS; five products (h0 x0, h0 x1, h1 x0, h1 x1,
// h2 x0) carry every term above 2^-24 of the largest.  bf16 has f32's exponent range: no scaling.  The f32-input MFMA,
// which would be bit-identical to kModeFast, shares the VALU's datapath and measured 0.66x; an f16 form with a
// power-of-two scale per sub-tile measured at parity with the VALU kernel, this one 6 % behind it (fast
// mode is bound by the tile's HBM round trip, not by the FIRs).  The work-rate stages behind are kModeFast's.  Tap COUNT
// is a run-time quantity here (the table is zero-padded to the kernel's K): a tuned resample_atten / resample_delta_freq
// stays on a specialised kernel while its taps per branch fit — what this mode is kept for.
constexpr int kModeMfma = 3;
// kModeStrictPad (round 6): kModeStrict's arithmetic in a SPLIT kernel compiled for a tap-count BOUND: the chunk-major table
// is laid out for the bound and holds zeros behind the filter's last tap.  sum + 0 * x = sum exactly for finite x (the sum
// starts at +0 and never becomes -0), so the results are bit-identical to the reference's, which skips those taps; a
// tile whose results are not all finite — where 0 * inf would have put a NaN the reference does not have — is evaluated
// again sample by sample from HBM, in the reference's order.  default_settings.toml:108-140 is a user-editable file:
// a tuned resample_atten / resample_delta_freq changes the tap count, and until round 6 such a plan fell to k_fused_any.
constexpr int kModeStrictPad = 4;
// PHASE stage 1 with ONE branch per thread (l <= 256: 44 100 Hz, 48 kHz at the fast profile, 8 / 16 / 24 / 32 kHz ...): the
// paired input tile goes through LDS in two halves of eight windows each (round 6; kernel and phase_geom must agree, hence
// a macro).  Its 45-52 KB were what held these kernels at three workgroups per CU.
#ifndef APT_PHASE_HALVES
#define APT_PHASE_HALVES 1
#endif
// (not the standard profile's kernel in APTGPU_MODE_FAST: its 68-tap branch is then fetched twice, in segments, by a kernel
// that has half the arithmetic to hide it under — 0.80 -> 0.86 ms per 16 at 44 100 Hz, profiles/r06_phase_halves_ab.txt)
constexpr bool phase_halves(int nq, bool stream, int nthr, int t2, bool fast)
{
    return APT_PHASE_HALVES != 0 && nq == 1 && !stream && nthr == 256 && !(fast && t2 == 37);
}
// kModeStrictPad2 (round 6): kModeStrictPad whose LOW-PASS length is a bound too (T2 = kPadT2Max: h2 / h2p hold zeros behind
// the filter's last tap — the taps an output meets last, since stage 3 walks them in ascending order — and a tile whose F
// values are not all finite is filtered again from D in LDS with the run-time tap count).  demodulation_atten is as
// user-editable as the resampler's settings (default_settings.toml:116) and moves the Kaiser length of the low-pass
// (25 dB: 37 taps; 24: 35; 26: 39): until this mode such a plan fell to k_fused_any, 7 x the stock step at 48 kHz.
constexpr int kModeStrictPad2 = 5;
// low-pass taps the kModeStrictPad2 instantiations are compiled for (standard profile; four pre-halo threads hold up to 51)
constexpr int kPadT2Max = 45;
// FusedGeom's variant argument for a mode
constexpr int fused_geom_var(int mode)
{
    return mode == kModeF16Taps ? 1 : mode == kModeMfma ? 2 : (mode == kModeStrictPad || mode == kModeStrictPad2) ? 3 : 0;
}
// tap counts the padded strict instantiations are compiled for, about an eighth above the stock profiles' counts
// (standard 48 / 96 kHz: 83 / 165 taps per branch, stock 74 / 148; slow 48 / 96 kHz: 241 / 481, stock 215 / 429; fast
// profile at 96 kHz: 56, stock 50)
constexpr int kPadT1Max48k = 1079, kPadT1Max96k = 2145;
constexpr int kPadT1Max48kSlow = 3133, kPadT1Max96kSlow = 6253, kPadT1Max96kFastp = 727;
// tap counts the MFMA instantiations are compiled for (window of a tile's last branch + taps per branch <= K = 128 / 256)
constexpr int kMfmaT1Max48k = 1053, kMfmaT1Max96k = 2119;

// Window samples per stage-1 chunk (one scalar-load wait per chunk) of the specialised kernels; host (table builder)
// and device agree through this and the layout functions below.  (Rounds 1-3: two samples x all 13 branches per chunk
// at 48 kHz, three at 96 kHz, every thread a whole window — 51.5 KB of input tile per 256 / 128 threads.)
constexpr int fused_chunk(int m, int mode)
{
    return 4;  // kSplitChunk
}

// SPLIT stage 1.  An input tile of 256 windows is 51.5 KB of LDS at 48 kHz (three workgroups per CU) and would be 102 KB
// at 96 kHz.  A 256-thread workgroup instead runs stage 1 over TWO sub-tiles of 128 windows, one after the other through
// the same LDS; in each, thread (half h, window a) computes the branches [b0, b0 + nbr) of window a only — h = 0: the
// first (l + 1) / 2 branches, h = 1: the rest.  All 256 threads then share a 26 KB (48 kHz) / 52 KB (96 kHz) tile:
// six workgroups per CU at 48 kHz (the work-rate stages' 26.1 KB set the footprint; round 3: 28.5 KB, five), three at 96 kHz where the
// 128-thread workgroups of rounds 1-3 reached 1.5 waves per SIMD.  The stages behind it see 256 threads x l outputs.  A half's taps: its own chunk-major table over
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
    int lds_pad = 0;           // APTGPU_FUSED_LDS_PAD (A/B switch, read at plan creation): more dynamic LDS than the kernel uses
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
// APTGPU_MODE_FAST on the matrix cores (kModeMfma): 48 / 96 kHz, standard profile, any tap count up to kMfmaT1Max*
void fused_launch_48k_mfma_f32(const FusedLaunch &a);
void fused_launch_48k_mfma_i16(const FusedLaunch &a);
void fused_launch_96k_mfma_f32(const FusedLaunch &a);
void fused_launch_96k_mfma_i16(const FusedLaunch &a);
// strict, any tap count up to kPadT1Max* (kModeStrictPad): 48 / 96 kHz, standard profile
void fused_launch_48k_pad_f32(const FusedLaunch &a);
void fused_launch_48k_pad_i16(const FusedLaunch &a);
void fused_launch_96k_pad_f32(const FusedLaunch &a);
void fused_launch_96k_pad_i16(const FusedLaunch &a);
void fused_launch_48k_slow_pad_f32(const FusedLaunch &a);
void fused_launch_48k_slow_pad_i16(const FusedLaunch &a);
void fused_launch_96k_slow_pad_f32(const FusedLaunch &a);
void fused_launch_96k_slow_pad_i16(const FusedLaunch &a);
void fused_launch_96k_fastp_pad_f32(const FusedLaunch &a);  // (odd m: f32 input only, as the exact-count kernel)
// ... and any low-pass length up to kPadT2Max as well (kModeStrictPad2): 48 / 96 kHz, standard profile
void fused_launch_48k_pad2_f32(const FusedLaunch &a);
void fused_launch_48k_pad2_i16(const FusedLaunch &a);
void fused_launch_96k_pad2_f32(const FusedLaunch &a);
void fused_launch_96k_pad2_i16(const FusedLaunch &a);
// ... and the standard profile's PHASE kernels (every rate a sound card records at) with a low-pass of up to kPadT2Max taps
void fused_launch_phase_std_pad2_f32(const FusedLaunch &a);
void fused_launch_phase_std_pad2_i16(const FusedLaunch &a);
void fused_launch_phase2_std_pad2_f32(const FusedLaunch &a);
void fused_launch_phase2_std_pad2_i16(const FusedLaunch &a);
void fused_launch_phase4_std_pad2_f32(const FusedLaunch &a);
void fused_launch_phase4_std_pad2_i16(const FusedLaunch &a);
// 48 kHz at the slow profile (13 / 30, 2783 taps; 61-tap low-pass, pixel width 5): the same SPLIT form
void fused_launch_48k_slow_f32(const FusedLaunch &a);
void fused_launch_48k_slow_i16(const FusedLaunch &a);
void fused_launch_48k_slow_fast_f32(const FusedLaunch &a);
void fused_launch_48k_slow_fast_i16(const FusedLaunch &a);
// 96 kHz at the slow profile (13 / 60, 5565 taps): the same SPLIT form, strict instantiations only
void fused_launch_96k_slow_f32(const FusedLaunch &a);
void fused_launch_96k_slow_i16(const FusedLaunch &a);
// 96 kHz at the fast profile (13 / 75, 639 taps; odd m: 4-byte window reads, f32 input only — PCM16 payloads are staged)
void fused_launch_96k_fastp_f32(const FusedLaunch &a);
// phase-resident taps + the FAST PROFILE's work-rate stages (43-tap low-pass, pixel width 4), 256-thread workgroups:
// 48 kHz (l = 26), 96 kHz (l = 13, m = 75) and the other rates whose l <= 256 at work rate 16 640
void fused_launch_phase_fastp_f32(const FusedLaunch &a);
void fused_launch_phase_fastp_i16(const FusedLaunch &a);
void fused_launch_phase_fastp_fast_f32(const FusedLaunch &a);
void fused_launch_phase_fastp_fast_i16(const FusedLaunch &a);
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
// ... 256-thread workgroups whose threads hold two / four branches (256 < l <= 512: 22 050 Hz; 512 < l <= 1024: 11 025 Hz)
void fused_launch_phase2_std_f32(const FusedLaunch &a);
void fused_launch_phase2_std_i16(const FusedLaunch &a);
void fused_launch_phase2_std_fast_f32(const FusedLaunch &a);
void fused_launch_phase2_std_fast_i16(const FusedLaunch &a);
void fused_launch_phase4_std_f32(const FusedLaunch &a);
void fused_launch_phase4_std_i16(const FusedLaunch &a);
void fused_launch_phase4_std_fast_f32(const FusedLaunch &a);
void fused_launch_phase4_std_fast_i16(const FusedLaunch &a);
// ... the fast profile's work-rate stages with four / eight branches per thread (44 100 Hz: l = 832; 22 050 Hz: l = 1664)
void fused_launch_phase4_fastp_f32(const FusedLaunch &a);
void fused_launch_phase4_fastp_i16(const FusedLaunch &a);
void fused_launch_phase8_fastp_f32(const FusedLaunch &a);
void fused_launch_phase8_fastp_i16(const FusedLaunch &a);
void fused_launch_phase16_fastp_f32(const FusedLaunch &a);  // (11 025 Hz: l = 3328)
void fused_launch_phase16_fastp_i16(const FusedLaunch &a);
// ... the slow profile's work-rate stages (61-tap low-pass, pixel width 5), taps streamed from the table (197 per branch at
// 44 100 / 22 050 / 11 025 Hz: l = 208 / 416 / 832, m = 441)
void fused_launch_phase_slowp_f32(const FusedLaunch &a);
void fused_launch_phase_slowp_i16(const FusedLaunch &a);
void fused_launch_phase2_slowp_f32(const FusedLaunch &a);
void fused_launch_phase2_slowp_i16(const FusedLaunch &a);
void fused_launch_phase4_slowp_f32(const FusedLaunch &a);
void fused_launch_phase4_slowp_i16(const FusedLaunch &a);
// ... 1024-thread workgroups (512 < l <= 1024), one branch per thread
void fused_launch_phase1024_std_f32(const FusedLaunch &a);
void fused_launch_phase1024_std_i16(const FusedLaunch &a);
void fused_launch_phase1024_std_fast_f32(const FusedLaunch &a);
void fused_launch_phase1024_std_fast_i16(const FusedLaunch &a);
#ifdef APT_WITH_PROBES
// timing probes (make PROBES=1; APTGPU_PROBE_STOP=1..7; sources under tools/probes/): the fast 48 kHz f32
// kernel cut off after a stage (1..5)
void fused_launch_probe1(const FusedLaunch &a);
void fused_launch_probe2(const FusedLaunch &a);
void fused_launch_probe3(const FusedLaunch &a);
void fused_launch_probe4(const FusedLaunch &a);
void fused_launch_probe5(const FusedLaunch &a);
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
