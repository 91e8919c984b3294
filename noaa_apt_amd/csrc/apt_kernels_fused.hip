// apt_kernels_fused.hip — the fused, specialised front end of decode() for gfx950.
//
//   x (input rate) --polyphase FIR--> R --AM envelope--> D --low-pass--> F --sync corr--> C
//
// One launch produces F (the filtered work-rate signal, the only intermediate that goes
// back to HBM) and GM (the maximum of the sync cross-correlation over groups of GS
// consecutive positions — all the peak picker needs; see apt_kernels_sync.hip).  R, D
// and C never leave the CU.  Reference: src/decode.rs:77-110, src/dsp.rs:186-289,
// 350-410, src/decode.rs:225-233.
//
// Behind stage 1 thread t of a 256-thread workgroup owns the 13 consecutive work-rate samples 13 t .. 13 t + 12 of a
// tile of 3328 (halo threads either side: the low-pass and the envelope look back, the correlation looks ahead); R, D, F
// and the pulse sums go through two LDS regions.  Stage 1 — the polyphase resampler — comes in three forms:
//   SPLIT (M > 0: 48 / 96 kHz)   taps wave-uniform: scalar loads into pinned SGPR tuples, every window sample read once
//                                from LDS and broadcast against a pair of taps feeding a pair of accumulators;
//   PHASE (M < 0: every other rate a sound card records at)   taps thread-resident, samples paired in LDS;
//   TABLE (M == 0: fallback)     taps and samples from LDS.
// Every product and sum is a separate multiplication / addition in the reference's order: bit-identical to the scalar
// Rust loop (no FMA; #pragma fp contract(off)).  This file: the host side — which kernel serves which geometry, the
// tap tables in the layouts the kernels read, the launch dispatch.
#include "apt_kernels_fused_launch.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#pragma clang fp contract(off)

namespace apt::gpu {


bool fused_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    if (l == 13 && m == 50 && t1 == 959 && t2 == 37 && pw == 3) return true;    // 48 kHz, standard
    if (l == 13 && m == 100 && t1 == 1915 && t2 == 37 && pw == 3) return true;  // 96 kHz, standard
    if (l == 13 && m == 30 && t1 == 2783 && t2 == 61 && pw == 5) return true;    // 48 kHz, slow profile
    if (l == 13 && m == 60 && t1 == 5565 && t2 == 61 && pw == 5) return true;    // 96 kHz, slow profile (strict instantiations only)
    if (l == 13 && m == 75 && t1 == 639 && t2 == 43 && pw == 4) return true;     // 96 kHz, fast profile (strict, f32 input only: odd m)
    return false;
}

uint32_t fused_group_size(uint32_t l) { return 4 * l; }

bool fused_takes_pcm16(uint32_t l, uint32_t m) { (void)l; return m % 2 == 0; }  // (sample pairs are read as dwords)

// floats in the stage-1 tap table: the SPLIT layout (apt_kernels_fused_launch.hpp), one chunk-major table per half
uint32_t fused_tap_table_floats(uint32_t l, uint32_t m, uint32_t t1, int ch)
{
    (void)ch;  // kSplitChunk
    // (t1: the tap count the kernel is compiled for — the filter's own, or kModeStrictPad's bound)
    const int li = static_cast<int>(l), mi = static_cast<int>(m), ti = static_cast<int>(t1);
    return static_cast<uint32_t>(fused_split_table_offset(li, mi, ti, 1) + (fused_split_nch(li, mi, ti, 1) + 1) * kSplitChunkDwords);
}

// host: per half h, chunk c holds for its window samples q = w0 + 4 c + e (e < 4) the pairs (tap of branch b0 + 2k at
// q, tap of branch b0 + 2k + 1 at q) at dwords 6 e + 2 k, k < 3, then (a half with an odd number of branches) the last
// branch's taps at q = w0 + 4 c .. + 3 at dwords 24 .. 27; 0 where a branch has no tap for q; one zero row behind
// each half.  Tap of branch b at window sample q: coeff[p_b + (q - c_b) l], c_b = ceil(b m / l), p_b = c_b l - b m
// (dsp.rs:252-263).
// t1_layout (0: t1): the tap count the kernel is compiled for when that is a bound (kModeStrictPad) — the table then has
// that kernel's chunks and zeros behind the filter's last tap.
void fused_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, int ch, float *hs, uint32_t t1_layout)
{
    (void)ch;
    if (t1_layout == 0) t1_layout = t1;
    const uint32_t tp = (t1_layout + l - 1) / l;
    const uint32_t clast = ((l - 1) * m + l - 1) / l;
    const uint32_t win = clast + tp;
    auto tap = [&](uint32_t b, int64_t q) -> float {
        const uint32_t cb = (b * m + l - 1) / l;
        const uint32_t pb = cb * l - b * m;
        if (q < static_cast<int64_t>(cb) || q >= static_cast<int64_t>(win)) return 0.f;
        const uint64_t j = pb + static_cast<uint64_t>(q - cb) * l;
        return j < t1 ? coeff[j] : 0.f;
    };
    const int li = static_cast<int>(l), mi = static_cast<int>(m), ti = static_cast<int>(t1_layout);
    for (int h = 0; h < 2; ++h) {
        const int b0 = fused_split_b0(li, h), nbr = fused_split_nbr(li, h), w0 = fused_split_w0(li, mi, h);
        const int nch_h = fused_split_nch(li, mi, ti, h);
        float *base = hs + fused_split_table_offset(li, mi, ti, h);
        for (int c = 0; c <= nch_h; ++c) {
            float *row = base + static_cast<size_t>(c) * kSplitChunkDwords;
            for (int i = 0; i < kSplitChunkDwords; ++i) row[i] = 0.f;
            if (c == nch_h) break;
            for (int e = 0; e < kSplitChunk; ++e) {
                const int64_t q = w0 + static_cast<int64_t>(kSplitChunk) * c + e;
                for (int k = 0; k < nbr / 2; ++k) {
                    row[6 * e + 2 * k] = tap(static_cast<uint32_t>(b0 + 2 * k), q);
                    row[6 * e + 2 * k + 1] = tap(static_cast<uint32_t>(b0 + 2 * k + 1), q);
                }
                if (nbr & 1) row[24 + e] = tap(static_cast<uint32_t>(b0 + nbr - 1), q);
            }
        }
    }
}

int fused_chunk_of(uint32_t m, bool fast)
{
    return fused_chunk(static_cast<int>(m), fast ? kModeFast : kModeStrict);
}

// host: stage-3 table h2p[k] = (h2[k-1], h2[k]) for k = 0 .. t2 (0 outside the filter)
void fused_lowpass_pairs(const float *h2, uint32_t t2, float *h2p)
{
    for (uint32_t k = 0; k <= t2; ++k) {
        h2p[2 * k] = k >= 1 ? h2[k - 1] : 0.f;
        h2p[2 * k + 1] = k < t2 ? h2[k] : 0.f;
    }
}

uint32_t fused_f16_table_dwords(uint32_t l, uint32_t m, uint32_t t1)
{
    const uint32_t tp = (t1 + l - 1) / l;
    const uint32_t clast = ((l - 1) * m + l - 1) / l;
    return ((clast + tp + 1) / 2) * 16;
}

// host: [ceil(WIN/2)][16] half2 bit patterns — entry (qp, b) = taps of branch b at window samples
// (2qp, 2qp+1), prescaled by a power of two into the fp16 normal range; returns 2^-s
float fused_f16_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, uint32_t *table)
{
    const uint32_t tp = (t1 + l - 1) / l;
    const uint32_t clast = ((l - 1) * m + l - 1) / l;
    const uint32_t win = clast + tp;
    auto tap = [&](uint32_t b, uint32_t q) -> float {
        const uint32_t cb = (b * m + l - 1) / l;
        const uint32_t pb = cb * l - b * m;
        if (q < cb || q >= win) return 0.f;
        const uint64_t j = pb + static_cast<uint64_t>(q - cb) * l;
        return j < t1 ? coeff[j] : 0.f;
    };
    float mx = 0.f;
    for (uint32_t j = 0; j < t1; ++j) mx = fmaxf(mx, fabsf(coeff[j]));
    int e = 0;
    if (mx > 0.f) (void)frexpf(mx, &e);
    const int sh = -e + 1;  // scaled maximum in [1, 2)
    const float scale = ldexpf(1.f, sh);
    auto bits = [](float v) -> uint32_t {
        const _Float16 h = static_cast<_Float16>(v);
        uint16_t u;
        __builtin_memcpy(&u, &h, 2);
        return u;
    };
    const uint32_t nqp = (win + 1) / 2;
    for (uint32_t qp = 0; qp < nqp; ++qp)
        for (uint32_t b = 0; b < 16; ++b)
            table[qp * 16 + b] = b < l ? (bits(tap(b, 2 * qp) * scale) | (bits(tap(b, 2 * qp + 1) * scale) << 16)) : 0u;
    return ldexpf(1.f, -sh);
}

bool fused_f16_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    return l == 13 && m == 50 && t1 == 959 && t2 == 37 && pw == 3;
}

// Bound on |a - c| / sum|F| for a = the pulse-sum evaluation and c = the reference's sequential chain of
// the 38*pw-term sync correlation: gamma(38 pw - 1) + gamma(pw + 18) with gamma(k) = k u / (1 - k u),
// u = 2^-24, plus 3 % for the rounding of the bound's own arithmetic (the sum of |F|, the product, the
// two additions that form lo and hi: < 40 u relative).  `scale` (APTGPU_GM_SLACK_SCALE, tests; read at plan creation —
// LaunchSwitches) widens it so that the picker's exact settlement of open comparisons is exercised on ordinary inputs.
float fused_gm_slack(uint32_t pw, float scale)
{
    return static_cast<float>(38u * pw - 1u + pw + 18u) * 1.03f * 0x1p-24f * 1.0001f * (scale >= 1.f ? scale : 1.f);
}

bool fused_fast_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    if (l == 13 && m == 60 && t1 == 5565) return false;  // (100 KB of unrolled taps per instantiation: the strict kernel serves fast mode)
    if (l == 13 && m == 75) return false;
    return fused_supported(l, m, t1, t2, pw);
}

// ---- kModeStrictPad: the strict SPLIT kernels compiled for a tap-count bound (apt_kernels_fused_launch.hpp)
// kModeStrictPad2: the low-pass bound of the instantiation that serves a low-pass of t2 taps other than the profile's
// (a tuned demodulation_atten), or 0 — the standard profile's work-rate stages at 48 / 96 kHz
uint32_t fused_pad_t2(uint32_t l, uint32_t m, uint32_t t2, uint32_t pw)
{
    if (l != 13 || pw != 3 || (m != 50 && m != 100)) return 0;
    if ((t2 & 1u) == 0 || t2 == 37 || t2 > static_cast<uint32_t>(kPadT2Max)) return 0;
    return static_cast<uint32_t>(kPadT2Max);
}
uint32_t fused_pad_t1(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    if (l != 13 || (t1 & 1u) == 0) return 0;  // (Kaiser lengths are odd: filters.rs:164-167)
    uint32_t bound = 0;
    if ((t2 == 37 || fused_pad_t2(l, m, t2, pw) != 0) && pw == 3) bound = m == 50 ? kPadT1Max48k : m == 100 ? kPadT1Max96k : 0;  // standard profile
    else if (t2 == 61 && pw == 5) bound = m == 30 ? kPadT1Max48kSlow : m == 60 ? kPadT1Max96kSlow : 0;  // slow profile
    else if (t2 == 43 && pw == 4) bound = m == 75 ? kPadT1Max96kFastp : 0;                            // fast profile, 96 kHz
    return t1 <= bound ? bound : 0;
}

// ---- kModeMfma: the FIRs as banded Toeplitz products on the matrix cores (apt_kernels_fused_launch.hpp)
bool fused_mfma_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    if (l != 13 || t2 != 37 || pw != 3 || (t1 & 1u) == 0) return false;  // (Kaiser lengths are odd: filters.rs:164-167)
    if (m == 50) return t1 <= static_cast<uint32_t>(kMfmaT1Max48k);
    if (m == 100) return t1 <= static_cast<uint32_t>(kMfmaT1Max96k);
    return false;
}

namespace {
uint32_t mfma_kpad(uint32_t l, uint32_t m)
{
    const uint32_t t1max = m == 50 ? kMfmaT1Max48k : kMfmaT1Max96k;
    const uint32_t win = ((l - 1) * m + l - 1) / l + (t1max + l - 1) / l;
    return (win + 31u) / 32u * 32u;
}
// a row-major [rows <= 16][K] matrix -> A fragments of v_mfma_f32_16x16x32_bf16, three bf16 pieces per entry:
// table[(piece * nks + s) * 64 + lane] = 8 halves = entries (row = lane & 15, k = 32 s + 8 (lane >> 4) + j), j < 8, of
// piece 0 = the upper half of v's f32 pattern, piece 1 = that of v - piece 0, piece 2 = v - piece 0 - piece 1: three 8-bit
// pieces of the 24-bit significand, v = p0 + p1 + p2 exactly (subnormal taps aside: their last piece is truncated)
template <typename At>
void mfma_fragments(At &&at, uint32_t nks, uint32_t *table)
{
    auto upper = [](float v) -> uint32_t {
        uint32_t u;
        __builtin_memcpy(&u, &v, 4);
        return u >> 16;
    };
    auto value = [](uint32_t h) -> float {
        const uint32_t u = h << 16;
        float v;
        __builtin_memcpy(&v, &u, 4);
        return v;
    };
    for (uint32_t s = 0; s < nks; ++s)
        for (uint32_t lane = 0; lane < 64; ++lane)
            for (uint32_t jp = 0; jp < 4; ++jp) {
                uint32_t w[3] = {0, 0, 0};
                for (uint32_t h = 0; h < 2; ++h) {
                    const float v = at(lane & 15u, 32u * s + 8u * (lane >> 4) + 2u * jp + h);
                    const uint32_t p0 = upper(v);
                    const float r1 = v - value(p0);
                    const uint32_t p1 = upper(r1);
                    const float r2 = r1 - value(p1);
                    const uint32_t p2 = upper(r2);
                    w[0] |= p0 << (16u * h);
                    w[1] |= p1 << (16u * h);
                    w[2] |= p2 << (16u * h);
                }
                for (uint32_t pc = 0; pc < 3; ++pc) table[((pc * nks + s) * 64u + lane) * 4u + jp] = w[pc];
            }
}
}  // namespace

uint32_t fused_mfma_table_dwords(uint32_t l, uint32_t m) { return 3u * (mfma_kpad(l, m) / 32u) * 64u * 4u; }

// host: H[b][q] = coeff[p_b + (q - c_b) l] (0 outside the branch's taps): branch b of a window against window sample q
// (dsp.rs:252-263; c_b = ceil(b m / l), p_b = c_b l - b m), rows 13 .. 15 zero
void fused_mfma_table(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, uint32_t *table)
{
    auto at = [&](uint32_t b, uint32_t q) -> float {
        if (b >= l) return 0.f;
        const uint32_t cb = (b * m + l - 1) / l, pb = cb * l - b * m;
        if (q < cb) return 0.f;
        const uint64_t j = pb + static_cast<uint64_t>(q - cb) * l;
        return j < t1 ? coeff[j] : 0.f;
    };
    mfma_fragments(at, mfma_kpad(l, m) / 32u, table);  // (no prescale: bf16 pieces carry f32's exponent)
}

bool fused_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, int mode,
                     bool pcm16, const CallArgs &call, const FusedParams *d_prm, uint64_t max_w, int lds_pad)
{
    if (call.count == 0 || call.count > static_cast<uint32_t>(kMaxCall)) return false;
    if (pcm16)  // dword loads of sample pairs
        for (uint32_t i = 0; i < call.count; ++i)
            if (reinterpret_cast<uintptr_t>(call.rec[i].x) & 3u) return false;
    const FusedLaunch a{s, &call, d_prm, max_w, 0, lds_pad};
    if (mode == kModeStrictPad) {
        if (fused_pad_t1(l, m, t1, t2, pw) == 0) return false;
        if (fused_pad_t2(l, m, t2, pw) != 0) {  // the low-pass length a bound too (kModeStrictPad2)
            if (m == 50) pcm16 ? fused_launch_48k_pad2_i16(a) : fused_launch_48k_pad2_f32(a);
            else pcm16 ? fused_launch_96k_pad2_i16(a) : fused_launch_96k_pad2_f32(a);
        } else
        if (m == 50) pcm16 ? fused_launch_48k_pad_i16(a) : fused_launch_48k_pad_f32(a);
        else if (m == 100) pcm16 ? fused_launch_96k_pad_i16(a) : fused_launch_96k_pad_f32(a);
        else if (m == 30) pcm16 ? fused_launch_48k_slow_pad_i16(a) : fused_launch_48k_slow_pad_f32(a);
        else if (m == 60) pcm16 ? fused_launch_96k_slow_pad_i16(a) : fused_launch_96k_slow_pad_f32(a);
        else if (m == 75 && !pcm16) fused_launch_96k_fastp_pad_f32(a);
        else return false;
        return true;
    }
    if (mode == kModeMfma) {
        if (!fused_mfma_supported(l, m, t1, t2, pw)) return false;
        if (m == 50) pcm16 ? fused_launch_48k_mfma_i16(a) : fused_launch_48k_mfma_f32(a);
        else pcm16 ? fused_launch_96k_mfma_i16(a) : fused_launch_96k_mfma_f32(a);
        return true;
    }
    if (l == 13 && m == 50 && t1 == 959 && t2 == 37 && pw == 3) {
        if (mode == kModeF16Taps)  // fp16-tap stage 1: hb is the half2 table of fused_f16_branch_taps
            pcm16 ? fused_launch_48k_f16taps_i16(a) : fused_launch_48k_f16taps_f32(a);
        else if (mode == kModeFast) {
#ifdef APT_WITH_PROBES
            static const int probe = [] {
                const char *e = std::getenv("APTGPU_PROBE_STOP");
                return e ? std::atoi(e) : 0;
            }();
            if (!pcm16 && probe >= 1 && probe <= 9 && probe != 6 && probe != 7) {  // (6, 7: the 128 / 192-thread forms, gone with the unsplit stage 1)
                void (*const fn[9])(const FusedLaunch &) = {fused_launch_probe1, fused_launch_probe2, fused_launch_probe3,
                                                            fused_launch_probe4, fused_launch_probe5, nullptr,
                                                            nullptr, fused_launch_probe8, fused_launch_probe9};
                fn[probe - 1](a);
            } else
#endif
            {
                pcm16 ? fused_launch_48k_fast_i16(a) : fused_launch_48k_fast_f32(a);
            }
        }
        else {
#ifdef APT_WITH_PROBES
            static const int sprobe = [] {
                const char *e = std::getenv("APTGPU_PROBE_STOP");
                return e ? std::atoi(e) : 0;
            }();
            if (!pcm16 && sprobe >= 11 && sprobe <= 17) {
                void (*const fn[7])(const FusedLaunch &) = {fused_launch_probe11, fused_launch_probe12, fused_launch_probe13,
                                                            fused_launch_probe14, fused_launch_probe15, fused_launch_probe16, fused_launch_probe17};
                fn[sprobe - 11](a);
            } else
#endif
            {
                pcm16 ? fused_launch_48k_i16(a) : fused_launch_48k_f32(a);
            }
        }
        return true;
    }
    if (l == 13 && m == 30 && t1 == 2783 && t2 == 61 && pw == 5 && mode != kModeF16Taps) {
        if (mode == kModeFast)
            pcm16 ? fused_launch_48k_slow_fast_i16(a) : fused_launch_48k_slow_fast_f32(a);
        else
            pcm16 ? fused_launch_48k_slow_i16(a) : fused_launch_48k_slow_f32(a);
        return true;
    }
    if (l == 13 && m == 75 && t1 == 639 && t2 == 43 && pw == 4 && mode == kModeStrict && !pcm16) {
        fused_launch_96k_fastp_f32(a);
        return true;
    }
    if (l == 13 && m == 60 && t1 == 5565 && t2 == 61 && pw == 5 && mode == kModeStrict) {
        pcm16 ? fused_launch_96k_slow_i16(a) : fused_launch_96k_slow_f32(a);
        return true;
    }
    if (l == 13 && m == 100 && t1 == 1915 && t2 == 37 && pw == 3 && mode != kModeF16Taps) {
        // twice the input per work sample: 128-thread workgroups keep the x tile at 51.8 KB
        if (mode == kModeFast)
            pcm16 ? fused_launch_96k_fast_i16(a) : fused_launch_96k_fast_f32(a);
        else
            pcm16 ? fused_launch_96k_i16(a) : fused_launch_96k_f32(a);
        return true;
    }
    return false;
}


// ---- table-driven stage 1 (k_fused in TABLE mode)
bool fused_table_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, TableGeom *geom)
{
    constexpr uint32_t kThreads = 512, kPerThread = 13, kTile = kThreads * kPerThread;
    // 80 KB: two workgroups per CU.  APTGPU_TABLE_LDS_KB (A/B switch, <= 160) lets larger tables / tiles in at
    // one workgroup per CU.
    uint32_t kLdsFloats = 20480;
    if (const char *e = std::getenv("APTGPU_TABLE_LDS_KB")) {
        const long kb = std::atol(e);
        if (kb >= 16 && kb <= 160) kLdsFloats = static_cast<uint32_t>(kb) * 256u;
    }
    if (l < 2 || m == 0 || t1 == 0 || t2 != 37 || pw != 3) return false;  // (work-rate stages: standard profile)
    if (static_cast<uint64_t>(kTile + 4096) * m + l > 0x7fffffffull) return false;  // 32-bit in-tile index math
    TableGeom g{};
    g.l = l;
    g.m = m;
    g.jlim = 2 * ((t1 - 1) / 2) + 1;
    const uint32_t per_phase = (g.jlim + l - 1) / l;
    g.tpp = per_phase | 1u;  // odd stride: rows of consecutive phases start in different banks
    g.xt = (static_cast<uint32_t>((static_cast<uint64_t>(kTile) * m + l - 1) / l) + per_phase + 8 + 3) & ~3u;
    g.off_x = (l * g.tpp + 3u) & ~3u;
    g.step_q = static_cast<uint32_t>((static_cast<uint64_t>(kThreads) * m) / l);
    g.step_r = static_cast<uint32_t>((static_cast<uint64_t>(kThreads) * m) % l);
    g.jl_a = g.jlim / l;
    g.jl_b = g.jlim % l;
    if (static_cast<uint64_t>(g.off_x) + g.xt > kLdsFloats) return false;
    if (geom) *geom = g;
    return true;
}

bool fused_table_front_end(hipStream_t s, const TableGeom &geom, int mode, bool pcm16, const CallArgs &call,
                           const FusedParams *d_prm, uint64_t max_w)
{
    if (call.count == 0 || call.count > static_cast<uint32_t>(kMaxCall)) return false;
    if (pcm16)
        for (uint32_t i = 0; i < call.count; ++i)
            if (reinterpret_cast<uintptr_t>(call.rec[i].x) & 1u) return false;
    const FusedLaunch a{s, &call, d_prm, max_w, static_cast<size_t>(geom.off_x) + geom.xt};
    if (mode == kModeFast)
        pcm16 ? fused_launch_tab_std_fast_i16(a) : fused_launch_tab_std_fast_f32(a);
    else if (mode == kModeStrict)
        pcm16 ? fused_launch_tab_std_i16(a) : fused_launch_tab_std_f32(a);
    else
        return false;
    return true;
}


// ---- phase-resident stage 1 (k_fused in PHASE mode)
namespace {
constexpr uint32_t kPhaseOutputs = 16;
constexpr uint32_t kPhaseMaxPerm = 8;
uint32_t phase_tpp(uint32_t l, uint32_t t1, bool stream = false)
{
    const uint32_t jlim = 2 * ((t1 - 1) / 2) + 1;
    return stream ? ((jlim + l - 1) / l + 15u) & ~15u : ((jlim + l - 1) / l + 3u) & ~3u;  // (streamed: whole chunks of 16)
}
// the tile of a PHASE workgroup (FusedGeom with TABLE geometry: whole groups of four halo threads either side)
struct PhaseTile {
    uint32_t pre_k, own_k;
};
PhaseTile phase_tile(uint32_t threads, uint32_t t2, uint32_t pw)
{
    const uint32_t pre = ((t2 + 1 + 12) / 13 + 3) / 4 * 4, post = ((38 * pw - 1 + 12) / 13 + 3) / 4 * 4;
    const uint32_t own = (threads - pre - post) / 4 * 4;
    return PhaseTile{pre * 13, own * 13};
}
uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    while (b) {
        const uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
// taps per branch a thread's registers hold (TPPM in k_fused)
uint32_t phase_tap_regs(uint32_t threads, uint32_t t2, uint32_t nq)
{
    if (threads == 1024) return 24;
    if (threads == 512) return 40;
    if (t2 == 43) return 28;
    return nq == 1 ? 76u : nq == 2 ? 36u : 20u;
}
// geometry for workgroups of `threads` (256: three per CU; 512: two; 1024: one) whose threads hold nq branches each
bool phase_geom(uint32_t threads, uint32_t nq, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, TableGeom *geom,
                bool stream = false)
{
    const uint32_t tile = threads * 13;
    if (nq == 1 ? l > threads : (l % nq != 0 || l / nq > threads || l <= threads)) return false;
    if (static_cast<uint64_t>(tile + 4096) * m + l > 0x7fffffffull) return false;  // 32-bit in-tile index math
    TableGeom g{};
    g.l = l;
    g.m = m;
    g.nq = nq;
    g.jlim = 2 * ((t1 - 1) / 2) + 1;
    const uint32_t per_phase = (g.jlim + l - 1) / l;
    if (!stream && per_phase > phase_tap_regs(threads, t2, nq)) return false;
    g.stream = stream ? 1u : 0u;
    g.tpp = phase_tpp(l, t1, stream);
    // outputs between a branch's consecutive outputs: as many whole periods of l as the threads hold (nq = 1), or l
    const uint32_t stride = nq == 1 ? l * (threads / l) : l;
    if ((tile + stride - 1) / stride > std::max(1u, kPhaseOutputs / nq)) return false;
    g.step_r = stride;
    g.step_q = static_cast<uint32_t>(static_cast<uint64_t>(stride) * m / l);  // exact: l divides stride
    // slot stride (TableGeom::sq): four / eight branches per thread — every thread takes slots, `threads` apart, and the
    // slots >= l do not exist; else l / nq slots of nq branches each (A/B switch, plan creation: APTGPU_PHASE_BALANCED=0 / 1)
    {
        const char *bal = std::getenv("APTGPU_PHASE_BALANCED");
        // (default: where it measured faster — profiles/r06_phase_balanced_ab.txt: four branches at the standard and slow
        // profiles, eight at the fast one; not the fast profile's four (44 100 Hz) and sixteen, not two branches)
        const bool balanced = (bal && (bal[0] == '0' || bal[0] == '1')) ? bal[0] == '1' : ((nq == 4 && t2 != 43) || nq == 8);
        g.sq = nq == 1 ? stride : (balanced ? threads : stride / nq);
    }
    // paired input tile: kPhaseOutputs / nq / 2 regions of off_x f2 entries — a branch's window starts at most
    // step_q + 4 entries into its region and is per_phase long
    g.off_x = g.step_q + per_phase + 8;
    // (the tile loader covers a region in 1024 / threads rounds; the fast profile's forms with four / eight branches per
    // thread — one or two regions only — in nine)
    if (g.off_x > ((t2 == 43 && nq > 2) ? 2304u : 1024u)) return false;
    g.xt = std::max(1u, kPhaseOutputs / nq / 2) * 2 * g.off_x;  // (sixteen branches: one region whose second half stays zero)
    // LDS: three 256-thread workgroups (53 KB each), two 512-thread ones (80 KB) or one of 1024 threads per CU
    if (g.xt > (threads == 256 ? 13600u : threads == 512 ? 20480u : 36000u)) return false;
    g.jl_a = g.jlim / l;
    g.jl_b = g.jlim % l;
    // the thread assignment lists: one per distinct tile phase rb = ((tile * own_k - pre_k) * m) mod l
    const PhaseTile pt = phase_tile(threads, t2, pw);
    const uint32_t step = static_cast<uint32_t>((static_cast<uint64_t>(pt.own_k) * m) % l);
    uint32_t nperm = step ? l / gcd_u32(l, step) : 1u;
    g.exact = (nperm <= kPhaseMaxPerm && (nperm & (nperm - 1)) == 0) ? 1u : 0u;
    if (!g.exact) nperm = 1;  // (then the list of tile 0 serves every tile: same results, more bank conflicts, divisions in the kernel)
    g.nthr = threads;
    g.nperm = nperm;
    g.perm_off = (l * g.tpp + 3u) & ~3u;
    g.cp_off = (g.perm_off + nperm * threads + 3u) & ~3u;
    g.tt_off = (g.cp_off + nperm * threads * nq + 3u) & ~3u;
    if (const char *e = std::getenv("APTGPU_PHASE_TT"); !g.exact || (e && e[0] == '0')) g.tt_off = 0;  // (A/B switch: 0 = phase-major rows only)
    for (g.perm_shift = 0; (1u << g.perm_shift) < nperm; ++g.perm_shift) {}
    if (g.exact) {
        g.xd = static_cast<uint32_t>(static_cast<uint64_t>(nperm) * pt.own_k * m / l);  // exact: the period's definition
        for (uint32_t r = 0; r < nperm; ++r) {
            const int64_t km = (static_cast<int64_t>(r) * pt.own_k - static_cast<int64_t>(pt.pre_k)) * static_cast<int64_t>(m);
            int64_t x0 = km / static_cast<int64_t>(l);
            if (x0 * static_cast<int64_t>(l) > km) --x0;  // floor
            g.x0r[r] = static_cast<int32_t>(x0);
            g.rbr[r] = static_cast<uint32_t>(km - x0 * static_cast<int64_t>(l));
        }
    }
    if (geom) *geom = g;
    return true;
}
}  // namespace

// kModeStrictPad2 on the PHASE kernels: the low-pass bound of the instantiation that serves a low-pass of t2 taps other than
// the standard profile's 37 (a tuned demodulation_atten at a sound-card rate), or 0
namespace {
uint32_t phase_pad_t2_bound(uint32_t t2, uint32_t pw)
{
    if (pw != 3 || (t2 & 1u) == 0 || t2 == 37 || t2 > static_cast<uint32_t>(kPadT2Max)) return 0;
    return static_cast<uint32_t>(kPadT2Max);
}
}  // namespace
uint32_t fused_phase_pad_t2(uint32_t t2, uint32_t pw)  // (plan creation: reads the A/B switch)
{
    const char *off = std::getenv("APTGPU_FUSED_PAD");  // (tests: 0 = as until round 6, k_fused_any)
    if (off && off[0] == '0') return 0;
    return phase_pad_t2_bound(t2, pw);
}

bool fused_phase_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, TableGeom *geom)
{
    if (l < 2 || m == 0 || t1 == 0) return false;
    // work-rate stages: the standard profile's (256 threads with one, two or four branches each; 512- and 1024-thread
    // workgroups), the fast profile's (256 threads, one branch)
    if (t2 == 43 && pw == 4)
        return phase_geom(256, 1, l, m, t1, t2, pw, geom) || phase_geom(256, 4, l, m, t1, t2, pw, geom) ||
               phase_geom(256, 8, l, m, t1, t2, pw, geom) || phase_geom(256, 16, l, m, t1, t2, pw, geom);
    // the slow profile's (61-tap low-pass, pixel width 5): its resampling filters (197 taps per branch at the sound-card
    // rates) do not fit the registers — the streamed form
    if (t2 == 61 && pw == 5)
        return phase_geom(256, 1, l, m, t1, t2, pw, geom, true) || phase_geom(256, 2, l, m, t1, t2, pw, geom, true) ||
               phase_geom(256, 4, l, m, t1, t2, pw, geom, true);
    if (pw != 3) return false;
    if (t2 != 37) {
        // a tuned demodulation_atten: the instantiations compiled for a low-pass BOUND (kModeStrictPad2), 256 threads only
        // (geometry by the kernel's T2, the bound: phase_geom tells the profiles apart by their low-pass lengths)
        const uint32_t tb = fused_phase_pad_t2(t2, pw);
        if (tb == 0) return false;
        return phase_geom(256, 1, l, m, t1, tb, pw, geom) || phase_geom(256, 2, l, m, t1, tb, pw, geom) ||
               phase_geom(256, 4, l, m, t1, tb, pw, geom);
    }
    const char *wide = std::getenv("APTGPU_PHASE_WIDE");  // A/B switch (plan creation): the 512- / 1024-thread forms first
    if (wide && wide[0] == '1' && (phase_geom(512, 1, l, m, t1, t2, pw, geom) || phase_geom(1024, 1, l, m, t1, t2, pw, geom)))
        return true;
    return phase_geom(256, 1, l, m, t1, t2, pw, geom) || phase_geom(256, 2, l, m, t1, t2, pw, geom) ||
           phase_geom(256, 4, l, m, t1, t2, pw, geom) || phase_geom(512, 1, l, m, t1, t2, pw, geom) ||
           phase_geom(1024, 1, l, m, t1, t2, pw, geom);
}

uint32_t fused_phase_table_floats(const TableGeom &g)
{
    // (+ slack behind both tap tables: the streamed form prefetches one chunk past a row's end)
    const uint32_t base = g.tt_off ? g.tt_off : ((g.cp_off + g.nperm * g.nthr * (g.nq ? g.nq : 1u) + 3u) & ~3u);
    return base + (g.tt_off ? g.nperm * (g.nq ? g.nq : 1u) * g.tpp * g.nthr : 0u) + 64u * g.nthr / 4u + 80u;
}

// Thread -> branch-slot lists.  Thread t of a tile computes the outputs u, u + S, ... (S = step_r) for its slot u; the
// window of slot u starts c(u) = ceil((rb + u m) / l) entries into a region of the paired input tile, and every
// 8-byte LDS read of stage 1 is "entry c(u) + constant".  An 8-byte wave-read takes two cycles when the entries of each
// of its half-waves (lanes 0-31, 32-63) are distinct mod 32 — in any order, at any multiple of 32 — and half a cycle
// more per extra pass a half-wave needs (tools/ubench/lds_pat.hip, profiles/r05_ubench_lds_pat.txt: one pair 2.5, a pair
// in both halves 3.0); with slot = thread the starts advance m / l = 3.53 entries per lane at 44 100 Hz and up to
// four lanes of a half-wave share a residue (4.5 cycles measured).  The lists below deal the slots out over the
// half-waves so that as few of them as the residue class sizes allow need extra passes, for all of a thread's branches.  Which thread computes which slot changes
// nothing else: the tile loader indexes by thread, the results go to R in LDS by slot.
void fused_phase_table(const TableGeom &g, uint32_t t2, uint32_t pw, const float *coeff, uint32_t t1, float *table)
{
    const uint32_t l = g.l, tpp = g.tpp;
    for (uint32_t p = 0; p < l; ++p)
        for (uint32_t i = 0; i < tpp; ++i) {
            const uint64_t j = p + static_cast<uint64_t>(i) * l;
            table[static_cast<size_t>(p) * tpp + i] = j < g.jlim ? coeff[j] : 0.f;
        }
    (void)t1;
    const PhaseTile pt = phase_tile(g.nthr, t2, pw);
    const uint32_t nq = g.nq ? g.nq : 1u;
    const uint32_t S = g.sq;                           // slot stride = threads with work; thread slot u holds u + q S < step_r
    const uint32_t NH = 2 * ((S + 63) / 64);           // half-waves of the waves with work (the other waves skip stage 1's arithmetic)
    // does slot u's branch q exist?  (nq == 1: S = step_r, always; S = step_r / nq: always; S = threads: not the last ones)
    auto exists = [&](uint32_t u, uint32_t q) -> bool { return u + q * S < g.step_r; };
    auto branches_of = [&](uint32_t u) -> uint32_t {
        uint32_t n = 0;
        for (uint32_t q = 0; q < nq; ++q) n += exists(u, q) ? 1u : 0u;
        return n;
    };
    const char *ident = std::getenv("APTGPU_PHASE_IDENTITY");  // A/B switch: slot = thread, as until round 4
    std::vector<uint32_t> list(g.nthr), ident_list(g.nthr);
    for (uint32_t t = 0; t < g.nthr; ++t) ident_list[t] = t < S ? t : 0xFFFFFFFFu;
    for (uint32_t r = 0; r < g.nperm; ++r) {
        const int64_t k0 = static_cast<int64_t>(r) * pt.own_k - static_cast<int64_t>(pt.pre_k);
        int64_t rb = (k0 * static_cast<int64_t>(g.m)) % static_cast<int64_t>(l);
        if (rb < 0) rb += l;
        // window start of slot u's branch q: its entry in a region, and that mod the 32 entry positions of the LDS banks
        auto ent = [&](uint32_t u, uint32_t q) -> uint32_t {  // (a slot that does not exist reads entry 0: every such lane the same one)
            if (!exists(u, q)) return 0u;
            return static_cast<uint32_t>((static_cast<uint64_t>(rb) + static_cast<uint64_t>(u + q * S) * g.m + l - 1) / l);
        };
        auto res = [&](uint32_t u, uint32_t q) -> uint32_t { return ent(u, q) & 31u; };
        // distinct entries among `have` (slots of one half-wave) that share the bank position of slot u's branch q and
        // differ from it (lanes that read the SAME entry are served together)
        auto rivals = [&](const std::vector<uint32_t> &have, uint32_t u, uint32_t q) -> uint32_t {
            const uint32_t e = ent(u, q);
            uint32_t n = 0;
            for (size_t i = 0; i < have.size(); ++i) {
                const uint32_t o = ent(have[i], q);
                if (o == e) return 0;  // u rides along with that lane
                if (((o ^ e) & 31u) != 0) continue;
                bool seen = false;
                for (size_t j = 0; j < i && !seen; ++j) seen = ent(have[j], q) == o;
                n += seen ? 0 : 1;
            }
            return n;
        };
        // passes a half-wave's read of branch q takes beyond the first: (most distinct entries on one bank position) - 1
        auto extra_of = [&](const std::vector<uint32_t> &have, uint32_t q) -> uint32_t {
            uint32_t mx = 0;
            for (uint32_t p = 0; p < 32; ++p) {
                std::vector<uint32_t> es;
                for (uint32_t u : have)
                    if (res(u, q) == p && std::find(es.begin(), es.end(), ent(u, q)) == es.end()) es.push_back(ent(u, q));
                mx = std::max<uint32_t>(mx, static_cast<uint32_t>(es.size()));
            }
            return mx ? mx - 1 : 0;
        };
        auto cost = [&](const std::vector<uint32_t> &ls) -> uint32_t {
            uint32_t total = 0;
            for (uint32_t h = 0; h < g.nthr / 32; ++h) {
                std::vector<uint32_t> have;
                for (uint32_t e = 0; e < 32; ++e)
                    if (ls[32 * h + e] < S) have.push_back(ls[32 * h + e]);
                for (uint32_t q = 0; q < nq; ++q) total += extra_of(have, q);
            }
            return total;
        };
        // greedy: slots of the largest residue classes (of branch 0) first, each to the half-wave where it adds the
        // fewest extra passes over all its branches; among equals, to the one where it meets the fewest rivals, then
        // to the emptiest
        // Slots with more branches first, into the first half-waves: a wave runs a branch when ANY of its lanes has it, so
        // the slots that have the last branch are kept together (l = 832: slots 0-63 in wave 0; a slot of another class
        // only lands in a class's half-waves when its own are full — results do not depend on the lists).
        std::vector<uint32_t> order(S);
        std::vector<uint32_t> csize(32, 0);
        for (uint32_t u = 0; u < S; ++u) ++csize[res(u, 0)];
        for (uint32_t u = 0; u < S; ++u) order[u] = u;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const uint32_t na = branches_of(a), nb = branches_of(b);
            return na != nb ? na > nb : csize[res(a, 0)] > csize[res(b, 0)];
        });
        std::vector<uint32_t> first_hw(nq + 1, 0);  // first half-wave of the slots with n branches
        {
            uint32_t before = 0;
            for (uint32_t nbr = nq; nbr >= 1; --nbr) {
                first_hw[nbr] = before / 32u;
                for (uint32_t u = 0; u < S; ++u) before += branches_of(u) == nbr ? 1u : 0u;
            }
        }
        std::vector<std::vector<uint32_t>> half(NH);
        std::vector<uint32_t> hextra(static_cast<size_t>(NH) * nq, 0);
        for (uint32_t u : order) {
            uint32_t best = 0;
            uint64_t best_key = ~0ull;
            const uint32_t h_lo = first_hw[branches_of(u)];
            bool room = false;
            for (uint32_t h = h_lo; h < NH; ++h) room = room || half[h].size() < 32;
            for (uint32_t h = room ? h_lo : 0u; h < NH; ++h) {
                if (half[h].size() >= 32) continue;
                uint32_t add = 0, mult = 0;
                for (uint32_t q = 0; q < nq; ++q) {
                    const uint32_t k = rivals(half[h], u, q);  // u would be entry number k + 1 on its bank position
                    if (k > hextra[static_cast<size_t>(h) * nq + q]) add += k - hextra[static_cast<size_t>(h) * nq + q];
                    mult += k;
                }
                const uint64_t key = (static_cast<uint64_t>(add) << 40) | (static_cast<uint64_t>(mult) << 20) | half[h].size();
                if (key < best_key) {
                    best_key = key;
                    best = h;
                }
            }
            for (uint32_t q = 0; q < nq; ++q) {
                uint32_t &hx = hextra[static_cast<size_t>(best) * nq + q];
                hx = std::max(hx, rivals(half[best], u, q));
            }
            half[best].push_back(u);
        }
        for (uint32_t t = 0; t < g.nthr; ++t) list[t] = 0xFFFFFFFFu;
        for (uint32_t h = 0; h < NH; ++h)
            for (uint32_t e = 0; e < half[h].size(); ++e) list[32 * h + e] = half[h][e];
        const bool use_ident = (ident && ident[0] == '1') || cost(ident_list) <= cost(list);
        if (const char *dbg = std::getenv("APTGPU_PHASE_LIST_DEBUG"); dbg && dbg[0] == '1')
            std::fprintf(stderr, "aptgpu: phase lists l=%u m=%u nq=%u threads=%u rb=%lld: extra passes per tap %u (slot = thread: %u)%s\n", l, g.m,
                         nq, g.nthr, static_cast<long long>(rb), cost(list), cost(ident_list), use_ident ? " -> slot = thread" : "");
        const std::vector<uint32_t> &chosen = use_ident ? ident_list : list;
        std::memcpy(table + g.perm_off + static_cast<size_t>(r) * g.nthr, chosen.data(), g.nthr * sizeof(uint32_t));
        // window start and branch of every slot of every thread (the kernel reads them when g.exact)
        std::vector<uint32_t> cp(static_cast<size_t>(g.nthr) * nq, 0u);
        for (uint32_t t = 0; t < g.nthr; ++t)
            for (uint32_t q = 0; q < nq; ++q) {
                const bool have = chosen[t] < S && exists(chosen[t], q);  // (else: window start 0, branch 0 — read, never stored)
                const uint64_t v = static_cast<uint64_t>(rb) + static_cast<uint64_t>(have ? chosen[t] + q * S : 0u) * g.m;
                const uint64_t c = have ? (v + l - 1) / l : 0u, ph = have ? c * l - v : 0u;
                cp[static_cast<size_t>(t) * nq + q] = static_cast<uint32_t>(c) | (static_cast<uint32_t>(ph) << 16);
            }
        std::memcpy(table + g.cp_off + static_cast<size_t>(r) * g.nthr * nq, cp.data(), cp.size() * sizeof(uint32_t));
        if (g.tt_off) {
            // the taps in thread order: [r][q][e][t][4]
            for (uint32_t q = 0; q < nq; ++q)
                for (uint32_t e = 0; e < tpp / 4; ++e)
                    for (uint32_t t = 0; t < g.nthr; ++t) {
                        const uint32_t ph = cp[static_cast<size_t>(t) * nq + q] >> 16;
                        float *dst = table + g.tt_off + (((static_cast<size_t>(r) * nq + q) * (tpp / 4) + e) * g.nthr + t) * 4;
                        const bool have = chosen[t] < S && exists(chosen[t], q);
                        for (uint32_t k = 0; k < 4; ++k) dst[k] = have ? table[static_cast<size_t>(ph) * tpp + 4 * e + k] : 0.f;
                    }
        }
    }
}

bool fused_phase_front_end(hipStream_t s, const TableGeom &geom, uint32_t t2, uint32_t pw, int mode, bool pcm16,
                           const CallArgs &call, const FusedParams *d_prm, uint64_t max_w)
{
    if (call.count == 0 || call.count > static_cast<uint32_t>(kMaxCall)) return false;
    if (pcm16)
        for (uint32_t i = 0; i < call.count; ++i)
            if (reinterpret_cast<uintptr_t>(call.rec[i].x) & 1u) return false;
    // (one branch per thread: the kernel takes the paired tile through LDS in two halves — apt_kernels_fused_launch.hpp)
    // (a tuned low-pass at the standard profile runs the strict kModeStrictPad2 instantiation whatever the mode)
    const bool fast_kernel = mode == kModeFast && !(pw == 3 && t2 != 37);
    const bool halves = phase_halves(static_cast<int>(geom.nq ? geom.nq : 1u), geom.stream != 0, static_cast<int>(geom.nthr),
                                     static_cast<int>(t2), fast_kernel);
    const FusedLaunch a{s, &call, d_prm, max_w, static_cast<size_t>(halves ? geom.xt / 2 : geom.xt)};
    const bool wide = geom.nthr == 512, huge = geom.nthr == 1024;
    if (t2 == 61 && pw == 5) {  // the slow profile's work-rate stages, streamed taps (strict instantiations only: they serve fast mode too)
        if (wide || huge || !geom.stream) return false;
        if (geom.nq == 1) pcm16 ? fused_launch_phase_slowp_i16(a) : fused_launch_phase_slowp_f32(a);
        else if (geom.nq == 2) pcm16 ? fused_launch_phase2_slowp_i16(a) : fused_launch_phase2_slowp_f32(a);
        else if (geom.nq == 4) pcm16 ? fused_launch_phase4_slowp_i16(a) : fused_launch_phase4_slowp_f32(a);
        else return false;
        return true;
    }
    if (t2 == 43 && pw == 4 && geom.nq > 1) {  // the fast profile's, four / eight branches per thread (strict instantiations only)
        if (wide || huge) return false;
        if (geom.nq == 4) pcm16 ? fused_launch_phase4_fastp_i16(a) : fused_launch_phase4_fastp_f32(a);
        else if (geom.nq == 8) pcm16 ? fused_launch_phase8_fastp_i16(a) : fused_launch_phase8_fastp_f32(a);
        else if (geom.nq == 16) pcm16 ? fused_launch_phase16_fastp_i16(a) : fused_launch_phase16_fastp_f32(a);
        else return false;
        return true;
    }
    if (t2 == 43 && pw == 4) {  // the fast profile's work-rate stages
        if (wide || huge) return false;
        if (mode == kModeFast) pcm16 ? fused_launch_phase_fastp_fast_i16(a) : fused_launch_phase_fastp_fast_f32(a);
        else if (mode == kModeStrict) pcm16 ? fused_launch_phase_fastp_i16(a) : fused_launch_phase_fastp_f32(a);
        else return false;
        return true;
    }
    if (pw == 3 && t2 != 37) {  // a tuned low-pass: kModeStrictPad2 (strict arithmetic: it serves APTGPU_MODE_FAST too)
        if (wide || huge || phase_pad_t2_bound(t2, pw) == 0) return false;
        if (geom.nq == 1 || geom.nq == 0) pcm16 ? fused_launch_phase_std_pad2_i16(a) : fused_launch_phase_std_pad2_f32(a);
        else if (geom.nq == 2) pcm16 ? fused_launch_phase2_std_pad2_i16(a) : fused_launch_phase2_std_pad2_f32(a);
        else if (geom.nq == 4) pcm16 ? fused_launch_phase4_std_pad2_i16(a) : fused_launch_phase4_std_pad2_f32(a);
        else return false;
        return true;
    }
    if (geom.nq == 2 || geom.nq == 4) {
        if (wide || huge) return false;
        const bool four = geom.nq == 4;
        if (mode == kModeFast) {
            if (four) pcm16 ? fused_launch_phase4_std_fast_i16(a) : fused_launch_phase4_std_fast_f32(a);
            else pcm16 ? fused_launch_phase2_std_fast_i16(a) : fused_launch_phase2_std_fast_f32(a);
        } else if (mode == kModeStrict) {
            if (four) pcm16 ? fused_launch_phase4_std_i16(a) : fused_launch_phase4_std_f32(a);
            else pcm16 ? fused_launch_phase2_std_i16(a) : fused_launch_phase2_std_f32(a);
        } else {
            return false;
        }
        return true;
    }
    if (mode == kModeFast) {
        if (huge) pcm16 ? fused_launch_phase1024_std_fast_i16(a) : fused_launch_phase1024_std_fast_f32(a);
        else if (wide) pcm16 ? fused_launch_phase512_std_fast_i16(a) : fused_launch_phase512_std_fast_f32(a);
        else pcm16 ? fused_launch_phase_std_fast_i16(a) : fused_launch_phase_std_fast_f32(a);
    } else if (mode == kModeStrict) {
        if (huge) pcm16 ? fused_launch_phase1024_std_i16(a) : fused_launch_phase1024_std_f32(a);
        else if (wide) pcm16 ? fused_launch_phase512_std_i16(a) : fused_launch_phase512_std_f32(a);
        else pcm16 ? fused_launch_phase_std_i16(a) : fused_launch_phase_std_f32(a);
    }
    else
        return false;
    return true;
}

}  // namespace apt::gpu
