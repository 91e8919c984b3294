// apt_kernels_fused_any.hip — fused front end for ANY (input rate, profile): resample ->
// envelope -> low-pass -> sync correlation (+ group maxima) in one launch, all run-time
// parameters.  Used when no compile-time specialisation of k_fused (apt_kernels_fused.hip)
// matches: 11 025 Hz (L = 832) and 44 100 Hz (L = 208) recordings, the fast / slow profiles,
// odd rates.  Same outputs, same bit-exact arithmetic (separately rounded products and sums in
// the reference's order); about 2x the instructions of the specialised kernel because nothing
// is packed or unrolled against compile-time tap positions.
//
// One workgroup = one tile of KT = NTHR*KPT work samples:
//
//   tile index:   0 ......... PRE ........................ PRE+OWN ....... KT
//                 | history    | owned: F, C, GM go to HBM  | look-ahead   |
//                 | (low-pass) |                            | (sync frame) |
//
//   stage 0  polyphase table (phase-major, row stride odd -> conflict-free) and the input
//            tile -> LDS, coalesced
//   stage 1  R[k] = sum_i coeff[p + i*l] * x[x0 + i]      (dsp.rs:252-263), one output per
//            thread and step, taps and samples from LDS
//   stage 2  D[k] = envelope(R[k-1], R[k])                (dsp.rs:369-377)
//   stage 3  F[k] = sum_{j<T2} D[k-j] * h[j]              (dsp.rs:396-404), KPT consecutive
//            outputs per thread: a sliding register window cuts LDS reads per tap to 1/KPT
//   stage 4  bounds of max C over groups of 52 positions, C[k] = sum_{j<G} +-F[k+j] (decode.rs:225-233): the
//            correlation from pulse sums (apt_sync_corr.hpp) widened by the rounding bound slack * sum|F|; the
//            picker evaluates the exact chain where it matters (apt_kernels_sync.hip)
//   stage 5  owned F -> HBM (coalesced), GM[g] = [lo, hi]
//
// LDS layout of A and B: logical index i lives at i + i/KPT (one pad word after every KPT), so a
// thread's block of KPT consecutive outputs starts at an ODD multiple-of-words stride from its
// neighbour's: the register-blocked stages read conflict-free (unpadded, a stride of 8 floats
// put 64 lanes on 4 banks and made stage 4 three times slower than everything else together).
//
// Why zero-filling is exact: a running sum that starts at +0.0 can never be -0.0 (x + y is
// -0.0 only if both are), so adding the +-0.0 product of a zero-filled sample (inputs at or
// past n, which the reference skips, dsp.rs:257; D[k <= 0], which the `i > j` guard of
// dsp.rs:399 excludes and which is 0.0 anyway) leaves every bit unchanged.
#include "apt_kernels_fused_any_launch.hpp"

#include <type_traits>

#pragma clang fp contract(off)

namespace apt::gpu {

namespace {

constexpr int kGS = 52;  // correlation positions per group (apt_kernels_sync.hip)

constexpr size_t kLdsLimit = 160 * 1024;

struct Candidate {
    int nthr, kpt;
};
// most resident waves per CU first (LDS permitting), then the larger tile
// (round 6: {256, 4} — a tile of 1024 work samples — for input rates beyond seventeen times the work rate's 1 / l share:
// 250 kHz SDR recordings, whose 2048-sample tile needs more input than the LDS holds, ran the unfused kernels)
constexpr Candidate kCandidates[] = {{256, 8}, {1024, 8}, {1024, 4}, {256, 4}};

bool make_geom(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, int nthr, int kpt, bool table_in_global,
               AnyGeom *out, size_t *lds_bytes)
{
    AnyGeom g{};
    g.l = l;
    g.m = m;
    g.jlim = 2 * ((t1 - 1) / 2) + 1;
    const uint32_t per_phase = (g.jlim + l - 1) / l;
    g.tpp = per_phase | 1u;  // odd stride: rows of consecutive phases start in different banks
    g.t2 = t2;
    g.g = 38 * pw;
    g.pulse = 2 * pw;
    g.kt = static_cast<uint32_t>(nthr * kpt);
    if (38 * pw > 256) return false;  // sign bitmap
    // 32-bit in-tile index math: (tile outputs + look-ahead) * m + l must stay below 2^31 (per launch shape since round 6:
    // the old blanket test for the largest tile sent every rate above 170 kHz that is coprime to the work rate to the
    // unfused kernels)
    if ((static_cast<uint64_t>(g.kt) + 4096u) * m + l > 0x7fffffffull) return false;
    g.pre = (t2 + 2 * static_cast<uint32_t>(kpt) + static_cast<uint32_t>(kpt) - 1) / static_cast<uint32_t>(kpt) *
            static_cast<uint32_t>(kpt);  // multiple of KPT (and of 4)
    if (g.kt < g.pre + g.g + kGS) return false;
    g.own = (g.kt - g.pre - (g.g - 1)) / kGS * kGS;
    if (g.own == 0) return false;
    g.xt = (static_cast<uint32_t>((static_cast<uint64_t>(g.kt) * m + l - 1) / l) + per_phase + 8 + 3) & ~3u;
    if (g.xt < static_cast<uint32_t>(nthr)) g.xt = static_cast<uint32_t>(nthr);  // (stage 4 keeps one float per thread there)
    const uint32_t slack = 64;
    // (a table that does not fit LDS beside the tile — 44 100 Hz at the slow profile: 29 k taps, 116 KB — stays in
    // HBM / L2 and stage 1 reads its rows from there: every thread 856 bytes per output, but fused: R, D and C still
    // never leave the CU)
    g.table_in_global = table_in_global ? 1u : 0u;
    g.off_x = table_in_global ? 0u : ((l * g.tpp + 3u) & ~3u);
    const uint32_t ab_len = ((g.kt + slack) + (g.kt + slack) / static_cast<uint32_t>(kpt) + 8u) & ~3u;  // padded
    g.off_a = g.off_x + g.xt;
    g.off_b = g.off_a + ab_len;
    g.step_q = static_cast<uint32_t>((static_cast<uint64_t>(nthr) * m) / l);
    g.step_r = static_cast<uint32_t>((static_cast<uint64_t>(nthr) * m) % l);
    g.jl_a = g.jlim / l;
    g.jl_b = g.jlim % l;
    for (uint32_t j = 0; j < g.g; ++j) {
        const bool plus = j >= g.pulse && j < 15 * g.pulse && (((j - g.pulse) / g.pulse) & 1) == 1;
        if (plus) g.sign[j >> 6] |= 1ull << (j & 63);
    }
    const size_t floats = static_cast<size_t>(g.off_b) + ab_len;
    *lds_bytes = floats * sizeof(float);
    *out = g;
    return *lds_bytes <= kLdsLimit;
}

// picks the launch shape: maximise resident waves per CU, then tile size
bool choose_with(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, bool in_global, Candidate *best, AnyGeom *geom,
                 size_t *lds)
{
    int best_score = 0;
    uint32_t best_kt = 0;
    bool found = false;
    for (const Candidate &c : kCandidates) {
        AnyGeom g;
        size_t bytes;
        if (!make_geom(l, m, t1, t2, pw, c.nthr, c.kpt, in_global, &g, &bytes)) continue;
        int wgs = static_cast<int>(kLdsLimit / bytes);
        int waves = wgs * (c.nthr / 64);
        if (waves > 32) waves = 32;  // 8 per SIMD is plenty
        // tile efficiency matters too: weight by the owned fraction
        const int score = waves * 1000 + static_cast<int>(1000.0 * g.own / g.kt);
        if (!found || score > best_score || (score == best_score && g.kt > best_kt)) {
            best_score = score;
            best_kt = g.kt;
            *best = c;
            *geom = g;
            *lds = bytes;
            found = true;
        }
    }
    return found;
}

// the table in LDS where a launch shape exists for that; else in HBM / L2
bool choose(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, Candidate *best, AnyGeom *geom, size_t *lds)
{
    return choose_with(l, m, t1, t2, pw, false, best, geom, lds) || choose_with(l, m, t1, t2, pw, true, best, geom, lds);
}

}  // namespace

bool fused_any_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw)
{
    if (l == 0 || m == 0 || t1 == 0 || t2 == 0 || pw == 0) return false;
    Candidate c;
    AnyGeom g;
    size_t lds;
    return choose(l, m, t1, t2, pw, &c, &g, &lds);
}

uint32_t fused_any_table_floats(uint32_t l, uint32_t t1)
{
    const uint32_t jlim = 2 * ((t1 - 1) / 2) + 1;
    return l * (((jlim + l - 1) / l) | 1u);
}

// host: phase-major table [l][tpp], row p = coeff[p], coeff[p + l], ... (zero padded)
void fused_any_table(uint32_t l, const float *coeff, uint32_t t1, float *table)
{
    const uint32_t jlim = 2 * ((t1 - 1) / 2) + 1;
    const uint32_t tpp = ((jlim + l - 1) / l) | 1u;
    for (uint32_t p = 0; p < l; ++p)
        for (uint32_t i = 0; i < tpp; ++i) {
            const uint64_t j = p + static_cast<uint64_t>(i) * l;
            table[static_cast<size_t>(p) * tpp + i] = j < jlim ? coeff[j] : 0.f;
        }
}

bool fused_any_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, bool pcm16,
                         const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, const float *table,
                         const float *h2, const float *h2p, float cosphi2, float sinphi, float inv_sinphi, bool want_gm,
                         float gm_slack_scale)
{
    Candidate c;
    AnyGeom g;
    size_t lds;
    if (call.count == 0) return true;
    if (!choose(l, m, t1, t2, pw, &c, &g, &lds)) return false;
    g.gm_slack_scale = gm_slack_scale;
    if (pcm16)
        for (uint32_t i = 0; i < call.count; ++i)
            if (reinterpret_cast<uintptr_t>(call.rec[i].x) & 1u) return false;
    const int prof = (t2 == 37 && pw == 3) ? 1 : (t2 == 43 && pw == 4) ? 2 : (t2 == 61 && pw == 5) ? 3 : 0;
    if (c.nthr == 256 && c.kpt == 8)
        fused_any_launch_256x8(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
    else if (c.nthr == 1024 && c.kpt == 8)
        fused_any_launch_1024x8(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
    else if (c.nthr == 1024 && c.kpt == 4)
        fused_any_launch_1024x4(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
    else if (c.nthr == 256 && c.kpt == 4)
        fused_any_launch_256x4(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
    else
        return false;
    return true;
}

}  // namespace apt::gpu
