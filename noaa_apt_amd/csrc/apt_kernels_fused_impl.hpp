// apt_kernels_fused_impl.hpp — the template of the specialised fused front end (k_fused) and its
// launch wrapper.  Included by the apt_kernels_fused_*.hip translation units, one instantiation
// each, so that the eight instantiations compile in parallel (one TU took seven minutes).
// See apt_kernels_fused.hip for the description of the kernel.
#pragma once

#include "apt_kernels_fused_launch.hpp"
#include "apt_envelope.hpp"
#include "apt_sync_corr.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

// clobber lists of the pinned tap tuples (stage 1 of k_fused)
#define APT_S1(n) "s" #n
#define APT_S8_(a, b, c, d, e, f, g, h) APT_S1(a), APT_S1(b), APT_S1(c), APT_S1(d), APT_S1(e), APT_S1(f), APT_S1(g), APT_S1(h)
#define APT_S8(n) APT_S8_X(n)
#define APT_S16(n) APT_S16_X(n)
// (the preprocessor cannot add: the tuples in use are spelled out)
#define APT_S8_X(n) APT_S8_##n
#define APT_S16_X(n) APT_S16_##n
#define APT_S8_52 APT_S8_(52, 53, 54, 55, 56, 57, 58, 59)
#define APT_S16_36 APT_S8_(36, 37, 38, 39, 40, 41, 42, 43), APT_S8_(44, 45, 46, 47, 48, 49, 50, 51)
// (16 + 8 + 4 dwords per buffer)
#define APT_S16_64 APT_S8_(64, 65, 66, 67, 68, 69, 70, 71), APT_S8_(72, 73, 74, 75, 76, 77, 78, 79)
#define APT_S8_80 APT_S8_(80, 81, 82, 83, 84, 85, 86, 87)
#define APT_S4_60 "s60", "s61", "s62", "s63"
#define APT_S4_88 "s88", "s89", "s90", "s91"

namespace apt::gpu {

namespace {


// Listing marks (tools/isa.sh -DAPT_FUSED_MARKS=1, tools/isa_budget.py --marks): comments in the assembly around the
// code an INTERIOR tile of the specialised kernels executes, so that the per-stage instruction budget can be read off
// a listing without following its branches by hand.  Never defined in a product build (an asm statement is a
// scheduling fence).
#ifdef APT_FUSED_MARKS
#define APT_MARK(name) asm volatile("; APTMARK " name)
#else
#define APT_MARK(name) ((void)0)
#endif

// Probe builds (apt_kernels_fused_probe*.hip, timing experiments only): the kernel stops after stage
// APT_FUSED_STOP — 1 tile in LDS, 2 resampler, 3 envelope, 4 low-pass, 5 F stored; 0 = the real kernel.
#ifndef APT_FUSED_STOP
#define APT_FUSED_STOP 0
#endif
#ifndef APT_FUSED_PERSIST
#define APT_FUSED_PERSIST 0
#endif
// waves per SIMD the kModeMfma kernels are compiled for (= workgroups per CU; the LDS would hold six, the registers do
// not: 119 VGPRs with sub-tile 1's loads in flight under sub-tile 0's products.  tools/probes/mfma_variants.sh builds the
// other values: five and six workgroups spill and measured 12 % / 22 % slower)
#ifndef APT_MFMA_WAVES
#define APT_MFMA_WAVES 4
#endif
// waves per SIMD the kModeStrictPad2 kernels are compiled for (the 45-tap low-pass's window does not fit the 80 registers
// of six: eleven of them spilled inside stage 3)
#ifndef APT_PAD2_WAVES
#define APT_PAD2_WAVES 5
#endif
// Halo threads of a tile (FusedGeom::kPreThreads / kPostThreads).  The low-pass and the envelope need T2 + 1 = 38 work
// samples of history, the correlation G - 1 = 113 of look-ahead: 3 + 9 threads of 13 samples in the specialised
// kernels (round 2: 4 + 12 — a whole group of four either side; 244 instead of 240 of 256 threads own samples, 116
// instead of 112 of the 96 kHz kernels' 128).  The table-driven and phase-resident forms keep 4 + 12: their
// stage 1 is laid out for a tile that starts on a group boundary.
// (Round 4: computed from T2 and PW — FusedGeom — so that the work-rate stages also serve the fast and slow profiles:
// 43 / 61-tap low-pass, pixel width 4 / 5.)
constexpr float kNegInfF = -__builtin_huge_valf();

// max(a, b, c) of values that are results of floating-point additions (never signalling NaNs), a quiet NaN
// counting as absent — fmaxf's semantics on such inputs — as ONE v_max3_f32.  The compiler cannot see where the
// values come from once they have been through LDS or a select and canonicalises every operand of an fmaxf first
// (v_max_f32 x, x, x): 25 instructions for the 13 + 12 values of stage 4's maxima.
__device__ __forceinline__ float max3_of_sums(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int L, int M>
__host__ __device__ constexpr int branch_first(int b)  // c_b = ceil(b*M / L)
{
    return (b * M + L - 1) / L;
}
template <int L, int M>
__host__ __device__ constexpr int branch_phase(int b)  // p_b = c_b*L - b*M
{
    return branch_first<L, M>(b) * L - b * M;
}

// XB: bytes per input sample in the LDS tile (4: f32 Signal; 2: PCM16 kept as int16 — exact, and
// half the tile)
// M == 0 selects the table-driven stage 1 (TABLE mode, see k_fused): the resampling factors, tap count and
// input tile are then run-time quantities (FusedParams::tab) and only the work-rate geometry is static.
// VAR (fused_geom_var): 0 the f32 stage 1, 1 the fp16-tap stage 1 (kModeF16Taps), 2 stage 1 on the matrix cores (kModeMfma:
// the input tile is two bf16 planes of K-padded windows; T1 is then the LARGEST tap count the instantiation serves)
template <int L, int M, int T1, int T2, int PW, int NTHR, int XB = 4, int VAR = 0>
struct FusedGeom {
    static constexpr bool F16TAPS = VAR == 1, MFMA = VAR == 2, PAD = VAR == 3;
    static constexpr bool TABLE = M <= 0;   // run-time resampling factors (table-driven or phase-resident stage 1)
    static constexpr bool PHASE = M < 0;    // ... with the taps of a thread's -M polyphase branches in registers
    static constexpr int kFusedThreads = NTHR;
    // threads of L samples in front of / behind the owned ones: T2 + 1 samples of history, 38 PW - 1 of look-ahead;
    // whole groups of four threads where stage 1 is laid out for a tile that starts on a group boundary (TABLE, PHASE);
    // the owned threads are whole groups of four in every form (standard profile: 3 + 9 / 4 + 12 as before)
    static constexpr int kPreNeed = (T2 + 1 + L - 1) / L, kPostNeed = (38 * PW - 1 + L - 1) / L;
    static constexpr int kPreThreads = TABLE ? (kPreNeed + 3) / 4 * 4 : kPreNeed;
    static constexpr int kOwnThreads = (NTHR - kPreThreads - (TABLE ? (kPostNeed + 3) / 4 * 4 : kPostNeed)) / 4 * 4;
    static constexpr int kPostThreads = NTHR - kPreThreads - kOwnThreads;
    // SPLIT stage 1 (apt_kernels_fused_launch.hpp): two sub-tiles of NS = NTHR / 2 windows through the same LDS
    static constexpr bool SPLIT = !TABLE && !F16TAPS && NTHR == 256;
    static constexpr int NS = SPLIT ? NTHR / 2 : NTHR;                // windows per input tile in LDS
    static constexpr int TP = (T1 + L - 1) / L;                       // taps per branch (max)
    static constexpr int CLAST = TABLE ? 0 : branch_first<L, (TABLE ? 1 : M)>(L - 1);  // last branch's first sample
    static constexpr int WIN = CLAST + TP;                            // input window per thread
    static constexpr int TILE_K = kFusedThreads * L;                  // work samples per tile
    static constexpr int OWN_K = kOwnThreads * L;                     // owned work samples
    static constexpr int PRE_K = kPreThreads * L;
    // The tile's first input sample is (first thread's group index) * M: a multiple of 4 only when kPreThreads * M is
    // (16-byte loads).  XSHIFT more samples are loaded in front of it, and every window read is XSHIFT further on.
    static constexpr int XSHIFT = TABLE ? 0 : (4 - (kPreThreads * M) % 4) % 4;
    static_assert(TABLE || (kOwnThreads * M) % 4 == 0, "every tile starts at the same offset from a 16-byte boundary");
    static_assert(XSHIFT % 2 == 0, "window reads stay 8-byte aligned (and PCM16 pairs whole)");
    static constexpr int KPAD = (WIN + 31) / 32 * 32;                 // MFMA: the window padded to whole K steps
    static constexpr int XT = TABLE ? 4 : (NS - 1) * M + (MFMA ? KPAD : WIN + 2) + XSHIFT;  // input floats per (sub-)tile
    static constexpr int XT_PAD = MFMA ? (XT + 7) & ~7 : (XT + 3) & ~3;
    static constexpr int G = 38 * PW;                                 // sync template length
    static constexpr int FWIN = L + G - 1;                            // F window per thread
    // one LDS region: the x tile, then R / F at [0, TILE_K) (+ the few words the last thread's pulse-sum window reads past
    // it: never used), D and later the pulse sums at D_OFF.  Until round 4 the R / F region kept G words of slack from
    // the days when stage 4 read 13 + G - 1 consecutive F values per thread, the pulse sums 36 PW (they need QTAIL),
    // and the per-thread |F| sums had 256 words of their own: 28.5 KB, five workgroups per CU.  Trimmed — the |F| sums
    // now land on the dead F region — the work-rate stages need 26.1 KB: SIX workgroups per CU for kernels of <= 80 VGPRs.
    static constexpr int D_OFF = (TILE_K + 2 * PW + 2 + 3) & ~3;
    // floats of LDS under the x tile (MFMA: two bf16 planes of XT_PAD samples, then the tile's non-finite flag)
    // (PAD: one more word behind the tile — the "results not all finite" flag of kModeStrictPad)
    static constexpr int XT_LDS = MFMA ? XT_PAD + 8 : (XB == 2 ? (XT_PAD / 2 + 4) : XT_PAD) + (PAD ? 4 : 0);
    // the pulse sums a thread of stage 4 reads: positions p0 + 2 PW n, n <= 31, p0 <= PRE_K + (blocks - 1) * 2 PW L + 2 PW - 1
    static constexpr int NBLK4 = (OWN_K + 2 * PW * L - 1) / (2 * PW * L);
    static constexpr int QMAX = PRE_K + (NBLK4 - 1) * 2 * PW * L + 2 * PW - 1 + 2 * PW * 31;
    static constexpr int QTAIL = QMAX + 1 > TILE_K ? ((QMAX + 1 - TILE_K + 3) & ~3) : 0;
    static constexpr int Q_FLOATS = TILE_K + QTAIL;
    // Stage 4 comes in two forms: COMPACT4 (pixel width 3: the block / group bookkeeping written out for 6-sample pulses,
    // partial maxima through LDS) and a general one (any pixel width: the correlation values themselves go through the
    // dead F region).  Per-thread |F| sums of the strict modes' bounds: COMPACT4 — inside the R / F region, behind the
    // partial maxima; general — a region of their own behind the pulse sums.
    static constexpr bool COMPACT4 = PW == 3 && L == 13;
    static constexpr int AB_OFF = COMPACT4 ? (((OWN_K / (4 * L) + 2) * 12 + 3) & ~3) : D_OFF + Q_FLOATS;
    static_assert(!COMPACT4 || AB_OFF + NTHR + 8 <= D_OFF, "|F| sums and partial maxima both fit the dead F region");
    static constexpr int W_LDS_FLOATS = D_OFF + Q_FLOATS + (COMPACT4 ? 0 : NTHR);  // what the work-rate stages need
    // (PAD: the word of kModeStrictPad2's "F not all finite" flag behind BOTH regions — it is cleared while sub-tile 0 lies
    // in LDS and read after stage 3)
    static constexpr int LP_FLAG_OFF = XT_LDS > W_LDS_FLOATS ? XT_LDS : W_LDS_FLOATS;
    static constexpr int LDS_FLOATS = LP_FLAG_OFF + (PAD ? 4 : 0);
    // workgroups a CU's 160 KB of LDS hold (the specialised kernels' occupancy; at most 8 waves per SIMD)
    static constexpr int WGS_PER_CU_LDS = TABLE ? 2 : (160 * 1024) / (LDS_FLOATS * 4);
    static constexpr int WGS_PER_CU = WGS_PER_CU_LDS * NTHR > 8 * 256 ? (8 * 256) / NTHR : WGS_PER_CU_LDS;
    static constexpr int GS = 4 * L;                                  // correlation group size
    static constexpr int NP = L / 2;                                  // accumulator pairs (+1 single if L odd)
    // (stage-1 tap table: chunk-major per thread half, fused_branch_taps / the layout functions in apt_kernels_fused_launch.hpp)
    static constexpr int DW = L + T2 - 1;                             // envelope window per thread
    static_assert(PRE_K >= T2 + 1, "pre-halo too small for the low-pass");
    static_assert(kOwnThreads > 0, "no owned threads");
    static_assert((kFusedThreads - kPreThreads - kOwnThreads) * L >= G - 1, "post-halo too small");
    static_assert(OWN_K % 4 == 0, "owned range must be float4-aligned");
    static_assert(kOwnThreads % 4 == 0, "whole correlation groups");
};

// does polyphase branch b use window sample q?  (tap index i = q - c_b, p_b + i*L < T1)
template <int L, int M, int T1>
__host__ __device__ constexpr bool branch_uses(int b, int q)
{
    const int i = q - branch_first<L, M>(b);
    return i >= 0 && branch_phase<L, M>(b) + i * L < T1;
}

// first branch that uses window sample 2c or 2c+1 (fp16 stage 1 works on sample pairs); L if none
template <int L, int M, int T1>
__host__ __device__ constexpr int first_branch_of_pair(int c)
{
    for (int b = 0; b < L; ++b)
        if (branch_uses<L, M, T1>(b, 2 * c) || branch_uses<L, M, T1>(b, 2 * c + 1)) return b;
    return L;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
// The tap tables and the slot table are never written while a decode runs: pointers fetched from the
// parameter block are cast to the constant address space, which is what makes the (wave-uniform)
// reads of them scalar loads — a generic pointer loaded from memory would turn every tap read
// into a per-lane flat load.
#define APT_CONST_AS __attribute__((address_space(4)))
typedef const f2 APT_CONST_AS *cf2_ptr;
typedef const float APT_CONST_AS *cfloat_ptr;

// sign of the sync template at index j (decode.rs:188-198): + inside the seven high pulses
template <int PW>
__host__ __device__ constexpr bool sync_plus(int j)
{
    const int pulse = 2 * PW;
    if (j < pulse || j >= pulse + 14 * pulse) return false;
    return (((j - pulse) / pulse) & 1) == 1;
}

// XT = float: the f32 Signal; XT = int16_t: mono PCM16 straight from the WAV data chunk
// (`*x as f32`, wav.rs:37), which halves the compulsory input bytes.
// MODE:
//   kModeStrict  every product and sum rounded separately in the reference's order: bit-identical
//                to the scalar Rust loops.
//   kModeF16Taps (APTGPU_MODE_FP16_TAPS, BASELINE config 5) stage 1 only runs on fp16 taps
//                (power-of-two prescaled) and fp16-rounded samples through v_dot2_f32_f16 with f32
//                accumulation; `hs` then holds [ceil(WIN/2)][16] half2 tap pairs.  Tolerance-based.
//   kModeFast    (APTGPU_MODE_FAST) f32 throughout, same taps, same tap order, but every
//                multiply-add is one fused v_pk_fma_f32 (stages 1 and 3), the envelope uses the
//                native v_sqrt_f32 and a multiplication by 1/sin(phi), and the +-1 correlation is
//                evaluated from pulse sums (apt_sync_corr.hpp): ~1/3 of the strict VALU
//                instructions.  Tolerance-based; deterministic.
// One launch covers the `count` recordings of a call; the per-recording arguments travel by value
// in the kernel-argument segment, the workspace pointers come from the plan's slot table.
// TABLE mode (M == 0; 11 025 Hz and other rates whose interpolation factor is far too large for one
// accumulator per polyphase branch): stage 1 runs from a phase-major tap table in LDS with run-time L / M /
// tap count — one output per thread and step, as in k_fused_any — and hands R to the SAME work-rate
// stages (envelope, packed low-pass, correlation) as the specialised kernels; the template's L is then
// just "work samples per thread" (13: four threads = one group of 52 positions).
// PHASE mode (M == -NQ; every rate a sound card records at — 44 100 Hz: l = 208, NQ = 1; 22 050 Hz: l = 416, NQ = 2;
// 11 025 Hz: l = 832, NQ = 4 — too many branches for one accumulator each): a thread holds NQ slots u + q S' of the S
// outputs after which the branches repeat (S = l * floor(NTHR / l) for NQ = 1, else l; S' = S / NQ) and computes the
// 16 / NQ outputs (u + q S') + a S of each.  They all use the SAME polyphase branch, so that branch's taps are fetched
// once per tile into REGISTERS (16-byte loads from the L2-resident table, in thread order where the tile phases
// repeat) and every tap then costs one LDS read of a sample pair per output pair and half a v_pk_mul_f32 + half a
// v_pk_add_f32 (or half a v_pk_fma_f32) per output, as in the specialised kernels.  Each output still accumulates its
// taps in ascending order: bit-identical in strict mode.  Which thread takes which slot is the host's choice
// (fused_phase_table: the lists that keep the 8-byte LDS reads of a half-wave on distinct bank positions).  T1 == 1 marks
// the STREAMED form (taps fetched sixteen at a time: the slow profile's 197 taps per branch do not fit the registers).
// One workgroup per tile: blockIdx.y picks the recording, whose tiles are blockIdx.x < ceil(w / OWN_K).
// The tile's input goes through registers (all loads issued before the first LDS write).  The specialised kernels
// (M > 0, f32 taps) run stage 1 in the SPLIT form — two sub-tiles of 128 windows through the same LDS, each thread
// half a window's branches (apt_kernels_fused_launch.hpp) — so that the work-rate stages' 26.1 KB, not the input
// tile, set the LDS footprint at 48 kHz: six workgroups per CU (<= 80 VGPRs; five until round 4), three at 96 kHz.  (A persistent form
// that walked the tiles with a fixed grid and kept the NEXT tile's input in registers while the stages ran was measured
// in rounds 2 and 3 and dropped: slower in every mode)
template <int L, int M, int T1, int T2, int PW, int NTHR, typename XT, int MODE>
__global__ void __launch_bounds__(NTHR, M < 0 ? ((NTHR > 512 ? 1 : NTHR > 256 ? 2 : (T1 == 1 ? (M == -1 ? 3 : 4) : phase_halves(-M, false, NTHR, T2, MODE == kModeFast) ? (MODE == kModeFast ? 4 : 5) : T2 == 43 ? 4 : M == -4 ? 5 : M == -2 ? 5 : 3)) * NTHR + 255) / 256  /* phase mode: three 256-thread (<= 170 VGPRs; the fast profile's and the streamed forms with several branches: four, <= 128; two or four branches per thread at the standard profile: five, <= 96), two 512-thread or one 1024-thread (<= 128) workgroups per CU */
                                     : M == 0 ? (2 * NTHR + 255) / 256  /* table mode: two workgroups per CU */
                                               /* specialised: as many workgroups as the CU's 160 KB of LDS hold (48 kHz SPLIT: 5, 96 kHz: 3) */
                                               : MODE == kModeMfma ? APT_MFMA_WAVES
                                               : MODE == kModeStrictPad2 ? (FusedGeom<L, M, T1, T2, PW, NTHR, static_cast<int>(sizeof(XT)), 3>::WGS_PER_CU < APT_PAD2_WAVES
                                                                                ? FusedGeom<L, M, T1, T2, PW, NTHR, static_cast<int>(sizeof(XT)), 3>::WGS_PER_CU : APT_PAD2_WAVES)
                                               : (FusedGeom<L, M, T1, T2, PW, NTHR, static_cast<int>(sizeof(XT)), fused_geom_var(MODE)>::WGS_PER_CU * NTHR + 255) / 256)
k_fused(const CallArgs call_by_value, const FusedParams *__restrict__ prm)
{
    // The call's arguments are read where they lie, in the kernel-argument segment (constant address
    // space, scalar loads with a run-time index).  (`call_by_value` is the first argument: offset 0
    // of the segment.)
    typedef const CallArgs APT_CONST_AS *ccall_ptr;
    const ccall_ptr callp = (ccall_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#define call (*callp)
    // Only the stage-1 tap table is fetched from `prm` here; everything the later stages need is
    // loaded after stage 1 (see `late`): an SGPR held across stage 1 is one its tap pipeline cannot
    // use, and the register allocator answered the extra pressure by spilling the table pointer
    // itself inside the hot loop.
    const cf2_ptr hs = (cf2_ptr)(prm->hs);  // [WIN][PS] tap pairs
    using Gm = FusedGeom<L, M, T1, T2, PW, NTHR, static_cast<int>(sizeof(XT)), fused_geom_var(MODE)>;
    constexpr int kFusedThreads = NTHR;
    constexpr int kOwnThreads = Gm::kOwnThreads, kPreThreads = Gm::kPreThreads, kPostThreads = Gm::kPostThreads;
    constexpr bool F16 = MODE == kModeF16Taps;
    constexpr bool MFMA = MODE == kModeMfma;         // the FIRs on the matrix cores; everything else as kModeFast
    constexpr bool PADLP = MODE == kModeStrictPad2;  // ... and T2 a bound too (zero-padded low-pass tables)
    constexpr bool PAD = MODE == kModeStrictPad || PADLP;  // strict arithmetic, T1 a bound (zero-padded table)
    constexpr bool FAST = MODE == kModeFast || MFMA;
    extern __shared__ float lds[];
    float *P = lds;                  // x tile -> R -> F
    float *Q = lds + Gm::D_OFF;      // D (inside the dead part of the x tile), later the pulse sums (fast mode)
    const int tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < NTHR);  // (the compiler only knows < 1024: it kept three always-true guards of tile loads)
    // A (wave-uniform) 64-bit offset clamped to +-2^30, in scalar instructions.  Written in C++ the compiler
    // evaluates it with VALU instructions (v_cmp_lt_i64, v_med3_i32, v_readfirstlane: there is no scalar 64-bit
    // signed compare and no scalar med3) — a dozen issue slots per tile for five uniform numbers.
    auto rel = [](int64_t v) -> int { return v < -(1 << 30) ? -(1 << 30) : (v > (1 << 30) ? (1 << 30) : static_cast<int>(v)); };
    auto rel_u = [&](int64_t v) -> int {  // v wave-uniform
        if constexpr (Gm::TABLE) return rel(v);
        const int32_t hi = static_cast<int32_t>(v >> 32), lo = static_cast<int32_t>(static_cast<uint32_t>(v));
        int32_t r, t;
        asm("s_ashr_i32 %0, %3, 31\n\t"          // sign of v
            "s_xor_b32 %0, %0, 0x7fffffff\n\t"   // -> INT_MAX / INT_MIN: v saturated to 32 bits
            "s_ashr_i32 %1, %2, 31\n\t"
            "s_cmp_eq_u32 %1, %3\n\t"            // v fits 32 bits?
            "s_cselect_b32 %0, %2, %0\n\t"
            "s_max_i32 %0, %0, 0xc0000000\n\t"
            "s_min_i32 %0, %0, 0x40000000"
            : "=&s"(r), "=&s"(t)
            : "s"(lo), "s"(hi)
            : "scc");
        return r;
    };

    // ---- stage 0a: a tile's input -> registers (coalesced 16-byte loads, zero outside [0, n));
    // f32: 4 samples per register quad, PCM16: 4 samples per register pair
    constexpr int NXR = (Gm::XT_PAD / 4 + kFusedThreads - 1) / kFusedThreads;
    using XReg = std::conditional_t<sizeof(XT) == 4, float4, uint2>;
    auto load_tile = [&](uint32_t ri, int64_t tile, int sub, XReg (&xr)[NXR]) {  // (sub: the sub-tile of a SPLIT stage 1, else 0)
        const XT *__restrict__ x = static_cast<const XT *>(call.rec[ri].x);
        const uint64_t n = call.rec[ri].n;
        const int64_t k0 = tile * Gm::OWN_K - Gm::PRE_K;   // first work sample of the tile (< 0 in tile 0)
        const int64_t xs0 = (k0 / L + sub * Gm::NS) * M - Gm::XSHIFT;  // first input sample the tile loads (16-byte aligned)
        const int x_lo = rel_u(-xs0);                               // tile index of input sample 0
        const int x_hi = rel_u(static_cast<int64_t>(n) - xs0);      // tile index of input sample n
        const XT *xt = x + xs0;  // only dereferenced inside [x_lo, x_hi)
        if (x_lo <= 0 && x_hi >= Gm::XT_PAD) {
            // interior tile (wave-uniform): every load is a whole, unguarded 16 / 8 bytes.  The address is a scalar
            // base, advanced per load in scalar registers (the empty asm keeps the compiler from folding the steps
            // back into one base + constants, which it then adds per lane: a 64-bit VALU add pair per load), plus
            // the lane's 32-bit byte offset.  Registers of lanes beyond the tile's end stay undefined:
            // tile_to_lds applies the same guard.
            APT_MARK("BEGIN load");
            typedef const char __attribute__((address_space(1))) *gchar_ptr;  // (global, not flat, after the asm)
            gchar_ptr sb = (gchar_ptr)(xt);
            const uint32_t voff = static_cast<uint32_t>(tid) * static_cast<uint32_t>(sizeof(XReg));
#pragma unroll
            for (int e = 0; e < NXR; ++e) {
                const int q = (tid + e * kFusedThreads) * 4;
                if constexpr (sizeof(XT) == 4) {
#ifdef APT_FUSED_NOLOAD  // timing probe: the kernel without its HBM reads (synthetic tile contents)
                    xr[e] = make_float4(1000.f + q, 900.f - q, 800.f + (q & 255), 700.f - (q & 127));
#else
                    if (q < Gm::XT_PAD) {
                        const f4 v = *(const f4 __attribute__((address_space(1))) *)(sb + voff);
                        xr[e] = make_float4(v.x, v.y, v.z, v.w);
                    }
#endif
                } else {
                    // PCM16: x is 4-byte aligned and xs0, q are even, so sample pairs move as dwords
                    if (q < Gm::XT_PAD) {
                        const u2 v = *(const u2 __attribute__((address_space(1))) *)(sb + voff);
                        xr[e] = make_uint2(v.x, v.y);
                    }
                }
                sb += kFusedThreads * sizeof(XReg);
                asm volatile("" : "+s"(sb));
            }
            APT_MARK("END load");
        } else {
            // first / last tiles of a recording: sample by sample, zero outside [0, n)
#pragma unroll
            for (int e = 0; e < NXR; ++e) {
                const int q = (tid + e * kFusedThreads) * 4;
                auto at = [&](int i) -> XT { return (i < Gm::XT_PAD && i >= x_lo && i < x_hi) ? xt[i] : XT(0); };
                if constexpr (sizeof(XT) == 4) {
                    xr[e] = make_float4(at(q), at(q + 1), at(q + 2), at(q + 3));
                } else {
                    auto u = [&](int i) -> uint32_t { return static_cast<uint32_t>(static_cast<uint16_t>(at(i))); };
                    xr[e] = make_uint2(u(q) | (u(q + 1) << 16), u(q + 2) | (u(q + 3) << 16));
                }
            }
        }
    };
    // ---- stage 0b: registers -> LDS (PCM16: the tile stays int16 in LDS; `*x as f32` happens when
    // stage 1 reads it)
    auto tile_to_lds = [&](const XReg (&xr)[NXR]) {
        APT_MARK("BEGIN to_lds");
#pragma unroll
        for (int e = 0; e < NXR; ++e) {
            const int q = (tid + e * kFusedThreads) * 4;
            if (q < Gm::XT_PAD) {
                if constexpr (sizeof(XT) == 4) *reinterpret_cast<float4 *>(P + q) = xr[e];
                else *reinterpret_cast<uint2 *>(reinterpret_cast<uint32_t *>(lds) + q / 2) = xr[e];
            }
        }
        APT_MARK("END to_lds");
    };

    // ---- one tile through the four stages
    auto run_tile = [&](uint32_t ri, int64_t tile, XReg (&xr)[NXR], auto &&after_tile_in_lds) {
    // (field by field: only what the stages below need is loaded, and the output pointers are
    // fetched from the slot table after stage 3)
    APT_MARK("BEGIN tile_prologue");
    const uint64_t w = call.rec[ri].w;
    const uint64_t n_corr = w - Gm::G;  // w >= 10 rows of samples > G (checked on the host)
    const int64_t o0 = tile * Gm::OWN_K;            // first owned work sample
    const int64_t k0 = o0 - Gm::PRE_K;              // first work sample of the tile (< 0 in tile 0)
    // everything below indexes relative to the tile with 32-bit integers; the global limits
    // become wave-uniform scalars
    const int k_lo = rel_u(-k0);                                    // tile index of work sample 0
    const int k_hi = rel_u(static_cast<int64_t>(w) - k0);           // tile index of work sample w
    const int c_hi = rel_u(static_cast<int64_t>(n_corr) - k0);      // tile index of position n_corr
    const int kq = tid * L;        // this thread's first work sample, tile-relative
    const int kt = kq - k_lo;      // ... and as a global work-sample index clamped to int
    // Interior tile (wave-uniform; all but the first and the last two tiles of a recording): every work
    // sample of the tile exists and is a correlation position, so the per-sample edge tests of the
    // stages below (a v_cmp + v_cndmask each, ~200 per thread) are compiled out of the copies of the
    // small loops that interior tiles run.
    const bool interior = k_lo < 0 && c_hi >= Gm::TILE_K;
    APT_MARK("END tile_prologue");
    float r[L];
    // (kModeStrictPad2's "F not all finite" flag: a word behind everything else in LDS — a constant in the SPLIT kernels, behind
    // the run-time input tile in the PHASE ones)
    [[maybe_unused]] uint32_t lp_flag_off = static_cast<uint32_t>(Gm::LP_FLAG_OFF);
    if constexpr (Gm::PHASE) {
        // ---- stages 0 + 1, taps of the thread's polyphase branches in registers (dsp.rs:252-263):
        // k*m - X0*l = v;  x0 - X0 = c = ceil(v / l);  phase p = c*l - v;  output k = sum_i h[p + i*l] * x[x0 + i]
        // A thread holds NQ = -M slots u + q S' (q < NQ, S' = S / NQ) of the S consecutive outputs after which the
        // branches repeat, and computes the NB / NQ outputs (u + q S') + a S of each: NQ = 1 where S = l floor(NTHR / l)
        // fits the workgroup (44 100 Hz: l = 208), NQ = 2 / 4 where l itself is two / four times too long for one slot
        // per thread (22 050 Hz: l = 416, 11 025 Hz: l = 832 — until round 5 512- and 1024-thread workgroups at two / one
        // per CU, and the table-driven stage 1): the same 256-thread kernel at three workgroups per CU for all of them.
        constexpr int NQ = -M;     // branches per thread
        // sixteen branches (l = 3328: the fast profile at 11 025 Hz, one output per branch and tile): the arithmetic stays
        // in pairs, the pair's second output — a tile further on — a phantom that is neither loaded for nor stored
        constexpr bool HALF = NQ == 16;
        constexpr int NB = HALF ? 32 : 16;  // outputs per thread (NB / NQ >= ceil(TILE_K / S): checked by fused_phase_supported)
        constexpr int NWIN = NB / NQ, NREG = NWIN / 2;  // outputs per branch; regions of the paired tile
        static_assert(NQ == 1 || NQ == 2 || NQ == 4 || NQ == 8 || NQ == 16, "one, two, four, eight or sixteen branches per thread");
        // T1 == 1 (unused otherwise in this mode): the STREAMED form for filters too long for the registers (the slow
        // profile: 197 taps per branch) — a branch's taps are fetched from the table sixteen at a time while the previous
        // sixteen are in use, instead of all of them before the first multiplication
        constexpr bool STREAM = T1 == 1;
        // taps per branch the registers hold (>= tpp; 128 / 170 VGPRs).  The fast profile's filters are short (639 taps at
        // 48 kHz: 25 per branch at every rate — its transition band is three times the standard profile's): 28 registers,
        // so that its kernels fit four workgroups per CU
        constexpr int TPPM = STREAM ? 4 : NTHR > 512 ? 24 : NTHR > 256 ? 40 : (T2 == 43 ? 28 : NQ == 1 ? 76 : NQ == 2 ? 36 : 20);  // (phase_tap_regs in apt_kernels_fused.hip)
        typedef const FusedParams APT_CONST_AS *cprm_tab_ptr;
        const cprm_tab_ptr tp = (cprm_tab_ptr)(prm);
        const uint32_t gl = tp->tab.l, gm_ = tp->tab.m, tpp = tp->tab.tpp;
        const uint32_t S = tp->tab.step_r, dq = tp->tab.step_q;  // stride of a branch's outputs; S*m / l input samples
        // slot stride = threads with work: thread slot u holds the slots u + q SQ that are < S (TableGeom::sq — S, S / NQ, or
        // the workgroup's threads)
        const uint32_t SQ = NQ == 1 ? S : tp->tab.sq;
        const uint32_t ZR = tp->tab.off_x;                        // f2 entries per region of the paired tile
        const uint32_t jl_a = tp->tab.jl_a, jl_b = tp->tab.jl_b;
        const XT *__restrict__ x = static_cast<const XT *>(call.rec[ri].x);
        const int64_t n = static_cast<int64_t>(call.rec[ri].n);
        // The tile starts at work sample k0, which is negative in tile 0: the samples before the recording
        // read as zero (like those at or past its end), the outputs before it are zeroed afterwards.
        // k0*m = X0*l + rb, 0 <= rb < l: tabulated per tile phase where the phases repeat (TableGeom::exact — every rate
        // recordings come in), a 64-bit division on the vector unit otherwise
        const uint32_t exact = tp->tab.exact;
        int64_t X0;
        uint32_t rb, rsel = 0;
        if (exact) {
            const uint32_t sh = tp->tab.perm_shift;
            rsel = static_cast<uint32_t>(tile) & ((1u << sh) - 1u);
            const uint32_t t1 = static_cast<uint32_t>(tile) >> sh;
            X0 = static_cast<int64_t>(tp->tab.x0r[rsel]) + static_cast<int64_t>(static_cast<uint64_t>(t1) * tp->tab.xd);
            rb = tp->tab.rbr[rsel];
        } else {
            const int64_t kbm = k0 * static_cast<int64_t>(gm_);
            X0 = kbm / static_cast<int64_t>(gl);
            if (X0 * static_cast<int64_t>(gl) > kbm) --X0;                // floor
            rb = static_cast<uint32_t>(kbm - X0 * static_cast<int64_t>(gl));
        }
        const int64_t xfirst = X0 + (rb ? 1 : 0);
        const int64_t xs0 = xfirst & ~static_cast<int64_t>(3);        // first input sample of the tile
        const uint32_t xrel0 = static_cast<uint32_t>(X0 - xs0);       // -1 (wrapped) only when rb > 0, and then c >= 1
        // this thread's slot u (its branches are the same for all their outputs: S*m is a multiple of l).  Which thread
        // takes which slot is the host's choice (fused_phase_table: the lists that keep the LDS reads below off each
        // other's banks; one list per tile phase rb), and where the lists are exact the window start c and the branch p
        // of each of its slots come with it
        const uint32_t lidx = rsel * static_cast<uint32_t>(kFusedThreads) + static_cast<uint32_t>(tid);
        const uint32_t u_slot = reinterpret_cast<const uint32_t *>(tp->table + tp->tab.perm_off)[lidx];
        const bool act = u_slot < SQ;
        // which of the thread's slots exist (the last ones of some threads do not where SQ NQ > S: their branch is skipped
        // by a wave none of whose lanes has it, and computed on window 0 / branch 0 and dropped by a lane beside one that has)
        bool vq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) vq[q] = act && (NQ == 1 || u_slot + static_cast<uint32_t>(q) * SQ < S);
        uint32_t cq[NQ], phq[NQ];
        if (exact) {
            typedef uint32_t uq __attribute__((ext_vector_type(NQ)));
            uint32_t cpv[NQ];
            if constexpr (NQ == 1) {
                cpv[0] = reinterpret_cast<const uint32_t *>(tp->table + tp->tab.cp_off)[lidx];
            } else {
                const uq v = reinterpret_cast<const uq *>(tp->table + tp->tab.cp_off)[lidx];
#pragma unroll
                for (int q = 0; q < NQ; ++q) cpv[q] = v[q];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                cq[q] = cpv[q] & 0xFFFFu;
                phq[q] = cpv[q] >> 16;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t v = rb + (vq[q] ? u_slot + static_cast<uint32_t>(q) * SQ : 0u) * gm_;
                cq[q] = (v + gl - 1) / gl;
                phq[q] = cq[q] * gl - v;
            }
        }
        // The input tile in LDS, PAIRED: outputs a = 2jj and 2jj+1 of a branch read windows exactly dq samples
        // apart, so region jj holds Z[jj][s] = (x[2jj*dq + s], x[(2jj+1)*dq + s]), s < ZR = dq + window + slack:
        // one 8-byte LDS read then delivers the two samples a packed multiply needs, already in a register
        // pair (two 4-byte reads would be merged by the compiler with their NEIGHBOURS in the window, and
        // the pairs rebuilt with a v_mov per sample).  The windows' overlap (~10 % at NQ = 1) is stored twice.
        f2 *Z = reinterpret_cast<f2 *>(lds);
        // One branch per thread (HALVES): the tile goes through LDS in two passes of NREG / 2 regions — windows 0-7, then 8-15
        // — so that the work-rate stages' 26 KB, not the 45-52 KB of sixteen windows' input, set the footprint (three
        // workgroups per CU until round 6); the taps stay in their registers across both passes.
        constexpr bool HALVES = phase_halves(NQ, STREAM, NTHR, T2, FAST);
        constexpr int NPASS = HALVES ? 2 : 1, RPP = NREG / NPASS;  // passes; regions per pass
        static_assert(!HALVES || NREG % 2 == 0, "two passes of whole regions");
        if constexpr (PADLP && sizeof(XT) == 4) {
            const uint32_t xt_used = HALVES ? tp->tab.xt / 2u : tp->tab.xt;
            lp_flag_off = xt_used > static_cast<uint32_t>(Gm::W_LDS_FLOATS) ? xt_used : static_cast<uint32_t>(Gm::W_LDS_FLOATS);
            if (tid == 0) reinterpret_cast<uint32_t *>(lds)[lp_flag_off] = 0u;  // (read behind several barriers, after stage 3)
        }
        // every load of a pass issued before its first LDS write (regions x rounds unrolled: a loop
        // that waited for each round's loads cost 26 HBM latencies per tile)
        // ceil(ZR / NTHR) at most (phase_geom: ZR <= 1024; the fast profile's long periods — 44 100 Hz: m = 2205 — 2304,
        // with one or two regions only)
        constexpr int ZROUNDS = (T2 == 43 && NQ > 2) ? 9 : 1024 / kFusedThreads;
        const XT *xt0 = x + xs0;    // only dereferenced inside [x_lo, x_hi)
        const int x_lo = rel(-xs0), x_hi = rel(n - xs0);
        const bool x_interior = x_lo <= 0 && x_hi >= static_cast<int>((HALF ? 0 : NWIN - 1) * dq + ZR);
        auto load_pass = [&](auto pp) {
            constexpr int J0 = decltype(pp)::value * RPP;  // first region of the pass
            XT za[RPP][ZROUNDS], zb[RPP][ZROUNDS];
            if (x_interior) {
                // interior tile (wave-uniform): every sample exists.  A scalar base per region half, advanced in scalar
                // registers (the empty asm keeps the steps from being folded into per-lane 64-bit adds, as in load_tile),
                // plus the lane's 32-bit byte offset — the guarded form below spends ten VALU instructions per sample
                // on index arithmetic and range tests, 700 per thread and tile at NQ = 1.
                typedef const char __attribute__((address_space(1))) *gchar_ptr;
                typedef const XT __attribute__((address_space(1))) *gx_ptr;
                const uint32_t dqb = dq * static_cast<uint32_t>(sizeof(XT));
#pragma unroll
                for (int rr = 0; rr < ZROUNDS; ++rr) {
                    const uint32_t s_in = static_cast<uint32_t>(tid + rr * kFusedThreads);
                    const uint32_t voff = s_in * static_cast<uint32_t>(sizeof(XT));
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) {
                        za[jj][rr] = XT(0);
                        zb[jj][rr] = XT(0);
                    }
                    if (s_in < ZR) {
                        // (xt0 is wave-uniform, but the 64-bit division behind it runs on the vector unit)
                        const uint64_t xa = reinterpret_cast<uint64_t>(xt0);
                        // (the builtin returns int: through uint32_t, or the low half is sign-extended over the high one)
                        const uint32_t xa_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(xa))));
                        const uint32_t xa_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(xa >> 32))));
                        const uint64_t xu = static_cast<uint64_t>(xa_lo) | (static_cast<uint64_t>(xa_hi) << 32);
                        gchar_ptr sb = (gchar_ptr)(xu);
                        if constexpr (J0 > 0) {
                            sb += static_cast<uint64_t>(2 * J0) * dqb;
                            asm volatile("" : "+s"(sb));
                        }
#pragma unroll
                        for (int jj = 0; jj < RPP; ++jj) {
                            za[jj][rr] = *(gx_ptr)(sb + voff);
                            sb += dqb;
                            asm volatile("" : "+s"(sb));
                            if constexpr (!HALF) zb[jj][rr] = *(gx_ptr)(sb + voff);
                            sb += dqb;
                            asm volatile("" : "+s"(sb));
                        }
                    }
                }
            } else {
#pragma unroll
            for (int jj = 0; jj < RPP; ++jj) {
#pragma unroll
                for (int rr = 0; rr < ZROUNDS; ++rr) {
                    const uint32_t s_in = static_cast<uint32_t>(tid + rr * kFusedThreads);
                    const int ia = static_cast<int>(static_cast<uint32_t>(2 * (J0 + jj)) * dq + s_in);
                    const int ib = ia + static_cast<int>(dq);
                    za[jj][rr] = (s_in < ZR && ia >= x_lo && ia < x_hi) ? xt0[ia] : XT(0);
                    zb[jj][rr] = (!HALF && s_in < ZR && ib >= x_lo && ib < x_hi) ? xt0[ib] : XT(0);
                }
            }
            }
#pragma unroll
            for (int jj = 0; jj < RPP; ++jj) {
#pragma unroll
                for (int rr = 0; rr < ZROUNDS; ++rr) {
                    const uint32_t s_in = static_cast<uint32_t>(tid + rr * kFusedThreads);
                    if (s_in < ZR) Z[jj * ZR + s_in] = (f2){static_cast<float>(za[jj][rr]), static_cast<float>(zb[jj][rr])};
                }
            }
        };
        load_pass(std::integral_constant<int, 0>{});
        // its taps -> registers (16-byte loads from the L2-resident table; the first ones in flight across the barrier).
        // The branches are worked through one after the other.
        typedef float f4v __attribute__((ext_vector_type(4)));
        // The taps of a branch are held one SEGMENT of SEG taps at a time, in two ping-pong buffers: the segment after
        // next is requested when a segment has been used up.  One branch per thread: the whole branch is one segment,
        // loaded before the first multiplication (one buffer).  Two branches (tpp <= 36): segments of 20 taps — 40
        // registers where both branches' rows took 72.  Four / eight: a branch per segment.
        // (round 6, HALVES with a long filter — 44 100 Hz at the standard profile, 68 taps: segments of 20 as well, fetched
        // again for the second pass; a resident branch, 76 registers, left no room for the second pass's loads at four waves
        // per SIMD: 228 registers spilled)
        constexpr int SEG = (NQ == 2 || (HALVES && TPPM > 40)) ? 20 : TPPM;
        constexpr int SPB = (TPPM + SEG - 1) / SEG;      // segments per branch
        constexpr int NSEG = NQ * SPB;
        constexpr int NTB = NSEG > 1 ? 2 : 1;
        f4v tq[NTB][SEG / 4];
        static_assert(SEG % 4 == 0 && TPPM % 4 == 0, "segments of whole quads");
        auto load_taps = [&](auto ss) {
            constexpr int sidx = decltype(ss)::value;
            constexpr int q = sidx / SPB, k = sidx % SPB;
            // (exact lists: the thread-order copy — quad e of this thread's branch q is element e * NTHR + tid of its
            // [tpp / 4][NTHR] block, a wave's load 1 KB of consecutive memory; else its row of the phase-major table)
            const uint32_t tt_off = tp->tab.tt_off;  // (0: no such copy)
            const f4v *row = tt_off ? reinterpret_cast<const f4v *>(tp->table + tt_off) +
                                          (static_cast<size_t>(rsel) * NQ + q) * (tpp / 4) * kFusedThreads + tid
                                    : reinterpret_cast<const f4v *>(tp->table) + static_cast<size_t>(act ? phq[q] : 0u) * (tpp / 4);
            const uint32_t estep = tt_off ? static_cast<uint32_t>(kFusedThreads) : 1u;
            // (quads past the row's end are never multiplied — the tap loop stops at jl_a <= tpp — and are not loaded: a
            // wave-uniform test per quad; as a select per register it cost a v_cndmask per tap)
#pragma unroll
            for (int e = 0; e < SEG / 4; ++e) {
                const int eq = k * (SEG / 4) + e;  // quad of the branch
                if (eq < TPPM / 4 && static_cast<uint32_t>(4 * eq) < tpp) tq[sidx % NTB][e] = row[static_cast<size_t>(eq) * estep];
            }
        };
        if constexpr (!STREAM) static_for<0, NTB>(load_taps);
        __syncthreads();
        if constexpr (APT_FUSED_STOP == 1) return;
        f2 acc[NQ][NREG];  // branch q, output pair (2jj, 2jj+1)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int jj = 0; jj < NREG; ++jj) acc[q][jj] = (f2){0.f, 0.f};
        auto compute_pass = [&](auto pp) {
        constexpr int J0 = decltype(pp)::value * RPP;  // first region (= output pair) of the pass
        if (act) {
            static_for<0, NQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            // (a slot that exists for no lane of the wave — then neither do the later ones: nothing waits for the tap
            // segments this branch's loop would have requested)
            if constexpr (NQ > 1) {
                if (__builtin_amdgcn_ballot_w64(vq[q]) == 0ull) return;
            }
            const f2 *zw[RPP];  // window of the branch's output pair (2jj, 2jj+1)
#pragma unroll
            for (int jj = 0; jj < RPP; ++jj) zw[jj] = Z + jj * ZR + (xrel0 + cq[q]);
            // Software pipeline over the taps: the 8-byte reads of tap i + 1 are issued before the arithmetic of tap i
            // (until round 5 a chunk's reads were issued and waited for in place: one exposed LDS latency per two taps, at
            // three waves per SIMD).  Two buffers of RPP pairs (NQ = 1: the same registers a two-tap chunk held).
            f2 xb[2][RPP];
            if constexpr (STREAM) {
                constexpr int TCH = 16;  // taps per chunk: four 16-byte loads of the branch's row (rows are padded to chunks)
                const uint32_t tt_off = tp->tab.tt_off;
                const f4v *row = tt_off ? reinterpret_cast<const f4v *>(tp->table + tt_off) +
                                              (static_cast<size_t>(rsel) * NQ + q) * (tpp / 4) * kFusedThreads + tid
                                        : reinterpret_cast<const f4v *>(tp->table) + static_cast<size_t>(phq[q]) * (tpp / 4);
                const uint32_t estep = tt_off ? static_cast<uint32_t>(kFusedThreads) : 1u;
                f4v tb[2][TCH / 4];
                const f2 *zp[RPP];
#pragma unroll
                for (int jj = 0; jj < RPP; ++jj) zp[jj] = zw[jj];
                auto fetch = [&](auto bb, uint32_t ch) {
                    constexpr int b = decltype(bb)::value;
#pragma unroll
                    for (int e = 0; e < TCH / 4; ++e) tb[b][e] = row[static_cast<size_t>(ch * (TCH / 4) + e) * estep];
                };
                // one chunk: tap i0 + k against the pairs at zp[jj][k]; the reads of tap k + 1 issued before the arithmetic
                // of tap k; a wave-uniform test per tap (all lanes run taps i < jl_a)
                auto chunk = [&](auto bb, uint32_t i0) {
                    constexpr int b = decltype(bb)::value;
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) xb[0][jj] = zp[jj][0];
                    static_for<0, TCH>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        if (i0 + static_cast<uint32_t>(k) < jl_a) {
                            if constexpr (k + 1 < TCH) {
#pragma unroll
                                for (int jj = 0; jj < RPP; ++jj) xb[(k + 1) & 1][jj] = zp[jj][k + 1];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            const f4v q4 = tb[b][k / 4];
                            const float t = (k & 3) == 0 ? q4.x : (k & 3) == 1 ? q4.y : (k & 3) == 2 ? q4.z : q4.w;
                            if constexpr (FAST) {
#pragma unroll
                                for (int jj = 0; jj < RPP; ++jj) acc[q][J0 + jj] = __builtin_elementwise_fma((f2){t, t}, xb[k & 1][jj], acc[q][J0 + jj]);
                            } else {
                                f2 pr[RPP];
#pragma unroll
                                for (int jj = 0; jj < RPP; ++jj) pr[jj] = (f2){t, t} * xb[k & 1][jj];
#pragma unroll
                                for (int jj = 0; jj < RPP; ++jj) acc[q][J0 + jj] = acc[q][J0 + jj] + pr[jj];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) zp[jj] += TCH;
                };
                const uint32_t nch = (jl_a + TCH - 1) / TCH;
                fetch(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
                for (uint32_t ch = 0; ch < nch; ch += 2) {
                    fetch(std::integral_constant<int, 1>{}, ch + 1);  // (past the row's last chunk: the next row or the table's slack, never used)
                    chunk(std::integral_constant<int, 0>{}, ch * TCH);
                    if (ch + 1 < nch) {
                        fetch(std::integral_constant<int, 0>{}, ch + 2);
                        chunk(std::integral_constant<int, 1>{}, (ch + 1) * TCH);
                    }
                }
            }
            auto issue = [&](auto ii) {
                constexpr int i = decltype(ii)::value;
#pragma unroll
                for (int jj = 0; jj < RPP; ++jj) xb[i & 1][jj] = zw[jj][i];
            };
            auto tap = [&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const f4v q4 = tq[(q * SPB + i / SEG) % NTB][(i % SEG) / 4];
                const float t = (i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w;
                if constexpr (FAST) {
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) acc[q][J0 + jj] = __builtin_elementwise_fma((f2){t, t}, xb[i & 1][jj], acc[q][J0 + jj]);
                } else {
                    f2 pr[RPP];
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) pr[jj] = (f2){t, t} * xb[i & 1][jj];
#pragma unroll
                    for (int jj = 0; jj < RPP; ++jj) acc[q][J0 + jj] = acc[q][J0 + jj] + pr[jj];
                }
            };
            // taps 0 .. jl_a - 1 for every branch, tap jl_a for the branches p < jl_b (the reference's
            // `p + i*l < jlim`): the tap count is wave-uniform up to that last, lane-predicated one.
            bool go = !STREAM;  // (wave-uniform)
            if constexpr (!STREAM) issue(std::integral_constant<int, 0>{});
            static_for<0, STREAM ? 0 : TPPM>([&](auto ee) {
                constexpr int i0 = decltype(ee)::value;
                go = go && static_cast<uint32_t>(i0 + 1) <= jl_a;
                if (go) {
                    // (a read past the branch's last tap stays inside its region's slack or the next region: never used)
                    if constexpr (i0 + 1 < TPPM) issue(std::integral_constant<int, i0 + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    tap(ee);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // a segment's last tap is behind us (or was never needed): its buffer takes the segment after next
                if constexpr ((i0 + 1) % SEG == 0 || i0 + 1 == TPPM) {
                    constexpr int sidx = q * SPB + i0 / SEG;
                    if constexpr (sidx + NTB < NSEG) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_taps(std::integral_constant<int, sidx + NTB>{});
                    }
                }
            });
            // the predicated last tap, read from the table again (a run-time index)
            if (phq[q] < jl_b) {
                const float t = tp->table[static_cast<size_t>(phq[q]) * tpp + jl_a];
                f2 xp[RPP];
#pragma unroll
                for (int jj = 0; jj < RPP; ++jj) xp[jj] = zw[jj][jl_a];
#pragma unroll
                for (int jj = 0; jj < RPP; ++jj) {
                    if constexpr (FAST) acc[q][J0 + jj] = __builtin_elementwise_fma((f2){t, t}, xp[jj], acc[q][J0 + jj]);
                    else acc[q][J0 + jj] = acc[q][J0 + jj] + (f2){t, t} * xp[jj];
                }
            }

            });
        }
        };
        compute_pass(std::integral_constant<int, 0>{});
        if constexpr (HALVES) {
            if constexpr (NSEG > 1) static_for<0, NTB>(load_taps);  // (the branch's first segments again: both buffers are free)
            __syncthreads();  // everyone is done with windows 0-7
            load_pass(std::integral_constant<int, 1>{});
            __syncthreads();
            compute_pass(std::integral_constant<int, 1>{});
        }
        __syncthreads();  // everyone is done with the input tile: R may land on it
        if (act) {
            // (wave-uniform: every output of every slot lies inside the tile — l = 208, 416, 832 — and exists: plain stores)
            constexpr int NST = HALF ? 1 : NWIN;  // outputs of a branch that exist
            const bool plain = interior && static_cast<uint32_t>(NST) * S <= static_cast<uint32_t>(Gm::TILE_K);
            if (plain) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!vq[q]) continue;
#pragma unroll
                    for (int a = 0; a < NST; ++a)
                        P[static_cast<int>(u_slot + static_cast<uint32_t>(q) * SQ) + a * static_cast<int>(S)] = (a & 1) ? acc[q][a / 2].y : acc[q][a / 2].x;
                }
            } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!vq[q]) continue;
#pragma unroll
                for (int a = 0; a < NST; ++a) {
                    const int idx = static_cast<int>(u_slot + static_cast<uint32_t>(q) * SQ) + a * static_cast<int>(S);
                    const float val = (a & 1) ? acc[q][a / 2].y : acc[q][a / 2].x;
                    // (outputs before the recording or at / past its end: zero; an interior tile has none)
                    if (idx < Gm::TILE_K) P[idx] = (interior || (idx >= k_lo && idx < k_hi)) ? val : 0.f;
                }
            }
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < L; ++b) r[b] = P[kq + b];
    } else
    if constexpr (Gm::TABLE) {
        // ---- stages 0 + 1, table-driven (dsp.rs:252-263): k*m - X0*l = v;  x0 - X0 = c = ceil(v / l);
        // phase p = c*l - v; output k = sum_i table[p][i] * x[x0 + i] over the taps the reference uses
        // (jl_a + (p < jl_b) of them), inputs at or past n zero-filled (exact: see k_fused_any).
        typedef const FusedParams APT_CONST_AS *cprm_tab_ptr;
        const cprm_tab_ptr tp = (cprm_tab_ptr)(prm);
        TableGeom G;  // (field by field: a struct cannot be copied out of the constant address space)
        G.l = tp->tab.l;
        G.m = tp->tab.m;
        G.jlim = tp->tab.jlim;
        G.tpp = tp->tab.tpp;
        G.xt = tp->tab.xt;
        G.off_x = tp->tab.off_x;
        G.step_q = tp->tab.step_q;
        G.step_r = tp->tab.step_r;
        G.jl_a = tp->tab.jl_a;
        G.jl_b = tp->tab.jl_b;
        const cfloat_ptr table = (cfloat_ptr)(tp->table);
        const XT *__restrict__ x = static_cast<const XT *>(call.rec[ri].x);
        const uint64_t n = call.rec[ri].n;
        float *T = lds;
        float *X = lds + G.off_x;
        const int64_t kb = k0 > 0 ? k0 : 0;                           // first work sample that exists
        const int idx0 = static_cast<int>(kb - k0);
        const uint64_t kbm = static_cast<uint64_t>(kb) * G.m;         // kb*m = X0*l + rb
        const uint64_t X0 = kbm / G.l;
        const uint32_t rb = static_cast<uint32_t>(kbm - X0 * G.l);
        const uint64_t xfirst = X0 + (rb ? 1 : 0);
        const uint64_t xs0 = xfirst & ~3ull;                          // tile's first input, 16-byte aligned
        const uint32_t xrel0 = static_cast<uint32_t>(X0 - xs0);       // may wrap by -1: only used with c >= 1 or rb == 0
        {
            const uint32_t nt = G.l * G.tpp, nt4 = nt / 4;
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef const f4v APT_CONST_AS *cf4_ptr;
            const cf4_ptr t4 = (cf4_ptr)(tp->table);
            f4v *l4 = reinterpret_cast<f4v *>(T);
#pragma unroll 4
            for (uint32_t q = tid; q < nt4; q += kFusedThreads) l4[q] = t4[q];
            for (uint32_t q = 4 * nt4 + tid; q < nt; q += kFusedThreads) T[q] = table[q];
        }
        // the input tile: batches of four loads per thread in flight before anything is written to LDS (a loop that
        // waited for each round's load paid one HBM round trip per 2048 / 8192 samples: eight of them per tile at
        // 32 kHz, where the phase-resident stage 1 — all loads first — was 19 % faster for that reason alone)
        if constexpr (sizeof(XT) == 4) {
            const float *xf = reinterpret_cast<const float *>(x);
            if ((reinterpret_cast<uintptr_t>(xf) & 15u) == 0) {
                constexpr uint32_t KB = 4;
                for (uint32_t q0 = tid * 4; q0 < G.xt; q0 += KB * kFusedThreads * 4) {
                    float4 v[KB];
#pragma unroll
                    for (uint32_t e = 0; e < KB; ++e) {
                        const uint32_t q = q0 + e * kFusedThreads * 4;
                        const uint64_t i = xs0 + q;
                        v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (q < G.xt) {
                            if (i + 3 < n) {
                                v[e] = *reinterpret_cast<const float4 *>(xf + i);
                            } else {
                                v[e].x = i < n ? xf[i] : 0.f;
                                v[e].y = i + 1 < n ? xf[i + 1] : 0.f;
                                v[e].z = i + 2 < n ? xf[i + 2] : 0.f;
                                v[e].w = i + 3 < n ? xf[i + 3] : 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t e = 0; e < KB; ++e) {
                        const uint32_t q = q0 + e * kFusedThreads * 4;
                        if (q < G.xt) *reinterpret_cast<float4 *>(X + q) = v[e];
                    }
                }
            } else {
                for (uint32_t q = tid; q < G.xt; q += kFusedThreads) X[q] = xs0 + q < n ? xf[xs0 + q] : 0.f;
            }
        } else {
            // mono PCM16 payload (wav.rs:37: `*x as f32`)
            constexpr uint32_t KB = 8;
            for (uint32_t q0 = tid; q0 < G.xt; q0 += KB * kFusedThreads) {
                float v[KB];
#pragma unroll
                for (uint32_t e = 0; e < KB; ++e) {
                    const uint32_t q = q0 + e * kFusedThreads;
                    v[e] = (q < G.xt && xs0 + q < n) ? static_cast<float>(x[xs0 + q]) : 0.f;
                }
#pragma unroll
                for (uint32_t e = 0; e < KB; ++e) {
                    const uint32_t q = q0 + e * kFusedThreads;
                    if (q < G.xt) X[q] = v[e];
                }
            }
        }
        __syncthreads();
        if constexpr (APT_FUSED_STOP == 1) return;
        float s1[L];  // outputs tid, tid + NTHR, ...: consecutive lanes = consecutive outputs
        {
            uint32_t c = 0, ph = 0;
            bool primed = false;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const int idx = tid + i * kFusedThreads;
                float sum = 0.f;
                if (idx >= idx0) {
                    if (!primed) {
                        const uint32_t v = rb + static_cast<uint32_t>(idx - idx0) * G.m;
                        c = (v + G.l - 1) / G.l;
                        ph = c * G.l - v;
                        primed = true;
                    } else {
                        c += G.step_q;
                        if (ph >= G.step_r) {
                            ph -= G.step_r;
                        } else {
                            ph += G.l - G.step_r;
                            c += 1;
                        }
                    }
                    if (idx < k_hi) {
                        const uint32_t cnt = G.jl_a + (ph < G.jl_b ? 1u : 0u);
                        const float *row = T + ph * G.tpp;
                        const float *xs = X + (xrel0 + c);
                        // batches of 8 taps: sixteen LDS reads in flight, then the eight MACs in tap order
                        uint32_t j = 0;
                        for (; j + 8 <= cnt; j += 8) {
                            float tv[8], xv[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                tv[e] = row[j + e];
                                xv[e] = xs[j + e];
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if constexpr (FAST) sum = __builtin_fmaf(tv[e], xv[e], sum);
                                else sum = sum + tv[e] * xv[e];
                            }
                        }
                        for (; j < cnt; ++j) {
                            if constexpr (FAST) sum = __builtin_fmaf(row[j], xs[j], sum);
                            else sum = sum + row[j] * xs[j];
                        }
                    }
                }
                s1[i] = sum;
            }
        }
        __syncthreads();  // everyone is done with the table and the input tile: R may land on them
#pragma unroll
        for (int i = 0; i < L; ++i) P[tid + i * kFusedThreads] = s1[i];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < L; ++b) r[b] = P[kq + b];
    } else {
    if constexpr (!MFMA) {  // (the matrix-core stage 1 writes its own LDS image of the tile)
    tile_to_lds(xr);
    __syncthreads();
    }
    if constexpr (APT_FUSED_STOP == 1) return;
    APT_MARK("BEGIN stage1");

    // ---- stage 1: polyphase resampler, L outputs per thread (dsp.rs:252-263)
    // Sample-stationary form: window sample q is broadcast (op_sel) against a PAIR of taps
    // (one scalar-loaded SGPR pair) feeding a pair of accumulators, so every tap costs half
    // a v_pk_mul_f32 + half a v_pk_add_f32 and no register shuffling.  Each branch still
    // accumulates its own taps in ascending order, products and sums rounded separately.
    if constexpr (F16) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        constexpr int NQP = (Gm::WIN + 1) / 2;  // window sample pairs
        auto xsrc = [&](int q) -> float {
            if constexpr (sizeof(XT) == 2) return static_cast<float>(reinterpret_cast<const int16_t *>(lds)[tid * M + Gm::XSHIFT + q]);
            else return P[tid * M + Gm::XSHIFT + q];
        };
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        typedef const u4v APT_CONST_AS *cu4v_ptr;
        const cu4v_ptr ht = (cu4v_ptr)(hs);  // [NQP][16] half2 bit patterns (4 x 4 dwords per row)
        float acc[L];
#pragma unroll
        for (int b = 0; b < L; ++b) acc[b] = 0.f;
        u4v tb[2][4];  // tap pairs of the sample pair in use / in flight (16 SGPRs each)
        float xa[2], xb[2];
        static_assert(L <= 16, "one 16-dword table row per sample pair");
        auto dot2 = [](float &a, uint32_t tap_pair /*SGPR*/, h2v x_pair) {
            asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a) : "s"(tap_pair), "v"(x_pair));
        };
        auto tapw = [&](int buf, int b) -> uint32_t {
            const u4v v = tb[buf][b >> 2];
            return (b & 3) == 0 ? v.x : (b & 3) == 1 ? v.y : (b & 3) == 2 ? v.z : v.w;
        };
        auto issue = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int buf = c & 1;
            xa[buf] = xsrc(2 * c);
            xb[buf] = (2 * c + 1 < Gm::WIN) ? xsrc(2 * c + 1) : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) tb[buf][k] = ht[c * 4 + k];
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NQP>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int buf = c & 1;
            const h2v xh = {static_cast<_Float16>(xa[buf]), static_cast<_Float16>(xb[buf])};
            // the first dot forces the wait for the loads issued one pair ago
            // (asm volatile pins the dots between the scheduling barriers: as plain intrinsics the
            // compiler sank all 793 of them behind the loads and spilled 750 SGPRs)
            // only the branches whose taps reach one of the two samples (the others hold zeros)
            constexpr int b0 = first_branch_of_pair<L, M, T1>(c);
            if constexpr (b0 < L) dot2(acc[b0], tapw(buf, b0), xh);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (c + 1 < NQP) issue(std::integral_constant<int, c + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, L>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                if constexpr (b > b0 && (branch_uses<L, M, T1>(b, 2 * c) || branch_uses<L, M, T1>(b, 2 * c + 1)))
                    dot2(acc[b], tapw(buf, b), xh);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        const float f16_unscale = prm->f16_unscale;  // 2^-s of the fp16 tap prescale
#pragma unroll
        for (int b = 0; b < L; ++b) r[b] = (kq + b < k_lo || kq + b >= k_hi) ? 0.f : acc[b] * f16_unscale;
    } else if constexpr (MFMA) {
        // ---- stage 1 on the matrix cores (kModeMfma, apt_kernels_fused_launch.hpp): the SPLIT frame — two sub-tiles of NS
        // windows through the same LDS — with each sub-tile as two bf16 planes (x = x0 + x1 + a remainder below 2^-16 |x|:
        // none for 16-bit samples) and, per group of 16 windows, R[branch][window] = sum over K of H[branch][k] X[k][window]
        // as five chains of v_mfma_f32_16x16x32_bf16 — h2 x0, h1 x1, h1 x0, h0 x1, h0 x0 with h = h0 + h1 + h2 exactly
        // (three 8-bit pieces of the 24-bit tap) — accumulated in f32.  bf16 has f32's exponent: no scaling anywhere.
        static_assert(Gm::SPLIT && L <= 16 && M % 2 == 0 && Gm::XSHIFT % 2 == 0, "windows start on sample pairs");
        constexpr int NS = Gm::NS, NKS = Gm::KPAD / 32, PLANE = Gm::XT_PAD / 2;  // (dwords per plane)
        constexpr int NGW = NS / 16 / (NTHR / 64);                                // window groups per wave and sub-tile
        static_assert(NGW * 16 * (NTHR / 64) == NS, "whole groups per wave");
        uint32_t *l32 = reinterpret_cast<uint32_t *>(lds);
        uint32_t *nf_flag = l32 + 2 * PLANE;  // a non-finite result somewhere in the tile (see below)
        const int lane = tid & 63, wave = tid >> 6;
        typedef uint32_t u4m __attribute__((ext_vector_type(4)));
        typedef __bf16 b8m __attribute__((ext_vector_type(8)));
        auto as_b8 = [](u4m v) -> b8m { return __builtin_bit_cast(b8m, v); };
        // H's fragments: [3 pieces][NKS][64 lanes] (fused_mfma_table), read per K step (L1-resident: 6 KB)
        typedef const u4m __attribute__((address_space(1))) *gu4_ptr;
        const gu4_ptr atab = (gu4_ptr)(prm->hs) + lane;
        if (tid == 0) *nf_flag = 0u;
        // registers -> the two planes: x0 = the upper half of the f32 pattern, x1 = the upper half of x - x0 (exact
        // subtraction); PCM16: `*x as f32` (wav.rs:37) first
        auto planes_to_lds = [&](const XReg (&xv)[NXR]) {
#pragma unroll
            for (int e = 0; e < NXR; ++e) {
                const int q = (tid + e * kFusedThreads) * 4;
                if (q < Gm::XT_PAD) {
                    float v[4];
                    if constexpr (sizeof(XT) == 4) {
                        v[0] = xv[e].x, v[1] = xv[e].y, v[2] = xv[e].z, v[3] = xv[e].w;
                    } else {
                        v[0] = static_cast<float>(static_cast<int16_t>(xv[e].x & 0xFFFFu)), v[1] = static_cast<float>(static_cast<int32_t>(xv[e].x) >> 16);
                        v[2] = static_cast<float>(static_cast<int16_t>(xv[e].y & 0xFFFFu)), v[3] = static_cast<float>(static_cast<int32_t>(xv[e].y) >> 16);
                    }
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = v[i] - __uint_as_float(__float_as_uint(v[i]) & 0xFFFF0000u);
                    const u2 w0 = (u2){__builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u),
                                       __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u)};
                    const u2 w1 = (u2){__builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u),
                                       __builtin_amdgcn_perm(__float_as_uint(r[3]), __float_as_uint(r[2]), 0x07060302u)};
                    *reinterpret_cast<u2 *>(l32 + q / 2) = w0;
                    *reinterpret_cast<u2 *>(l32 + PLANE + q / 2) = w1;
                }
            }
        };
        f4 racc[2][NGW];
        auto mfma_sub = [&](auto subc) {
            constexpr int SUB = decltype(subc)::value;
            int base[NGW];
#pragma unroll
            for (int g = 0; g < NGW; ++g) {
                racc[SUB][g] = (f4){0.f, 0.f, 0.f, 0.f};
                const int wl = (wave * NGW + g) * 16 + (lane & 15);
                base[g] = wl * (M / 2) + Gm::XSHIFT / 2 + 4 * (lane >> 4);
            }
            // (H's fragments one K step ahead of the products that use them)
            u4m an0 = atab[0], an1 = atab[NKS * 64], an2 = atab[2 * NKS * 64];
#pragma unroll
            for (int s_ = 0; s_ < NKS; ++s_) {
                const u4m a0 = an0, a1 = an1, a2 = an2;
                if (s_ + 1 < NKS) {
                    an0 = atab[(s_ + 1) * 64];
                    an1 = atab[(NKS + s_ + 1) * 64];
                    an2 = atab[(2 * NKS + s_ + 1) * 64];
                }
                u4m b0[NGW], b1[NGW];
#pragma unroll
                for (int g = 0; g < NGW; ++g) {
                    const uint32_t *bp = l32 + base[g] + 16 * s_;
                    b0[g] = (u4m){bp[0], bp[1], bp[2], bp[3]};
                    b1[g] = (u4m){bp[PLANE], bp[PLANE + 1], bp[PLANE + 2], bp[PLANE + 3]};
                }
                // (the small terms first; the groups alternate so that no MFMA waits for the one before it)
#pragma unroll
                for (int g = 0; g < NGW; ++g) racc[SUB][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_b8(a2), as_b8(b0[g]), racc[SUB][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NGW; ++g) racc[SUB][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_b8(a1), as_b8(b1[g]), racc[SUB][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NGW; ++g) racc[SUB][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_b8(a1), as_b8(b0[g]), racc[SUB][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NGW; ++g) racc[SUB][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_b8(a0), as_b8(b1[g]), racc[SUB][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NGW; ++g) racc[SUB][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_b8(a0), as_b8(b0[g]), racc[SUB][g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // sub-tile 0 is in registers; sub-tile 1's loads are in flight under sub-tile 0's products
        XReg xr1[NXR];
        planes_to_lds(xr);
        __builtin_amdgcn_sched_barrier(0);  // (sub-tile 1's loads behind sub-tile 0's conversion: 28 registers fewer)
        load_tile(ri, tile, 1, xr1);
        __syncthreads();
        mfma_sub(std::integral_constant<int, 0>{});
        __syncthreads();  // everyone is done reading sub-tile 0
        planes_to_lds(xr1);
        __syncthreads();
        mfma_sub(std::integral_constant<int, 1>{});
        APT_MARK("END stage1");
        // A NaN or an infinity among the samples spreads over whole groups of the matrix product (0 x inf): the sum of a
        // lane's results is then not finite, and the tile is evaluated again sample by sample below.  (An overflowing sum
        // of finite results takes that path too: the same values, slowly.)
        {
            float sum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int g = 0; g < NGW; ++g) sum += (racc[sub][g].x + racc[sub][g].y) + (racc[sub][g].z + racc[sub][g].w);
            if ((__float_as_uint(sum) & 0x7F800000u) == 0x7F800000u) *nf_flag = 1u;
        }
        __syncthreads();  // everyone is done reading sub-tile 1: R may land on it
        const bool tile_nf = *nf_flag != 0u;
        {
            // D: column = lane & 15 (window), row = 4 (lane >> 4) + register (branch)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int g = 0; g < NGW; ++g) {
                    const int wdw = sub * NS + (wave * NGW + g) * 16 + (lane & 15);
#pragma unroll
                    for (int rr_ = 0; rr_ < 4; ++rr_) {
                        const int b = 4 * (lane >> 4) + rr_;
                        if (b < L) P[wdw * L + b] = racc[sub][g][rr_];
                    }
                }
        }
        if (tile_nf) {
            // (wave-uniform; never on recordings) kModeFast's chain of fused multiply-adds, from HBM
            __syncthreads();
            const XT *__restrict__ xg = static_cast<const XT *>(call.rec[ri].x);
            const uint64_t n_in = call.rec[ri].n;
            const float *__restrict__ cf = prm->coeff;
            const uint32_t t1r = prm->t1;
            for (int b = 0; b < L; ++b) {
                const int64_t k = (tile * Gm::OWN_K - Gm::PRE_K) + kq + b;
                float sum = 0.f;
                if (k >= 0) {
                    const uint64_t v = static_cast<uint64_t>(k) * static_cast<uint64_t>(M);
                    uint64_t x0 = (v + L - 1) / L;
                    for (uint64_t j = x0 * L - v; j < t1r; j += L, ++x0)
                        if (x0 < n_in) sum = __builtin_fmaf(cf[j], static_cast<float>(xg[x0]), sum);
                }
                P[tid * L + b] = sum;
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < L; ++b) r[b] = P[tid * L + b];
        if (!interior) {
            bool any = false;
#pragma unroll
            for (int b = 0; b < L; ++b)
                if (kq + b < k_lo || kq + b >= k_hi) {
                    r[b] = 0.f;
                    any = true;
                }
            if (any) {
#pragma unroll
                for (int b = 0; b < L; ++b) P[tid * L + b] = r[b];
            }
            __syncthreads();
        }
    } else if constexpr (Gm::SPLIT) {
        // ---- SPLIT stage 1 (apt_kernels_fused_launch.hpp): two sub-tiles of NS windows through the same LDS; in each,
        // the threads of half h = tid / NS compute the branches [B0, B0 + NBR) of window wl = tid % NS.  Same
        // software pipeline as the unsplit form below — a chunk's taps are one contiguous run of the half's table,
        // fetched by scalar loads written as assembly into pinned tuples, waited for one chunk later — with chunks of
        // four window samples = one 16-byte LDS read (PCM16: 8-byte), 24 + 4 tap dwords.
        static_assert(L == 13 && Gm::XSHIFT % 2 == 0 && (M % 2 == 0 || sizeof(XT) == 4),
                      "SPLIT: halves of 7 and 6 branches (three pairs each); window reads of aligned sample pairs, or — odd M, f32 input only — of single samples");
        constexpr bool WIDE = M % 4 == 0 && Gm::XSHIFT % 4 == 0;  // a chunk's four samples are one 16-byte LDS read (else two of 8)
        constexpr int NS = Gm::NS;
        const int wl = tid & (NS - 1);
        constexpr int NH0 = fused_split_nbr(L, 0);   // 7: results a thread holds per sub-tile (half 1: 6)
        float rh[2][NH0];
        if constexpr (PAD) {
#pragma unroll
            for (int j = 0; j < NH0; ++j) rh[0][j] = rh[1][j] = 0.f;
        }
        auto half = [&](auto hc, auto subc) {
            constexpr int H = decltype(hc)::value, SUB = decltype(subc)::value;
            constexpr int B0 = fused_split_b0(L, H), NBR = fused_split_nbr(L, H);
            constexpr int NPH = NBR / 2;
            constexpr bool ODDH = (NBR & 1) != 0;
            static_assert(NPH == 3, "three branch pairs per half");
            constexpr int W0 = fused_split_w0(L, M, H), NCH = fused_split_nch(L, M, T1, H);
            constexpr int CHW = kSplitChunkDwords;
            const cf2_ptr hsp = hs + fused_split_table_offset(L, M, T1, H) / 2;
            typedef uint32_t u16s __attribute__((ext_vector_type(16)));
            typedef uint32_t u8s __attribute__((ext_vector_type(8)));
            typedef uint32_t u4s __attribute__((ext_vector_type(4)));
            u16s ta[2];
            u8s tb[2];
            u4s tc[2];
            typedef float f4w __attribute__((ext_vector_type(4)));
            using XRaw = std::conditional_t<sizeof(XT) == 4, f4w, u2>;  // four window samples as they lie in LDS
            XRaw xraw[2];
            f2 acc[NPH];
            float accl = 0.f;
#pragma unroll
            for (int k = 0; k < NPH; ++k) acc[k] = (f2){0.f, 0.f};
            auto issue = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // (clobbers, not outputs: see the unsplit form)
                if constexpr ((c & 1) == 0)
                    asm volatile("s_load_dwordx16 s[36:51], %0, %1\n\ts_load_dwordx8 s[52:59], %0, %2\n\ts_load_dwordx4 s[60:63], %0, %3"
                                 :: "s"(hsp), "n"(c * CHW * 4), "n"(c * CHW * 4 + 64), "n"(c * CHW * 4 + 96)
                                 : APT_S16(36), APT_S8(52), APT_S4_60);
                else
                    asm volatile("s_load_dwordx16 s[64:79], %0, %1\n\ts_load_dwordx8 s[80:87], %0, %2\n\ts_load_dwordx4 s[88:91], %0, %3"
                                 :: "s"(hsp), "n"(c * CHW * 4), "n"(c * CHW * 4 + 64), "n"(c * CHW * 4 + 96)
                                 : APT_S16(64), APT_S8(80), APT_S4_88);
                constexpr int q0 = Gm::XSHIFT + W0 + kSplitChunk * c;
                if constexpr (sizeof(XT) == 4) {
                    if constexpr (WIDE) {
                        xraw[c & 1] = *reinterpret_cast<const f4w *>(P + wl * M + q0);
                    } else if constexpr (M % 2 != 0) {
                        // (odd lane stride — 96 kHz at the fast profile, M = 75: windows start at any word; 4-byte reads,
                        // conflict-free at an odd stride)
                        const float *wp = P + wl * M + q0;
                        xraw[c & 1] = (f4w){wp[0], wp[1], wp[2], wp[3]};
                    } else {
                        // (lane stride M = 50 words: conflict-free as 8-byte reads, see the unsplit form)
                        const f2 lo = *reinterpret_cast<const f2 *>(P + wl * M + q0), hi = *reinterpret_cast<const f2 *>(P + wl * M + q0 + 2);
                        xraw[c & 1] = (f4w){lo.x, lo.y, hi.x, hi.y};
                    }
                } else if constexpr (WIDE) {
                    xraw[c & 1] = *reinterpret_cast<const u2 *>(reinterpret_cast<const int16_t *>(lds) + wl * M + q0);
                } else {
                    // (four PCM16 samples at an even sample index: two dwords, 4-byte aligned)
                    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(lds) + (wl * M + q0) / 2;
                    xraw[c & 1] = (u2){w32[0], w32[1]};
                }
            };
            auto wait_taps = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                u16s &ra = ta[c & 1];
                u8s &rb = tb[c & 1];
                u4s &rc = tc[c & 1];
                XRaw &xr_ = xraw[c & 1];
                if constexpr ((c & 1) == 0)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "={s[36:51]}"(ra), "={s[52:59]}"(rb), "={s[60:63]}"(rc), "+v"(xr_));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "={s[64:79]}"(ra), "={s[80:87]}"(rb), "={s[88:91]}"(rc), "+v"(xr_));
            };
            auto tapd = [&](auto cc, auto ii) -> float {
                constexpr int buf = decltype(cc)::value & 1, i = decltype(ii)::value;
                if constexpr (i < 16) return __uint_as_float(ta[buf][i]);
                else if constexpr (i < 24) return __uint_as_float(tb[buf][i - 16]);
                else return __uint_as_float(tc[buf][i - 24]);
            };
            issue(std::integral_constant<int, 0>{});
            static_for<0, NCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                wait_taps(cc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (c + 1 < NCH) issue(std::integral_constant<int, c + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                f4w xv;
                if constexpr (sizeof(XT) == 4) {
                    xv = xraw[c & 1];
                } else {
                    // `*x as f32` (wav.rs:37) of the four PCM16 samples
                    const u2 u = xraw[c & 1];
                    xv = (f4w){static_cast<float>(static_cast<int16_t>(u.x & 0xFFFFu)), static_cast<float>(static_cast<int32_t>(u.x) >> 16),
                               static_cast<float>(static_cast<int16_t>(u.y & 0xFFFFu)), static_cast<float>(static_cast<int32_t>(u.y) >> 16)};
                }
                static_for<0, kSplitChunk>([&](auto ee) {
                    constexpr int e = decltype(ee)::value;
                    constexpr int q = W0 + kSplitChunk * c + e;
                    const float xq = xv[e];
                    // The sample, broadcast over both lanes of a packed instruction, is an op_sel on the aligned register
                    // pair it lies in.  The compiler does that for three of a chunk's four samples and builds a pair with
                    // a v_mov_b32 for the fourth (50 issue slots per tile): the packed multiplies / fmas that take a
                    // broadcast sample are written out.
                    const f2 xpair_e = (e < 2) ? __builtin_shufflevector(xv, xv, 0, 1) : __builtin_shufflevector(xv, xv, 2, 3);
                    auto mul_bcast = [&](f2 t) -> f2 {
                        f2 p;
                        if constexpr ((e & 1) == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(xpair_e), "s"(t));
                        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(p) : "v"(xpair_e), "s"(t));
                        return p;
                    };
                    auto fma_bcast = [&](f2 t, f2 a) -> f2 {
                        if constexpr ((e & 1) == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(xpair_e), "s"(t));
                        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a) : "v"(xpair_e), "s"(t));
                        return a;
                    };
                    f2 pr[NPH];
                    // all products of the sample first, then the dependent adds
                    static_for<0, NPH>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        constexpr bool va = branch_uses<L, M, T1>(B0 + 2 * k, q), vb = branch_uses<L, M, T1>(B0 + 2 * k + 1, q);
                        const f2 t = (f2){tapd(cc, std::integral_constant<int, 6 * e + 2 * k>{}), tapd(cc, std::integral_constant<int, 6 * e + 2 * k + 1>{})};
                        pr[k] = (f2){0.f, 0.f};
                        if constexpr (FAST) {
                            if constexpr (va && vb) acc[k] = fma_bcast(t, acc[k]);
                            else if constexpr (va) acc[k].x = __builtin_fmaf(t.x, xq, acc[k].x);
                            else if constexpr (vb) acc[k].y = __builtin_fmaf(t.y, xq, acc[k].y);
                        } else {
                            if constexpr (va && vb) pr[k] = mul_bcast(t);
                            else if constexpr (va) pr[k].x = t.x * xq;
                            else if constexpr (vb) pr[k].y = t.y * xq;
                        }
                    });
                    if constexpr (!FAST) {
                        static_for<0, NPH>([&](auto kk) {
                            constexpr int k = decltype(kk)::value;
                            constexpr bool va = branch_uses<L, M, T1>(B0 + 2 * k, q), vb = branch_uses<L, M, T1>(B0 + 2 * k + 1, q);
                            if constexpr (va && vb) acc[k] = acc[k] + pr[k];
                            else if constexpr (va) acc[k].x = acc[k].x + pr[k].x;
                            else if constexpr (vb) acc[k].y = acc[k].y + pr[k].y;
                        });
                    }
                });
                if constexpr (ODDH) {
                    // the half's last branch: the products of an aligned sample pair in one packed multiply, the additions
                    // one after the other in tap order
                    static_for<0, kSplitChunk / 2>([&](auto pp) {
                        constexpr int e = 2 * decltype(pp)::value;
                        constexpr int q = W0 + kSplitChunk * c + e;
                        constexpr bool u0 = branch_uses<L, M, T1>(B0 + NBR - 1, q), u1 = branch_uses<L, M, T1>(B0 + NBR - 1, q + 1);
                        const f2 t = (f2){tapd(cc, std::integral_constant<int, 24 + e>{}), tapd(cc, std::integral_constant<int, 25 + e>{})};
                        const f2 xp = (f2){xv[e], xv[e + 1]};
                        if constexpr (FAST) {
                            if constexpr (u0) accl = __builtin_fmaf(t.x, xp.x, accl);
                            if constexpr (u1) accl = __builtin_fmaf(t.y, xp.y, accl);
                        } else if constexpr (u0 && u1) {
                            const f2 po = t * xp;
                            accl = accl + po.x;
                            accl = accl + po.y;
                        } else if constexpr (u0) {
                            accl = accl + t.x * xp.x;
                        } else if constexpr (u1) {
                            accl = accl + t.y * xp.y;
                        }
                    });
                }
                // (pins the accumulators: see the unsplit form)
                static_for<0, NPH>([&](auto kk) {
                    f2 &a = acc[decltype(kk)::value];
                    asm volatile("" : "+v"(a));
                });
                asm volatile("" : "+v"(accl));
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int k = 0; k < NPH; ++k) {
                rh[SUB][2 * k] = acc[k].x;
                rh[SUB][2 * k + 1] = acc[k].y;
            }
            if constexpr (ODDH) rh[SUB][NBR - 1] = accl;
        };
        // Which thread half takes which branch half alternates with the tile: half 0 is 7 branches, one of them unpaired
        // (24 % more instructions than half 1's three pairs), and a workgroup's waves 0, 1 / 2, 3 land on the same
        // SIMDs in every workgroup — without the swap two SIMDs of a CU would carry the heavy half of every tile.
        const bool first_half = (tid < NS) != ((tile & 1) != 0);
        // kModeStrictPad: "this tile's results are not all finite" (one word behind the x tile; re-armed per tile, two
        // barriers before it can be set)
        [[maybe_unused]] uint32_t *pad_flag = reinterpret_cast<uint32_t *>(lds) + (Gm::XT_LDS - 4);
        if constexpr (PAD && sizeof(XT) == 4) {
            if (tid == 0) *pad_flag = 0u;
        }
        // kModeStrictPad2: "this tile's F values are not all finite" (a word behind everything else in LDS; set after
        // stage 3, read behind the barrier that follows it)
        if constexpr (PADLP && sizeof(XT) == 4) {
            if (tid == 0) reinterpret_cast<uint32_t *>(lds)[Gm::LP_FLAG_OFF] = 0u;
        }
        // sub-tile 0 is in LDS; sub-tile 1's loads are in flight under its stage 1
        XReg xr1[NXR];
        load_tile(ri, tile, 1, xr1);
        if (first_half) half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        else half(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        __syncthreads();  // everyone is done reading sub-tile 0
        tile_to_lds(xr1);
        __syncthreads();
        if (first_half) half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        else half(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        APT_MARK("END stage1");
        if constexpr (PAD && sizeof(XT) == 4) {
            // 0 * inf / 0 * NaN behind the filter's last tap (the reference skips those taps): any non-finite result
            // sends the tile to the sample-by-sample evaluation below.  (A sum of finite results that overflows does too:
            // the same values, slowly.)
            float chk = 0.f;
#pragma unroll
            for (int j = 0; j < fused_split_nbr(L, 0); ++j) chk = chk + (rh[0][j] + rh[1][j]);  // (half 1: its unused seventh slot is 0-initialised below)
            if ((__float_as_uint(chk) & 0x7F800000u) == 0x7F800000u) *pad_flag = 1u;
        }
        __syncthreads();  // everyone is done reading sub-tile 1: R may land on it
        if (first_half) {
#pragma unroll
            for (int j = 0; j < fused_split_nbr(L, 0); ++j) {
                P[wl * L + j] = rh[0][j];
                P[(NS + wl) * L + j] = rh[1][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < fused_split_nbr(L, 1); ++j) {
                P[wl * L + fused_split_b0(L, 1) + j] = rh[0][j];
                P[(NS + wl) * L + fused_split_b0(L, 1) + j] = rh[1][j];
            }
        }
        if constexpr (PAD && sizeof(XT) == 4) {
            // (the flag lies behind the x tile: R, which has just landed on the tile, does not reach it)
            static_assert(Gm::XT_LDS - 4 >= Gm::TILE_K, "the flag lies behind R");
            if (*pad_flag != 0u) {
                // (workgroup-uniform; never on recordings) the reference's loop, sample by sample from HBM (dsp.rs:252-263)
                __syncthreads();
                const XT *__restrict__ xg = static_cast<const XT *>(call.rec[ri].x);
                const uint64_t n_in = call.rec[ri].n;
                const float *__restrict__ cf = prm->coeff;
                const uint32_t t1r = prm->t1;
                for (int b = 0; b < L; ++b) {
                    const int64_t k = (tile * Gm::OWN_K - Gm::PRE_K) + kq + b;
                    float sum = 0.f;
                    if (k >= 0) {
                        const uint64_t v = static_cast<uint64_t>(k) * static_cast<uint64_t>(M);
                        uint64_t x0 = (v + L - 1) / L;
                        for (uint64_t j = x0 * L - v; j < t1r; j += L, ++x0)
                            if (x0 < n_in) sum = sum + cf[j] * static_cast<float>(xg[x0]);
                    }
                    P[tid * L + b] = sum;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < L; ++b) r[b] = P[tid * L + b];
        if (!interior) {
            bool any = false;
#pragma unroll
            for (int b = 0; b < L; ++b)
                if (kq + b < k_lo || kq + b >= k_hi) {
                    r[b] = 0.f;
                    any = true;
                }
            // (the stages behind read R from LDS too: the envelope's predecessor pairs)
            if (any) {
#pragma unroll
                for (int b = 0; b < L; ++b) P[tid * L + b] = r[b];
            }
            __syncthreads();
        }
    } else {
        // (Rounds 1-3 ran an unsplit form here — every thread all 13 branches of its own window, 256 windows per
        // 51.5 KB tile at 48 kHz, 128 at 96 kHz, two- / three-sample chunks — until the SPLIT form above made the
        // tile a quarter of that per thread: git history)
        static_assert(Gm::SPLIT || F16, "the specialised kernels are 256-thread workgroups");
    }
    APT_MARK("BEGIN r_store");
    if constexpr (!Gm::SPLIT) {  // (SPLIT: R went through LDS already)
    __syncthreads();  // everyone is done reading the x tile
#pragma unroll
    for (int b = 0; b < L; ++b) P[tid * L + b] = r[b];
    __syncthreads();
    }
    after_tile_in_lds();  // (persistent form: the NEXT tile's loads, in flight under stages 2 to 4)
    }  // !TABLE

    if constexpr (APT_FUSED_STOP == 2) return;
    // the parameters of the later stages, fetched now (the empty asm keeps the loads from being hoisted)
    typedef const FusedParams APT_CONST_AS *cprm_ptr;
    cprm_ptr late = (cprm_ptr)(prm);
    asm volatile("" : "+s"(late));
    const float cosphi2 = late->cosphi2, sinphi = late->sinphi;
    const float inv_sinphi = late->inv_sinphi;  // strict: verified RN(1/sinphi) or 0; fast: RN(1/sinphi)
    const cfloat_ptr h2 = (cfloat_ptr)(late->h2);   // [T2]
    const cf2_ptr h2p = (cf2_ptr)(late->h2p);       // [T2+1] (h2[m-1], h2[m])
    typedef const SlotPtrs APT_CONST_AS *cslot_ptr;
    const cslot_ptr slots = (cslot_ptr)(late->slots);
    const int want_gm = late->want_gm;
    const float gm_slack = late->gm_slack;  // (now: the pointer pair otherwise stays live, and is spilled, across stage 3)
    APT_MARK("END r_store");

    // ---- stage 2: AM envelope from consecutive samples (dsp.rs:369-377)
    auto envelope = [&](auto interior_tag) {
    constexpr bool INT = decltype(interior_tag)::value;  // interior tile: no edge tests
    if constexpr (FAST) {
        // native v_sqrt_f32 (1 ulp) and a multiplication by RN(1/sin(phi)): ~2 ulp from the
        // correctly rounded value, no range checks (the radicand is >= (1-|cos phi|)(p^2+c^2) >= 0)
        float prev = (tid > 0) ? P[tid * L - 1] : 0.f;
        if constexpr (INT) {
            // two values per instruction, as in the strict path below
            const f2 cos2 = (f2){cosphi2, cosphi2}, inv2 = (f2){inv_sinphi, inv_sinphi};
#pragma unroll
            for (int p2 = 0; p2 < (L + 1) / 2; ++p2) {
                const f2 A = (f2){r[2 * p2], (2 * p2 + 1 < L) ? r[2 * p2 + 1] : r[2 * p2]};
                const f2 S = p2 == 0 ? (f2){prev, r[0]} : (f2){P[tid * L + 2 * p2 - 1], P[tid * L + 2 * p2]};  // (see strict)
                const f2 rad = __builtin_elementwise_fma(-(S * A), cos2, (S * S) + (A * A));
                const f2 q = (f2){__builtin_amdgcn_sqrtf(rad.x), __builtin_amdgcn_sqrtf(rad.y)} * inv2;
                Q[tid * L + 2 * p2] = q.x;
                if (2 * p2 + 1 < L) Q[tid * L + 2 * p2 + 1] = q.y;
            }
        } else {
        float prev_sq = prev * prev;
#pragma unroll
        for (int b = 0; b < L; ++b) {
            const float curr = r[b];
            const float curr_sq = curr * curr;
            const float rad = __builtin_fmaf(-(prev * curr), cosphi2, prev_sq + curr_sq);
            Q[tid * L + b] = (kq + b > k_lo) ? __builtin_amdgcn_sqrtf(rad) * inv_sinphi : 0.f;
            prev = curr;
            prev_sq = curr_sq;
        }
        }
    } else {
        // (the constants in vector registers: an instruction with a scalar operand issues at half the rate of one
        // without, tools/ubench/rates3.hip)
        float cosv = cosphi2, sinv = sinphi, invv = inv_sinphi;
        asm volatile("" : "+v"(cosv), "+v"(sinv), "+v"(invv));
        float prev = (tid > 0) ? P[tid * L - 1] : 0.f;
        float xr[L];
        bool in_range = inv_sinphi != 0.f;  // 0: the fast root / divide did not verify on this device / for this sin(phi)
        if constexpr (INT) {
            // Interior tile, two values per instruction: A[p] = (r[2p], r[2p+1]) are the stage-1 accumulator pairs,
            // S[p] = (r[2p-1], r[2p]) their predecessors; radicand = (prev^2 + curr^2) - (prev*curr)*cos2 in the
            // reference's order (envelope_radicand), element by element.
            APT_MARK("BEGIN envelope_radicands");
            constexpr int NPE = (L + 1) / 2;
            f2 rad[NPE];
            uint32_t umin = 0xFFFFFFFFu, umax = 0u;
            const f2 cos2 = (f2){cosv, cosv};
#pragma unroll
            for (int p2 = 0; p2 < NPE; ++p2) {
                const f2 A = (f2){r[2 * p2], (2 * p2 + 1 < L) ? r[2 * p2 + 1] : r[2 * p2]};
                // (the predecessor pairs straddle the accumulator pairs: read back from R in LDS — one two-address
                // read each, no issue slot — where building them from the accumulators takes two v_mov_b32 apiece)
                const f2 S = p2 == 0 ? (f2){prev, r[0]} : (f2){P[tid * L + 2 * p2 - 1], P[tid * L + 2 * p2]};
                const f2 ss = (S * S) + (A * A);
                const f2 cc = (S * A) * cos2;
                rad[p2] = ss - cc;
                // every value of the thread inside [2^-96, 2^100]: as bit patterns, minimum and maximum (a NaN or a
                // negative radicand has a pattern above the range's end)
                const uint32_t u0 = __float_as_uint(rad[p2].x), u1 = __float_as_uint(rad[p2].y);
                umin = u0 < umin ? u0 : umin;
                umin = u1 < umin ? u1 : umin;
                umax = u0 > umax ? u0 : umax;
                umax = u1 > umax ? u1 : umax;
            }
            in_range = in_range && umin >= 0x0F800000u && umax <= 0x71800000u;
            APT_MARK("END envelope_radicands");
            // wave-uniform choice: the exactly rounded fast path (apt_envelope.hpp) when every value
            // of the wave is in its range, the compiler's general sequences otherwise
            if (__all(in_range)) {
                APT_MARK("BEGIN envelope_roots");
                const f2 sin2 = (f2){sinv, sinv}, inv2 = (f2){invv, invv}, half2 = (f2){0.5f, 0.5f};
#pragma unroll
                for (int p2 = 0; p2 < NPE; ++p2) {
                    // exact_sqrt_inrange + fast_divide (apt_envelope.hpp), both lanes of a pair at once
                    const f2 x = rad[p2];
                    const f2 y = (f2){__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
                    const f2 g = x * y;
                    const f2 h = half2 * y;
                    const f2 d = __builtin_elementwise_fma(-g, g, x);
                    const f2 root = __builtin_elementwise_fma(d, h, g);
                    const f2 q0 = root * inv2;
                    const f2 rem = __builtin_elementwise_fma(-sin2, q0, root);
                    const f2 q = __builtin_elementwise_fma(rem, inv2, q0);
                    Q[tid * L + 2 * p2] = q.x;
                    if (2 * p2 + 1 < L) Q[tid * L + 2 * p2 + 1] = q.y;
                }
                APT_MARK("END envelope_roots");
            } else {
#pragma unroll
                for (int b = 0; b < L; ++b) {
                    const float x = (b & 1) ? rad[b / 2].y : rad[b / 2].x;
                    Q[tid * L + b] = (kq + b > k_lo) ? envelope_general(x, sinphi) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < L; ++b) {
                const float curr = r[b];
                xr[b] = envelope_radicand(prev, curr, cosv);
                // (outputs outside the recording are zeroed below whatever their radicand is)
                in_range = in_range && (envelope_in_range(xr[b]) || kq + b <= k_lo || kq + b >= k_hi);
                prev = curr;
            }
            if (__all(in_range)) {
#pragma unroll
                for (int b = 0; b < L; ++b) {
                    // (positions at or past the end of the recording hold garbage: never read)
                    Q[tid * L + b] = (kq + b > k_lo) ? envelope_fast(xr[b], sinv, invv) : 0.f;
                }
            } else {
#pragma unroll
                for (int b = 0; b < L; ++b) Q[tid * L + b] = (kq + b > k_lo) ? envelope_general(xr[b], sinphi) : 0.f;
            }
        }
    }
    };
    if (interior) envelope(std::true_type{});
    else envelope(std::false_type{});
    __syncthreads();
    if constexpr (APT_FUSED_STOP == 3) return;
    // ---- stage 3: causal low-pass with the `i > j` guard (dsp.rs:396-404)
    // Same sample-stationary pairing: envelope sample d = D[kt-(T2-1)+qq] meets output b at
    // tap j = (T2-1)+b-qq, so outputs (b, b+1) take the tap pair (h2[m], h2[m+1]); walking qq
    // downwards gives every output its taps in ascending j.
    float f[L];
    {
        const int base = tid * L - (T2 - 1);
        if (kt >= T2) {
            // (the window's first word as one opaque register: the reads below then differ by immediate offsets
            // only, where the compiler otherwise rebuilds "lane base + D_OFF + constant" with a v_add_u32 per read)
            APT_MARK("BEGIN stage3");
            int qofs = Gm::D_OFF + base;
            asm volatile("" : "+v"(qofs));
            f2 fa[Gm::NP > 0 ? Gm::NP : 1];
            float fl = 0.f;
#pragma unroll
            for (int pp = 0; pp < Gm::NP; ++pp) fa[pp] = (f2){0.f, 0.f};
            constexpr int CH3 = 8;
            static_for<0, (Gm::DW + CH3 - 1) / CH3>([&](auto cc) {
                constexpr int hi = Gm::DW - 1 - decltype(cc)::value * CH3;  // walk qq downwards
                float dv[CH3];
#pragma unroll
                for (int e = 0; e < CH3; ++e) dv[e] = (hi - e >= 0) ? lds[qofs + hi - e] : 0.f;
                // The odd output (L - 1) has no partner: its products two samples at a time — one packed multiply of the
                // tap pair (h2[ml], h2[ml + 1]) with the sample pair (d[qq], d[qq - 1]) — and the additions one after the
                // other in tap order (strict modes; -18 instructions per thread at T2 = 37).
                [[maybe_unused]] f2 plp[(CH3 + 1) / 2];
                if constexpr ((L & 1) && !FAST) {
                    static_for<0, CH3 / 2>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        constexpr int qa = hi - 2 * k, qb = qa - 1;
                        constexpr int ma = (T2 - 1) + (L - 1) - qa;  // tap of sample qa; sample qb: ma + 1
                        if constexpr (qb >= 0 && ma >= 0 && ma + 1 < T2) plp[k] = h2p[ma + 1] * (f2){dv[2 * k], dv[2 * k + 1]};
                    });
                }
                static_for<0, CH3>([&](auto ee) {
                    constexpr int qq = hi - decltype(ee)::value;
                    if constexpr (qq >= 0) {
                        const float d = dv[decltype(ee)::value];
                        // all products of this sample first, then the dependent adds (a
                        // v_pk_add right behind the v_pk_mul it reads costs a wait state)
                        f2 pr[Gm::NP > 0 ? Gm::NP : 1];
                        float pl = 0.f;
                        static_for<0, Gm::NP>([&](auto pc) {
                            constexpr int pp = decltype(pc)::value;
                            constexpr int m = (T2 - 1) + 2 * pp - qq;       // tap of lane x; lane y: m+1
                            constexpr bool va = m >= 0 && m < T2;
                            constexpr bool vb = m + 1 >= 0 && m + 1 < T2;
                            pr[pp] = (f2){0.f, 0.f};
                            if constexpr (FAST) {
                                if constexpr (va && vb) {
                                    fa[pp] = __builtin_elementwise_fma(h2p[m + 1], (f2){d, d}, fa[pp]);
                                } else if constexpr (va) {
                                    fa[pp].x = __builtin_fmaf(h2[m], d, fa[pp].x);
                                } else if constexpr (vb) {
                                    fa[pp].y = __builtin_fmaf(h2[m + 1], d, fa[pp].y);
                                }
                            } else if constexpr (va && vb) {
                                pr[pp] = h2p[m + 1] * (f2){d, d};
                            } else if constexpr (va) {
                                pr[pp].x = h2[m] * d;
                            } else if constexpr (vb) {
                                pr[pp].y = h2[m + 1] * d;
                            }
                        });
                        if constexpr (L & 1) {
                            constexpr int ml = (T2 - 1) + (L - 1) - qq;
                            if constexpr (ml >= 0 && ml < T2) {
                                // (is this sample half of a packed pair?  its partner is the other sample of (2k, 2k + 1))
                                constexpr int e_ = decltype(ee)::value, k_ = e_ / 2;
                                constexpr int qa_ = hi - 2 * k_, ma_ = (T2 - 1) + (L - 1) - qa_;
                                constexpr bool paired = k_ < CH3 / 2 && qa_ - 1 >= 0 && ma_ >= 0 && ma_ + 1 < T2;
                                if constexpr (FAST) fl = __builtin_fmaf(h2[ml], d, fl);
                                else if constexpr (paired) pl = (e_ & 1) ? plp[k_].y : plp[k_].x;
                                else pl = h2[ml] * d;
                            }
                        }
                        if constexpr (!FAST) {
                        static_for<0, Gm::NP>([&](auto pc) {
                            constexpr int pp = decltype(pc)::value;
                            constexpr int m = (T2 - 1) + 2 * pp - qq;
                            constexpr bool va = m >= 0 && m < T2;
                            constexpr bool vb = m + 1 >= 0 && m + 1 < T2;
                            if constexpr (va && vb) {
                                fa[pp] = fa[pp] + pr[pp];
                            } else if constexpr (va) {
                                fa[pp].x = fa[pp].x + pr[pp].x;
                            } else if constexpr (vb) {
                                fa[pp].y = fa[pp].y + pr[pp].y;
                            }
                        });
                        if constexpr (L & 1) {
                            constexpr int ml = (T2 - 1) + (L - 1) - qq;
                            if constexpr (ml >= 0 && ml < T2) fl = fl + pl;
                        }
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int pp = 0; pp < Gm::NP; ++pp) {
                f[2 * pp] = fa[pp].x;
                f[2 * pp + 1] = fa[pp].y;
            }
            if constexpr (L & 1) f[L - 1] = fl;
            APT_MARK("END stage3");
        } else {
            // first samples of the recording (tile 0 only): the reference's `i > j` guard
#pragma unroll
            for (int b = 0; b < L; ++b) f[b] = 0.f;
#pragma unroll 1
            for (int j = 0; j < T2; ++j) {
                const float hj = h2[j];
#pragma unroll
                for (int b = 0; b < L; ++b) {
                    const int qi = base + (T2 - 1) + b - j;
                    if (kt + b > j && qi >= 0) f[b] = f[b] + Q[qi] * hj;
                }
            }
        }
    }
    APT_MARK("BEGIN f_to_lds");
#pragma unroll
    for (int b = 0; b < L; ++b) P[tid * L + b] = f[b];  // R is dead: P now holds F
    if constexpr (PADLP && sizeof(XT) == 4) {
        // 0 * inf / 0 * NaN behind the low-pass's last tap (the reference stops at it): any non-finite F of a thread whose
        // window lies inside the tile sends the tile to the run-time loop below.  (PCM16 input: the envelope is finite.)
        float chk = 0.f;
#pragma unroll
        for (int b = 0; b < L; ++b) chk = chk + f[b];
        if (tid >= kPreThreads && (__float_as_uint(chk) & 0x7F800000u) == 0x7F800000u)
            reinterpret_cast<uint32_t *>(lds)[lp_flag_off] = 1u;
    }
    __syncthreads();
    if constexpr (PADLP && sizeof(XT) == 4) {
        if (reinterpret_cast<const uint32_t *>(lds)[lp_flag_off] != 0u) {
            // (workgroup-uniform; never on recordings) the reference's loop over the filter's own taps (dsp.rs:396-404),
            // from D, which still lies in Q
            const uint32_t t2r = late->t2;
#pragma unroll
            for (int b = 0; b < L; ++b) f[b] = 0.f;
#pragma unroll 1
            for (uint32_t j = 0; j < t2r; ++j) {
                const float hj = h2[j];
#pragma unroll
                for (int b = 0; b < L; ++b) {
                    const int qi = tid * L + b - static_cast<int>(j);
                    if (kt + b > static_cast<int>(j) && qi >= 0) f[b] = f[b] + Q[qi] * hj;
                }
            }
#pragma unroll
            for (int b = 0; b < L; ++b) P[tid * L + b] = f[b];
            __syncthreads();
        }
    }
    APT_MARK("END f_to_lds");
    if constexpr (APT_FUSED_STOP == 4) return;
    // owned F -> HBM, coalesced 16-byte stores
    uint32_t slot_late = call.rec[ri].slot;
    asm volatile("" : "+s"(slot_late));  // keeps the loads below from being hoisted above stage 1
    float *__restrict__ f_out = slots[slot_late].f;
    if (interior) {
        // every owned sample exists: whole 16-byte stores at a scalar base advanced in scalar registers plus the
        // lane's 32-bit offset (as the tile loads), no per-lane 64-bit address arithmetic
        APT_MARK("BEGIN f_store");
        typedef char __attribute__((address_space(1))) *gchar_wptr;
        typedef f4 __attribute__((address_space(1))) *gfloat4_wptr;
        gchar_wptr sb = (gchar_wptr)(f_out + o0);
        const uint32_t voff = static_cast<uint32_t>(tid) * 16u;
#pragma unroll
        for (int e = 0; e * kFusedThreads * 4 < Gm::OWN_K; ++e) {
            const int q = (tid + e * kFusedThreads) * 4;
            // (PRE_K is odd: the LDS side is four 4-byte-aligned words, read as two two-address reads)
            if (q < Gm::OWN_K) {
                const float *src = P + Gm::PRE_K + q;
                *(gfloat4_wptr)(sb + voff) = (f4){src[0], src[1], src[2], src[3]};
            }
            sb += kFusedThreads * 16;
            asm volatile("" : "+s"(sb));
        }
        APT_MARK("END f_store");
    } else {
        float *ft = f_out + o0;
        for (int q = tid * 4; q < Gm::OWN_K; q += kFusedThreads * 4) {
            if (Gm::PRE_K + q + 3 < k_hi) {
                const float *src = P + Gm::PRE_K + q;
                *reinterpret_cast<float4 *>(ft + q) = make_float4(src[0], src[1], src[2], src[3]);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (Gm::PRE_K + q + e < k_hi) ft[q + e] = P[Gm::PRE_K + q + e];
            }
        }
    }
    if constexpr (APT_FUSED_STOP == 5) return;
    // ---- stage 4: sync cross-correlation (decode.rs:225-233) -> per-group bounds of its maximum.
    // The correlation itself never leaves the CU: k_sync_words evaluates it (apt_sync_corr.hpp) for
    // the few candidate groups the picker has to look at, so all the front end owes the picker is, per
    // group of GS positions, an interval [lo, hi] that holds the group's maximum.  Every mode gets it from
    // pulse sums (22 operations per position instead of the 114 of the reference's chain):
    //   fast    the pulse-sum value IS that mode's correlation: lo = hi = max.
    //   strict  the pulse-sum value a and the reference's sequentially rounded chain c are two
    //           floating-point evaluations of the same sum of 38*PW terms +-F[i+j]; with u = 2^-24,
    //           |c - S| <= gamma(38*PW - 1) * sum|F|  (first term exact) and |a - S| <= gamma(21) * sum|F|
    //           (an F passes through at most 3 + 18 rounded additions), so |a - c| <= slack * A with
    //           slack = 138 u at PW = 3 (fused_gm_slack: 134 u plus 3 % for the bound's own roundings) and
    //           A = sum of |F| over the group's whole window.  No underflow term: floating-point
    //           additions of subnormals are exact.  lo = max - slack*A, hi = max + slack*A; k_sync_words
    //           prunes with lo against hi and settles what the bounds cannot with the exact chain.
    //   a group whose window holds a non-finite F (NaN, +-Inf, or an overflowing sum) — where the two
    //   evaluations may disagree about WHERE the NaNs are — gets [-inf, +inf]: always evaluated, never
    //   pruning anything.
    if (want_gm) {
        GroupMax *__restrict__ gm_out = slots[slot_late].gm;
        constexpr int PUL = 2 * PW;
        static_assert(Gm::GS == 4 * L, "a group is four threads' samples");
        // [NTHR] per-thread sums of |F|: written once F (region P) is dead, behind the partial maxima
        float ab_mine = 0.f;
        {
            APT_MARK("BEGIN pulse_sums");
            // pulse sums of the thread's own L positions -> Q (D is dead).  B[p] = F[p] + ... + F[p + 2 PW - 1] as
            // PW sums of neighbours b2[e] = F[e] + F[e + 1], two positions per instruction: the neighbour sums
            // themselves are packed too — (b2[2p], b2[2p+1]) = (F[2p], F[2p+1]) + (F[2p+1], F[2p+2]) — with the
            // thread's F window read TWICE from LDS, as even- and as odd-aligned pairs (a second two-address read
            // costs no issue slot; through an offset the compiler cannot see through, or it rebuilds the second set
            // of pairs from the first with a v_mov_b32 per value).
            int fofs = tid * L;            // word offsets into lds (P = lds): the window, ...
            int qofs4 = Gm::D_OFF + tid * L;  // ... and where the thread's pulse sums go (Q = lds + D_OFF)
            int fofs_odd = fofs + 1;
            asm volatile("" : "+v"(fofs), "+v"(fofs_odd), "+v"(qofs4));
            constexpr int NFW = L + PUL - 1;           // 18 window values: F[0 .. 17]
            constexpr int NB2 = (NFW + 1) / 2;         // 9 pairs of neighbour sums: b2[0 .. 17] (b2[17] unused garbage)
            float fw[NFW];
            f2 b2p[NB2];
#pragma unroll
            for (int p2 = 0; p2 < NB2; ++p2) {
                const f2 ev = (f2){lds[fofs + 2 * p2], lds[fofs + 2 * p2 + 1]};           // (past the tile: unused garbage)
                const f2 od = (f2){lds[fofs_odd + 2 * p2], lds[fofs_odd + 2 * p2 + 1]};
                fw[2 * p2] = ev.x;
                if (2 * p2 + 1 < NFW) fw[2 * p2 + 1] = ev.y;
                b2p[p2] = ev + od;
            }
            // (the sums of PW pairs two positions at a time: packed)
#pragma unroll
            for (int b = 0; b < L; b += 2) {
                f2 bs = b2p[b / 2] + b2p[b / 2 + 1];
#pragma unroll
                for (int t = 2; t < PW; ++t) bs = bs + b2p[b / 2 + t];
                lds[qofs4 + b] = bs.x;
                if (b + 1 < L) lds[qofs4 + b + 1] = bs.y;
            }
            if constexpr (!FAST) {
                float a = 0.f;
                if (interior) {
#pragma unroll
                    for (int b = 0; b < L; ++b) a = a + __builtin_fabsf(fw[b]);
                } else {
#pragma unroll
                    for (int b = 0; b < L; ++b)
                        a = a + ((kq + b >= k_lo && kq + b < k_hi) ? __builtin_fabsf(fw[b]) : 0.f);
                }
                ab_mine = a;
            }
        }
        __syncthreads();  // pulse sums complete; F (region P) is dead from here on
        if constexpr (!FAST) lds[Gm::AB_OFF + tid] = ab_mine;  // (read after the next barrier, by the threads that write group records)
        APT_MARK("END pulse_sums");
        // ---- the correlation of the owned positions, REMAPPED: thread t = 6 blk + r takes the 13 positions
        // p_j = PRE_K + 78 blk + r + 6 j (one pulse apart), whose 19 pulse sums each are V[j + k] with
        // V[n] = B[p_0 + 6 n], n < 31: 31 LDS reads per thread where 13 CONSECUTIVE positions need a window of 121
        // (each pulse sum now serves up to 13 positions of the thread instead of 2), all of them 4-byte reads at
        // compile-time offsets (a thread's window of consecutive positions starts at 13 t words: unaligned for
        // wider reads, and the ds_read2_b32 pairs the compiler then picks cost one v_add_u32 each for an address
        // beyond their 8-bit offsets).  And with a thread's values one pulse apart, the template's alternating
        // part telescopes inside the thread (apt_sync_corr.hpp: E = differences of neighbouring pulse sums,
        // S2 / S4 = sums of 2 / 4 of them, T2 / T4 the template's tail): 139 additions per thread, not 247.
        // A block of 78 positions is 1.5 groups of 52: the thread reduces its values to two partial maxima,
        // for the group its first positions lie in and for the next one, and the block pair's six partial
        // lists go through LDS (region P) to the thread that writes a group's record.
        constexpr int NB6 = Gm::NBLK4 * PUL;  // threads with a block (the last may be partial)
        static_assert(NB6 <= NTHR, "one thread per block column");
        float *PMX = P;  // COMPACT4: [groups][12] partial maxima by group (6 from each of the two blocks that reach it, or -inf)
        float *CV = P;   // general form: the correlation values themselves, by tile position (F is dead)
        APT_MARK("BEGIN correlation");
        if constexpr (!Gm::COMPACT4) {
            // ---- general form (any pixel width; the fast / slow profiles' 8- and 10-sample pulses): the same evaluation
            // at positions one pulse apart — thread t = PUL blk + r takes p_j = PRE_K + PUL L blk + r + PUL j — and the
            // values go to LDS where the thread that writes a group's record takes the maximum of its 52.
            if (tid < NB6) {
                const uint32_t blk = static_cast<uint32_t>(tid) / static_cast<uint32_t>(PUL);
                const int rr = tid - static_cast<int>(blk) * PUL;
                const int p0 = Gm::PRE_K + static_cast<int>(blk) * (PUL * L) + rr;
                const float *vsrc = Q + p0;
                constexpr int NP2 = (L + 1) / 2;
                f2 VA[NP2 + 9], VS[NP2 + 8], cp[NP2];
#pragma unroll
                for (int p2 = 0; p2 < NP2 + 9; ++p2) VA[p2] = (f2){vsrc[PUL * (2 * p2)], vsrc[PUL * (2 * p2 + 1)]};
                int odd_ofs = PUL;
                asm volatile("" : "+v"(odd_ofs));  // (see the compact form)
                const float *vsrc_odd = vsrc + odd_ofs;
#pragma unroll
                for (int p2 = 0; p2 < NP2 + 8; ++p2) VS[p2] = (f2){vsrc_odd[PUL * (2 * p2)], vsrc_odd[PUL * (2 * p2 + 1)]};
                sync_corr_pulse_stride<NP2>(VA, VS, cp);
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    float cj = (j & 1) ? cp[j / 2].y : cp[j / 2].x;
                    const int pq = p0 + PUL * j;
                    if (!interior) {
                        if (pq == k_lo && !(cj > 0.f)) cj = 0.f;     // the picker starts from the peak (0, 0.)
                        if (pq < k_lo || pq >= c_hi) cj = kNegInfF;  // not a correlation position
                    }
                    if (pq < Gm::PRE_K + Gm::OWN_K) CV[pq] = cj;
                }
            }
        } else
        if (tid < NB6) {
            const uint32_t blk = static_cast<uint32_t>(tid) / 6u;
            const int rr = tid - static_cast<int>(blk) * 6;
            const int p0 = Gm::PRE_K + static_cast<int>(blk) * (6 * L) + rr;  // tile-relative position of j = 0
            const float *vsrc = Q + p0;
            float c[L];
            {
                // V[n] = vsrc[6 n] twice: even-aligned pairs (V[2p], V[2p+1]) and odd-aligned pairs (V[2p+1], V[2p+2]),
                // one two-address LDS read each (apt_sync_corr.hpp: every addition of the evaluation is then packed)
                constexpr int NP2 = (L + 1) / 2;
                f2 VA[NP2 + 9], VS[NP2 + 8], cp[NP2];
#pragma unroll
                for (int p2 = 0; p2 < NP2 + 9; ++p2) VA[p2] = (f2){vsrc[PUL * (2 * p2)], vsrc[PUL * (2 * p2 + 1)]};
                // (through a pointer the compiler cannot identify with vsrc: it would reuse the values already loaded
                // and build these pairs with 28 v_mov — a second LDS read costs no VALU issue slot)
                int odd_ofs = PUL;
                asm volatile("" : "+v"(odd_ofs));  // (the offset, not the pointer: that would lose its address space)
                const float *vsrc_odd = vsrc + odd_ofs;
#pragma unroll
                for (int p2 = 0; p2 < NP2 + 8; ++p2) VS[p2] = (f2){vsrc_odd[PUL * (2 * p2)], vsrc_odd[PUL * (2 * p2 + 1)]};
                sync_corr_pulse_stride<NP2>(VA, VS, cp);
#pragma unroll
                for (int j = 0; j < L; ++j) c[j] = (j & 1) ? cp[j / 2].y : cp[j / 2].x;
            }
            const bool odd = (blk & 1u) != 0u;
            APT_MARK("END correlation");
            if (!interior) {
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    const int pq = p0 + PUL * j;
                    if (pq == k_lo && !(c[j] > 0.f)) c[j] = 0.f;     // the picker starts from the peak (0, 0.)
                    if (pq < k_lo || pq >= c_hi) c[j] = kNegInfF;    // not a correlation position
                }
            }
            // a NaN position is a terminal of the picker whatever the finite maximum of its group is
            // (decode.rs:250): fast mode reports it through the bounds here — any NaN among the thread's values
            // (their sum is one) opens both of its groups; the strict modes see it in the sum of |F| below
            bool any_nan = false;
            if constexpr (FAST) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < L; ++j) sum = sum + ((c[j] == kNegInfF) ? 0.f : c[j]);
                any_nan = sum != sum;
            }
            // positions of the first group: r + 6 j < 52 in an even block (j <= 7, and j = 8 for r < 4), < 26 in an
            // odd one (j <= 3, and j = 4 for r < 2)
            APT_MARK("BEGIN partial_maxima");
            const float m57 = max3_of_sums(c[5], c[6], c[7]);
            const bool a4 = !odd || rr < 2, a8 = !odd && rr < 4;
            float mA = max3_of_sums(c[0], c[1], c[2]);
            mA = max3_of_sums(mA, c[3], a4 ? c[4] : kNegInfF);
            mA = max3_of_sums(mA, odd ? kNegInfF : m57, a8 ? c[8] : kNegInfF);
            float mB = max3_of_sums(c[9], c[10], c[11]);
            mB = max3_of_sums(mB, c[12], a4 ? kNegInfF : c[4]);
            mB = max3_of_sums(mB, odd ? m57 : kNegInfF, a8 ? kNegInfF : c[8]);
            if (any_nan) {
                mA = __builtin_huge_valf();
                mB = __builtin_huge_valf();
            }
            // group of the first part: 3 (blk / 2) + (odd ? 1 : 0); slots 0..5 of a group belong to the even
            // block that reaches it — except that group 3u+1 is reached by the even block's SECOND part (slots
            // 0..5) and the odd block's FIRST (slots 6..11); what no block fills is -inf
            const uint32_t u = blk >> 1;
            float *g0 = PMX + (3u * u + (odd ? 1u : 0u)) * 12u + (odd ? 6u : 0u) + rr;
            g0[0] = mA;
            g0[odd ? 6 : 12] = mB;                    // even: group 3u+1, slot r;  odd: group 3u+2, slot r
            g0[odd ? 12 : 6] = kNegInfF;              // even: group 3u, slot 6+r;  odd: group 3u+2, slot 6+r
        }
        __syncthreads();
        APT_MARK("END partial_maxima");
        // the group's record
        auto group_bounds = [&](auto interior_tag) {
        constexpr bool INT = decltype(interior_tag)::value;  // interior tile: no edge tests
        if constexpr (INT) APT_MARK("BEGIN group_record");
        if (((tid - kPreThreads) & 3) == 0 && tid >= kPreThreads && tid < kPreThreads + kOwnThreads && (INT || kq < c_hi)) {
            const int gl = (tid - kPreThreads) / 4;
            float mx;
            bool open;
            if constexpr (Gm::COMPACT4) {
                typedef float f4g __attribute__((ext_vector_type(4)));
                const f4g *pm = reinterpret_cast<const f4g *>(PMX + gl * 12);
                const f4g q0 = pm[0], q1 = pm[1], q2 = pm[2];
                mx = max3_of_sums(q0.x, q0.y, q0.z);
                mx = max3_of_sums(mx, q0.w, q1.x);
                mx = max3_of_sums(mx, q1.y, q1.z);
                mx = max3_of_sums(mx, q1.w, q2.x);
                mx = max3_of_sums(mx, q2.y, q2.z);
                mx = max3_of_sums(mx, q2.w, q2.w);
                open = mx == __builtin_huge_valf();  // fast mode's NaN mark (or a maximum that IS +inf: same record)
            } else {
                // the group's 52 values from LDS (NaNs drop out of v_max3_f32; fast mode finds them in the sum)
                float cv[Gm::GS];
                int cofs = Gm::PRE_K + gl * Gm::GS;
                asm volatile("" : "+v"(cofs));
#pragma unroll
                for (int e = 0; e < Gm::GS; ++e) cv[e] = lds[cofs + e];
                mx = max3_of_sums(cv[0], cv[1], cv[2]);
#pragma unroll
                for (int e = 3; e + 1 < Gm::GS; e += 2) mx = max3_of_sums(mx, cv[e], cv[e + 1]);
                if constexpr ((Gm::GS & 1) == 0) mx = max3_of_sums(mx, cv[Gm::GS - 1], cv[Gm::GS - 1]);
                open = mx == __builtin_huge_valf();
                if constexpr (FAST) {
                    float sum = 0.f;
#pragma unroll
                    for (int e = 0; e < Gm::GS; ++e) sum = sum + ((cv[e] == kNegInfF) ? 0.f : cv[e]);
                    if (sum != sum) open = true;  // a NaN position (decode.rs:250), or +inf and -inf values: always evaluated
                }
            }
            float hi = mx, lo = mx;
            if constexpr (!FAST) {
                // |F| over the group's window: the threads that hold positions kq .. kq + GS + G - 2
                constexpr int NT = (Gm::GS + Gm::G - 1 + L - 1) / L;
                static_assert(NT - 1 <= 3 + kPostThreads, "the |F| window must end inside the tile");
                float av[NT];
                int abofs = Gm::AB_OFF + tid;  // (AB + tid as one opaque register: immediate offsets below)
                asm volatile("" : "+v"(abofs));
#pragma unroll
                for (int e = 0; e < NT; ++e) av[e] = lds[abofs + e];
                float A = av[0];
#pragma unroll
                for (int e = 1; e < NT; ++e) A = A + av[e];
                const float err = A * gm_slack;
                if (!(err < __builtin_huge_valf())) open = true;  // NaN or Inf somewhere in the window
                hi = mx + err;
                lo = mx - err;
            }
            if (open) {
                hi = __builtin_huge_valf();
                lo = kNegInfF;
            }
            gm_out[o0 / Gm::GS + gl] = GroupMax{hi, lo};
        }
        if constexpr (INT) APT_MARK("END group_record");
        };
        if (interior) group_bounds(std::true_type{});
        else group_bounds(std::false_type{});
    }
    };  // run_tile

    XReg xr[NXR];
#if APT_FUSED_PERSIST
    if constexpr (!Gm::TABLE) {
        // Persistent form (experiment, probe builds): a fixed grid walks the (recording, tile) pairs of the call; the next
        // tile's input is requested right after this tile's has been written to LDS — into the same registers — and is
        // in flight under the four stages.
        const uint32_t tiles_x = call.tiles_x;  // tiles of the longest recording (what the plain form's grid.x is)
        const uint32_t total = tiles_x * call.count;
        // first (recording, tile) pair at or after `from`, in steps of the grid, whose tile exists (wave-uniform)
        auto first = [&](uint32_t from, uint32_t *r, int64_t *t, uint32_t *at) -> bool {
            for (uint32_t q = from; q < total; q += gridDim.x) {
                const uint32_t rr = q / tiles_x;
                const uint32_t tt = q - rr * tiles_x;
                if (static_cast<uint64_t>(tt) * Gm::OWN_K < call.rec[rr].w) {
                    *r = rr;
                    *t = static_cast<int64_t>(tt);
                    *at = q;
                    return true;
                }
            }
            return false;
        };
        uint32_t ri = 0, ri2 = 0, at = 0;
        int64_t tile = 0, tile2 = 0;
        if (!first(blockIdx.x, &ri, &tile, &at)) return;
        load_tile(ri, tile, 0, xr);
        while (true) {
            uint32_t at2 = 0;
            const bool more = first(at + gridDim.x, &ri2, &tile2, &at2);
            run_tile(ri, tile, xr, [&] {
                if (more) load_tile(ri2, tile2, 0, xr);
            });
            if (!more) break;
            ri = ri2;
            tile = tile2;
            at = at2;
            __syncthreads();  // the slowest wave is done with this tile's LDS before the next tile lands on it
        }
    } else
#endif
    {
    const uint32_t ri = blockIdx.y;
    const int64_t tile = blockIdx.x;
    if (static_cast<uint64_t>(tile) * Gm::OWN_K >= call.rec[ri].w) return;
    if constexpr (!Gm::TABLE) load_tile(ri, tile, 0, xr);
    run_tile(ri, tile, xr, [] {});
    }
#undef call
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: set it once
// per (kernel instantiation, device) — plans on different devices, and host threads creating them
// concurrently, all pass through here
template <auto Kern>
inline void ensure_dynamic_lds(size_t lds)
{
    constexpr int kMaxDevices = 64;
    static std::atomic<size_t> have[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > 48 * 1024 && lds > have[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        have[dev].store(lds, std::memory_order_release);
    }
}

template <int L, int M, int T1, int T2, int PW, int NTHR, int MODE, typename XT>
void launch_fused_args(const FusedLaunch &a)
{
    using Gm = FusedGeom<L, M, T1, T2, PW, NTHR, static_cast<int>(sizeof(XT)), fused_geom_var(MODE)>;
    size_t lds = static_cast<size_t>(Gm::LDS_FLOATS) * sizeof(float);
    if constexpr (Gm::TABLE) lds = (std::max<size_t>(a.table_lds_floats, Gm::W_LDS_FLOATS) + (Gm::PAD ? 4 : 0)) * sizeof(float);
    // APTGPU_FUSED_LDS_PAD=bytes (A/B switch, read at plan creation): more dynamic LDS than the kernel uses = fewer workgroups
    // per CU (2048 takes the 48 kHz kernels from six back to five)
    lds += static_cast<size_t>(a.lds_pad > 0 ? a.lds_pad : 0);
    constexpr auto kern = k_fused<L, M, T1, T2, PW, NTHR, XT, MODE>;
    ensure_dynamic_lds<kern>(lds);
    const unsigned tiles = static_cast<unsigned>((a.max_w + Gm::OWN_K - 1) / Gm::OWN_K);
#if APT_FUSED_PERSIST
    if constexpr (!Gm::TABLE) {
        CallArgs c = *a.call;
        c.tiles_x = tiles;
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint64_t total = static_cast<uint64_t>(tiles) * c.count;  // (< 2^32: kMaxCall recordings of < 2^26 tiles)
        const unsigned wgs = static_cast<unsigned>(std::min<uint64_t>(total, static_cast<uint64_t>(cus) * Gm::WGS_PER_CU));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), lds, a.s, c, a.prm);
        return;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(tiles, a.call->count), dim3(NTHR), lds, a.s, *a.call, a.prm);
}

}  // namespace

}  // namespace apt::gpu
