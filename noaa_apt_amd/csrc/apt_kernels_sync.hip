// apt_kernels_sync.hip — the peak picker of find_sync() (src/decode.rs:204-263) as a
// parallel computation.
//
// What the reference does, sequentially over the cross-correlation corr[0..n):
//   peaks = [(0, 0.)]
//   for i: if i - last.idx > md:  while i/spr > len(peaks): push (i, corr[i])
//          elif corr[i] > last.val: replace last by (i, corr[i])
//
// Facts used here (proved in DESIGN.md §Peak picker, checked against the oracle on
// adversarial inputs by tests/test_oracle_numpy_crosscheck.py and tests/test_gpu_parity.py):
//  (1) A tracking phase that starts at position s ends on the first TERMINAL t >= s, where
//      T[i] <=> no j in (i, i+md] has corr[j] > corr[i]   (corr[0] clamped to >= 0 for
//      the initial (0, 0.) peak).  The chain of strict records can never step over a
//      terminal.
//  (2) The picker is therefore the orbit  s -> u = firstT(s) -> s' = max(u+md+1,
//      (cell(s)+1)*spr)  over start positions; pushes fill peaks[cell(prev) .. cell(s)-2]
//      with s and peaks[cell(s)-1] with u.
//  (3) Starts are grid points c*spr or u+md+1 for a terminal u, and firstT(s) is either s
//      itself (then s is a terminal with s-md-1 terminal, or a grid terminal) or the head of
//      a run of terminals.  So only NODE TERMINALS matter: heads, terminals t with t-md-1
//      terminal, and terminals on the grid — a few per image row.
//
// NaN correlations (float WAVs can carry NaN / Inf samples): `corr > last` is false with a NaN on
// either side (decode.rs:250).  A NaN position is therefore never a record — a tracking phase
// passes over it as if it held -inf — and, once it IS the peak (which only happens when a phase
// STARTS on it), it is never replaced.  So: terminals are computed with NaN read as -inf, and
// firstT(s) = s when corr[s] is NaN (s > 0; the root's peak is (0, 0.) whatever corr[0] is).  NaN
// positions that can be starts (on the grid, or md+1 behind a terminal or another NaN) travel in
// the node-terminal lists with bit 31 set: such an entry only matches a search that starts on it.
//
// Kernels (GS = 52 correlation positions per group; md = 32*pw groups, spr = 40*pw groups):
//   k_group_max   corr -> GM (unfused path only; the fused front end emits GM itself)
//   k_sync_words  coarse: GM[g] = [lo, hi] bounds the maximum of group g (lo = hi where the front
//                 end evaluated the exact arithmetic; [-inf, +inf] = holds a NaN / not finite).  A
//                 group can hold a terminal only if its hi is not exceeded by the lo of one of the
//                 next md/GS-1 groups; fine: exact test on the few candidates, whose correlation
//                 values are EVALUATED from F (apt_sync_corr.hpp: the reference's chain, or the
//                 fast mode's pulse sums) — the fused front ends never write the correlation to
//                 HBM.  Where the bounds of the groups in between leave a candidate position's
//                 comparison open (lo <= corr < hi: bounds from the strict front ends' pulse sums,
//                 ~1e-3 of a typical correlation value wide, or NaN groups), those groups are
//                 evaluated exactly too: the result never depends on the bounds.  Emits the
//                 terminal words and NaN words of its 128 groups
//   k_sync_slots  terminal / NaN words -> ordered node-terminal list per chunk of 128 groups
//   k_sync_orbit  one workgroup per recording: builds the functional graph over start nodes,
//                 extracts the orbit of the root (directly when the recording is confluent,
//                 else by pointer doubling, in LDS when the visited nodes fit), writes the peak list
//                 and the result record;
//                 falls back to a sequential walk over the terminal words when a per-chunk
//                 list overflowed (pathological inputs).
// All three take the recordings of one decode_device call in one launch (CallArgs by value,
// slot table in HBM; apt_kernels.hpp).
#include "apt_kernels.hpp"
#include "apt_sync_corr.hpp"

#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdlib>
#include <type_traits>

namespace apt::gpu {

namespace {

constexpr int GS = 52;
constexpr uint64_t kGroupMask = (1ull << GS) - 1;
constexpr float kNegInf = -__builtin_huge_valf();

// ------------------------------------------------------------------ k_group_max
__global__ void __launch_bounds__(256)
k_group_max(const float *__restrict__ corr, uint64_t n_corr, GroupMax *__restrict__ gm, uint32_t ng)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const uint64_t base = static_cast<uint64_t>(g) * GS;
    float mx = kNegInf;
    bool has_nan = false;
    for (int o = 0; o < GS; ++o) {
        const uint64_t i = base + o;
        if (i < n_corr) {
            float v = corr[i];
            if (i == 0 && !(v > 0.f)) v = 0.f;
            mx = fmaxf(mx, v);
            has_nan = has_nan || (v != v);
        }
    }
    gm[g] = has_nan ? GroupMax{-kNegInf, kNegInf} : GroupMax{mx, mx};  // exact values: lo = hi
}

// ------------------------------------------------------------------ k_sync_words + k_sync_slots
constexpr int kNodesThreads = 128;
constexpr int kNodesWaves = kNodesThreads / 64;
constexpr int kChunkGroups = 128;  // own groups per workgroup
constexpr int kSlotCap = 64;       // node terminals kept per chunk before "overflow"
static_assert(kNodesThreads == kChunkGroups, "one thread per own group");
constexpr uint32_t kNanStartTag = 0x80000000u;  // slot entry = NaN position: only matches a search starting on it
constexpr uint32_t kPosMask = 0x7FFFFFFFu;      // (positions are < 2^31: longer recordings take the walk)

// floats of LDS one wave needs to re-evaluate the correlation of one group: an F window of
// GS + 38*pw - 1 samples
__host__ __device__ constexpr uint32_t nodes_window(uint32_t pw) { return (GS + 38u * pw - 1u + 3u) & ~3u; }

// Wave-wide scans of maxima with DPP row operations (no LDS traffic, no address arithmetic: a ds_bpermute step costs a
// v_add for its address, the LDS round trip and a compare + select for the wave's edge — six of them per scan, two
// scans per candidate group, were a quarter of this kernel's VALU instructions and most of a candidate's latency).
// The values are never NaN here (NaN correlations were replaced by -inf).  All 64 lanes must be active.
// "s_nop 1": a DPP operand written by the previous VALU instruction needs two wait states, and the compiler's hazard
// recogniser does not look inside an asm statement.
// inclusive prefix maximum: lane i <- max(x[0 .. i])
__device__ __forceinline__ float wave_prefix_max_dpp(float x)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   // row 0 -> 1, row 2 -> 3
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"   // rows 0-1 -> 2, 3
                 "s_nop 1"
                 : "+v"(x));
    return x;
}
// exclusive suffix maximum: lane i <- max(x[i+1 .. 63]), -inf in lane 63
__device__ __forceinline__ float wave_suffix_max_excl_dpp(float x, int lane)
{
    // y[i] = x[i+1] (lane 63: -inf), then the inclusive suffix maximum of y inside every row of 16 lanes ...
    float y = -__builtin_huge_valf();
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(y)
                 : "v"(x));
    // ... and the rows behind: their totals sit in their first lanes
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 48));
    const float m23 = fmaxf(r2, r3), m123 = fmaxf(r1, m23);
    const float tail = lane < 16 ? m123 : lane < 32 ? m23 : lane < 48 ? r3 : -__builtin_huge_valf();
    return fmaxf(y, tail);
}

// bytes of dynamic LDS of k_sync_words for r = md/GS groups of look-ahead: terminal and NaN words of the own
// groups | hi, lo bounds and two buffers of running maxima over own + look-ahead groups | window maxima |
// candidate list | one F window per wave
__host__ __device__ constexpr uint32_t words_f32_ofs() { return kChunkGroups * 16u; }
__host__ __device__ constexpr uint32_t words_win_ofs(uint32_t r)
{
    return (words_f32_ofs() + (kChunkGroups + r + 1) * 16u + kChunkGroups * 4u + kChunkGroups * 2u + 15u) & ~15u;
}

// correlation of the 52 positions of a group from its F window in `win` (LDS, owned by one wave: its
// LDS operations complete in program order, and the compiler keeps stores and loads of the same array
// in order, so no barrier is needed between filling a window and reading it; wave_barrier() only pins
// the phases for the scheduler).  fast: from pulse sums formed in place, exactly as the fast front end
// does; else the reference's chain.  Lanes with !on return -inf.
template <int NL, int PWC>
__device__ __forceinline__ float nodes_eval_window(float *win, bool on, int lane, uint32_t pw, int fast)
{
    constexpr int NLR = NL > 0 ? NL : 1;
    float c = kNegInf;
    if (fast) {
        // pulse sums in place: every value a lane needs is read before anything is written
        const uint32_t blen = GS + 36u * pw;  // positions whose pulse sum is used
        if constexpr (NL > 0) {
            float bs[NLR];
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const uint32_t pq = lane + 64u * t;
                bs[t] = pq < blen ? sync_pulse_sum(pw, [&](uint32_t j) { return win[pq + j]; }) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const uint32_t pq = lane + 64u * t;
                if (pq < blen) win[pq] = bs[t];
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // any pulse width: ascending sweeps of 64 positions; a sweep only overwrites positions
            // below the ones later sweeps read
            for (uint32_t p0 = 0; p0 < blen; p0 += 64) {
                const uint32_t pq = p0 + lane;
                const float bs = pq < blen ? sync_pulse_sum(pw, [&](uint32_t j) { return win[pq + j]; }) : 0.f;
                __builtin_amdgcn_wave_barrier();
                if (pq < blen) win[pq] = bs;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (on) c = sync_corr_from_pulses([&](int k) { return win[lane + k * 2 * pw]; });
    } else {
        if constexpr (PWC >= 4) {
            if (on) c = sync_corr_strict_rolled(pw, [&](uint32_t j) { return win[lane + j]; });
        } else {
            if (on) c = sync_corr_strict(pw, [&](uint32_t j) { return win[lane + j]; });
        }
    }
    return c;
}

// k_sync_words: terminal words and NaN words of the picker for the 128 groups of a chunk.
// NL: 64-lane loads per F window kept in registers while a wave has four candidates in flight
// (window <= 64*NL samples); 0: any window, one candidate at a time straight into LDS.
// PWC: the pixel width at compile time (the stock profiles: 3, 4, 5) — the 114..190-term correlation
// chains then unroll and their LDS reads pipeline (run-time loops wait for every read: 8x slower);
// 0: run-time pw.
// (Until round 2 one kernel also built the node-terminal lists, for which a workgroup needs the words of
// the md/GS + 1 groups BEHIND its own: it evaluated the candidates of that look-behind halo a second
// time, 76 % more work.  The lists are now a second, trivially cheap launch, k_sync_slots, that reads
// the words back — a kernel boundary instead of an inter-workgroup hand-over inside one launch.)
// DPP: the two scans of a candidate's terminal test with DPP row operations (the default; APTGPU_WORDS_DPP=0 keeps
// the ds_bpermute form for A/B).
template <int NL, int PWC, bool DPP>
__global__ void __launch_bounds__(kNodesThreads, 4)  // <= 128 VGPRs: must fit beside the front end's waves
k_sync_words(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t pw_arg, uint32_t r_groups /* md/GS */,
             int fast, int use_corr)
{
    const uint32_t pw = PWC > 0 ? static_cast<uint32_t>(PWC) : pw_arg;
    const RecArgs rec = call.rec[blockIdx.y];
    const uint64_t w = rec.w;
    const uint64_t n_corr = w - 38ull * pw;
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    if (blockIdx.x * static_cast<uint32_t>(kChunkGroups) >= ng) return;
    const SlotPtrs sp = slots[rec.slot];
    const GroupMax *__restrict__ gm = sp.gm;
    const float *__restrict__ corr = use_corr ? sp.corr : nullptr;  // nullptr: re-evaluate from F
    const float *__restrict__ fsig = sp.f;
    uint32_t *__restrict__ flags = sp.flags;

    // groups [g0, g0 + CG) are the workgroup's own; bounds are needed up to R groups ahead.  LDS is sized
    // at launch for the actual R and pw (8 KB at R = 96, pw = 3) so these workgroups fit beside the front
    // end of the next call, which leaves only ~9 KB of LDS free per CU.
    extern __shared__ uint64_t lds_nodes[];
    const int R = static_cast<int>(r_groups);
    const int N = kChunkGroups + R + 1;                                       // bounds held: own + look-ahead
    uint64_t *s_words = lds_nodes;                                            // [CG]
    uint64_t *s_nanw = s_words + kChunkGroups;                                // [CG]
    float *s_hi = reinterpret_cast<float *>(reinterpret_cast<char *>(lds_nodes) + words_f32_ofs());  // [N] upper bounds
    float *s_lo = s_hi + N;                                                   // [N] lower bounds
    float *s_ma = s_lo + N;                                                   // [N] running maxima of lo ...
    float *s_mb = s_ma + N;                                                   // [N] ... double-buffered
    float *s_wm = s_mb + N;                                                   // [CG]
    uint16_t *s_cand = reinterpret_cast<uint16_t *>(s_wm + kChunkGroups);     // [CG]
    const uint32_t wlen = nodes_window(pw);
    float *s_win = reinterpret_cast<float *>(reinterpret_cast<char *>(lds_nodes) + words_win_ofs(r_groups));
    __shared__ uint32_t s_ncand;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kChunkGroups;
    const int64_t gw0 = g0;  // group of window index 0
    const uint64_t md = static_cast<uint64_t>(R) * GS;

    if (tid == 0) s_ncand = 0;
    if (blockIdx.x == 0 && tid == 0) {
        // an unfinished chain is detectable: the orbit kernel overwrites this when it completes
        sp.res->status = -1;
        sp.res->reason = -1;
    }
    for (int q = tid; q < N; q += kNodesThreads) {
        const int64_t g = g0 + q;
        const GroupMax v = g < static_cast<int64_t>(ng) ? gm[g] : GroupMax{kNegInf, kNegInf};
        s_hi[q] = v.hi;
        s_lo[q] = v.lo;
        s_ma[q] = v.lo;
    }
    s_words[tid] = 0ull;  // (kNodesThreads == kChunkGroups)
    s_nanw[tid] = 0ull;
    __syncthreads();

    // coarse: WM[q] = max(lo[q+1 .. q+W]), W = R-1 — the full groups inside every window of group q's
    // positions — from running maxima over power-of-two spans (doubling, log2 W steps of one max per
    // entry): a window is the union of the two spans of K = 2^floor(log2 W) entries at its two ends.
    {
        const int W = R - 1;  // >= 1 for every legal md
        float *a = s_ma, *bq = s_mb;
        int K = 1;
        for (; 2 * K <= W; K *= 2) {
            for (int q = tid; q < N; q += kNodesThreads) bq[q] = (q + K < N) ? fmaxf(a[q], a[q + K]) : a[q];
            __syncthreads();
            float *t = a; a = bq; bq = t;
        }
        // a[q] = max(lo[q .. q+K-1]) (clipped at N)
        const int q = tid;
        const int64_t g = g0 + q;
        const float wm = fmaxf(a[q + 1], a[q + 1 + W - K]);  // q + W <= CG - 1 + R - 1 < N
        s_wm[q] = wm;
        // pruned only when a later group certainly holds more than this one possibly does (a group
        // that holds a NaN has hi = +inf: its NaN positions may be starts, see the top)
        if (g < static_cast<int64_t>(ng) && !(wm > s_hi[q])) s_cand[atomicAdd(&s_ncand, 1u)] = static_cast<uint16_t>(q);
        __syncthreads();
    }

    // fine: exact terminal test for the candidate groups; each wave takes four candidates at
    // a time so their loads are in flight together.  The group md ahead (whose first positions close
    // the window of this group's positions) is only evaluated when its maximum could matter.
    const uint32_t ncand = s_ncand;
    constexpr int kBatch = NL > 0 ? 4 : 1;
    constexpr int NLR = NL > 0 ? NL : 1;
    constexpr uint16_t kDeferred = 0x8000u;  // s_cand entry: to be settled in the second pass
    float *wa = s_win + static_cast<size_t>(wave) * wlen;  // this wave's F window
    const uint32_t wneed = GS + 38u * pw - 1u;  // samples of a window
    // F window of the group at `base` -> wa -> its correlation values
    auto eval_group = [&](uint64_t base, bool on) -> float {
        for (uint32_t t = lane; t < wlen; t += 64) {
            const uint64_t j = base + t;
            wa[t] = (t < wneed && j < w) ? fsig[j] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const float v = nodes_eval_window<NL, PWC>(wa, on, lane, pw, fast);
        __builtin_amdgcn_wave_barrier();
        return v;
    };
    // Terminal word of candidate group q (window-relative) from its correlation values cv (c2v: those of
    // the positions md ahead if already known, else -inf).  The full groups in between, g+1 .. g+R-1,
    // exceed corr[i] for certain when the largest of their lower bounds does, and do not for certain
    // when none of their upper bounds does; what the bounds leave open is settled by evaluating every
    // group in between that could matter — with EXACT; without, the candidate is handed back (false) for
    // the second pass, so that the four-candidates-in-flight loop stays free of that (rare) code.
    auto settle = [&](auto exact_tag, int q, float cv, float c2v, bool in_v) -> bool {
        constexpr bool EXACT = decltype(exact_tag)::value;
        const uint64_t base = static_cast<uint64_t>(gw0 + q) * GS;
        const bool in2 = in_v && base + lane + md < n_corr;
        if (in_v && gw0 + q == 0 && lane == 0 && !(cv > 0.f)) cv = 0.f;  // the peak (0, 0.)
        // NaN: remembered for the start test, -inf for every comparison
        const unsigned long long nanword = __ballot(in_v && cv != cv) & kGroupMask;
        if (cv != cv) cv = kNegInf;
        // suffix max over lanes > lane (rest of this group)
        float sfx_ex;
        if constexpr (DPP) {
            sfx_ex = wave_suffix_max_excl_dpp(cv, lane);
        } else {
            float sfx = cv;
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_down(sfx, d, 64);
                if (lane + d < 64) sfx = fmaxf(sfx, o);
            }
            sfx_ex = __shfl_down(sfx, 1, 64);
            if (lane == 63) sfx_ex = kNegInf;
        }
        float wm = s_wm[q];
        const bool open_lo = in_v && !(sfx_ex > cv) && !(wm > cv);
        if (__ballot(open_lo) != 0ull) {
            float wm_hi = kNegInf;
            for (int h = q + 1 + lane; h < q + R; h += 64) wm_hi = fmaxf(wm_hi, s_hi[h]);
            for (int d = 32; d >= 1; d >>= 1) wm_hi = fmaxf(wm_hi, __shfl_xor(wm_hi, d, 64));
            const bool open = open_lo && wm_hi > cv;
            if (__ballot(open) != 0ull) {
                if constexpr (!EXACT) {
                    return false;
                } else {
                    // exact maximum (NaNs left out) over the groups in between whose upper bound exceeds the
                    // smallest open value: the others cannot exceed any open position
                    float thr = open ? cv : -kNegInf;
                    for (int d = 32; d >= 1; d >>= 1) thr = fminf(thr, __shfl_xor(thr, d, 64));
                    float ex = kNegInf;
#pragma unroll 1
                    for (int h0 = q + 1; h0 < q + R; h0 += 64) {
                        // 64 groups per look: on APT data one or two of the 95 in between are above the threshold
                        unsigned long long todo = __ballot(h0 + lane < q + R && s_hi[h0 + lane] > thr);
                        while (todo) {
                            const int h = h0 + __ffsll(static_cast<long long>(todo)) - 1;
                            todo &= todo - 1;
                            const int64_t gh = gw0 + h;  // >= 1
                            if (gh >= static_cast<int64_t>(ng)) break;
                            const uint64_t hb = static_cast<uint64_t>(gh) * GS;
                            const bool on = lane < GS && hb + lane < n_corr;
                            float v = kNegInf;
                            if (corr != nullptr) {
                                if (on) v = corr[hb + lane];
                            } else {
                                v = eval_group(hb, on);
                            }
                            if (v != v) v = kNegInf;
                            ex = fmaxf(ex, v);
                        }
                    }
                    for (int d = 32; d >= 1; d >>= 1) ex = fmaxf(ex, __shfl_xor(ex, d, 64));
                    if (open) wm = ex;
                    if (lane == 0) atomicAdd(&flags[7], 1u);  // how often the bounds were not enough (tests, tuning)
                }
            }
        }
        float wmax = fmaxf(sfx_ex, wm);
        // the group md ahead: its positions up to this lane's offset belong to the window.  None of them
        // can exceed corr[i] unless the group's maximum does, so it is only evaluated when some lane
        // that is still a terminal so far is below the upper bound of that maximum (on APT data: almost
        // never).
        const float gm_ahead = s_hi[q + R];
        if (corr == nullptr && __ballot(in2 && !(wmax > cv) && gm_ahead > cv) != 0ull) c2v = eval_group(base + md, in2);
        if (c2v != c2v) c2v = kNegInf;
        // prefix max over lanes <= lane of the group md positions ahead
        float pfx = c2v;
        if constexpr (DPP) {
            pfx = wave_prefix_max_dpp(pfx);
        } else {
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_up(pfx, d, 64);
                if (lane >= d) pfx = fmaxf(pfx, o);
            }
        }
        wmax = fmaxf(wmax, pfx);
        const bool term = in_v && !(wmax > cv);
        const unsigned long long word = __ballot(term) & kGroupMask;
        if (lane == 0) {
            s_words[q] = word;
            s_nanw[q] = nanword;
        }
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    bool any_deferred = false;
    for (uint32_t c0 = wave * kBatch; c0 < ncand; c0 += kBatch * kNodesWaves) {
        int qv[kBatch];
        float cv[kBatch], c2v[kBatch];
        bool inv[kBatch];
        float fa[kBatch][NLR];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const uint32_t ci = c0 + e;
            qv[e] = (ci < ncand) ? s_cand[ci] : -1;
            const int64_t g = gw0 + (qv[e] < 0 ? 0 : qv[e]);
            const uint64_t i = static_cast<uint64_t>(g) * GS + lane;
            inv[e] = qv[e] >= 0 && lane < GS && i < n_corr;
            cv[e] = kNegInf;
            c2v[e] = kNegInf;
            if (corr != nullptr) {
                if (inv[e]) {
                    cv[e] = corr[i];
                    if (i + md < n_corr) c2v[e] = corr[i + md];
                }
            } else if constexpr (NL > 0) {
#pragma unroll
                for (int t = 0; t < NL; ++t) {
                    const uint64_t j = static_cast<uint64_t>(g) * GS + lane + 64u * t;
                    fa[e][t] = (qv[e] >= 0 && lane + 64u * t < wneed && j < w) ? fsig[j] : 0.f;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            if (qv[e] < 0) continue;  // wave-uniform
            if (corr == nullptr) {
                // F window of the candidate group -> LDS -> its 52 correlation values
                if constexpr (NL > 0) {
#pragma unroll
                    for (int t = 0; t < NL; ++t)
                        if (lane + 64u * t < wlen) wa[lane + 64 * t] = fa[e][t];
                    __builtin_amdgcn_wave_barrier();
                    cv[e] = nodes_eval_window<NL, PWC>(wa, inv[e], lane, pw, fast);
                    __builtin_amdgcn_wave_barrier();  // the window is dead from here on: wa is reused
                } else {
                    cv[e] = eval_group(static_cast<uint64_t>(gw0 + qv[e]) * GS, inv[e]);
                }
            }
            if (!settle(std::false_type{}, qv[e], cv[e], c2v[e], inv[e])) {
                if (lane == 0) s_cand[c0 + e] = static_cast<uint16_t>(qv[e]) | kDeferred;
                any_deferred = true;
            }
        }
    }
    // second pass: the candidates whose comparison with the groups in between the bounds left open (each
    // wave its own; one at a time, from scratch)
    if (any_deferred) {
#pragma unroll 1
        for (uint32_t c0 = wave * kBatch; c0 < ncand; c0 += kBatch * kNodesWaves) {
#pragma unroll 1
            for (uint32_t ci = c0; ci < c0 + kBatch && ci < ncand; ++ci) {
                const uint16_t ent = s_cand[ci];
                if (!(ent & kDeferred)) continue;  // wave-uniform
                const int q = ent & (kDeferred - 1);
                const uint64_t base = static_cast<uint64_t>(gw0 + q) * GS;
                const bool in_v = lane < GS && base + lane < n_corr;
                float cv = kNegInf, c2v = kNegInf;
                if (corr != nullptr) {
                    if (in_v) {
                        cv = corr[base + lane];
                        if (base + lane + md < n_corr) c2v = corr[base + lane + md];
                    }
                } else {
                    cv = eval_group(base, in_v);
                }
                (void)settle(std::true_type{}, q, cv, c2v, in_v);
            }
        }
    }
    __syncthreads();

    // own groups' words -> HBM (every group of the chunk: zero where nothing was evaluated)
    {
        const int64_t g = g0 + tid;
        if (g < static_cast<int64_t>(ng)) {
            sp.words[g] = s_words[tid];
            sp.nanw[g] = s_nanw[tid];
        }
    }
}

// k_sync_slots: the ordered node-terminal list of every chunk of 128 groups from the terminal / NaN words:
// heads of runs of terminals, terminals whose (t - md - 1) is a terminal or a NaN position, terminals on the
// grid, and the NaN positions a phase can start on (tagged with bit 31).  One thread per group.
__global__ void __launch_bounds__(kNodesThreads)
k_sync_slots(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t pw, uint32_t r_groups /* md/GS */,
             uint32_t grid_groups /* spr/GS */)
{
    const RecArgs rec = call.rec[blockIdx.y];
    const uint64_t n_corr = rec.w - 38ull * pw;
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    if (blockIdx.x * static_cast<uint32_t>(kChunkGroups) >= ng) return;
    const SlotPtrs sp = slots[rec.slot];
    const uint64_t *__restrict__ words = sp.words;
    const uint64_t *__restrict__ nanw = sp.nanw;
    uint32_t *__restrict__ slot_nt = sp.slot_nt;
    __shared__ uint32_t s_scan[kNodesWaves];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t R = r_groups;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * kChunkGroups + tid;
    auto word_at = [&](const uint64_t *p, int64_t gi) -> uint64_t {
        return (gi >= 0 && gi < static_cast<int64_t>(ng)) ? p[gi] : 0ull;
    };
    uint64_t nw = 0, ns = 0;
    if (g < static_cast<int64_t>(ng)) {
        const uint64_t wd = words[g];
        const uint64_t nb = nanw[g];
        const uint64_t prev_bit = word_at(words, g - 1) >> (GS - 1);
        const uint64_t heads = wd & ~(((wd << 1) | prev_bit) & kGroupMask);
        // positions md+1 behind a terminal or behind a NaN position, and the grid: where a phase can start
        const uint64_t shifted = ((word_at(words, g - R) << 1) | (word_at(words, g - R - 1) >> (GS - 1))) & kGroupMask;
        const uint64_t shifted_nan = ((word_at(nanw, g - R) << 1) | (word_at(nanw, g - R - 1) >> (GS - 1))) & kGroupMask;
        const uint64_t on_grid = (g % grid_groups == 0) ? 1ull : 0ull;
        const uint64_t starts = shifted | shifted_nan | on_grid;
        nw = wd & (heads | starts);
        ns = nb & starts & ~wd;  // NaN positions a phase can start on (position 0 was clamped: never NaN)
    }
    // ordered compaction of the node-terminal positions of this chunk (NaN starts tagged with bit 31)
    const uint32_t cnt = static_cast<uint32_t>(__popcll(nw | ns));
    uint32_t inc = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int wv = 0; wv < wave; ++wv) base += s_scan[wv];
    uint32_t total = 0;
    for (int wv = 0; wv < kNodesWaves; ++wv) total += s_scan[wv];
    uint32_t ofs = base + inc - cnt;
    uint64_t bitsleft = nw | ns;
    while (bitsleft) {
        const int bpos = __ffsll(static_cast<long long>(bitsleft)) - 1;
        bitsleft &= bitsleft - 1;
        if (ofs < kSlotCap)
            slot_nt[static_cast<uint64_t>(blockIdx.x) * kSlotCap + ofs] =
                static_cast<uint32_t>(static_cast<uint64_t>(g) * GS + bpos) | (((ns >> bpos) & 1ull) ? kNanStartTag : 0u);
        ++ofs;
    }
    if (tid == 0) {
        sp.slot_cnt[blockIdx.x] = total;
        if (total > kSlotCap) atomicOr(&sp.flags[0], 1u);
    }
}

// ------------------------------------------------------------------ k_sync_orbit
constexpr int kOrbitThreadsMax = 1024;
constexpr int kMaxLevels = 64;    // breadth-first levels before giving up on the parallel path

struct OrbitGeom {
    uint64_t n_corr, work_len;
    uint32_t spr, md;
};

// first terminal at or after s in the 52-bit group words (the fallback walk)
__device__ __forceinline__ uint64_t first_terminal52(const uint64_t *__restrict__ words,
                                                     uint64_t n_groups, uint64_t s)
{
    const int lane = threadIdx.x & 63;
    uint64_t g0 = s / GS;
    const uint32_t sh = static_cast<uint32_t>(s % GS);
    bool first = true;
    while (true) {
        const uint64_t gi = g0 + lane;
        uint64_t word = (gi < n_groups) ? words[gi] : 0ull;
        if (first && lane == 0) word &= (~0ull) << sh;
        const unsigned long long any = __ballot(word != 0ull);
        if (any) {
            const int src = __ffsll(static_cast<long long>(any)) - 1;
            const uint64_t pos = gi * GS + (__ffsll(static_cast<long long>(word)) - 1);
            return __shfl(pos, src, 64);
        }
        first = false;
        g0 += 64;
        if (g0 >= n_groups) return ~0ull;  // cannot happen: position n_corr-1 is a terminal
    }
}

// sequential orbit over the terminal words, one wave (any input, any size)
__device__ void orbit_walk52(const uint64_t *__restrict__ words, const uint64_t *__restrict__ nanw,
                             const OrbitGeom &gq, uint32_t *__restrict__ peaks, uint32_t peaks_cap,
                             Result *__restrict__ res)
{
    // a phase that starts on a NaN position keeps it (see the top of the file)
    auto nan_at = [&](uint64_t s) -> bool { return (nanw[s / GS] >> (s % GS)) & 1ull; };
    const int lane = threadIdx.x & 63;
    const uint64_t n_groups = (gq.n_corr + GS - 1) / GS;
    const uint64_t spr = gq.spr, md = gq.md;
    uint64_t len = 1;
    uint64_t u = 0;
    if (gq.n_corr > 0) u = first_terminal52(words, n_groups, 0);
    if (lane == 0 && peaks_cap > 0) peaks[0] = static_cast<uint32_t>(u);
    uint64_t fit = (u + spr < gq.work_len) ? 1 : 0;
    uint64_t last_fit = fit;
    while (gq.n_corr > 0) {
        const uint64_t a = u + md + 1;
        const uint64_t b = (len + 1) * spr;
        const uint64_t s = a > b ? a : b;
        if (s >= gq.n_corr) break;
        const uint64_t c = s / spr;
        for (uint64_t q = len + lane; q + 1 < c; q += 64)
            if (q < peaks_cap) peaks[q] = static_cast<uint32_t>(s);
        if (s + spr < gq.work_len) fit += c - len - 1;
        u = nan_at(s) ? s : first_terminal52(words, n_groups, s);
        if (lane == 0 && c - 1 < peaks_cap) peaks[c - 1] = static_cast<uint32_t>(u);
        last_fit = (u + spr < gq.work_len) ? 1 : 0;
        fit += last_fit;
        len = c;
    }
    if (lane == 0) {
        const bool few = len < 5;  // decode.rs:112-118
        res->status = few ? 1 : 0;
        res->reason = few ? 2 : 0;
        res->n_sync = static_cast<uint32_t>(len);
        res->n_rows = few ? 0 : static_cast<uint32_t>(fit - last_fit);
        res->work_len = gq.work_len;
        res->n_out = few ? 0 : (fit - last_fit) * 2080u;
    }
}


// Relaxed agent-scope accesses: the tables below live in global memory (L2) and are written
// and re-read inside one launch; these bypass the CU's L1 so a __syncthreads() is enough to
// hand data between the threads of the (single) workgroup.
__device__ __forceinline__ uint32_t gld(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gst(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One workgroup per recording (blockIdx.x); every table in the global scratch `ws` (sized by
// sync_orbit_ws_words()), so the kernel needs almost no LDS and can run beside the next call's
// front end.  Node terminals are looked up straight in the per-chunk slots k_sync_slots wrote
// (no gather pass): chunk = position / (128*52), then the first entry >= s of that slot or of the
// next non-empty one.  Node ids: 0 root, 1..n_grid grid cells 2..kc, base_d + slot entry, END.
// NT threads: 1024 (the latency-bound phases finish soonest when the kernel has a CU's slots to itself) or 256 (a
// workgroup that still finds room on a CU whose LDS and wave slots five front-end workgroups have taken).
template <int NT>
__global__ void __launch_bounds__(NT, NT >= 1024 ? 8 : 2)
k_sync_orbit_global(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t spr_in, uint32_t md_in,
                    uint32_t pw, int force_walk, uint32_t lds_entries, int alg)
{
    const RecArgs rec = call.rec[blockIdx.x];
    const SlotPtrs sp = slots[rec.slot];
    const uint64_t *__restrict__ words = sp.words;
    const uint32_t *__restrict__ slot_nt = sp.slot_nt;
    const uint32_t *__restrict__ slot_cnt = sp.slot_cnt;
    uint32_t *__restrict__ flags = sp.flags;
    uint32_t *__restrict__ ws = sp.orbit_ws;
    uint32_t *__restrict__ peaks = sp.peaks;
    const uint32_t peaks_cap = sp.peaks_cap;
    Result *__restrict__ res = sp.res;
    OrbitGeom gq;
    gq.work_len = rec.w;
    gq.n_corr = rec.w - 38ull * pw;
    gq.spr = spr_in;
    gq.md = md_in;
    const uint32_t n_chunks =
        static_cast<uint32_t>(((gq.n_corr + GS - 1) / GS + kChunkGroups - 1) / kChunkGroups);
    const uint32_t nt_cap = n_chunks * kSlotCap;  // node ids of slot entries: base_d + chunk*kSlotCap + k
    __shared__ uint32_t s_count, s_plen, s_conflict, s_endcell;
    __shared__ unsigned long long s_fit;
    __shared__ uint32_t s_wtot[kOrbitThreadsMax / 64];
    extern __shared__ uint16_t lds_orbit[];
    // this latency-bound workgroup shares its CU with VALU-saturated front-end waves of the next
    // recording: let its few instructions issue first
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const uint64_t n_corr = gq.n_corr;
    const uint32_t spr = gq.spr, md = gq.md;
    const uint32_t spr_magic = static_cast<uint32_t>((1ull << 32) / spr);
    auto div_spr = [&](uint32_t sv) -> uint32_t {  // sv / spr by multiply-high + two corrections
        uint32_t q = __umulhi(sv, spr_magic);
        uint32_t r = sv - q * spr;
        if (r >= spr) { ++q; r -= spr; }
        if (r >= spr) ++q;
        return q;
    };
    const uint64_t t_begin = __builtin_readcyclecounter();
    auto stamp = [&](int k) {
        if (tid == 0) flags[8 + k] = static_cast<uint32_t>(__builtin_readcyclecounter() - t_begin);
    };
    // finer stamps of the closure-free path (flags[16 ..]: tools/orbit_stamps.py prints them)
    auto fine = [&](int k) {
        if (tid == 0) flags[16 + k] = static_cast<uint32_t>(__builtin_readcyclecounter() - t_begin);
    };

    const uint64_t kc64 = n_corr ? (n_corr - 1) / spr : 0;  // cells that can hold a start: 2 .. kc
    const uint32_t flags7 = flags[7];  // (read here, beside flags[0]: the result record's writer would wait a round trip for it)
    bool walk = force_walk == 1 || flags[0] != 0 || n_corr == 0 || gq.work_len >= (1ull << 31);
    const uint32_t kc = static_cast<uint32_t>(kc64);

    // ---- workspace carve-up (uint32 words)
    const uint32_t n_grid = kc >= 2 ? kc - 1 : 0;
    const uint32_t base_d = 1 + n_grid;
    const uint32_t END = base_d + nt_cap;
    const uint32_t n_nodes = END + 1;
    uint32_t *w_ja = ws;                      // [n_nodes] next / jump table
    uint32_t *w_jb = w_ja + n_nodes;          // [n_nodes] double buffer
    uint32_t *w_u = w_jb + n_nodes;           // [n_nodes] terminal reached from the node's start
    uint32_t *w_list = w_u + n_nodes;         // [n_nodes] visited nodes
    uint32_t *w_mark = w_list + n_nodes;      // [n_nodes/32 + 1]
    uint32_t *w_path = w_mark + (n_nodes / 32 + 1);  // [kc + 2]
    uint32_t *w_succ = w_path + (kc + 2);            // [kc + 3] common successor per cell
    const uint32_t nc32 = static_cast<uint32_t>(n_corr);
    const uint32_t wl32 = static_cast<uint32_t>(gq.work_len);
    constexpr uint32_t kChunkSpan = kChunkGroups * GS;  // positions per chunk

    // first node terminal at or after sv: (slot entry index, value); slots were written by the
    // previous kernel, so plain (cached) loads are fine
    auto first_node_terminal = [&](uint32_t sv, uint32_t *uval) -> uint32_t {
        uint32_t ch = sv / kChunkSpan;
        for (; ch < n_chunks; ++ch) {
            const uint32_t cnt = slot_cnt[ch];
            if (cnt == 0) continue;
            const uint4 *src = reinterpret_cast<const uint4 *>(slot_nt + static_cast<uint64_t>(ch) * kSlotCap);
            const uint32_t lim = cnt < kSlotCap ? cnt : kSlotCap;
            for (uint32_t j = 0; 4 * j < lim; j += 4) {  // 16 entries per round trip
                uint4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (4 * (j + e) < lim) ? src[j + e] : make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t k0 = 4 * (j + e);
                    const uint32_t vals[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {  // 4 entries per uint4
                        const uint32_t pos = vals[t] & kPosMask;  // a tagged entry only matches its own position
                        if (k0 + t < lim && pos >= sv && (!(vals[t] & kNanStartTag) || pos == sv)) {
                            *uval = pos;
                            return ch * kSlotCap + k0 + t;
                        }
                    }
                }
            }
        }
        *uval = nc32 - 1;  // cannot happen (fact 3)
        return nt_cap - 1;
    };
    auto node_start = [&](uint32_t v, uint32_t *cell) -> uint32_t {
        if (v == 0) { *cell = 1; return 0; }
        if (v < base_d) { *cell = v + 1; return (v + 1) * spr; }
        const uint32_t sv = (slot_nt[v - base_d] & kPosMask) + md + 1;
        *cell = div_spr(sv);
        return sv;
    };

    // ---- alg 1 (the default): NO closure.  The successor of EVERY possible start — the root, the grid nodes, md + 1 behind every
    // node terminal of the recording — is one independent lookup in the (compacted, LDS-resident) node-terminal list;
    // the root's orbit then follows by doubling the known prefix, path[m + 2^r] = J_r[path[m]], J_{r+1} = J_r o J_r,
    // with 16-bit jump tables in LDS: log2(cells) rounds whatever the recording looks like (the breadth-first closure
    // below takes one global round trip per level: 18 and 34 levels for 2 of the bench's 8 recordings).  Needs
    // nodes < 65535 and (chunks + entries) * 4 + nodes * 2 + cells * 2 bytes of LDS; else the closure path runs.
    bool all_done = false;
    bool peaks_done = false;  // ... and the peak list written from LDS already (round 6)
    uint32_t n_all = 0;
    fine(0);  // arguments, slot pointers and the overflow flag are here
    if (!walk && alg == 1) {
        const uint32_t lds_bytes = lds_entries * 2u;
        uint32_t *s_pref = reinterpret_cast<uint32_t *>(lds_orbit);  // [n_chunks + 1] entries before chunk ch
        bool ok = (n_chunks + 1u) * 4u <= lds_bytes;                 // (uniform)
        uint32_t n_ent = 0;
        // a contiguous run of chunks per thread.  Round 6: where that is at most kPre chunks (recordings up to 18 minutes
        // at 1024 threads) their entry counts AND their first sixteen entries are fetched in ONE round trip — the
        // addresses do not depend on the counts — where the scan and the gather below used to take three in a row
        constexpr uint32_t kPre = 2;
        const uint32_t cpt = (n_chunks + NT - 1) / NT;
        const uint32_t c0 = static_cast<uint32_t>(tid) * cpt;
        const bool pre = cpt <= kPre;  // (uniform)
        uint32_t cnt_r[kPre] = {};
        uint4 ent_r[kPre][4] = {};
        uint32_t run_r[kPre] = {};
        if (ok) {
            uint32_t local = 0;
            if (pre) {
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) {
                    const uint32_t ch = c0 + j;
                    if (j < cpt && ch < n_chunks) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(slot_nt + static_cast<uint64_t>(ch) * kSlotCap);
                        cnt_r[j] = slot_cnt[ch];
#pragma unroll
                        for (int e = 0; e < 4; ++e) ent_r[j][e] = src[e];
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) {
                    cnt_r[j] = cnt_r[j] < static_cast<uint32_t>(kSlotCap) ? cnt_r[j] : static_cast<uint32_t>(kSlotCap);
                    local += cnt_r[j];
                }
            } else {
                for (uint32_t j = 0; j < cpt; ++j) {
                    const uint32_t ch = c0 + j;
                    if (ch < n_chunks) {
                        const uint32_t c = slot_cnt[ch];
                        local += c < kSlotCap ? c : kSlotCap;
                    }
                }
            }
            uint32_t incl = local;
            const int ln = tid & 63;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (ln >= d) incl += o;
            }
            if (ln == 63) s_wtot[wave] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (int wq = 0; wq < wave; ++wq) before += s_wtot[wq];
            uint32_t run = before + incl - local;
            if (pre) {
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) {
                    const uint32_t ch = c0 + j;
                    run_r[j] = run;
                    if (j < cpt && ch < n_chunks) s_pref[ch] = run;
                    run += cnt_r[j];
                }
            } else {
                for (uint32_t j = 0; j < cpt; ++j) {
                    const uint32_t ch = c0 + j;
                    if (ch < n_chunks) {
                        s_pref[ch] = run;
                        const uint32_t c = slot_cnt[ch];
                        run += c < kSlotCap ? c : kSlotCap;
                    }
                }
            }
            if (tid == NT - 1) s_pref[n_chunks] = run;  // the last thread's run ends at the total
            __syncthreads();
            n_ent = s_pref[n_chunks];
            fine(1);  // counts and entries fetched, counts scanned
        }
        n_all = base_d + n_ent;
        const uint32_t path_cap_a = kc + 2;
        // LDS: s_pref | s_ent [n_ent] (uint32; the pruned jump tables — or the second full one — lie over it once the
        // successors are known) | la [n_all + 2] | pth [path_cap] | nid [n_all + 2] (all uint16)
        const uint32_t la_len = (n_all + 2u) & ~1u, pth_len = (path_cap_a + 1u) & ~1u;
        const uint64_t need = 4ull * (n_chunks + 1u) + 4ull * n_ent + 2ull * la_len + 2ull * pth_len + 2ull * la_len + 8u;
        ok = ok && n_ent > 0 && n_all < 0xFFFEu && 2ull * (n_all + 2u) <= 4ull * n_ent && need <= lds_bytes;
        if (ok) {
            uint32_t *s_ent = s_pref + (n_chunks + 1);
            uint16_t *la = reinterpret_cast<uint16_t *>(s_ent + n_ent);
            uint16_t *pth = la + la_len;
            uint16_t *nid = pth + pth_len;
            uint16_t *lb = reinterpret_cast<uint16_t *>(s_ent);
            const uint16_t ENDC = static_cast<uint16_t>(n_all);
            // the node terminals of the whole recording, in order
            if (pre) {
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) {
                    const uint32_t ch = c0 + j;
                    if (j < cpt && ch < n_chunks) {
                        const uint32_t lim = cnt_r[j], at = run_r[j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t vals[4] = {ent_r[j][e].x, ent_r[j][e].y, ent_r[j][e].z, ent_r[j][e].w};
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (4u * e + t < lim) s_ent[at + 4 * e + t] = vals[t];
                        }
                        if (lim > 16u) {  // (rare: more than sixteen node terminals in 128 groups)
                            const uint4 *src = reinterpret_cast<const uint4 *>(slot_nt + static_cast<uint64_t>(ch) * kSlotCap);
                            for (uint32_t k4 = 4; 4 * k4 < lim; ++k4) {
                                const uint4 v = src[k4];
                                const uint32_t vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int t = 0; t < 4; ++t)
                                    if (4 * k4 + t < lim) s_ent[at + 4 * k4 + t] = vals[t];
                            }
                        }
                    }
                }
            } else {
                for (uint32_t ch = tid; ch < n_chunks; ch += NT) {
                    const uint32_t c = slot_cnt[ch];
                    const uint32_t lim = c < kSlotCap ? c : kSlotCap;
                    const uint32_t at = s_pref[ch];
                    const uint4 *src = reinterpret_cast<const uint4 *>(slot_nt + static_cast<uint64_t>(ch) * kSlotCap);
                    for (uint32_t k4 = 0; 4 * k4 < lim; ++k4) {
                        const uint4 v = src[k4];
                        const uint32_t vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (4 * k4 + t < lim) s_ent[at + 4 * k4 + t] = vals[t];
                    }
                }
            }
            __syncthreads();
            fine(2);  // list in LDS
            // first node terminal at or after sv, from the list in LDS: a binary search inside sv's chunk (the list is in
            // position order; s_pref gives the chunk's run: <= 64 entries, ten on recordings) — a handful of instructions per
            // step.  (Round 5 walked the chunk entry by entry from its first one, 46 k cycles for a ten-minute recording on
            // this one CU; counting sixteen entries at a time without a dependent chain took 29 k: 150 VALU instructions
            // per node — the phase is bound by the CU's issue slots, not by LDS latency.)  A tagged (NaN) entry only
            // matches its own position.
            auto lds_lower_bound = [&](uint32_t sv) -> uint32_t {  // first list index whose position is >= sv
                const uint32_t c = sv / kChunkSpan;
                uint32_t lo = s_pref[c], hi = s_pref[c + 1];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((s_ent[mid] & kPosMask) < sv) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            auto lds_match_from = [&](uint32_t j, uint32_t sv, uint32_t *uval) -> uint32_t {  // (every entry from j on is >= sv)
                for (; j < n_ent; ++j) {
                    const uint32_t e = s_ent[j];
                    const uint32_t pos = e & kPosMask;
                    if (!(e & kNanStartTag) || pos == sv) { *uval = pos; return j; }
                }
                *uval = nc32 - 1;  // cannot happen (fact 3)
                return n_ent - 1;
            };
            auto lds_first_terminal = [&](uint32_t sv, uint32_t *uval) -> uint32_t { return lds_match_from(lds_lower_bound(sv), sv, uval); };
            auto successor = [&](uint32_t cell, uint32_t u, uint32_t j) -> uint32_t {
                const uint32_t a = u + md + 1;
                const uint32_t b = (cell + 1) * spr;
                const uint32_t s2 = a > b ? a : b;
                return s2 < nc32 ? ((a >= b) ? base_d + j : cell) : n_all;  // grid(cell+1) has id `cell`; n_all: END
            };
            // successor of every node (ids: 0 root, 1 .. n_grid grid cells 2 .. kc, base_d + list index), one search each, the
            // nodes dealt round-robin.  (Measured, cycles of this phase for a ten-minute recording: entry-by-entry walk from
            // the chunk's first entry 46 k (round 5); sixteen entries at a time, counted without a dependent chain, 29 k — 150
            // VALU instructions per node; runs of consecutive list nodes per thread with a carried lower bound 40 k — one
            // long dependent chain per thread; this form 21 k.)
            const uint32_t npt = (n_all + NT - 1) / NT;
            const uint32_t i_lo = static_cast<uint32_t>(tid) * npt;
            const uint32_t i_hi = i_lo + npt < n_all ? i_lo + npt : n_all;
            for (uint32_t i = tid; i < n_all; i += NT) {
                uint32_t cell, sv;
                if (i == 0) { cell = 1; sv = 0; }
                else if (i < base_d) { cell = i + 1; sv = cell * spr; }
                else { sv = (s_ent[i - base_d] & kPosMask) + md + 1; cell = div_spr(sv); }
                uint32_t nx = n_all;  // END
                if (sv < nc32) {
                    uint32_t u;
                    const uint32_t j = lds_first_terminal(sv, &u);
                    nx = successor(cell, u, j);
                }
                la[i] = static_cast<uint16_t>(nx);
            }
            if (tid == 0) { la[n_all] = ENDC; pth[0] = 0; }
            for (uint32_t i = tid; i <= n_all; i += NT) nid[i] = 0;
            __syncthreads();
            stamp(0);  // successors known
            // ---- round 6: only a node that is SOME node's successor (or the root) can lie on the orbit, and recordings
            // are confluent — most of a row's ten or so candidate starts lead to the same next start — so the doubling
            // below runs over the few nodes with a predecessor, renumbered densely, not over all of them (its cost is
            // LDS gathers: nodes x rounds).
            for (uint32_t i = tid; i < n_all; i += NT) {
                const uint16_t t = la[i];
                if (t != ENDC) nid[t] = 1;  // (the same value from every writer)
            }
            if (tid == 0) nid[0] = 1;
            __syncthreads();
            uint32_t n2 = 0;
            {
                uint32_t local = 0;
                for (uint32_t i = i_lo; i < i_hi; ++i) local += nid[i];
                uint32_t incl = local;
                const int ln = tid & 63;
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d, 64);
                    if (ln >= d) incl += o;
                }
                __syncthreads();  // (s_wtot: every wave has read the entry scan's totals)
                if (ln == 63) s_wtot[wave] = incl;
                __syncthreads();
                uint32_t before = 0;
                for (int wq = 0; wq < NT / 64; ++wq) {
                    if (wq < wave) before += s_wtot[wq];
                    n2 += s_wtot[wq];
                }
                uint32_t k = before + incl - local;
                for (uint32_t i = i_lo; i < i_hi; ++i) nid[i] = nid[i] ? static_cast<uint16_t>(k++) : static_cast<uint16_t>(0xFFFFu);
            }
            __syncthreads();
            fine(3);  // nodes with a predecessor numbered
            const uint32_t l2_len = (n2 + 2u) & ~1u;
            // (uniform) the three pruned tables behind everything else, so that the list stays readable: the path's starts and
            // terminals are then looked up in LDS too and the peak list below is written from LDS (path arrays over `la`)
            const bool fastp = need + 6ull * l2_len <= lds_bytes && 12ull * path_cap_a <= 2ull * la_len;
            const bool pruned = 3ull * 2ull * l2_len <= 4ull * n_ent;  // (uniform) else: three tables over s_ent
            if (fastp) {
                uint16_t *la2 = nid + la_len, *lb2 = la2 + l2_len, *orig = lb2 + l2_len;
                const uint16_t END2 = static_cast<uint16_t>(n2);
                for (uint32_t i = i_lo; i < i_hi; ++i) {
                    const uint16_t me = nid[i];
                    if (me != 0xFFFFu) {
                        const uint16_t t = la[i];
                        la2[me] = t == ENDC ? END2 : nid[t];  // (a successor has a predecessor: it is numbered)
                        orig[me] = static_cast<uint16_t>(i);
                    }
                }
                if (tid == 0) { la2[n2] = END2; lb2[n2] = END2; }  // (pth[0] = 0: the root is the first numbered node)
                __syncthreads();
                // path[m + q 4^r] = J^q[path[m]], q = 1 .. 3;  J <- J^4: half the rounds (and barriers) of plain doubling
                for (uint32_t span = 1; span < path_cap_a; span <<= 2) {
                    for (uint32_t mI = tid; mI < span && mI + span < path_cap_a; mI += NT) {
                        const uint16_t t1 = la2[pth[mI]], t2 = la2[t1], t3 = la2[t2];
                        pth[mI + span] = t1;
                        if (mI + 2 * span < path_cap_a) pth[mI + 2 * span] = t2;
                        if (mI + 3 * span < path_cap_a) pth[mI + 3 * span] = t3;
                    }
                    // (four entries per thread and pass, their four-deep chains side by side: the phase is LDS latency)
                    for (uint32_t i = tid; i < n2; i += 4 * NT) {
                        uint16_t t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = la2[i + e * NT < n2 ? i + e * NT : n2];
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int e = 0; e < 4; ++e) t[e] = la2[t[e]];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (i + e * NT < n2) lb2[i + e * NT] = t[e];
                    }
                    __syncthreads();
                    uint16_t *t = la2; la2 = lb2; lb2 = t;
                }
                if (tid == 0) flags[13] = n2;
                fine(4);  // orbit known
                // the path's starts, cells and terminals -> LDS (over `la`: dead)
                uint32_t *p_sv = reinterpret_cast<uint32_t *>(la), *p_cell = p_sv + path_cap_a, *p_u = p_cell + path_cap_a;
                for (uint32_t k = tid; k < path_cap_a; k += NT) {
                    const uint16_t q = pth[k];
                    uint32_t sv = 0xFFFFFFFFu, cell = 0, u = 0;
                    if (q != END2) {
                        const uint32_t i = orig[q];
                        if (i == 0) { cell = 1; sv = 0; }
                        else if (i < base_d) { cell = i + 1; sv = cell * spr; }
                        else { sv = (s_ent[i - base_d] & kPosMask) + md + 1; cell = div_spr(sv); }
                        if (sv < nc32) (void)lds_first_terminal(sv, &u);
                    }
                    p_sv[k] = sv;
                    p_cell[k] = cell;
                    p_u[k] = u;
                }
                if (tid == 0) { s_plen = 1; s_fit = 0ull; }
                __syncthreads();
                stamp(1);  // orbit extracted
                unsigned long long fit_l = 0;
                for (uint32_t k = tid; k < path_cap_a; k += NT) {
                    const uint32_t sv = p_sv[k];
                    if (sv == 0xFFFFFFFFu) continue;
                    const uint32_t u = p_u[k];
                    const bool is_last = (k + 1 >= path_cap_a) || p_sv[k + 1] == 0xFFFFFFFFu;
                    if (k == 0) {
                        if (peaks_cap > 0) peaks[0] = u;
                        if (!is_last && u + spr < wl32) ++fit_l;
                        if (is_last) s_plen = 1;
                        continue;
                    }
                    const uint32_t c_prev = (k - 1 == 0) ? 1u : p_cell[k - 1];  // the root leaves one entry
                    const uint32_t c = p_cell[k];                               // = s / spr for k >= 1
                    for (uint32_t qv = c_prev; qv + 1 < c; ++qv)
                        if (qv < peaks_cap) peaks[qv] = sv;
                    if (c - 1 < peaks_cap) peaks[c - 1] = u;
                    if (sv + spr < wl32) fit_l += c - c_prev - 1;
                    if (!is_last && u + spr < wl32) ++fit_l;  // the last peak is dropped
                    if (is_last) s_plen = c;
                }
                // (one LDS atomic per wave: a thousand of them on one address serialise — seven microseconds)
                {
                    uint32_t f32v = static_cast<uint32_t>(fit_l);  // (< 2^32: rows of one recording)
                    for (int d = 32; d >= 1; d >>= 1) f32v += __shfl_down(f32v, d, 64);
                    if ((tid & 63) == 0 && f32v) atomicAdd(&s_fit, static_cast<unsigned long long>(f32v));
                }
                __syncthreads();
                peaks_done = true;
            } else if (pruned) {
                uint16_t *la2 = reinterpret_cast<uint16_t *>(s_ent), *lb2 = la2 + l2_len, *orig = lb2 + l2_len;
                const uint16_t END2 = static_cast<uint16_t>(n2);
                for (uint32_t i = i_lo; i < i_hi; ++i) {
                    const uint16_t me = nid[i];
                    if (me != 0xFFFFu) {
                        const uint16_t t = la[i];
                        la2[me] = t == ENDC ? END2 : nid[t];  // (a successor has a predecessor: it is numbered)
                        orig[me] = static_cast<uint16_t>(i);
                    }
                }
                if (tid == 0) { la2[n2] = END2; lb2[n2] = END2; }  // (pth[0] = 0: the root is the first numbered node)
                __syncthreads();
                for (uint32_t span = 1; span < path_cap_a; span <<= 1) {
                    for (uint32_t mI = tid; mI < span && mI + span < path_cap_a; mI += NT) pth[mI + span] = la2[pth[mI]];
                    for (uint32_t i = tid; i < n2; i += NT) lb2[i] = la2[la2[i]];
                    __syncthreads();
                    uint16_t *t = la2; la2 = lb2; lb2 = t;
                }
                for (uint32_t k = tid; k < path_cap_a; k += NT) {
                    const uint16_t q = pth[k];
                    pth[k] = q == END2 ? ENDC : orig[q];
                }
                __syncthreads();
            } else {
                if (tid == 0) lb[n_all] = ENDC;  // (over s_ent: dead from here on)
                for (uint32_t span = 1; span < path_cap_a; span <<= 1) {
                    for (uint32_t mI = tid; mI < span && mI + span < path_cap_a; mI += NT) pth[mI + span] = la[pth[mI]];
                    for (uint32_t i = tid; i < n_all; i += NT) lb[i] = la[la[i]];
                    __syncthreads();
                    uint16_t *t = la; la = lb; lb = t;
                }
            }
            if (tid == 0 && !fastp) flags[13] = n2;  // nodes with a predecessor (diagnostics)
            if (!fastp) fine(4);  // orbit known (LDS ids)
            // the path in the kernel's node ids (base_d + chunk * kSlotCap + k for list entries), and the terminal
            // each of its nodes reaches — what the peak-list code below reads
            for (uint32_t k = tid; !fastp && k < path_cap_a; k += NT) {
                const uint32_t i = pth[k];
                uint32_t v = END;
                if (i != ENDC) {
                    v = i;
                    if (i >= base_d) {
                        // chunk of list index q: the last ch with s_pref[ch] <= q
                        const uint32_t q = i - base_d;
                        uint32_t lo_c = 0, hi_c = n_chunks;  // s_pref[lo_c] <= q < s_pref[hi_c]
                        while (hi_c - lo_c > 1) {
                            const uint32_t mid = (lo_c + hi_c) >> 1;
                            if (s_pref[mid] <= q) lo_c = mid; else hi_c = mid;
                        }
                        v = base_d + lo_c * kSlotCap + (q - s_pref[lo_c]);
                    }
                    uint32_t cell, u = 0;
                    const uint32_t sv = node_start(v, &cell);
                    if (sv < nc32) (void)first_node_terminal(sv, &u);
                    gst(w_u + v, u);
                }
                gst(w_path + k, v);
            }
            __syncthreads();
            all_done = true;
        }
    }

    // ---- reachable set by breadth-first marking from the root and every grid node
    uint32_t count = 0;
    if (!walk && !all_done) {
        for (uint32_t wq = tid; wq < n_nodes / 32 + 1; wq += NT) gst(w_mark + wq, 0u);
        for (uint32_t c = tid; c < kc + 3; c += NT) gst(w_succ + c, 0xFFFFFFFFu);
        if (tid == 0) { s_count = base_d; s_conflict = 0; s_endcell = kc + 2; }
        __syncthreads();
        for (uint32_t v = tid; v < base_d; v += NT) {
            gst(w_list + v, v);
            atomicOr(w_mark + (v >> 5), 1u << (v & 31));
        }
        __syncthreads();
        uint32_t lo = 0, hi = base_d;
        int level = 0;
        for (; level < kMaxLevels && lo < hi; ++level) {
            for (uint32_t idx = lo + tid; idx < hi; idx += NT) {
                const uint32_t v = gld(w_list + idx);
                uint32_t cell, u = 0, nx = END, nxcell = 0;
                const uint32_t sv = node_start(v, &cell);
                if (sv < nc32) {
                    const uint32_t e = first_node_terminal(sv, &u);
                    const uint32_t a = u + md + 1;
                    const uint32_t b = (cell + 1) * spr;
                    const uint32_t s2 = a > b ? a : b;
                    if (s2 < nc32) {
                        nx = (a >= b) ? base_d + e : cell;  // grid(cell+1) has id `cell`
                        nxcell = (a >= b) ? div_spr(a) : cell + 1;
                    }
                }
                gst(w_ja + v, nx);
                gst(w_u + v, u);
                // confluence bookkeeping for the direct path: every visited start of a cell must
                // agree on the successor, and the successor must sit in the very next cell
                if (cell < kc + 3) {
                    if (v < base_d) {
                        // seeds (root, grid nodes) are alone in their cell: plain store, no round trip
                        gst(w_succ + cell, nx);
                    } else {
                        const uint32_t old = atomicCAS(w_succ + cell, 0xFFFFFFFFu, nx);
                        if (old != 0xFFFFFFFFu && old != nx) s_conflict = 1;
                    }
                }
                if (nx == END) atomicMin(&s_endcell, cell);
                else if (nxcell != cell + 1) s_conflict = 1;
                if (nx != END && nx >= base_d) {  // grid targets are seeds: already visited
                    const uint32_t bit = 1u << (nx & 31);
                    if (!(atomicOr(w_mark + (nx >> 5), bit) & bit)) {
                        const uint32_t at = atomicAdd(&s_count, 1u);
                        gst(w_list + at, nx);
                        gst(w_jb + nx, at);  // its index in the visited list (w_jb is free until the doubling)
                    }
                }
            }
            __syncthreads();
            lo = hi;
            hi = s_count;
            __syncthreads();
        }
        if (lo < hi) walk = true;  // not closed within the level budget: take the general path
        count = hi;
        if (tid == 0) flags[12] = static_cast<uint32_t>(level);  // breadth-first levels (diagnostics)
    }
    if (!all_done) stamp(0);  // reachable set closed

    if (walk) {
        if (wave == 0) orbit_walk52(words, sp.nanw, gq, peaks, peaks_cap, res);
        if (tid == 0) {
            flags[1] = 1u;  // report which path ran
            flags[0] = 0u;  // re-arm the overflow flag for the next decode
            flags[5] = 0u;
            flags[11] = flags7;  // candidates k_sync_words settled with exact window maxima; re-armed
            flags[7] = 0u;
        }
        return;
    }
    if (tid == 0 && !all_done) { gst(w_ja + END, END); gst(w_jb + END, END); gst(w_path, 0u); }
    __syncthreads();

    const uint32_t path_cap = kc + 2;  // root + at most one start per cell
    const bool direct = !all_done && s_conflict == 0;  // uniform (shared)
    if (direct) {
        // ---- confluent recording: the start in cell k+1 is the common successor of cell k, so the
        // orbit is read off without any pointer chasing: path[0] = root (acts as cell 1),
        // path[k] = succ[k] up to the first cell whose successor is END
        const uint32_t endc = s_endcell;
        for (uint32_t k = tid + 1; k < path_cap; k += NT)
            gst(w_path + k, k < endc ? gld(w_succ + k) : END);
        __syncthreads();
    }
    // ---- otherwise: orbit of the root by pointer doubling over the visited nodes:
    // path[m + 2^r] = J_r[path[m]],  J_{r+1} = J_r o J_r (double-buffered)
    uint32_t *ja = w_ja, *jb = w_jb;
    // (a) in LDS when the visited nodes fit (they do unless the recording is pathological): jump tables
    // over the nodes' indices in the visited list, 16 bits each — log2(cells) rounds of LDS reads and
    // barriers, ~10 us instead of the ~130 us the same rounds cost through L2
    const uint32_t lds_cap = lds_entries;  // uint16 entries of dynamic LDS
    const bool in_lds = !all_done && !direct && path_cap + 2 * (count + 1) <= lds_cap && count < 0xFFFFu;
    if (in_lds) {
        uint16_t *pth = lds_orbit;              // [path_cap]
        uint16_t *la = pth + path_cap;          // [count + 1]
        uint16_t *lb = la + (count + 1);        // [count + 1]
        const uint16_t ENDC = static_cast<uint16_t>(count);
        for (uint32_t idx = tid; idx < count; idx += NT) {
            const uint32_t nx = gld(w_ja + gld(w_list + idx));
            la[idx] = nx == END ? ENDC : static_cast<uint16_t>(nx < base_d ? nx : gld(w_jb + nx));  // seeds: index == id
        }
        if (tid == 0) { la[count] = ENDC; lb[count] = ENDC; pth[0] = 0; }
        __syncthreads();
        for (uint32_t span = 1; span < path_cap; span <<= 1) {
            for (uint32_t mI = tid; mI < span && mI + span < path_cap; mI += NT) pth[mI + span] = la[pth[mI]];
            for (uint32_t idx = tid; idx < count; idx += NT) lb[idx] = la[la[idx]];
            __syncthreads();
            uint16_t *t = la; la = lb; lb = t;
        }
        for (uint32_t k = tid; k < path_cap; k += NT)
            gst(w_path + k, pth[k] == ENDC ? END : gld(w_list + pth[k]));
        __syncthreads();
    }
    // (b) through global memory otherwise
    if (!all_done && !direct && !in_lds) {
    constexpr int kKeep = 4;  // visited ids (and their current jump) kept in registers
    uint32_t vk[kKeep], jk[kKeep];
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
        const uint32_t idx = tid + j * NT;
        vk[j] = (idx < count) ? gld(w_list + idx) : END;
    }
#pragma unroll
    for (int j = 0; j < kKeep; ++j) jk[j] = gld(ja + vk[j]);
    for (uint32_t span = 1; span < path_cap; span <<= 1) {
        for (uint32_t mI = tid; mI < span && mI + span < path_cap; mI += NT)
            gst(w_path + mI + span, gld(ja + gld(w_path + mI)));
#pragma unroll
        for (int j = 0; j < kKeep; ++j) jk[j] = gld(ja + jk[j]);  // J[J[v]], one round trip
#pragma unroll
        for (int j = 0; j < kKeep; ++j) gst(jb + vk[j], jk[j]);
        for (uint32_t idx = tid + kKeep * NT; idx < count; idx += NT) {
            const uint32_t v = gld(w_list + idx);
            gst(jb + v, gld(ja + gld(ja + v)));
        }
        __syncthreads();
        uint32_t *t = ja; ja = jb; jb = t;
    }
    }  // !direct && !in_lds
    if (!peaks_done) stamp(1);  // orbit extracted

    // ---- peak list: path[k] (k >= 1) starts at s in cell c; pushes fill
    // peaks[cell(prev) .. c-2] with s and peaks[c-1] with u = firstT(s)
    if (!peaks_done) {
    if (tid == 0) { s_plen = 1; s_fit = 0ull; }
    __syncthreads();
    }
    unsigned long long fit_local = 0;
    for (uint32_t k = tid; !peaks_done && k < path_cap; k += NT) {
        const uint32_t v = gld(w_path + k);
        if (v == END) continue;
        uint32_t cell;
        const uint32_t sv = node_start(v, &cell);
        const uint32_t u = gld(w_u + v);
        const bool is_last = (k + 1 >= path_cap) || gld(w_path + k + 1) == END;
        if (k == 0) {
            if (peaks_cap > 0) peaks[0] = u;
            if (!is_last && u + spr < wl32) ++fit_local;
            if (is_last) s_plen = 1;
            continue;
        }
        uint32_t pcell;
        (void)node_start(gld(w_path + k - 1), &pcell);
        const uint32_t c_prev = (k - 1 == 0) ? 1u : pcell;  // the root leaves one entry
        const uint32_t c = cell;                             // = s / spr for k >= 1
        for (uint32_t qv = c_prev; qv + 1 < c; ++qv)
            if (qv < peaks_cap) peaks[qv] = sv;
        if (c - 1 < peaks_cap) peaks[c - 1] = u;
        if (sv + spr < wl32) fit_local += c - c_prev - 1;
        if (!is_last && u + spr < wl32) ++fit_local;  // the last peak is dropped
        if (is_last) s_plen = c;
    }
    if (!peaks_done) {
    {
        uint32_t f32v = static_cast<uint32_t>(fit_local);
        for (int d = 32; d >= 1; d >>= 1) f32v += __shfl_down(f32v, d, 64);
        if ((tid & 63) == 0 && f32v) atomicAdd(&s_fit, static_cast<unsigned long long>(f32v));
    }
    __syncthreads();
    }
    if (tid == 0) {
        const uint32_t len = s_plen;
        const bool few = len < 5;  // decode.rs:112-118
        const uint64_t rows = s_fit;
        res->status = few ? 1 : 0;
        res->reason = few ? 2 : 0;
        res->n_sync = len;
        res->n_rows = few ? 0 : static_cast<uint32_t>(rows);
        res->work_len = gq.work_len;
        res->n_out = few ? 0 : rows * 2080u;
        flags[1] = 2u;  // the global-memory kernel ran
        flags[0] = 0u;
        flags[5] = 0u;
        flags[2] = nt_cap;
        flags[3] = n_nodes;
        flags[4] = all_done ? n_all : count;
        // orbit: 1 read off directly, 2 doubling in LDS over the visited nodes, 0 the same through L2, 3 doubling over all nodes (alg 1)
        flags[6] = all_done ? 3u : direct ? 1u : (in_lds ? 2u : 0u);
        if (all_done) flags[12] = 0u;  // no breadth-first levels
        flags[11] = flags7;  // candidates k_sync_words settled with exact window maxima; re-armed
        flags[7] = 0u;
    }
    stamp(2);  // peaks written
}

}  // namespace

uint32_t sync_group_size() { return GS; }
uint32_t sync_chunk_groups() { return kChunkGroups; }
uint32_t sync_slot_cap() { return kSlotCap; }

void group_max(hipStream_t s, const float *corr, uint64_t n_corr, GroupMax *gm)
{
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    if (ng == 0) return;
    hipLaunchKernelGGL(k_group_max, dim3((ng + 255) / 256), dim3(256), 0, s, corr, n_corr, gm, ng);
}

void sync_nodes(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, uint32_t pw,
                uint32_t spr, uint32_t md, bool fast, bool use_corr, const LaunchSwitches &sw)
{
    if (call.count == 0 || max_w <= 38ull * pw) return;
    const uint64_t n_corr = max_w - 38ull * pw;
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    const uint32_t chunks = (ng + kChunkGroups - 1) / kChunkGroups;
    const uint32_t r = md / GS;
    const size_t lds = words_win_ofs(r) + static_cast<size_t>(kNodesWaves) * nodes_window(pw) * sizeof(float);
    const dim3 grid(chunks, call.count);
    const uint32_t wneed = GS + 38u * pw - 1u;
    const bool dpp = sw.words_dpp;  // (APTGPU_WORDS_DPP, read at plan creation)
#define APT_WORDS_LAUNCH(NL, PWC)                                                                                       \
    do {                                                                                                                \
        if (dpp)                                                                                                        \
            hipLaunchKernelGGL((k_sync_words<NL, PWC, true>), grid, dim3(kNodesThreads), lds, s, call, d_slots, pw, r,  \
                               fast ? 1 : 0, use_corr ? 1 : 0);                                                         \
        else                                                                                                            \
            hipLaunchKernelGGL((k_sync_words<NL, PWC, false>), grid, dim3(kNodesThreads), lds, s, call, d_slots, pw, r, \
                               fast ? 1 : 0, use_corr ? 1 : 0);                                                         \
    } while (0)
    if (pw == 3) APT_WORDS_LAUNCH(3, 3);        // standard profile
    else if (pw == 4) APT_WORDS_LAUNCH(4, 4);   // fast profile
    else if (pw == 5) APT_WORDS_LAUNCH(4, 5);   // slow profile
    else if (wneed <= 192) APT_WORDS_LAUNCH(3, 0);
    else if (wneed <= 256) APT_WORDS_LAUNCH(4, 0);
    else APT_WORDS_LAUNCH(0, 0);
#undef APT_WORDS_LAUNCH
    hipLaunchKernelGGL(k_sync_slots, grid, dim3(kNodesThreads), 0, s, call, d_slots, pw, r, spr / GS);
}

// node-terminal capacity and uint32 words of scratch k_sync_orbit needs for a work signal of w samples
uint32_t sync_nt_cap(uint64_t w)
{
    const uint64_t ng = (w + GS - 1) / GS;
    const uint64_t chunks = (ng + kChunkGroups - 1) / kChunkGroups;
    return static_cast<uint32_t>(chunks * kSlotCap);
}

size_t sync_orbit_ws_words(uint64_t w, uint32_t spr)
{
    const uint64_t nt_cap = sync_nt_cap(w);
    const uint64_t kc = (spr ? w / spr : 0) + 2;
    const uint64_t node_cap = 1 + kc + nt_cap + 1;
    return 4 * node_cap + (node_cap / 32 + 1) + (kc + 2) + (kc + 3) + 64;
}

// can this device give a workgroup the 128 KB of dynamic LDS the closure-free orbit kernel wants?  Asked once per device
// (the attribute is a per-device property of the function); where it cannot — an ARCH override, a smaller part — the
// 24 KB closure form serves every recording
static bool orbit_all_nodes_available(uint32_t bytes)
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> state[kMaxDevices];  // 0 unknown, 1 yes, 2 no
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    int st = state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        int max_lds = 0;
        const bool ok = hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
                        max_lds >= static_cast<int>(bytes) &&
                        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sync_orbit_global<kOrbitThreadsMax>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess;
        (void)hipGetLastError();
        st = ok ? 1 : 2;
        state[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}

void sync_orbit(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t md,
                uint32_t pw, int force, const LaunchSwitches &sw)
{
    if (call.count == 0) return;
    // force: 0 = parallel picker, 1 = sequential walk over the terminal words
    // 24 KB of LDS for the doubling path's jump tables: a recording's visited nodes (about one per image row plus
    // the seeds) fit unless it is hours long; the kernel falls back to tables in global memory when they do not
    // (APTGPU_ORBIT_LDS=0, tests: always through global memory)
    const uint32_t kOrbitLdsEntries = sw.orbit_lds ? 12288u : 0u;
    // the default: successors of all nodes + doubling from the root, no breadth-first closure; 128 KB of LDS
    // (recordings whose tables do not fit take the closure path inside the same launch).  APTGPU_ORBIT_ALG=0 (A/B
    // switch, tests), APTGPU_ORBIT_LDS=0, or a device without 128 KB of LDS per workgroup: the closure path for every
    // recording.  (All switches are read at plan creation: LaunchSwitches.)
    constexpr uint32_t kAllEntries = 65536u;  // uint16 entries: 128 KB
    const int alg = (sw.orbit_alg == 0 || kOrbitLdsEntries == 0u || !orbit_all_nodes_available(kAllEntries * sizeof(uint16_t))) ? 0 : 1;
    if (sw.orbit_threads == 256) {
        hipLaunchKernelGGL(k_sync_orbit_global<256>, dim3(call.count), dim3(256), kOrbitLdsEntries * sizeof(uint16_t), s,
                           call, d_slots, spr, md, pw, force == 1 ? 1 : 0, kOrbitLdsEntries, 0);
    } else if (alg == 1) {
        hipLaunchKernelGGL(k_sync_orbit_global<kOrbitThreadsMax>, dim3(call.count), dim3(kOrbitThreadsMax), kAllEntries * sizeof(uint16_t), s,
                           call, d_slots, spr, md, pw, force == 1 ? 1 : 0, kAllEntries, 1);
    } else {
        hipLaunchKernelGGL(k_sync_orbit_global<kOrbitThreadsMax>, dim3(call.count), dim3(kOrbitThreadsMax), kOrbitLdsEntries * sizeof(uint16_t), s,
                           call, d_slots, spr, md, pw, force == 1 ? 1 : 0, kOrbitLdsEntries, 0);
    }
}

}  // namespace apt::gpu
