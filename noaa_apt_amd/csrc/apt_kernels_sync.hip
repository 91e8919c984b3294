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
// Kernels (GS = 52 correlation positions per group; md = 32*pw groups, spr = 40*pw groups):
//   k_group_max   corr -> GM (unfused path only; the fused front end emits GM itself)
//   k_sync_nodes  coarse: a group can hold a terminal only if GM[g] is not exceeded by the
//                 next md/GS-1 group maxima; fine: exact test on the few candidates;
//                 emits terminal words (fallback input) and ordered node-terminal lists
//   k_sync_orbit  one workgroup: gathers the node terminals into LDS, builds the
//                 functional graph over start nodes, extracts the orbit of the root by
//                 pointer doubling, writes the peak list and the result record; falls
//                 back to a sequential walk over the terminal words when a capacity is
//                 exceeded (pathological inputs, very long recordings).
#include "apt_kernels.hpp"

#include <hip/hip_runtime.h>

namespace apt::gpu {

namespace {

constexpr int GS = 52;
constexpr uint64_t kGroupMask = (1ull << GS) - 1;
constexpr float kNegInf = -__builtin_huge_valf();

// ------------------------------------------------------------------ k_group_max
__global__ void __launch_bounds__(256)
k_group_max(const float *__restrict__ corr, uint64_t n_corr, float *__restrict__ gm, uint32_t ng)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const uint64_t base = static_cast<uint64_t>(g) * GS;
    float mx = kNegInf;
    for (int o = 0; o < GS; ++o) {
        const uint64_t i = base + o;
        if (i < n_corr) {
            float v = corr[i];
            if (i == 0 && !(v > 0.f)) v = 0.f;
            mx = fmaxf(mx, v);
        }
    }
    gm[g] = mx;
}

// ------------------------------------------------------------------ k_sync_nodes
constexpr int kNodesThreads = 128;
constexpr int kChunkGroups = 128;  // own groups per workgroup
constexpr int kSlotCap = 64;       // node terminals kept per chunk before "overflow"

__global__ void __launch_bounds__(kNodesThreads)
k_sync_nodes(const float *__restrict__ gm, uint32_t ng, const float *__restrict__ corr,
             uint64_t n_corr, uint32_t r_groups /* md/GS */, uint32_t grid_groups /* spr/GS */,
             uint64_t *__restrict__ words_out, uint32_t *__restrict__ slot_nt,
             uint32_t *__restrict__ slot_cnt, uint32_t *__restrict__ flags)
{
    // window of groups [gw0, gw0 + nwin): gw0 = g0 - R - 1, nwin = CG + R + 1.  LDS is sized
    // at launch for the actual R (4.4 KB at R = 96) so these workgroups fit beside the front
    // end of the next recording, which leaves only ~5 KB of LDS free per CU.
    extern __shared__ uint64_t lds_nodes[];
    const int R = static_cast<int>(r_groups);
    uint64_t *s_words = lds_nodes;                                           // [CG + R + 1]
    float *s_gm = reinterpret_cast<float *>(s_words + (kChunkGroups + R + 1));  // [CG + 2R + 2]
    float *s_wm = s_gm + (kChunkGroups + 2 * R + 2);                         // [CG + R + 1]
    uint16_t *s_cand = reinterpret_cast<uint16_t *>(s_wm + (kChunkGroups + R + 1));  // [CG + R + 1]
    __shared__ uint32_t s_ncand;
    __shared__ uint32_t s_scan[kNodesThreads / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kChunkGroups;
    const int64_t gw0 = g0 - R - 1;
    const int nwin = kChunkGroups + R + 1;
    const uint64_t md = static_cast<uint64_t>(R) * GS;

    if (tid == 0) s_ncand = 0;
    for (int q = tid; q < nwin + R; q += kNodesThreads) {
        const int64_t g = gw0 + q;
        s_gm[q] = (g >= 0 && g < static_cast<int64_t>(ng)) ? gm[g] : kNegInf;
    }
    __syncthreads();

    // coarse: WM[g] = max(GM[g+1 .. g+R-1]) — the full groups inside every window of group g
    for (int q = tid; q < nwin; q += kNodesThreads) {
        const int64_t g = gw0 + q;
        float w0 = kNegInf, w1 = kNegInf, w2 = kNegInf, w3 = kNegInf;
        int d = 1;
#pragma unroll 4
        for (; d + 3 < R; d += 4) {
            w0 = fmaxf(w0, s_gm[q + d]);
            w1 = fmaxf(w1, s_gm[q + d + 1]);
            w2 = fmaxf(w2, s_gm[q + d + 2]);
            w3 = fmaxf(w3, s_gm[q + d + 3]);
        }
        for (; d < R; ++d) w0 = fmaxf(w0, s_gm[q + d]);
        const float wm = fmaxf(fmaxf(w0, w1), fmaxf(w2, w3));
        s_wm[q] = wm;
        s_words[q] = 0ull;
        const bool valid = g >= 0 && g < static_cast<int64_t>(ng);
        if (valid && !(wm > s_gm[q])) s_cand[atomicAdd(&s_ncand, 1u)] = static_cast<uint16_t>(q);
    }
    __syncthreads();

    // fine: exact terminal test for the candidate groups; each wave takes four candidates at
    // a time so their eight corr loads are in flight together
    const uint32_t ncand = s_ncand;
    constexpr int kBatch = 4;
    for (uint32_t c0 = wave * kBatch; c0 < ncand; c0 += kBatch * (kNodesThreads / 64)) {
        int qv[kBatch];
        float cv[kBatch], c2v[kBatch];
        bool inv[kBatch];
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            const uint32_t ci = c0 + e;
            qv[e] = (ci < ncand) ? s_cand[ci] : -1;
            const int64_t g = gw0 + (qv[e] < 0 ? 0 : qv[e]);
            const uint64_t i = static_cast<uint64_t>(g) * GS + lane;
            inv[e] = qv[e] >= 0 && lane < GS && i < n_corr;
            cv[e] = kNegInf;
            c2v[e] = kNegInf;
            if (inv[e]) {
                cv[e] = corr[i];
                if (i == 0 && !(cv[e] > 0.f)) cv[e] = 0.f;
                if (i + md < n_corr) c2v[e] = corr[i + md];
            }
        }
#pragma unroll
        for (int e = 0; e < kBatch; ++e) {
            if (qv[e] < 0) continue;  // wave-uniform
            // suffix max over lanes > lane (rest of this group)
            float sfx = cv[e];
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_down(sfx, d, 64);
                if (lane + d < 64) sfx = fmaxf(sfx, o);
            }
            float sfx_ex = __shfl_down(sfx, 1, 64);
            if (lane == 63) sfx_ex = kNegInf;
            // prefix max over lanes <= lane of the group md positions ahead
            float pfx = c2v[e];
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_up(pfx, d, 64);
                if (lane >= d) pfx = fmaxf(pfx, o);
            }
            const float wmax = fmaxf(fmaxf(sfx_ex, s_wm[qv[e]]), pfx);
            const bool term = inv[e] && !(wmax > cv[e]);
            const unsigned long long word = __ballot(term) & kGroupMask;
            if (lane == 0) s_words[qv[e]] = word;
        }
    }
    __syncthreads();

    // node terminals of the own groups: heads, terminals whose (t - md - 1) is a terminal,
    // terminals on the grid
    const int q = tid + R + 1;  // own group index inside the window
    const int64_t g = g0 + tid;
    uint64_t nw = 0;
    if (g < static_cast<int64_t>(ng)) {
        const uint64_t w = s_words[q];
        const uint64_t prev_bit = s_words[q - 1] >> (GS - 1);
        const uint64_t heads = w & ~(((w << 1) | prev_bit) & kGroupMask);
        const uint64_t shifted = ((s_words[q - R] << 1) | (s_words[q - R - 1] >> (GS - 1))) & kGroupMask;
        const uint64_t on_grid = (g % grid_groups == 0) ? 1ull : 0ull;
        nw = w & (heads | shifted | on_grid);
        words_out[g] = w;
    }
    // ordered compaction of the node-terminal positions of this chunk
    const uint32_t cnt = static_cast<uint32_t>(__popcll(nw));
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
    for (int wv = 0; wv < kNodesThreads / 64; ++wv) total += s_scan[wv];
    uint32_t ofs = base + inc - cnt;
    uint64_t bitsleft = nw;
    while (bitsleft) {
        const int b = __ffsll(static_cast<long long>(bitsleft)) - 1;
        bitsleft &= bitsleft - 1;
        if (ofs < kSlotCap)
            slot_nt[static_cast<uint64_t>(blockIdx.x) * kSlotCap + ofs] =
                static_cast<uint32_t>(static_cast<uint64_t>(g) * GS + b);
        ++ofs;
    }
    if (tid == 0) {
        slot_cnt[blockIdx.x] = total;
        if (total > kSlotCap) atomicOr(&flags[0], 1u);
    }
}

// ------------------------------------------------------------------ k_sync_orbit
constexpr int kOrbitThreads = 1024;
constexpr int kMaxLevels = 64;    // breadth-first levels before giving up on the fast path
// capacities of the LDS-resident kernel (the global-memory kernel has none)
constexpr int kNtCap = 16384;     // node terminals held in LDS
constexpr int kCellCap = 8192;    // image rows (grid cells)
constexpr int kNodeCap = kNtCap + kCellCap + 2;
constexpr int kListCap = 8192;    // nodes the marked (reachable) set may hold

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
__device__ void orbit_walk52(const uint64_t *__restrict__ words, const OrbitGeom &gq,
                             uint32_t *__restrict__ peaks, uint32_t peaks_cap,
                             Result *__restrict__ res)
{
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
        u = first_terminal52(words, n_groups, s);
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

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- LDS-resident picker: fastest, bounded capacity (147 KB of LDS, one workgroup)
__global__ void __launch_bounds__(kOrbitThreads)
k_sync_orbit_lds(const uint64_t *__restrict__ words, const uint32_t *__restrict__ slot_nt,
             const uint32_t *__restrict__ slot_cnt, uint32_t n_chunks,
             uint32_t *__restrict__ flags, OrbitGeom gq, uint32_t *__restrict__ peaks,
             uint32_t peaks_cap, Result *__restrict__ res, int force_walk)
{
    extern __shared__ uint32_t lds_u32[];
    uint32_t *s_nt = lds_u32;                                         // [kNtCap] node terminals
    uint32_t *s_mark = s_nt + kNtCap;                                 // [kNodeCap/32 + 1] visited bits
    uint16_t *s_ja = reinterpret_cast<uint16_t *>(s_mark + (kNodeCap / 32 + 1));  // [kNodeCap] next / jump
    uint16_t *s_list = s_ja + kNodeCap;                               // [kListCap] visited node ids
    uint16_t *s_path = s_list + kListCap;                             // [kCellCap + 2]
    __shared__ uint32_t s_wave[kOrbitThreads / 64];
    __shared__ uint32_t s_total, s_count, s_plen, s_bad;
    __shared__ unsigned long long s_fit;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t n_corr = gq.n_corr;
    const uint32_t spr = gq.spr, md = gq.md;
    // s / spr for s < 2^32 by multiply-high with floor(2^32/spr) and two corrections
    const uint32_t spr_magic = static_cast<uint32_t>((1ull << 32) / spr);
    auto div_spr = [&](uint32_t s) -> uint32_t {
        uint32_t q = __umulhi(s, spr_magic);
        uint32_t r = s - q * spr;
        if (r >= spr) { ++q; r -= spr; }
        if (r >= spr) ++q;
        return q;
    };
    const uint64_t t_begin = __builtin_readcyclecounter();
    auto stamp = [&](int k) {
        if (tid == 0) flags[8 + k] = static_cast<uint32_t>(__builtin_readcyclecounter() - t_begin);
    };

    // number of grid cells that can hold a start: c in [2, kc]
    const uint64_t kc = n_corr ? (n_corr - 1) / spr : 0;
    bool walk = force_walk == 1 || flags[0] != 0 || n_corr == 0 || gq.work_len >= (1ull << 31);
    // beyond the LDS capacities the global-memory kernel (launched next) takes over
    bool defer = !walk && (force_walk == 2 || kc + 2 > kCellCap || n_chunks > kOrbitThreads * 16);

    // ---- gather the per-chunk node-terminal lists into one sorted LDS array
    uint32_t total = 0;
    if (!walk && !defer) {
        const uint32_t per = (n_chunks + kOrbitThreads - 1) / kOrbitThreads;
        const uint32_t c_lo = tid * per;
        uint32_t mine = 0;
        for (uint32_t e = 0; e < per; ++e)
            if (c_lo + e < n_chunks) mine += slot_cnt[c_lo + e];
        uint32_t inc = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int wv = 0; wv < wave; ++wv) base += s_wave[wv];
        if (tid == kOrbitThreads - 1) s_total = base + inc;
        __syncthreads();
        total = s_total;
        if (total > kNtCap) {
            defer = true;  // uniform: s_total is shared
        } else if (total == 0) {
            walk = true;
        } else {
            uint32_t ofs = base + inc - mine;
            for (uint32_t e = 0; e < per; ++e) {
                const uint32_t ch = c_lo + e;
                if (ch >= n_chunks) break;
                const uint32_t cnt = (per == 1) ? mine : slot_cnt[ch];
                const uint4 *src = reinterpret_cast<const uint4 *>(slot_nt + static_cast<uint64_t>(ch) * kSlotCap);
                uint4 v[kSlotCap / 4];
#pragma unroll
                for (int j = 0; j < kSlotCap / 4; ++j)
                    if (static_cast<uint32_t>(4 * j) < cnt) v[j] = src[j];
#pragma unroll
                for (int j = 0; j < kSlotCap / 4; ++j) {
                    if (static_cast<uint32_t>(4 * j) < cnt) s_nt[ofs + 4 * j] = v[j].x;
                    if (static_cast<uint32_t>(4 * j + 1) < cnt) s_nt[ofs + 4 * j + 1] = v[j].y;
                    if (static_cast<uint32_t>(4 * j + 2) < cnt) s_nt[ofs + 4 * j + 2] = v[j].z;
                    if (static_cast<uint32_t>(4 * j + 3) < cnt) s_nt[ofs + 4 * j + 3] = v[j].w;
                }
                ofs += cnt;
            }
        }
        __syncthreads();
    }
    stamp(0);  // node terminals gathered
    if (defer) {
        if (tid == 0) flags[5] = 1u;  // k_sync_orbit_global runs the picker for this recording
        return;
    }

    // ---- nodes: 0 = root, 1 .. n_grid = grid cells 2 .. kc, then one per node terminal, END
    const uint32_t n_grid = kc >= 2 ? static_cast<uint32_t>(kc - 1) : 0;
    const uint32_t base_d = 1 + n_grid;
    const uint32_t n_nodes = base_d + total + 1;
    const uint32_t END = n_nodes - 1;
    const uint32_t nc32 = static_cast<uint32_t>(n_corr);
    const uint32_t wl32 = static_cast<uint32_t>(gq.work_len);

    auto node_start = [&](uint32_t v, uint32_t *cell) -> uint32_t {
        // start position and the cell used for the "(cell+1)*spr" term
        if (v == 0) { *cell = 1; return 0; }
        if (v < base_d) { *cell = v + 1; return (v + 1) * spr; }
        const uint32_t s = s_nt[v - base_d] + md + 1;
        *cell = div_spr(s);
        return s;
    };
    auto first_node_terminal = [&](uint32_t s) -> uint32_t {
        uint32_t ui = lower_bound_u32(s_nt, total, s);
        if (ui >= total) ui = total - 1;  // cannot happen (fact 3); keeps reads in bounds
        return ui;
    };
    auto next_of = [&](uint32_t v) -> uint32_t {
        uint32_t cell;
        const uint32_t s = node_start(v, &cell);
        if (s >= nc32) return END;
        const uint32_t ui = first_node_terminal(v == 0 ? 0u : s);
        const uint32_t a = s_nt[ui] + md + 1;
        const uint32_t b = (cell + 1) * spr;
        const uint32_t s2 = a > b ? a : b;
        if (s2 >= nc32) return END;
        return (a >= b) ? base_d + ui : cell;  // grid(cell+1) has id `cell`
    };

    // ---- reachable set by breadth-first marking from the root and every grid node: on real
    // recordings all chains merge within a step or two, so a handful of levels closes it.
    uint32_t count = 0;
    if (!walk) {
        for (uint32_t wq = tid; wq < kNodeCap / 32 + 1; wq += kOrbitThreads) s_mark[wq] = 0u;
        if (tid == 0) { s_bad = 0; }
        __syncthreads();
        for (uint32_t v = tid; v < base_d; v += kOrbitThreads) {
            if (v < kListCap) s_list[v] = static_cast<uint16_t>(v);
            atomicOr(&s_mark[v >> 5], 1u << (v & 31));
        }
        if (tid == 0) {
            s_count = base_d;
            if (base_d > kListCap) s_bad = 1;
        }
        __syncthreads();
        uint32_t lo = 0, hi = base_d;
        for (int level = 0; level < kMaxLevels && lo < hi && !s_bad; ++level) {
            for (uint32_t idx = lo + tid; idx < hi; idx += kOrbitThreads) {
                const uint32_t v = s_list[idx];
                const uint32_t nx = next_of(v);
                s_ja[v] = static_cast<uint16_t>(nx);
                if (nx != END) {
                    const uint32_t bit = 1u << (nx & 31);
                    if (!(atomicOr(&s_mark[nx >> 5], bit) & bit)) {
                        const uint32_t pos = atomicAdd(&s_count, 1u);
                        if (pos < kListCap) s_list[pos] = static_cast<uint16_t>(nx); else s_bad = 1;
                    }
                }
            }
            __syncthreads();
            lo = hi;
            hi = s_count < kListCap ? s_count : kListCap;
            __syncthreads();
        }
        if (lo < hi || s_bad) {  // not closed within the LDS budget: the global kernel takes over
            if (tid == 0) flags[5] = 1u;
            return;
        }
        count = hi;
    }
    stamp(1);  // reachable set closed

    if (walk) {
        if (wave == 0) orbit_walk52(words, gq, peaks, peaks_cap, res);
        if (tid == 0) {
            flags[1] = 1u;  // report which path ran
            flags[0] = 0u;  // re-arm the overflow flag for the next decode
        }
        return;
    }
    if (tid == 0) { s_ja[END] = static_cast<uint16_t>(END); s_path[0] = 0; }
    __syncthreads();

    // ---- orbit of the root by pointer doubling over the visited nodes only:
    // path[m + 2^r] = J_r[path[m]],  J_{r+1} = J_r o J_r  (staged in registers, in place)
    const uint32_t path_cap = static_cast<uint32_t>(kc) + 2;  // root + at most one start per cell
    constexpr int kPerThread = (kListCap + kOrbitThreads - 1) / kOrbitThreads;
    for (uint32_t span = 1; span < path_cap; span <<= 1) {
        for (uint32_t mI = tid; mI < span && mI + span < path_cap; mI += kOrbitThreads)
            s_path[mI + span] = s_ja[s_path[mI]];
        uint16_t nv[kPerThread];
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const uint32_t idx = tid + j * kOrbitThreads;
            nv[j] = (idx < count) ? s_ja[s_ja[s_list[idx]]] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const uint32_t idx = tid + j * kOrbitThreads;
            if (idx < count) s_ja[s_list[idx]] = nv[j];
        }
        __syncthreads();
    }
    stamp(2);  // orbit extracted

    // ---- peak list: path[k] (k >= 1) starts at s in cell c; pushes fill
    // peaks[cell(prev) .. c-2] with s and peaks[c-1] with u = firstT(s)
    if (tid == 0) { s_plen = 1; s_fit = 0ull; }
    __syncthreads();
    unsigned long long fit_local = 0;
    for (uint32_t k = tid; k < path_cap; k += kOrbitThreads) {
        const uint32_t v = s_path[k];
        if (v == END) continue;
        uint32_t cell;
        const uint32_t s = node_start(v, &cell);
        const uint32_t u = s_nt[first_node_terminal(v == 0 ? 0u : s)];
        const bool is_last = (k + 1 >= path_cap) || s_path[k + 1] == END;
        if (k == 0) {
            if (peaks_cap > 0) peaks[0] = u;
            if (!is_last && u + spr < wl32) ++fit_local;
            if (is_last) s_plen = 1;
            continue;
        }
        uint32_t pcell;
        (void)node_start(s_path[k - 1], &pcell);
        const uint32_t c_prev = (k - 1 == 0) ? 1u : pcell;  // the root leaves one entry
        const uint32_t c = cell;                             // = s / spr for k >= 1
        for (uint32_t qv = c_prev; qv + 1 < c; ++qv)
            if (qv < peaks_cap) peaks[qv] = s;
        if (c - 1 < peaks_cap) peaks[c - 1] = u;
        if (s + spr < wl32) fit_local += c - c_prev - 1;
        if (!is_last && u + spr < wl32) ++fit_local;  // the last peak is dropped
        if (is_last) s_plen = c;
    }
    atomicAdd(&s_fit, fit_local);
    __syncthreads();
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
        flags[1] = 0u;
        flags[0] = 0u;
        flags[2] = total;
        flags[3] = n_nodes;
        flags[4] = count;
    }
    stamp(3);  // peaks written
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
// One workgroup; every table in the global scratch `ws` (sized by sync_orbit_ws_words()), so
// the kernel needs almost no LDS and can run beside the next recording's front end.  Node
// terminals are looked up straight in the per-chunk slots k_sync_nodes wrote (no gather pass):
// chunk = position / (128*52), then the first entry >= s of that slot or of the next
// non-empty one.  Node ids: 0 root, 1..n_grid grid cells 2..kc, base_d + slot entry, END.
__global__ void __launch_bounds__(kOrbitThreads, 8)
k_sync_orbit_global(const uint64_t *__restrict__ words, const uint32_t *__restrict__ slot_nt,
             const uint32_t *__restrict__ slot_cnt, uint32_t n_chunks, uint32_t *__restrict__ flags,
             OrbitGeom gq, uint32_t *__restrict__ ws, uint32_t nt_cap, uint32_t *__restrict__ peaks,
             uint32_t peaks_cap, Result *__restrict__ res, int force_walk)
{
    // after the LDS-resident kernel (force_walk == 3) it runs only if that kernel handed the
    // recording over (flags[5]); on its own it is the default picker
    if (force_walk == 3) {
        if (flags[5] == 0) return;
        force_walk = 0;
    }
    __shared__ uint32_t s_count, s_plen, s_conflict, s_endcell;
    __shared__ unsigned long long s_fit;
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

    const uint64_t kc64 = n_corr ? (n_corr - 1) / spr : 0;  // cells that can hold a start: 2 .. kc
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
                    for (int t = 0; t < 4; ++t)
                        if (k0 + t < lim && vals[t] >= sv) { *uval = vals[t]; return ch * kSlotCap + k0 + t; }
                }
            }
        }
        *uval = nc32 - 1;  // cannot happen (fact 3)
        return nt_cap - 1;
    };
    auto node_start = [&](uint32_t v, uint32_t *cell) -> uint32_t {
        if (v == 0) { *cell = 1; return 0; }
        if (v < base_d) { *cell = v + 1; return (v + 1) * spr; }
        const uint32_t sv = slot_nt[v - base_d] + md + 1;
        *cell = div_spr(sv);
        return sv;
    };

    // ---- reachable set by breadth-first marking from the root and every grid node
    uint32_t count = 0;
    if (!walk) {
        for (uint32_t wq = tid; wq < n_nodes / 32 + 1; wq += kOrbitThreads) gst(w_mark + wq, 0u);
        for (uint32_t c = tid; c < kc + 3; c += kOrbitThreads) gst(w_succ + c, 0xFFFFFFFFu);
        if (tid == 0) { s_count = base_d; s_conflict = 0; s_endcell = kc + 2; }
        __syncthreads();
        for (uint32_t v = tid; v < base_d; v += kOrbitThreads) {
            gst(w_list + v, v);
            atomicOr(w_mark + (v >> 5), 1u << (v & 31));
        }
        __syncthreads();
        uint32_t lo = 0, hi = base_d;
        for (int level = 0; level < kMaxLevels && lo < hi; ++level) {
            for (uint32_t idx = lo + tid; idx < hi; idx += kOrbitThreads) {
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
                    if (!(atomicOr(w_mark + (nx >> 5), bit) & bit)) gst(w_list + atomicAdd(&s_count, 1u), nx);
                }
            }
            __syncthreads();
            lo = hi;
            hi = s_count;
            __syncthreads();
        }
        if (lo < hi) walk = true;  // not closed within the level budget: take the general path
        count = hi;
    }
    stamp(0);  // reachable set closed

    if (walk) {
        if (wave == 0) orbit_walk52(words, gq, peaks, peaks_cap, res);
        if (tid == 0) {
            flags[1] = 1u;  // report which path ran
            flags[0] = 0u;  // re-arm the overflow flag for the next decode
            flags[5] = 0u;
        }
        return;
    }
    if (tid == 0) { gst(w_ja + END, END); gst(w_jb + END, END); gst(w_path, 0u); }
    __syncthreads();

    const uint32_t path_cap = kc + 2;  // root + at most one start per cell
    const bool direct = s_conflict == 0;  // uniform (shared)
    if (direct) {
        // ---- confluent recording: the start in cell k+1 is the common successor of cell k, so the
        // orbit is read off without any pointer chasing: path[0] = root (acts as cell 1),
        // path[k] = succ[k] up to the first cell whose successor is END
        const uint32_t endc = s_endcell;
        for (uint32_t k = tid + 1; k < path_cap; k += kOrbitThreads)
            gst(w_path + k, k < endc ? gld(w_succ + k) : END);
        __syncthreads();
    }
    // ---- otherwise: orbit of the root by pointer doubling over the visited nodes:
    // path[m + 2^r] = J_r[path[m]],  J_{r+1} = J_r o J_r (double-buffered)
    uint32_t *ja = w_ja, *jb = w_jb;
    if (!direct) {
    constexpr int kKeep = 4;  // visited ids (and their current jump) kept in registers
    uint32_t vk[kKeep], jk[kKeep];
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
        const uint32_t idx = tid + j * kOrbitThreads;
        vk[j] = (idx < count) ? gld(w_list + idx) : END;
    }
#pragma unroll
    for (int j = 0; j < kKeep; ++j) jk[j] = gld(ja + vk[j]);
    for (uint32_t span = 1; span < path_cap; span <<= 1) {
        for (uint32_t mI = tid; mI < span && mI + span < path_cap; mI += kOrbitThreads)
            gst(w_path + mI + span, gld(ja + gld(w_path + mI)));
#pragma unroll
        for (int j = 0; j < kKeep; ++j) jk[j] = gld(ja + jk[j]);  // J[J[v]], one round trip
#pragma unroll
        for (int j = 0; j < kKeep; ++j) gst(jb + vk[j], jk[j]);
        for (uint32_t idx = tid + kKeep * kOrbitThreads; idx < count; idx += kOrbitThreads) {
            const uint32_t v = gld(w_list + idx);
            gst(jb + v, gld(ja + gld(ja + v)));
        }
        __syncthreads();
        uint32_t *t = ja; ja = jb; jb = t;
    }
    }  // !direct
    stamp(1);  // orbit extracted

    // ---- peak list: path[k] (k >= 1) starts at s in cell c; pushes fill
    // peaks[cell(prev) .. c-2] with s and peaks[c-1] with u = firstT(s)
    if (tid == 0) { s_plen = 1; s_fit = 0ull; }
    __syncthreads();
    unsigned long long fit_local = 0;
    for (uint32_t k = tid; k < path_cap; k += kOrbitThreads) {
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
    atomicAdd(&s_fit, fit_local);
    __syncthreads();
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
        flags[4] = count;
        flags[6] = direct ? 1u : 0u;
    }
    stamp(2);  // peaks written
}

}  // namespace

uint32_t sync_group_size() { return GS; }
uint32_t sync_chunk_groups() { return kChunkGroups; }
uint32_t sync_slot_cap() { return kSlotCap; }

void group_max(hipStream_t s, const float *corr, uint64_t n_corr, float *gm)
{
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    if (ng == 0) return;
    hipLaunchKernelGGL(k_group_max, dim3((ng + 255) / 256), dim3(256), 0, s, corr, n_corr, gm, ng);
}

void sync_nodes(hipStream_t s, const float *gm, const float *corr, uint64_t n_corr, uint32_t spr,
                uint32_t md, uint64_t *words, uint32_t *slot_nt, uint32_t *slot_cnt, uint32_t *flags)
{
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    if (ng == 0) return;
    const uint32_t chunks = (ng + kChunkGroups - 1) / kChunkGroups;
    const uint32_t r = md / GS;
    const size_t lds = static_cast<size_t>(kChunkGroups + r + 1) * 8 + static_cast<size_t>(kChunkGroups + 2 * r + 2) * 4 +
                       static_cast<size_t>(kChunkGroups + r + 1) * 4 + static_cast<size_t>(kChunkGroups + r + 1) * 2 + 16;
    hipLaunchKernelGGL(k_sync_nodes, dim3(chunks), dim3(kNodesThreads), lds, s, gm, ng, corr, n_corr,
                       r, spr / GS, words, slot_nt, slot_cnt, flags);
}

// uint32 words of scratch k_sync_orbit needs for a correlation of n_corr positions
size_t sync_orbit_ws_words(uint64_t n_corr, uint32_t spr)
{
    const uint64_t ng = (n_corr + GS - 1) / GS;
    const uint64_t chunks = (ng + kChunkGroups - 1) / kChunkGroups;
    const uint64_t nt_cap = chunks * kSlotCap;
    const uint64_t kc = (spr ? n_corr / spr : 0) + 2;
    const uint64_t node_cap = 1 + kc + nt_cap + 1;
    return 4 * node_cap + (node_cap / 32 + 1) + (kc + 2) + (kc + 3) + 64;
}

void sync_orbit(hipStream_t s, const uint64_t *words, const uint32_t *slot_nt,
                const uint32_t *slot_cnt, uint32_t *flags, uint64_t n_corr, uint64_t work_len,
                uint32_t spr, uint32_t md, uint32_t *ws, uint32_t *peaks, uint32_t peaks_cap,
                Result *res, int force)
{
    const uint32_t ng = static_cast<uint32_t>((n_corr + GS - 1) / GS);
    const uint32_t chunks = (ng + kChunkGroups - 1) / kChunkGroups;
    OrbitGeom gq{n_corr, work_len, spr, md};
    const size_t lds = static_cast<size_t>(kNtCap) * 4 + static_cast<size_t>(kNodeCap / 32 + 1) * 4 +
                       static_cast<size_t>(kNodeCap) * 2 + static_cast<size_t>(kListCap) * 2 +
                       static_cast<size_t>(kCellCap + 2) * 2 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync_orbit_lds),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        attr_set = true;
    }
    // force: 0 = global-memory kernel (default: tiny LDS, co-runs with the next front end, no
    // capacity limits), 1 = sequential walk, 2 = same as 0, 4 = LDS-resident kernel first (lowest
    // stand-alone latency) with the global-memory kernel as its overflow path
    if (force == 4) {
        hipLaunchKernelGGL(k_sync_orbit_lds, dim3(1), dim3(kOrbitThreads), lds, s, words, slot_nt, slot_cnt,
                           chunks, flags, gq, peaks, peaks_cap, res, 0);
        hipLaunchKernelGGL(k_sync_orbit_global, dim3(1), dim3(kOrbitThreads), 0, s, words, slot_nt,
                           slot_cnt, chunks, flags, gq, ws, chunks * kSlotCap, peaks, peaks_cap, res, 3);
    } else {
        hipLaunchKernelGGL(k_sync_orbit_global, dim3(1), dim3(kOrbitThreads), 0, s, words, slot_nt,
                           slot_cnt, chunks, flags, gq, ws, chunks * kSlotCap, peaks, peaks_cap, res,
                           force == 1 ? 1 : 0);
    }
}

}  // namespace apt::gpu
