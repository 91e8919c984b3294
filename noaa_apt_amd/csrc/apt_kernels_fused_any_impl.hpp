// apt_kernels_fused_any_impl.hpp — the run-time fused front end (k_fused_any) and its launch
// wrapper; included by the apt_kernels_fused_any_*.hip translation units (one launch shape each)
// so that the instantiations compile in parallel.  See apt_kernels_fused_any.hip.
#pragma once

#include "apt_kernels_fused_any_launch.hpp"
#include "apt_envelope.hpp"
#include "apt_sync_corr.hpp"

#include <atomic>
#include <type_traits>

#pragma clang fp contract(off)

namespace apt::gpu {


namespace {


constexpr int kGS = 52;  // correlation positions per group (apt_kernels_sync.hip)
constexpr float kNegInfAny = -__builtin_huge_valf();


typedef float f2 __attribute__((ext_vector_type(2)));

// padded LDS offset of logical element e relative to a block-aligned base (floor division)
template <int KPT>
__host__ __device__ constexpr int pad_ofs(int e)
{
    return e + (e >= 0 ? e / KPT : -((-e + KPT - 1) / KPT));
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// sign of the sync template at index j (decode.rs:188-198): + inside the seven high pulses
template <int PW>
__host__ __device__ constexpr bool sync_plus(int j)
{
    const int pulse = 2 * PW;
    if (j < pulse || j >= pulse + 14 * pulse) return false;
    return (((j - pulse) / pulse) & 1) == 1;
}


// T2C / PWC > 0: low-pass length and pixel width known at compile time (the standard profile:
// 37 taps, pw = 3) — stages 3 and 4 are then fully unrolled in the packed "sample-stationary"
// form of k_fused: every sample is broadcast against a PAIR of taps (one SGPR pair) or signs
// (neg modifiers) feeding a pair of accumulators, half the VALU instructions of the run-time
// loops.  0: run-time loops.
// One launch covers the recordings of a call (blockIdx.y; CallArgs by value, workspace pointers from the plan's slot
// table — as every k_fused; until round 4 this kernel was launched once per recording).
template <int NTHR, int KPT, typename XT, int T2C, int PWC>
__global__ void __launch_bounds__(NTHR)
k_fused_any(const CallArgs call, const SlotPtrs *__restrict__ slots, const float *__restrict__ table /*[l][tpp]*/,
            const float *__restrict__ h2, const f2 *__restrict__ h2p /*[t2+1] (h2[k-1], h2[k])*/,
            float cosphi2, float sinphi, float inv_sinphi, int want_gm, float gm_slack, AnyGeom G)
{
    const RecArgs rec = call.rec[blockIdx.y];
    const uint64_t w = rec.w;
    if (static_cast<uint64_t>(blockIdx.x) * G.own >= w) return;  // (the grid is sized for the call's longest recording)
    const XT *__restrict__ x = static_cast<const XT *>(rec.x);
    const uint64_t n = rec.n;
    const uint64_t n_corr = w - G.g;  // w >= 10 rows of samples > G (checked on the host)
    float *__restrict__ f_out = slots[rec.slot].f;
    GroupMax *__restrict__ gm_out = want_gm ? slots[rec.slot].gm : nullptr;
    extern __shared__ float lds[];
    float *T = lds;
    float *X = lds + G.off_x;
    float *A = lds + G.off_a;  // R, then F
    float *B = lds + G.off_b;  // D, then C
    auto P = [](int i) { return i + i / KPT; };  // padded LDS index (i >= 0)
    const int tid = threadIdx.x;
    const int64_t o0 = static_cast<int64_t>(blockIdx.x) * G.own;  // first owned work sample
    const int64_t t0 = o0 - G.pre;                                // work sample at tile index 0
    const int64_t kb = t0 > 0 ? t0 : 0;                           // first work sample that exists
    const int idx0 = static_cast<int>(kb - t0);
    // kb*m = X0*l + rb; the tile's first input sample, rounded down to a 16-byte boundary
    const uint64_t kbm = static_cast<uint64_t>(kb) * G.m;
    const uint64_t X0 = kbm / G.l;
    const uint32_t rb = static_cast<uint32_t>(kbm - X0 * G.l);
    const uint64_t xfirst = X0 + (rb ? 1 : 0);
    const uint64_t xs0 = xfirst & ~3ull;
    const uint32_t xrel0 = static_cast<uint32_t>(X0 - xs0);  // may wrap by -1..: used only with ceil >= 1 or rb == 0

    // ---- stage 0
    if (!G.table_in_global) {
        // the table is 16-byte aligned in HBM and in LDS: 16-byte copies, several in flight
        const uint32_t nt = G.l * G.tpp, nt4 = nt / 4;
        const float4 *t4 = reinterpret_cast<const float4 *>(table);
        float4 *l4 = reinterpret_cast<float4 *>(T);
#pragma unroll 4
        for (uint32_t q = tid; q < nt4; q += NTHR) l4[q] = t4[q];
        for (uint32_t q = 4 * nt4 + tid; q < nt; q += NTHR) T[q] = table[q];
    }
    if constexpr (sizeof(XT) == 4) {
        const float *xf = reinterpret_cast<const float *>(x);
        if ((reinterpret_cast<uintptr_t>(xf) & 15u) == 0) {
            for (uint32_t q = tid * 4; q < G.xt; q += NTHR * 4) {
                const uint64_t i = xs0 + q;
                float4 v;
                if (i + 3 < n) {
                    v = *reinterpret_cast<const float4 *>(xf + i);
                } else {
                    v.x = i < n ? xf[i] : 0.f;
                    v.y = i + 1 < n ? xf[i + 1] : 0.f;
                    v.z = i + 2 < n ? xf[i + 2] : 0.f;
                    v.w = i + 3 < n ? xf[i + 3] : 0.f;
                }
                *reinterpret_cast<float4 *>(X + q) = v;
            }
        } else {
            for (uint32_t q = tid; q < G.xt; q += NTHR) X[q] = xs0 + q < n ? xf[xs0 + q] : 0.f;
        }
    } else {
        // mono PCM16 payload (wav.rs:37: `*x as f32`)
        for (uint32_t q = tid; q < G.xt; q += NTHR) X[q] = xs0 + q < n ? static_cast<float>(x[xs0 + q]) : 0.f;
    }
    __syncthreads();

    // ---- stage 1: one output per thread and step, consecutive lanes = consecutive outputs.
    // k*m - X0*l = v;  x0 - X0 = c = ceil(v / l);  phase p = c*l - v.  One division for the
    // thread's first output, then v += NTHR*m is  c += step_q (+1),  p -= step_r (+l).
    {
        uint32_t c = 0, p = 0;
        bool primed = false;
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * NTHR;
            const int64_t k = t0 + idx;
            float sum = 0.f;
            if (idx >= idx0) {
                if (!primed) {
                    const uint32_t v = rb + static_cast<uint32_t>(idx - idx0) * G.m;
                    c = (v + G.l - 1) / G.l;
                    p = c * G.l - v;
                    primed = true;
                } else {
                    c += G.step_q;
                    if (p >= G.step_r) {
                        p -= G.step_r;
                    } else {
                        p += G.l - G.step_r;
                        c += 1;
                    }
                }
                if (k < static_cast<int64_t>(w)) {
                    const uint32_t cnt = G.jl_a + (p < G.jl_b ? 1u : 0u);
                    const float *row = (G.table_in_global ? table : T) + static_cast<size_t>(p) * G.tpp;
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
                        for (int e = 0; e < 8; ++e) sum = sum + tv[e] * xv[e];
                    }
                    for (; j < cnt; ++j) sum = sum + row[j] * xs[j];
                }
            }
            A[P(idx)] = sum;
        }
    }
    __syncthreads();

    // ---- stage 2: exactly rounded envelope, fast path when the whole wave is in range
    {
        float xr[KPT];
        bool in_range = inv_sinphi != 0.f;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * NTHR;
            const int64_t k = t0 + idx;
            const bool live = idx > 0 && k > 0 && k < static_cast<int64_t>(w);
            xr[i] = live ? envelope_radicand(A[P(idx - 1)], A[P(idx)], cosphi2) : 1.f;
            in_range = in_range && envelope_in_range(xr[i]);
        }
        if (__all(in_range)) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int idx = tid + i * NTHR;
                const int64_t k = t0 + idx;
                const bool live = idx > 0 && k > 0 && k < static_cast<int64_t>(w);
                B[P(idx)] = live ? envelope_fast(xr[i], sinphi, inv_sinphi) : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int idx = tid + i * NTHR;
                const int64_t k = t0 + idx;
                const bool live = idx > 0 && k > 0 && k < static_cast<int64_t>(w);
                B[P(idx)] = live ? envelope_general(xr[i], sinphi) : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- stage 3: KPT consecutive outputs per thread, taps ascending
    if constexpr (T2C > 0) {
        // sample D[b0 + e] meets output u at tap j = u - e, so the pair (u, u+1) takes
        // (h2[j], h2[j+1]) = h2p[j+1]; walking e downwards gives ascending taps per output
        const int b0 = static_cast<int>(G.pre) + tid * KPT;
        if (b0 < static_cast<int>(G.kt)) {
            constexpr int NP = KPT / 2;
            constexpr int DW = KPT + T2C - 1;  // samples e = KPT-1 ... -(T2C-1)
            constexpr int CH = 8;
            f2 fa[NP];
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) fa[pp] = (f2){0.f, 0.f};
            const float *src = B + P(b0);  // b0 is a multiple of KPT: element e sits at pad(e)
            static_for<0, (DW + CH - 1) / CH>([&](auto cc) {
                constexpr int hi = KPT - 1 - decltype(cc)::value * CH;
                float dv[CH];
                static_for<0, CH>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    constexpr int e = hi - q;
                    if constexpr (e >= -(T2C - 1)) dv[q] = src[pad_ofs<KPT>(e)];
                    else dv[q] = 0.f;
                });
                static_for<0, CH>([&](auto ee) {
                    constexpr int e = hi - decltype(ee)::value;
                    if constexpr (e >= -(T2C - 1)) {
                        const float d = dv[decltype(ee)::value];
                        static_for<0, NP>([&](auto pc) {
                            constexpr int pp = decltype(pc)::value;
                            constexpr int j = 2 * pp - e;  // tap of output 2pp; output 2pp+1: j+1
                            constexpr bool va = j >= 0 && j < T2C;
                            constexpr bool vb = j + 1 >= 0 && j + 1 < T2C;
                            if constexpr (va && vb) {
                                fa[pp] = fa[pp] + h2p[j + 1] * (f2){d, d};
                            } else if constexpr (va) {
                                fa[pp].x = fa[pp].x + h2[j] * d;
                            } else if constexpr (vb) {
                                fa[pp].y = fa[pp].y + h2[j + 1] * d;
                            }
                        });
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                A[P(b0) + 2 * pp] = fa[pp].x;
                A[P(b0) + 2 * pp + 1] = fa[pp].y;
            }
        }
    } else
    {
        const int b0 = static_cast<int>(G.pre) + tid * KPT;
        if (b0 < static_cast<int>(G.kt)) {
            float acc[KPT];
#pragma unroll
            for (int u = 0; u < KPT; ++u) acc[u] = 0.f;
            for (uint32_t j0 = 0; j0 < G.t2; j0 += KPT) {
                // 2*KPT logical elements from a block-aligned base (b0, j0 multiples of KPT)
                float win[2 * KPT];
                const float *src = B + P(b0 - static_cast<int>(j0) - KPT);
#pragma unroll
                for (int e = 0; e < 2 * KPT; ++e) win[e] = src[e + e / KPT];
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    if (j0 + jj < G.t2) {
                        const float hj = h2[j0 + jj];
#pragma unroll
                        for (int u = 0; u < KPT; ++u) acc[u] = acc[u] + win[KPT + u - jj] * hj;
                    }
                }
            }
            // A (R) was last read in stage 2, before the barrier above: safe to overwrite with F
#pragma unroll
            for (int u = 0; u < KPT; ++u) A[P(b0) + u] = acc[u];
        }
    }
    __syncthreads();

    // ---- stage 5a: owned F -> HBM
    for (uint32_t q = tid * 4; q < G.own; q += NTHR * 4) {
        const uint64_t k = static_cast<uint64_t>(o0) + q;
        const float *src = A + P(static_cast<int>(G.pre + q));  // 4 | KPT: the quad stays inside one block
        if (k + 3 < w) {
            *reinterpret_cast<float4 *>(f_out + k) = make_float4(src[0], src[1], src[2], src[3]);
        } else {
            for (int e = 0; e < 4; ++e)
                if (k + e < w) f_out[k + e] = src[e];
        }
    }
    if (gm_out == nullptr) return;  // no sync search wanted

    // ---- stage 4: bounds of the sync correlation's maximum per group of 52 positions (decode.rs:225-233).
    // Until round 4 this kernel evaluated the reference's whole chain at every position (38 pw additions: 114 / 152 /
    // 190) and handed the picker exact maxima.  Like the specialised kernels (apt_kernels_fused_impl.hpp, stage 4) it
    // now evaluates the correlation from pulse sums — sync_pulse_sum, sync_corr_from_pulses: 2 pw + 21 additions per
    // position whatever the pixel width — and widens the group's maximum by the rigorous rounding bound
    // |pulse-sum value - sequential chain| <= gm_slack * sum|F| over the group's window: k_sync_words prunes with lo
    // against hi and settles whatever the bounds leave open with the exact chain, so the picker's result does not
    // depend on them (tests/test_gpu_bounds.py).  A window that holds a non-finite F gets [-inf, +inf].
    const uint32_t pw = PWC > 0 ? static_cast<uint32_t>(PWC) : G.pulse / 2u;
    const uint32_t pulse = 2u * pw;
    float *AB = X;  // [NTHR] per-thread sums of |F| over the thread's KPT samples (the input tile is dead; xt >= NTHR)
    {
        // 4a: pulse sums of the thread's KPT positions -> B (D is dead), |F| partial sums -> AB
        const int b0 = static_cast<int>(G.pre) + tid * KPT;
        if (b0 < static_cast<int>(G.pre + G.own + 36u * pw)) {
            float bs[KPT];
            if constexpr (PWC > 0) {
                constexpr int NW = KPT + 2 * PWC - 1;
                float wv[NW];
                const float *src = A + P(b0);  // b0 is a multiple of KPT: element e sits at pad(e)
#pragma unroll
                for (int e = 0; e < NW; ++e) wv[e] = src[pad_ofs<KPT>(e)];
#pragma unroll
                for (int u = 0; u < KPT; ++u) bs[u] = sync_pulse_sum(pw, [&](uint32_t jj) { return wv[u + static_cast<int>(jj)]; });
            } else {
#pragma unroll
                for (int u = 0; u < KPT; ++u)
                    bs[u] = sync_pulse_sum(pw, [&](uint32_t jj) { return A[P(b0 + u + static_cast<int>(jj))]; });
            }
#pragma unroll
            for (int u = 0; u < KPT; ++u) B[P(b0) + u] = bs[u];
        }
        {
            float a = 0.f;
            const int s0 = static_cast<int>(G.pre) + tid * KPT;  // this thread's KPT samples of F (zero past the signal's end)
            if (s0 < static_cast<int>(G.kt)) {
#pragma unroll
                for (int u = 0; u < KPT; ++u) a = a + __builtin_fabsf(A[P(s0) + u]);
            }
            AB[tid] = a;
        }
    }
    __syncthreads();
    {
        // 4b: the correlation of the thread's owned positions from the pulse sums -> A (F is dead: it went to HBM above)
        const int b0 = static_cast<int>(G.pre) + tid * KPT;
        if (b0 < static_cast<int>(G.pre + G.own)) {
            float cvals[KPT];
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                if constexpr (PWC > 0) {
                    const float *src = B + P(b0);
                    cvals[u] = sync_corr_from_pulses([&](int k) { return src[pad_ofs<KPT>(u + k * 2 * PWC)]; });
                } else {
                    cvals[u] = sync_corr_from_pulses([&](int k) { return B[P(b0 + u + k * static_cast<int>(pulse))]; });
                }
            }
            // (region A was last read in 4a, before the barrier above)
#pragma unroll
            for (int u = 0; u < KPT; ++u) A[P(b0) + u] = cvals[u];
        }
    }
    __syncthreads();

    // ---- stage 5b: the groups' records.  NaNs are left out of the maximum (they show in the sum of |F|).
    for (uint32_t g = tid; g < G.own / kGS; g += NTHR) {
        const uint64_t k = static_cast<uint64_t>(o0) + static_cast<uint64_t>(g) * kGS;
        if (k >= n_corr) break;
        const int base = static_cast<int>(G.pre + g * kGS);
        float cv[kGS];
#pragma unroll
        for (int o = 0; o < kGS; ++o) cv[o] = A[P(base + o)];  // all 52 reads in flight
        float mx = kNegInfAny;
#pragma unroll
        for (int o = 0; o < kGS; ++o) {
            float v = cv[o];
            if (k + o == 0 && !(v > 0.f)) v = 0.f;  // the picker starts from the peak (0, 0.)
            if (k + o < n_corr) mx = fmaxf(mx, v);
        }
        // |F| over the group's window [base, base + 52 + 38 pw - 1): the threads whose KPT samples touch it
        const int t_lo = static_cast<int>(g * kGS) / KPT;
        const int t_hi = (static_cast<int>(g * kGS) + kGS + static_cast<int>(G.g) - 2) / KPT;
        float asum = 0.f;
        for (int t = t_lo; t <= t_hi && t < NTHR; ++t) asum = asum + AB[t];
        const float err = asum * gm_slack;
        const bool open = !(err < __builtin_huge_valf());  // NaN or Inf somewhere in the window
        gm_out[static_cast<uint64_t>(o0) / kGS + g] =
            open ? GroupMax{__builtin_huge_valf(), -__builtin_huge_valf()} : GroupMax{mx + err, mx - err};
    }
}

template <int NTHR, int KPT, int T2C, int PWC, typename XT>
void launch_any(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, const float *table,
                const float *h2, const float *h2p, float cosphi2, float sinphi, float inv_sinphi, bool want_gm,
                const AnyGeom &g, size_t lds)
{
    const float gm_slack = fused_gm_slack(g.pulse / 2u, g.gm_slack_scale);
    auto kern = k_fused_any<NTHR, KPT, XT, T2C, PWC>;
    // (a per-device property of the function: plans on several devices / threads pass through here)
    constexpr int kMaxDevices = 64;
    static std::atomic<size_t> attr_lds[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > attr_lds[dev].load(std::memory_order_acquire) && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        attr_lds[dev].store(lds, std::memory_order_release);
    }
    const unsigned tiles = static_cast<unsigned>((max_w + g.own - 1) / g.own);
    hipLaunchKernelGGL(kern, dim3(tiles, call.count), dim3(NTHR), lds, s, call, d_slots, table, h2,
                       reinterpret_cast<const f2 *>(h2p), cosphi2, sinphi, inv_sinphi, want_gm ? 1 : 0, gm_slack, g);
}


// all profiles x input types of one launch shape
template <int NT, int KP>
void launch_any_shape(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, bool pcm16,
                      const float *table, const float *h2, const float *h2p, float cosphi2, float sinphi,
                      float inv_sinphi, bool want_gm, const AnyGeom &g, size_t lds, int prof)
{
#define APT_ANY_LAUNCH(T2C, PWC)                                                                                \
    do {                                                                                                        \
        if (pcm16)                                                                                              \
            launch_any<NT, KP, T2C, PWC, int16_t>(s, call, d_slots, max_w, table, h2, h2p, cosphi2, sinphi,     \
                                                  inv_sinphi, want_gm, g, lds);                                 \
        else                                                                                                    \
            launch_any<NT, KP, T2C, PWC, float>(s, call, d_slots, max_w, table, h2, h2p, cosphi2, sinphi,       \
                                                inv_sinphi, want_gm, g, lds);                                   \
        return;                                                                                                 \
    } while (0)
    // the W-rate stages of the three stock profiles (default_settings.toml:108-140) at any input
    // rate are instantiated with compile-time lengths; anything else takes the run-time loops
    if (prof == 1) APT_ANY_LAUNCH(37, 3);
    if (prof == 2) APT_ANY_LAUNCH(43, 4);
    if (prof == 3) APT_ANY_LAUNCH(61, 5);
    APT_ANY_LAUNCH(0, 0);
#undef APT_ANY_LAUNCH
}

}  // namespace

}  // namespace apt::gpu
