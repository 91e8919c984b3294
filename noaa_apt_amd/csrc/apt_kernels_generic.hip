// apt_kernels_generic.hip — gfx950 kernels for ANY (l, m, tap-count) combination.
//
// One thread per output, reference summation order, every product and sum rounded
// separately (__fmul_rn/__fadd_rn are never contracted into FMAs), so each kernel is
// bit-identical to the reference loop it cites.  These are the fallback / A-B baseline
// for the fused specialised kernels in apt_kernels_fused.hip, and the path used when
// the caller asks for the intermediate "steps" (Context::step).
#include "apt_kernels.hpp"
#include "apt_envelope.hpp"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

// never fuse a*b+c: the reference rounds the product and the sum separately
#pragma clang fp contract(off)

namespace apt::gpu {

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap = 256u * 32u)
{
    uint64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return static_cast<unsigned>(g);
}

// ---------------------------------------------------------------------------------
// fast_resampling (dsp.rs:186-289) in polyphase closed form.
// Output k sits at t = off + k*m on the interpolated axis; its window holds input
// samples x0, x0+1, ... with x0 = ceil(k*m / l), multiplied by coeff[p + i*l],
// p = x0*l - k*m, while p + i*l <= 2*off; inputs at or beyond n are skipped (:257).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_resample_generic(const float *__restrict__ x, uint64_t n, const float *__restrict__ coeff,
                   uint32_t jlim, uint32_t l, uint32_t m, float *__restrict__ out, uint64_t w)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < w;
         k += stride) {
        const uint64_t km = k * m;
        uint64_t xi = (km + l - 1) / l;
        const uint32_t p = static_cast<uint32_t>(xi * l - km);
        float sum = 0.f;
        for (uint32_t j = p; j < jlim; j += l, ++xi) {
            if (xi < n) sum = __fadd_rn(sum, __fmul_rn(coeff[j], x[xi]));
        }
        out[k] = sum;
    }
}

// ---------------------------------------------------------------------------------
// fp16-tap variant of the polyphase resampler (BASELINE config 5; NOT bit-exact).
// Taps are stored phase-major as fp16 pre-scaled by 2^s (so the 1/L-sized taps use the
// fp16 normal range), samples are rounded to fp16 on the fly, and two taps at a time go
// through v_dot2_f32_f16 with f32 accumulation; the sum is scaled back by 2^-s (exact).
// Error sources: 11-bit taps and 11-bit samples -> about 1e-3 of the signal's peak.
// ---------------------------------------------------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(kBlock)
k_resample_f16taps(const float *__restrict__ x, uint64_t n, const h2 *__restrict__ hp /*[l][tp2]*/,
                   uint32_t tp2 /* tap pairs per phase */, uint32_t l, uint32_t m, float unscale,
                   float *__restrict__ out, uint64_t w)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < w;
         k += stride) {
        const uint64_t km = k * m;
        const uint64_t x0 = (km + l - 1) / l;
        const uint32_t p = static_cast<uint32_t>(x0 * l - km);
        const h2 *row = hp + static_cast<uint64_t>(p) * tp2;
        float acc = 0.f;
        for (uint32_t i = 0; i < tp2; ++i) {
            const uint64_t xi = x0 + 2ull * i;
            const float a = xi < n ? x[xi] : 0.f;
            const float b = xi + 1 < n ? x[xi + 1] : 0.f;
            const h2 xv = {static_cast<_Float16>(a), static_cast<_Float16>(b)};
            acc = __builtin_amdgcn_fdot2(xv, row[i], acc, false);
        }
        out[k] = acc * unscale;
    }
}

// ---------------------------------------------------------------------------------
// filter (dsp.rs:386-410) evaluated only at i = k*m (decimate, dsp.rs:294-307):
// out[k] = sum_{j < ntaps, j < i} x[i-j]*h[j], ascending j.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_fir_decimate(const float *__restrict__ x, const float *__restrict__ h, uint32_t ntaps,
               uint32_t m, float *__restrict__ out, uint64_t n_out)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n_out;
         k += stride) {
        const uint64_t i = k * m;
        const uint32_t jn = i < ntaps ? static_cast<uint32_t>(i) : ntaps;
        float sum = 0.f;
        for (uint32_t j = 0; j < jn; ++j) sum = __fadd_rn(sum, __fmul_rn(x[i - j], h[j]));
        out[k] = sum;
    }
}

// ---------------------------------------------------------------------------------
// demodulate (dsp.rs:350-383): y[0] = 0,
// y[i] = sqrt(x[i-1]^2 + x[i]^2 - x[i-1]*x[i]*cosphi2) / sinphi  in exactly that order.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float demod_pair(float prev, float curr, float cosphi2, float sinphi)
{
    const float a = __fadd_rn(__fmul_rn(prev, prev), __fmul_rn(curr, curr));
    const float b = __fmul_rn(__fmul_rn(prev, curr), cosphi2);
    // __fsqrt_rn is the *native approximate* sqrt in this HIP; __builtin_sqrtf and `/` are
    // IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt (set in the Makefile).
    return __builtin_sqrtf(__fsub_rn(a, b)) / sinphi;
}

__global__ void __launch_bounds__(kBlock)
k_demodulate(const float *__restrict__ x, uint64_t n, float cosphi2, float sinphi,
             float *__restrict__ y)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += stride) {
        y[i] = (i == 0) ? 0.f : demod_pair(x[i - 1], x[i], cosphi2, sinphi);
    }
}

// ---------------------------------------------------------------------------------
// cross-correlation with the +-1 sync template (decode.rs:225-233), sequential adds
// from 0.0 in template order: 2pw x (-), 7 x [2pw x (-), 2pw x (+)], 8pw x (-).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_correlate(const float *__restrict__ f, uint64_t n_corr, uint32_t pw, float *__restrict__ corr)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint32_t pulse = 2 * pw;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_corr;
         i += stride) {
        const float *p = f + i;
        float c = 0.f;
        for (uint32_t j = 0; j < pulse; ++j) c = __fsub_rn(c, *p++);
        for (int rep = 0; rep < 7; ++rep) {
            for (uint32_t j = 0; j < pulse; ++j) c = __fsub_rn(c, *p++);
            for (uint32_t j = 0; j < pulse; ++j) c = __fadd_rn(c, *p++);
        }
        for (uint32_t j = 0; j < 8 * pw; ++j) c = __fsub_rn(c, *p++);
        corr[i] = c;
    }
}

// ---------------------------------------------------------------------------------
// Terminal flags for the peak picker of find_sync (decode.rs:239-253).
//
// The reference's picker keeps (idx, val) of the last peak; while i - idx <= md it
// replaces the peak whenever corr[i] > val.  A tracking phase that starts at s therefore
// climbs the chain of strict prefix maxima until no larger value exists within md samples:
// it stops on the first "terminal" at or after s, where
//     T[i]  <=>  no j in (i, i+md] (j < n_corr) has corr[j] > corr[i].
// (The chain can never step over a terminal t: the record it would land on lies in t's
// window, so it would have to be <= corr[t] <= the value it exceeds.)
// corr[0] is clamped to max(corr[0], 0) because the picker starts from the peak (0, 0.).
//
// One workgroup per chunk of md positions: sliding-window max by the two-block method
// (suffix max of the chunk, prefix max of the following md samples), block-wide scans
// in LDS, 64 flags per wave packed by ballot.
// ---------------------------------------------------------------------------------
constexpr int kTermThreads = 1024;
constexpr float kNegInf = -__builtin_huge_valf();

// In-place inclusive max-scan of a[0..n) in LDS by the whole workgroup; `reverse` scans
// from the end (suffix max).  Strips of `per` contiguous elements per thread.
__device__ void block_scan_max(float *a, int n, bool reverse, float *wave_tot)
{
    const int tid = threadIdx.x;
    const int per = (n + kTermThreads - 1) / kTermThreads;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // logical index q in scan order -> physical index
    auto phys = [&](int q) { return reverse ? (n - 1 - q) : q; };

    const int q0 = tid * per;
    float run = kNegInf;
    for (int e = 0; e < per; ++e) {
        const int q = q0 + e;
        if (q < n) {
            run = fmaxf(run, a[phys(q)]);
            a[phys(q)] = run;
        }
    }
    // exclusive prefix of strip totals across the workgroup
    float inc = run;
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(inc, d, 64);
        if (lane >= d) inc = fmaxf(inc, o);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    float carry = kNegInf;
    for (int wv = 0; wv < wave; ++wv) carry = fmaxf(carry, wave_tot[wv]);
    const float up = __shfl_up(inc, 1, 64);
    if (lane > 0) carry = fmaxf(carry, up);
    for (int e = 0; e < per; ++e) {
        const int q = q0 + e;
        if (q < n) a[phys(q)] = fmaxf(a[phys(q)], carry);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kTermThreads)
k_terminals(const float *__restrict__ corr, uint64_t n_corr, uint32_t md,
            uint64_t *__restrict__ bits)
{
    extern __shared__ float lds[];
    float *own = lds;        // corr[b0 .. b0+md), later its suffix max
    float *nxt = lds + md;   // corr[b0+md .. b0+2md), later its prefix max
    __shared__ float wave_tot[kTermThreads / 64];

    const uint64_t b0 = static_cast<uint64_t>(blockIdx.x) * md;
    for (uint32_t i = threadIdx.x; i < 2 * md; i += kTermThreads) {
        const uint64_t g = b0 + i;
        float v = kNegInf;
        if (g < n_corr) {
            v = corr[g];
            if (g == 0 && !(v > 0.f)) v = 0.f;
            // a NaN is never a record (`corr > last` is false, decode.rs:250): a tracking phase passes
            // over it as if it held -inf; a phase that STARTS on one keeps it (k_orbit_walk)
            if (v != v) v = kNegInf;
        }
        lds[i] = v;
    }
    __syncthreads();

    block_scan_max(own, md, /*reverse=*/true, wave_tot);
    block_scan_max(nxt, md, /*reverse=*/false, wave_tot);

    // thread t handles positions t, t+1024, ...: a wave covers 64 consecutive flags
    const int iters = (md + kTermThreads - 1) / kTermThreads;
    for (int it = 0; it < iters; ++it) {
        const uint32_t i = it * kTermThreads + threadIdx.x;
        bool term = false;
        if (i < md && b0 + i < n_corr) {
            float v = corr[b0 + i];  // the scan overwrote own[]; L2-hot re-read
            if (b0 + i == 0 && !(v > 0.f)) v = 0.f;
            if (v != v) v = kNegInf;
            // window (i, i+md] = own[i+1 .. md) U nxt[0 .. i]
            const float wmax = fmaxf((i + 1 < md) ? own[i + 1] : kNegInf, nxt[i]);
            term = !(wmax > v);
        }
        const unsigned long long word = __ballot(term);
        if ((threadIdx.x & 63) == 0 && i < md) bits[(b0 + i) >> 6] = word;
    }
}

// ---------------------------------------------------------------------------------
// The orbit of the peak picker, one wave.
//   peaks = [firstT(0)]; len = 1
//   loop: s = max(last + md + 1, (len+1)*spr); stop if s >= n_corr
//         c = s / spr; push s (c - len - 1) times; push firstT(s); len = c
// where firstT(s) = first terminal >= s, found by a 64-word-wide scan of the bitmask.
// This is decode.rs:239-253 verbatim once tracking phases are replaced by firstT().
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t first_terminal(const uint64_t *__restrict__ bits,
                                                   uint64_t n_words, uint64_t s)
{
    const int lane = threadIdx.x & 63;
    uint64_t w0 = s >> 6;
    const uint32_t sh = static_cast<uint32_t>(s & 63);
    bool first = true;
    while (true) {
        const uint64_t wi = w0 + lane;
        uint64_t word = (wi < n_words) ? bits[wi] : 0ull;
        if (first && lane == 0) word &= (~0ull) << sh;
        const unsigned long long any = __ballot(word != 0ull);
        if (any) {
            const int src = __ffsll(static_cast<long long>(any)) - 1;
            const uint64_t pos = wi * 64 + (__ffsll(static_cast<long long>(word)) - 1);
            return __shfl(pos, src, 64);
        }
        first = false;
        w0 += 64;
        if (w0 >= n_words) return ~0ull;  // cannot happen: position n_corr-1 is a terminal
    }
}

__global__ void __launch_bounds__(64)
k_orbit_walk(const uint64_t *__restrict__ bits, const float *__restrict__ corr, uint64_t n_corr,
             uint64_t work_len, uint32_t spr, uint32_t md, uint32_t *__restrict__ peaks, uint32_t peaks_cap,
             Result *__restrict__ res)
{
    const int lane = threadIdx.x & 63;
    const uint64_t n_words = (n_corr + 63) >> 6;
    uint64_t len = 1;
    uint64_t u = 0;
    if (n_corr > 0) u = first_terminal(bits, n_words, 0);
    if (lane == 0 && peaks_cap > 0) peaks[0] = static_cast<uint32_t>(u);
    // Rows are peaks[0 .. len-1) whose row fits strictly inside the signal
    // (decode.rs:125-132).  Positions never decrease, so the rows kept are a prefix;
    // count them as we go and take the last peak back out at the end.
    uint64_t fit = (u + spr < work_len) ? 1 : 0;
    uint64_t last_fit = fit;
    while (n_corr > 0) {
        const uint64_t a = u + md + 1;
        const uint64_t b = (len + 1) * spr;
        const uint64_t s = a > b ? a : b;
        if (s >= n_corr) break;
        const uint64_t c = s / spr;
        for (uint64_t q = len + lane; q + 1 < c; q += 64)
            if (q < peaks_cap) peaks[q] = static_cast<uint32_t>(s);
        if (s + spr < work_len) fit += c - len - 1;
        // a phase that starts on a NaN keeps it: nothing is ever `>` a NaN peak (decode.rs:250)
        const float cs = corr[s];
        u = (cs != cs) ? s : first_terminal(bits, n_words, s);
        if (lane == 0 && c - 1 < peaks_cap) peaks[c - 1] = static_cast<uint32_t>(u);
        last_fit = (u + spr < work_len) ? 1 : 0;
        fit += last_fit;
        len = c;
    }
    if (lane == 0) {
        res->n_sync = static_cast<uint32_t>(len);
        res->n_rows = static_cast<uint32_t>(fit - last_fit);
        res->work_len = work_len;
        res->n_out = (fit - last_fit) * 2080u;
        if (len < 5) {  // decode.rs:112-118
            res->status = 1;
            res->reason = 2;
            res->n_rows = 0;
            res->n_out = 0;
        } else {
            res->status = 0;
            res->reason = 0;
        }
    }
}

// ---------------------------------------------------------------------------------
// Row gather (decode.rs:120-134) fused with the final resample_with_filter(NoFilter)
// to 4160 Hz (decode.rs:158-159 -> filter([1.]) + decimate(pw), dsp.rs:106-116):
//   px[r*2080 + c] = 0.0 + F[peaks[r] + pw*c] * 1.0,  and px[0] = 0 (the `i > j` guard).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void gather_rows_body(const float *__restrict__ f, const uint32_t *__restrict__ peaks,
                                                 Result *__restrict__ res, uint32_t spr, uint32_t pw, int raw,
                                                 float *__restrict__ rows, uint32_t rows_cap)
{
    // (relaxed atomics on both sides: one thread of this launch may lower the count while the others read it —
    // every reader takes the same minimum whichever value it sees, and the accesses are not a data race)
    uint32_t n_rows = __hip_atomic_load(&res->n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t px_per_row = spr / pw;  // 2080 when pw = work_rate / 4160
    if (n_rows > rows_cap) {
        // the caller's buffer holds fewer rows than the recording has: the record reports what
        // was written (reason 4), so a caller that trusts n_out never reads past its buffer.
        n_rows = rows_cap;
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            __hip_atomic_store(&res->n_rows, rows_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            res->n_out = static_cast<uint64_t>(rows_cap) * px_per_row;
            res->reason = 4;
        }
    }
    // Four pixels per thread and step (four independent loads in flight, one 16-byte store): the kernel co-runs with
    // the next call's VALU-bound front end, where what it costs is its instruction count — 17 VALU instructions per
    // pixel in the one-pixel-per-thread form this replaces, 5 now.
    const bool quads = (px_per_row & 3u) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0;
    for (uint32_t r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const float *src = f + peaks[r];
        float *dst = rows + static_cast<uint64_t>(r) * px_per_row;
        if (quads) {
            for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < px_per_row / 4u; k += gridDim.x * blockDim.x) {
                const float *p4 = src + static_cast<uint64_t>(4u * k) * pw;
                float4 v = make_float4(p4[0], p4[pw], p4[2u * pw], p4[3u * pw]);
                if (!raw) {  // filter([1.]): sum = 0.0 + x*1.0, and nothing at all for i == 0
                    v.x = __fadd_rn(0.f, __fmul_rn(v.x, 1.f));
                    v.y = __fadd_rn(0.f, __fmul_rn(v.y, 1.f));
                    v.z = __fadd_rn(0.f, __fmul_rn(v.z, 1.f));
                    v.w = __fadd_rn(0.f, __fmul_rn(v.w, 1.f));
                    if (r == 0 && k == 0) v.x = 0.f;
                }
                *reinterpret_cast<float4 *>(dst + 4u * k) = v;
            }
            continue;
        }
        for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < px_per_row;
             c += gridDim.x * blockDim.x) {
            float v = src[static_cast<uint64_t>(c) * pw];
            if (!raw) {  // filter([1.]): sum = 0.0 + x*1.0, and nothing at all for i == 0
                v = __fadd_rn(0.f, __fmul_rn(v, 1.f));
                if (r == 0 && c == 0) v = 0.f;
            }
            dst[c] = v;
        }
    }
}

__global__ void __launch_bounds__(kBlock)
k_gather_rows(const float *__restrict__ f, const uint32_t *__restrict__ peaks,
              Result *__restrict__ res, uint32_t spr, uint32_t pw, int raw,
              float *__restrict__ rows, uint32_t rows_cap)
{
    gather_rows_body(f, peaks, res, spr, pw, raw, rows, rows_cap);
}

// the recordings of one call: blockIdx.z picks the recording
__global__ void __launch_bounds__(kBlock)
k_gather_rows_call(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t spr, uint32_t pw)
{
    const RecArgs rec = call.rec[blockIdx.z];
    const SlotPtrs sp = slots[rec.slot];
    gather_rows_body(sp.f, sp.peaks, sp.res, spr, pw, 0, rec.rows, rec.rows_cap);
}

// The same rows with the (row, pixel quad) pairs of a recording as ONE flat index walked by a fixed number of
// workgroups (grid.x) per recording: no workgroup of a row's third, nearly empty block of quads (2080 px = 520 quads =
// 2.03 blocks of 256 threads), and `iters` quads per thread instead of one wave launch per 256 pixels — the kernel
// runs beside the next call's front end, whose workgroups are dispatched through the same pipe.  QPR: quads per row
// at compile time (520 whenever the work rate is a multiple of 4160 Hz: the division by it is a multiply).
template <uint32_t QPR>
__global__ void __launch_bounds__(kBlock)
k_gather_rows_flat(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t spr, uint32_t pw)
{
    const RecArgs rec = call.rec[blockIdx.y];
    const SlotPtrs sp = slots[rec.slot];
    Result *__restrict__ res = sp.res;
    uint32_t n_rows = __hip_atomic_load(&res->n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n_rows > rec.rows_cap) {  // (see gather_rows_body)
        n_rows = rec.rows_cap;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            __hip_atomic_store(&res->n_rows, rec.rows_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            res->n_out = static_cast<uint64_t>(rec.rows_cap) * (QPR * 4u);
            res->reason = 4;
        }
    }
    const float *__restrict__ f = sp.f;
    const uint32_t *__restrict__ peaks = sp.peaks;
    float *__restrict__ rows = rec.rows;
    const uint32_t total = n_rows * QPR;  // < 2^32: rows_cap <= 32768 rows of 520 quads
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < total; q += gridDim.x * kBlock) {
        const uint32_t r = q / QPR, k = q - r * QPR;
        const float *p4 = f + peaks[r] + static_cast<uint64_t>(4u * k) * pw;
        float4 v = make_float4(p4[0], p4[pw], p4[2u * pw], p4[3u * pw]);
        // filter([1.]): sum = 0.0 + x*1.0 (x*1.0 is x; the addition turns -0.0 into +0.0), and nothing at all for i == 0
        v.x = __fadd_rn(0.f, __fmul_rn(v.x, 1.f));
        v.y = __fadd_rn(0.f, __fmul_rn(v.y, 1.f));
        v.z = __fadd_rn(0.f, __fmul_rn(v.z, 1.f));
        v.w = __fadd_rn(0.f, __fmul_rn(v.w, 1.f));
        if (q == 0) v.x = 0.f;
        *reinterpret_cast<float4 *>(rows + static_cast<uint64_t>(q) * 4u) = v;
    }
}

__global__ void k_set_result(Result *res, Result value) { *res = value; }

// The no-sync branch's tail for the recordings of one call (decode.rs:135-159): F cropped to whole rows, then
// resample_with_filter(NoFilter) = filter([1.]) + decimate(m2) (dsp.rs:106-116, 294-307): out[k] = 0.0 + F[k m2] * 1.0 for
// k m2 >= 1, out[0] = 0 (the `i > j` guard) — k_fir_decimate's arithmetic with its one tap — and the result record.
// blockIdx.y = recording; `rows_cap` of the call's records counts FLOATS here.  (Until round 5: two launches per recording.)
__global__ void __launch_bounds__(kBlock)
k_nosync_rows_call(const CallArgs call, const SlotPtrs *__restrict__ slots, uint32_t spr, uint32_t m2)
{
    const RecArgs rec = call.rec[blockIdx.y];
    const float *__restrict__ f = slots[rec.slot].f;
    const uint64_t aligned = rec.w / spr * spr;
    uint64_t n_out = aligned / m2;
    if (n_out > rec.rows_cap) n_out = rec.rows_cap;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n_out; k += stride) {
        const uint64_t i = k * m2;
        rec.rows[k] = i < 1 ? 0.f : __fadd_rn(0.f, __fmul_rn(f[i], 1.f));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *slots[rec.slot].res = Result{0 /* APTGPU_OK */, 0, static_cast<uint32_t>(n_out / 2080u), 0, rec.w, n_out};
}

}  // namespace

void set_result(hipStream_t s, Result *res, Result value)
{
    hipLaunchKernelGGL(k_set_result, dim3(1), dim3(1), 0, s, res, value);
}

void nosync_rows_call(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t m2, uint64_t max_w)
{
    if (call.count == 0 || spr == 0 || m2 == 0) return;
    const uint64_t n = max_w / m2 + 1;
    hipLaunchKernelGGL(k_nosync_rows_call, dim3(grid_for(n, kBlock), call.count), dim3(kBlock), 0, s, call, d_slots, spr, m2);
}

void resample_generic(hipStream_t s, const float *x, uint64_t n, const float *coeff,
                      uint32_t ntaps, uint32_t l, uint32_t m, float *out, uint64_t w)
{
    if (w == 0) return;
    const uint32_t jlim = 2 * ((ntaps - 1) / 2) + 1;  // n <= t + offset  <=>  j <= 2*offset
    hipLaunchKernelGGL(k_resample_generic, dim3(grid_for(w, kBlock)), dim3(kBlock), 0, s, x, n,
                       coeff, jlim, l, m, out, w);
}

uint32_t f16taps_pairs_per_phase(uint32_t l, uint32_t ntaps) { return ((ntaps + l - 1) / l + 1) / 2; }

// host: phase-major fp16 table [l][tp2][2] (as raw uint16 pairs) and the power-of-two scale
float f16taps_pack(uint32_t l, const float *coeff, uint32_t ntaps, uint16_t *table)
{
    const uint32_t jlim = 2 * ((ntaps - 1) / 2) + 1;  // taps the reference actually uses
    const uint32_t tp2 = f16taps_pairs_per_phase(l, ntaps);
    float mx = 0.f;
    for (uint32_t j = 0; j < jlim; ++j) mx = fmaxf(mx, fabsf(coeff[j]));
    int e = 0;
    if (mx > 0.f) (void)frexpf(mx, &e);       // mx = f * 2^e, f in [0.5, 1)
    const int sh = -e + 1;                     // scaled max in [1, 2): well inside fp16 range
    const float scale = ldexpf(1.f, sh);
    for (uint32_t p = 0; p < l; ++p)
        for (uint32_t i = 0; i < 2 * tp2; ++i) {
            const uint64_t j = p + static_cast<uint64_t>(i) * l;
            const _Float16 h = static_cast<_Float16>(j < jlim ? coeff[j] * scale : 0.f);
            uint16_t bits;
            __builtin_memcpy(&bits, &h, 2);
            table[(static_cast<size_t>(p) * tp2 * 2) + i] = bits;
        }
    return ldexpf(1.f, -sh);
}

void resample_f16taps(hipStream_t s, const float *x, uint64_t n, const uint16_t *table, uint32_t ntaps,
                      uint32_t l, uint32_t m, float unscale, float *out, uint64_t w)
{
    if (w == 0) return;
    hipLaunchKernelGGL(k_resample_f16taps, dim3(grid_for(w, kBlock)), dim3(kBlock), 0, s, x, n,
                       reinterpret_cast<const h2 *>(table), f16taps_pairs_per_phase(l, ntaps), l, m, unscale,
                       out, w);
}

void fir_decimate(hipStream_t s, const float *x, uint64_t /*n*/, const float *coeff,
                  uint32_t ntaps, uint32_t m, float *out, uint64_t n_out)
{
    if (n_out == 0) return;
    hipLaunchKernelGGL(k_fir_decimate, dim3(grid_for(n_out, kBlock)), dim3(kBlock), 0, s, x, coeff,
                       ntaps, m, out, n_out);
}

void demodulate(hipStream_t s, const float *x, uint64_t n, float cosphi2, float sinphi, float *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_demodulate, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, x, n, cosphi2,
                       sinphi, out);
}

void correlate(hipStream_t s, const float *f, uint64_t n_corr, uint32_t pw, float *corr)
{
    if (n_corr == 0) return;
    hipLaunchKernelGGL(k_correlate, dim3(grid_for(n_corr, kBlock)), dim3(kBlock), 0, s, f, n_corr,
                       pw, corr);
}

void terminals(hipStream_t s, const float *corr, uint64_t n_corr, uint32_t md, uint64_t *bits)
{
    if (n_corr == 0) return;
    const unsigned blocks = static_cast<unsigned>((n_corr + md - 1) / md);
    const size_t lds = static_cast<size_t>(2) * md * sizeof(float);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_terminals),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
    hipLaunchKernelGGL(k_terminals, dim3(blocks), dim3(kTermThreads), lds, s, corr, n_corr, md,
                       bits);
}

void orbit_walk(hipStream_t s, const uint64_t *bits, const float *corr, uint64_t n_corr, uint64_t work_len,
                uint32_t spr, uint32_t md, uint32_t *peaks, uint32_t peaks_cap, Result *res)
{
    hipLaunchKernelGGL(k_orbit_walk, dim3(1), dim3(64), 0, s, bits, corr, n_corr, work_len, spr, md,
                       peaks, peaks_cap, res);
}

LaunchSwitches read_launch_switches()
{
    LaunchSwitches sw;
    if (const char *e = std::getenv("APTGPU_GATHER_ITERS")) sw.gather_iters = std::atoi(e);
    if (const char *e = std::getenv("APTGPU_WORDS_DPP")) sw.words_dpp = e[0] != '0';
    if (const char *e = std::getenv("APTGPU_ORBIT_LDS")) sw.orbit_lds = e[0] != '0';
    if (const char *e = std::getenv("APTGPU_ORBIT_THREADS")) sw.orbit_threads = std::atoi(e);
    if (const char *e = std::getenv("APTGPU_ORBIT_ALG")) sw.orbit_alg = e[0] == '0' ? 0 : 1;
    if (const char *e = std::getenv("APTGPU_GM_SLACK_SCALE")) {
        const float v = std::strtof(e, nullptr);
        if (v >= 1.f) sw.gm_slack_scale = v;
    }
    if (const char *e = std::getenv("APTGPU_FUSED_LDS_PAD")) sw.fused_lds_pad = std::max(0, std::atoi(e));
    return sw;
}

void gather_rows_call(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t pw,
                      uint32_t max_rows_cap, const LaunchSwitches &sw)
{
    if (call.count == 0) return;
    // The flat form wherever a row is 520 quads (every work rate that is a multiple of 4160 Hz) and the rows buffers are
    // 16-byte aligned: ONE quad per thread measured best beside the front end (pipelined step, 16 recordings per call:
    // 0.843 ms against 0.853 / 0.847 / 0.855 with 2 / 4 / 8 quads per thread and 0.869 with the form below;
    // profiles/r04_sweeps.txt).  APTGPU_GATHER_ITERS=n (A/B switch, read at plan creation): n quads per thread, 0 = the form below.
    {
        const int iters = sw.gather_iters;
        bool aligned = true;
        for (uint32_t i = 0; i < call.count; ++i) aligned = aligned && (reinterpret_cast<uintptr_t>(call.rec[i].rows) & 15u) == 0;
        if (iters > 0 && spr / pw == 2080u && spr % pw == 0 && aligned && max_rows_cap > 0) {
            const uint64_t quads = static_cast<uint64_t>(max_rows_cap < 32768u ? max_rows_cap : 32768u) * 520u;
            const uint64_t per_wg = static_cast<uint64_t>(kBlock) * static_cast<uint64_t>(iters);
            const unsigned gx = static_cast<unsigned>((quads + per_wg - 1) / per_wg);
            hipLaunchKernelGGL(k_gather_rows_flat<520u>, dim3(gx ? gx : 1, call.count), dim3(kBlock), 0, s, call, d_slots, spr, pw);
            return;
        }
    }
    // eight rows per workgroup (one row each made 57 600 workgroups of three loop iterations per thread for a call of
    // 16 recordings: the kernel ran at the rate workgroups can be dispatched, 118 us whatever it read)
    const unsigned rows_cap = max_rows_cap < 32768u ? max_rows_cap : 32768u;
    const unsigned gy = max_rows_cap == 0 ? 1u : (rows_cap + 7u) / 8u;
    const unsigned gx = (spr / pw + 4 * kBlock - 1) / (4 * kBlock);  // one thread per four pixels of a row
    hipLaunchKernelGGL(k_gather_rows_call, dim3(gx ? gx : 1, gy, call.count), dim3(kBlock), 0, s, call, d_slots,
                       spr, pw);
}

void gather_rows(hipStream_t s, const float *f, const uint32_t *peaks, Result *res,
                 uint32_t spr, uint32_t pw, bool raw, float *rows, uint32_t rows_cap)
{
    // (rows_cap == 0 still launches: the kernel then only clamps the result record)
    const unsigned gy = rows_cap == 0 ? 1u : (rows_cap < 4096u ? rows_cap : 4096u);
    const unsigned gx = (spr / pw + 4 * kBlock - 1) / (4 * kBlock);
    hipLaunchKernelGGL(k_gather_rows, dim3(gx ? gx : 1, gy), dim3(kBlock), 0, s, f, peaks, res, spr,
                       pw, raw ? 1 : 0, rows, rows_cap);
}


// ---------------------------------------------------------------------------------
// Exhaustive check of fast_divide (apt_envelope.hpp) for one divisor: every significand in the
// binades [1, 2) and [2, 4) — the quotient's rounding only depends on the significands, and two
// adjacent binades cover both relative positions of x's and c's significands.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_verify_fast_divide(float c, float rc, uint32_t *bad)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // 2^24 values
    const float x = __uint_as_float(0x3F800000u + i);     // 1.0 ... 4.0
    const float want = x / c;
    const float got = fast_divide(x, c, rc);
    if (__float_as_uint(want) != __float_as_uint(got)) atomicAdd(bad, 1u);
}

// every float of [2^-96, 2^100], 64 per thread
__global__ void __launch_bounds__(256) k_verify_fast_sqrt(uint32_t lo, uint32_t count, uint32_t *bad)
{
    const uint64_t base = (static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x) * 64u;
    uint32_t b = 0;
    for (uint32_t i = 0; i < 64u; ++i) {
        const uint64_t idx = base + i;
        if (idx >= count) break;
        const float x = __uint_as_float(lo + static_cast<uint32_t>(idx));
        b += __float_as_uint(exact_sqrt_inrange(x)) != __float_as_uint(__builtin_sqrtf(x));
    }
    if (b) atomicAdd(bad, b);
}

// the square-root half of the check: a property of the device, not of the plan
static bool verify_fast_sqrt(int device, hipStream_t s, uint32_t *d_bad)
{
    static std::mutex mu;
    static std::map<int, bool> cache;
    std::lock_guard<std::mutex> lock(mu);
    if (auto it = cache.find(device); it != cache.end()) return it->second;
    constexpr uint32_t lo = 0x0F800000u, hi = 0x71800000u;  // envelope_in_range()
    constexpr uint32_t count = hi - lo + 1u, threads = (count + 63u) / 64u;
    uint32_t bad = 1;
    bool ok = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_verify_fast_sqrt, dim3((threads + 255u) / 256u), dim3(256), 0, s, lo, count, d_bad);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipStreamSynchronize(s) == hipSuccess;
    }
    if (ok) cache[device] = bad == 0;  // a failed check (HIP error) is not cached
    return ok && bad == 0;
}

bool verify_fast_divide(int device, float c, float rc)
{
    if (!(c == c) || !(rc == rc) || c == 0.f || rc == 0.f) return false;
    // keep every intermediate of the check itself in the normal range
    const float ac = c < 0.f ? -c : c;
    if (!(ac > 1e-6f && ac < 1e6f)) return false;
    // The answer depends on (c, rc) only — sin(phi) is a function of work_rate — so it is computed
    // once per (device, divisor) and process; later plans (every one-shot aptgpu_decode builds one)
    // take it from the cache and never touch the device here.
    static std::mutex mu;
    static std::map<std::tuple<int, uint32_t, uint32_t>, bool> cache;
    uint32_t cb, rb;
    std::memcpy(&cb, &c, 4);
    std::memcpy(&rb, &rc, 4);
    const auto key = std::make_tuple(device, cb, rb);
    std::lock_guard<std::mutex> lock(mu);
    if (auto it = cache.find(key); it != cache.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
    uint32_t *d_bad = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_bad), sizeof(uint32_t)) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return false;
    }
    uint32_t bad = 1;
    bool ok = verify_fast_sqrt(device, s, d_bad) && hipMemsetAsync(d_bad, 0, sizeof(uint32_t), s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_verify_fast_divide, dim3((1u << 24) / 256u), dim3(256), 0, s, c, rc, d_bad);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipStreamSynchronize(s) == hipSuccess;
    }
    (void)hipFree(d_bad);
    (void)hipStreamDestroy(s);
    if (ok) cache[key] = bad == 0;  // a failed check (HIP error) is not cached
    return ok && bad == 0;
}

}  // namespace apt::gpu
