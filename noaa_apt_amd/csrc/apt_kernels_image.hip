// apt_kernels_image.hip — gfx950 kernels for the consumers of decode()'s pixel rows
// (SURVEY.md §8(f) N2, N3): contrast limits (min/max, 1000-bucket percentile), the u8
// mapping (with the optional 180-degree channel rotation) and the telemetry statistics.
//
// Everything here is bit-exact with the reference's scalar loops: f32 products and sums are
// rounded separately (contract off), sums that the reference accumulates sequentially are
// accumulated sequentially in the same order by ONE thread, and the order-independent parts
// (min/max with first-index ties, integer histogram) are the only ones done in parallel.
// The work is small (10 MB of pixels per 10-minute recording) and HBM/latency-bound.
//
// The pixel count is read from the decode result record on the device when one is given, so
// the kernels chain behind gather_rows on the same stream without a host round trip.
#include "apt_kernels.hpp"

#include <cfloat>
#include <cmath>

#pragma clang fp contract(off)

namespace apt::gpu {

namespace {

constexpr int kPx = 2080;                 // PX_PER_ROW, decode.rs:14
constexpr int kBuckets = 1000;            // misc.rs:129
constexpr int kTelemetryLen = 200;        // 25 wedges x 8 rows, telemetry.rs:134-141
constexpr int kMinMaxBlocks = 512;
constexpr int kThreads = 256;

// number of pixels this launch works on; 0 when the decode failed
__device__ inline uint64_t px_count(const Result *res, uint64_t n_host, uint64_t cap)
{
    uint64_t n = n_host;
    if (res) n = res->status == 0 ? res->n_out : 0;
    return n < cap ? n : cap;
}

struct Extreme {
    float v;
    unsigned long long i;
};

__device__ inline bool beats_max(float v, unsigned long long i, const Extreme &b)
{
    return v > b.v || (v == b.v && i < b.i);
}
__device__ inline bool beats_min(float v, unsigned long long i, const Extreme &b)
{
    return v < b.v || (v == b.v && i < b.i);
}

__device__ inline Extreme shfl_extreme(Extreme e, int delta)
{
    Extreme o;
    o.v = __shfl_down(e.v, delta);
    o.i = __shfl_down(e.i, delta);
    return o;
}

// block reduction of a (max, min) pair with lowest-index ties; result valid in thread 0
__device__ inline void block_extremes(Extreme &mx, Extreme &mn)
{
    __shared__ Extreme s_mx[kThreads / 64], s_mn[kThreads / 64];
    for (int d = 32; d >= 1; d >>= 1) {
        Extreme a = shfl_extreme(mx, d), b = shfl_extreme(mn, d);
        if (beats_max(a.v, a.i, mx)) mx = a;
        if (beats_min(b.v, b.i, mn)) mn = b;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_mx[wave] = mx;
        s_mn[wave] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; w++) {
            if (beats_max(s_mx[w].v, s_mx[w].i, mx)) mx = s_mx[w];
            if (beats_min(s_mn[w].v, s_mn[w].i, mn)) mn = s_mn[w];
        }
    }
}

// dsp::get_max / get_min (dsp.rs:20-54).  The strict comparisons keep the FIRST of equal
// values, which only shows when the extreme is zero (+0.0 == -0.0 with different bits), and a
// NaN never wins a comparison.  So the reduction carries plain max/min values plus the index of
// the first zero-valued element, whose sign the result takes when the extreme is zero.
struct Range {
    float mx, mn;
    unsigned long long zero;  // lowest index with x == 0, ~0 if none
};

__device__ inline void range_take(Range &r, float v, unsigned long long i)
{
    r.mx = v > r.mx ? v : r.mx;
    r.mn = v < r.mn ? v : r.mn;
    if (v == 0.f && i < r.zero) r.zero = i;
}

__device__ inline void range_merge(Range &r, const Range &o)
{
    r.mx = o.mx > r.mx ? o.mx : r.mx;
    r.mn = o.mn < r.mn ? o.mn : r.mn;
    r.zero = o.zero < r.zero ? o.zero : r.zero;
}

// result valid in thread 0
__device__ inline void block_range(Range &r)
{
    __shared__ Range s_r[kThreads / 64];
    for (int d = 32; d >= 1; d >>= 1) {
        Range o;
        o.mx = __shfl_down(r.mx, d);
        o.mn = __shfl_down(r.mn, d);
        o.zero = __shfl_down(r.zero, d);
        range_merge(r, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_r[wave] = r;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kThreads / 64; w++) range_merge(r, s_r[w]);
}

__global__ __launch_bounds__(kThreads) void k_minmax_partial(const float *__restrict__ x,
                                                             const Result *res, uint64_t n_host,
                                                             uint64_t cap, Range *partial, uint32_t *counts,
                                                             ImageResult *out)
{
    const uint64_t n = px_count(res, n_host, cap);
    if (blockIdx.x == 0) {
        // first kernel of the stage: fresh record, empty histogram
        if (threadIdx.x == 0) {
            ImageResult z{};
            z.channel_a = z.channel_b = -1;
            *out = z;
        }
        for (int b = threadIdx.x; b < kBuckets; b += kThreads) counts[b] = 0;
    }
    Range r{-INFINITY, INFINITY, ~0ull};
    const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const uint64_t n4 = n / 4;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (uint64_t q = gtid; q < n4; q += stride) {
            const float4 v = x4[q];
            range_take(r, v.x, 4 * q);
            range_take(r, v.y, 4 * q + 1);
            range_take(r, v.z, 4 * q + 2);
            range_take(r, v.w, 4 * q + 3);
        }
        for (uint64_t i = 4 * n4 + gtid; i < n; i += stride) range_take(r, x[i], i);
    } else {
        for (uint64_t i = gtid; i < n; i += stride) range_take(r, x[i], i);
    }
    block_range(r);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// limits[0] = min, limits[1] = max
__global__ __launch_bounds__(kThreads) void k_minmax_final(const float *__restrict__ x, const Result *res,
                                                           uint64_t n_host, uint64_t cap,
                                                           const Range *partial, int n_partial,
                                                           float *limits, ImageResult *out)
{
    const uint64_t n = px_count(res, n_host, cap);
    Range r{-INFINITY, INFINITY, ~0ull};
    for (int k = threadIdx.x; k < n_partial; k += kThreads) range_merge(r, partial[k]);
    block_range(r);
    if (threadIdx.x == 0) {
        if (n == 0) {
            // "Can't get minimum of a zero length vector" (dsp.rs:40-44; get_min is called first
            // at misc.rs:135 and noaa_apt.rs:160) — or the decode itself failed
            out->status = 1;
            out->reason = (res && res->status != 0) ? 4 : 1;
            limits[0] = limits[1] = 0.f;
        } else {
            // `best = x[0]`: a NaN first element is never replaced
            const float x0 = x[0];
            const bool nan0 = x0 != x0;
            float mn = r.mn, mx = r.mx;
            if (mn == 0.f) mn = x[r.zero];  // the first zero's sign
            if (mx == 0.f) mx = x[r.zero];
            limits[0] = nan0 ? x0 : mn;
            limits[1] = nan0 ? x0 : mx;
        }
    }
}

// Rust `f as usize` then .min(999): saturating, NaN and negatives -> 0
__device__ inline int bucket_of(float v, float mn, float range)
{
    const float t = truncf((v - mn) / range * 1000.f);  // misc.rs:140-144
    if (!(t > 0.f)) return 0;
    return t >= 999.f ? kBuckets - 1 : static_cast<int>(t);
}

__global__ __launch_bounds__(kThreads) void k_histogram(const float *__restrict__ x, const Result *res,
                                                        uint64_t n_host, uint64_t cap,
                                                        const float *limits, uint32_t *counts)
{
    __shared__ uint32_t s_counts[kBuckets];
    const uint64_t n = px_count(res, n_host, cap);
    for (int b = threadIdx.x; b < kBuckets; b += kThreads) s_counts[b] = 0;
    __syncthreads();
    const float mn = limits[0];
    const float range = limits[1] - limits[0];  // misc.rs:137
    const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const uint64_t n4 = n / 4;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (uint64_t q = gtid; q < n4; q += stride) {
            const float4 v = x4[q];
            atomicAdd(&s_counts[bucket_of(v.x, mn, range)], 1u);
            atomicAdd(&s_counts[bucket_of(v.y, mn, range)], 1u);
            atomicAdd(&s_counts[bucket_of(v.z, mn, range)], 1u);
            atomicAdd(&s_counts[bucket_of(v.w, mn, range)], 1u);
        }
        for (uint64_t i = 4 * n4 + gtid; i < n; i += stride) atomicAdd(&s_counts[bucket_of(x[i], mn, range)], 1u);
    } else {
        for (uint64_t i = gtid; i < n; i += stride) atomicAdd(&s_counts[bucket_of(x[i], mn, range)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kBuckets; b += kThreads)
        if (s_counts[b]) atomicAdd(&counts[b], s_counts[b]);
}

// The bucket scan of misc::percent (misc.rs:152-174) as a block-wide prefix sum.  With
// frac_b = accum_b as f32 / len as f32 (monotone in b) the sequential loop's result is
//   low  = first b with frac_b > remainder
//   high = first b != low with frac_b > 1 - remainder   (the `else if` skips b == low; for
//          b < low the second test cannot hold because remainder <= 0.5), else 999.
// The integer prefix sums are exact, the float compare is the reference's own expression.
__global__ __launch_bounds__(kThreads) void k_percent_final(const Result *res, uint64_t n_host, uint64_t cap,
                                                            float percent, const uint32_t *counts,
                                                            float *limits, ImageResult *out)
{
    __shared__ uint32_t s_wave[kThreads / 64];
    __shared__ int s_low[kThreads / 64], s_high[kThreads / 64], s_first[2];
    const uint64_t n = px_count(res, n_host, cap);
    if (n == 0) return;  // k_minmax_final has reported it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 4 consecutive buckets per thread
    uint32_t c[4], run = 0;
    for (int k = 0; k < 4; k++) {
        const int b = tid * 4 + k;
        c[k] = b < kBuckets ? counts[b] : 0u;
        run += c[k];
    }
    uint32_t incl = run;  // inclusive scan of the per-thread sums
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    const float remainder = (1.f - percent) / 2.f;
    const float len = static_cast<float>(n);
    int low = kBuckets, high_a = kBuckets;  // first b above remainder / above 1-remainder
    uint32_t accum = base;
    for (int k = 0; k < 4; k++) {
        const int b = tid * 4 + k;
        if (b >= kBuckets) break;
        accum += c[k];
        const float frac = static_cast<float>(accum) / len;
        if (frac > remainder && low == kBuckets) low = b;
        if (frac > 1.f - remainder && high_a == kBuckets) high_a = b;
    }
    // block minima
    for (int d = 32; d >= 1; d >>= 1) {
        low = min(low, __shfl_down(low, d));
        high_a = min(high_a, __shfl_down(high_a, d));
    }
    if (lane == 0) {
        s_low[wave] = low;
        s_high[wave] = high_a;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kThreads / 64; w++) {
            low = min(low, s_low[w]);
            high_a = min(high_a, s_high[w]);
        }
        s_first[0] = low;
        s_first[1] = high_a;
    }
    __syncthreads();
    if (tid != 0) return;
    int low_bucket = s_first[0], high_bucket = s_first[1];
    // `else if`: the bucket that set low cannot also set high; the next one can (frac is monotone)
    if (high_bucket == low_bucket && high_bucket < kBuckets) high_bucket = high_bucket + 1 < kBuckets ? high_bucket + 1 : kBuckets;
    if (high_bucket >= kBuckets) high_bucket = kBuckets - 1;  // misc.rs:165-169
    if (low_bucket >= kBuckets) {  // low_bucket.unwrap() panics in the reference (NaN-only input)
        out->status = 1;
        out->reason = 3;
        return;
    }
    const float mn = limits[0];
    const float total_range = limits[1] - limits[0];
    limits[0] = static_cast<float>(low_bucket) / 1000.f * total_range + mn;
    limits[1] = static_cast<float>(high_bucket) / 1000.f * total_range + mn;
}

// map_signal_u8 (noaa_apt.rs:249-259): ((x - low) / range * 255).max(0).min(255).round() as u8
__device__ inline uint32_t map_px(float v, float low, float range)
{
    float t = (v - low) / range * 255.f;
    t = fmaxf(t, 0.f);    // NaN -> 0
    t = fminf(t, 255.f);
    return static_cast<uint32_t>(roundf(t));  // half away from zero
}

// processing::rotate (processing.rs:21-37): both 909-px channel images turned by 180 degrees
// in place (sync, space and telemetry columns stay): source pixel of output (r, c)
__device__ inline uint64_t rotate_src(uint64_t r, uint32_t c, uint64_t rows)
{
    constexpr uint32_t kOff = 39 + 47, kW = 909, kCh = 1040;  // decode.rs:16-35
    uint32_t base = ~0u;
    if (c >= kOff && c < kOff + kW) base = kOff;
    else if (c >= kOff + kCh && c < kOff + kCh + kW) base = kOff + kCh;
    if (base == ~0u) return r * kPx + c;
    return (rows - 1 - r) * kPx + base + (kW - 1 - (c - base));
}

// 4 pixels per thread (2080 is a multiple of 4, so a quad never straddles a row)
__global__ __launch_bounds__(kThreads) void k_map_u8(const float *__restrict__ x, const Result *res,
                                                     uint64_t n_host, uint64_t cap, const float *limits,
                                                     int rotate, uint8_t *__restrict__ out, ImageResult *info)
{
    const uint64_t n = px_count(res, n_host, cap);
    const float low = limits[0];
    const float range = limits[1] - limits[0];
    const uint64_t rows = n / kPx;
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (q == 0) {
        info->low = limits[0];
        info->high = limits[1];
        info->height = static_cast<uint32_t>(rows);
        info->n_px = info->status == 0 ? n : 0;
    }
    if (info->status != 0) return;
    const uint64_t i0 = q * 4;
    if (i0 >= n) return;
    uint32_t packed = 0;
    if (!rotate && i0 + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i0);
        packed = map_px(v.x, low, range) | (map_px(v.y, low, range) << 8) |
                 (map_px(v.z, low, range) << 16) | (map_px(v.w, low, range) << 24);
        *reinterpret_cast<uint32_t *>(out + i0) = packed;
        return;
    }
    for (int k = 0; k < 4 && i0 + k < n; k++) {
        const uint64_t i = i0 + k;
        uint64_t src = i;
        // pixels past the last whole row (never produced by decode()) are not rotated
        if (rotate && i / kPx < rows) src = rotate_src(i / kPx, static_cast<uint32_t>(i % kPx), rows);
        out[i] = static_cast<uint8_t>(map_px(x[src], low, range));
    }
}

// ---------------------------------------------------------------------------- telemetry
// per-row band statistics, telemetry.rs:154-177: one thread per row, sequential sums
__global__ __launch_bounds__(kThreads) void k_telemetry_rows(const float *__restrict__ x, const Result *res,
                                                             uint64_t n_host, uint64_t cap, float *mean_a,
                                                             float *mean_b, float *variance, ImageResult *out)
{
    const uint64_t rows = px_count(res, n_host, cap) / kPx;
    const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (r == 0) {  // first kernel of the stage: fresh record
        ImageResult z{};
        z.channel_a = z.channel_b = -1;
        *out = z;
    }
    if (r >= rows) return;
    const float *a = x + r * kPx + 994;
    const float *b = x + r * kPx + 2034;
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < 44; i++) sa += a[i];
    for (int i = 0; i < 44; i++) sb += b[i];
    const float ma = sa / 44.f, mb = sb / 44.f;
    float va = 0.f, vb = 0.f;
    for (int i = 0; i < 44; i++) {
        const float d = a[i] - ma;
        va += d * d;
    }
    for (int i = 0; i < 44; i++) {
        const float d = b[i] - mb;
        vb += d * d;
    }
    mean_a[r] = ma;
    mean_b[r] = mb;
    variance[r] = (va + vb) / 88.f;
}

__device__ inline float telemetry_sample(int j)
{
    // telemetry.rs:134-141: wedges 1-9, 7 variable wedges (0), wedges 1-9; 8 rows each
    const int w = j >> 3;
    const int k = w < 9 ? w : (w < 16 ? -1 : w - 16);
    if (k < 0 || k == 8) return 0.f;
    if (k == 7) return 255.f;
    if (k == 6) return 224.f;
    return 31.f + 32.f * static_cast<float>(k);  // 31, 63, 95, 127, 159, 191
}

// correlation with the wedge pattern and the quality figure, telemetry.rs:210-231: one thread
// per start row, 400 sequential MACs + 200 sequential sqrt-adds, the three bands staged in LDS
__global__ __launch_bounds__(kThreads) void k_telemetry_corr(const Result *res, uint64_t n_host, uint64_t cap,
                                                             const float *__restrict__ mean_a,
                                                             const float *__restrict__ mean_b,
                                                             const float *__restrict__ variance,
                                                             float *corr, float *quality)
{
    __shared__ float s_a[kThreads + kTelemetryLen], s_b[kThreads + kTelemetryLen], s_sd[kThreads + kTelemetryLen];
    const uint64_t rows = px_count(res, n_host, cap) / kPx;
    if (rows < kTelemetryLen) return;
    const uint64_t nc = rows - kTelemetryLen;
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kThreads;
    if (base >= nc) return;
    for (int k = threadIdx.x; k < kThreads + kTelemetryLen; k += kThreads) {
        const uint64_t r = base + k;
        s_a[k] = r < rows ? mean_a[r] : 0.f;
        s_b[k] = r < rows ? mean_b[r] : 0.f;
        s_sd[k] = r < rows ? __builtin_sqrtf(variance[r]) : 0.f;
    }
    __syncthreads();
    const uint64_t i = base + threadIdx.x;
    if (i >= nc) return;
    float sum = 0.f, sd = 0.f;
    for (int j = 0; j < kTelemetryLen; j++) {
        const float t = telemetry_sample(j);
        sum += t * s_a[threadIdx.x + j];
        sum += t * s_b[threadIdx.x + j];
    }
    for (int j = 0; j < kTelemetryLen; j++) sd += s_sd[threadIdx.x + j];
    corr[i] = sum;
    quality[i] = sum / sd;
}

// best frame start (first strict maximum above 0, telemetry.rs:196,228-230), then
// Telemetry::from_bands (telemetry.rs:30-72), the contrast wedges (noaa_apt.rs:146-147) and the
// channel names (telemetry.rs:93-121).  One workgroup.
__global__ __launch_bounds__(kThreads) void k_telemetry_best(const Result *res, uint64_t n_host, uint64_t cap,
                                                             const float *__restrict__ mean_a,
                                                             const float *__restrict__ mean_b,
                                                             const float *__restrict__ quality,
                                                             float *limits, ImageResult *out, int set_limits)
{
    const uint64_t n = px_count(res, n_host, cap);
    const uint64_t rows = n / kPx;
    if (rows < kTelemetryLen) {
        if (threadIdx.x == 0) {
            // "Recording too short for telemetry decoding", telemetry.rs:199-203
            out->status = 1;
            out->reason = (res && res->status != 0) ? 4 : 2;
        }
        return;
    }
    const uint64_t nc = rows - kTelemetryLen;
    Extreme best{0.f, ~0ull}, unused{INFINITY, ~0ull};
    for (uint64_t i = threadIdx.x; i < nc; i += kThreads) {
        const float q = quality[i];
        if (q > best.v || (q == best.v && i < best.i && best.i != ~0ull)) best = Extreme{q, i};
    }
    // ties: a later equal q never replaces (strict >), and q == 0 never replaces the initial
    // (0, 0.); beats_max's index rule gives exactly that once "none" is the largest index
    block_extremes(best, unused);
    if (threadIdx.x != 0) return;
    const uint64_t row = best.i == ~0ull ? 0 : best.i;
    out->telemetry_row = static_cast<uint32_t>(row);
    out->telemetry_quality = best.i == ~0ull ? 0.f : best.v;
    float wa[25], wb[25];
    for (int w = 0; w < 25; w++) {
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < 8; r++) {
            sa += mean_a[row + w * 8 + r];
            sb += mean_b[row + w * 8 + r];
        }
        wa[w] = sa / 8.f;
        wb[w] = sb / 8.f;
    }
    for (int wedge = 1; wedge <= 16; wedge++) {
        out->values_a[wedge - 1] = wedge <= 9 ? (wa[wedge - 1] + wa[wedge + 15]) / 2.f : wa[wedge - 1];
        out->values_b[wedge - 1] = wedge <= 9 ? (wb[wedge - 1] + wb[wedge + 15]) / 2.f : wb[wedge - 1];
    }
    for (int ch = 0; ch < 2; ch++) {
        const float value = ch == 0 ? out->values_a[15] : out->values_b[15];
        int name = 0;
        float best_d = 0.f;
        bool nan = false;
        for (int i = 0; i < 9; i++) {
            const float d = fabsf((out->values_a[i] + out->values_b[i]) / 2.f - value);
            if (d != d) nan = true;
            if (i == 0 || d < best_d) {
                name = i;
                best_d = d;
            }
        }
        (ch == 0 ? out->channel_a : out->channel_b) = nan ? -1 : name;
    }
    if (set_limits) {
        limits[0] = (out->values_a[8] + out->values_b[8]) / 2.f;  // wedge 9, both channels
        limits[1] = (out->values_a[7] + out->values_b[7]) / 2.f;  // wedge 8
    }
}

__global__ void k_image_begin(ImageResult *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ImageResult z{};
        z.channel_a = z.channel_b = -1;
        *out = z;
    }
}

__global__ void k_set_limits(float *limits, float low, float high)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        limits[0] = low;
        limits[1] = high;
    }
}

inline unsigned blocks_for(uint64_t n, unsigned per_block, unsigned max_blocks)
{
    uint64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return static_cast<unsigned>(b < max_blocks ? b : max_blocks);
}

}  // namespace

size_t image_ws_bytes(uint64_t max_px)
{
    const uint64_t rows = max_px / kPx + 1;
    return kMinMaxBlocks * sizeof(Range) + 4 * sizeof(float) + kBuckets * sizeof(uint32_t) +
           5 * rows * sizeof(float) + 64;
}

namespace {
struct WsView {
    Range *partial;
    float *limits;
    uint32_t *counts;
    float *mean_a, *mean_b, *variance, *corr, *quality;
};
inline WsView carve(void *ws, uint64_t max_px)
{
    const uint64_t rows = max_px / kPx + 1;
    WsView v;
    char *p = static_cast<char *>(ws);
    v.partial = reinterpret_cast<Range *>(p);
    p += kMinMaxBlocks * sizeof(Range);
    v.limits = reinterpret_cast<float *>(p);
    p += 4 * sizeof(float);
    v.counts = reinterpret_cast<uint32_t *>(p);
    p += kBuckets * sizeof(uint32_t);
    v.mean_a = reinterpret_cast<float *>(p);
    v.mean_b = v.mean_a + rows;
    v.variance = v.mean_b + rows;
    v.corr = v.variance + rows;
    v.quality = v.corr + rows;
    return v;
}
}  // namespace

ImageWsPointers image_ws_pointers(void *ws, uint64_t max_px)
{
    const WsView v = carve(ws, max_px);
    return ImageWsPointers{v.limits, v.counts, v.mean_a, v.mean_b, v.variance, v.corr, v.quality};
}

void image_begin(hipStream_t s, ImageResult *out)
{
    hipLaunchKernelGGL(k_image_begin, dim3(1), dim3(64), 0, s, out);
}

void image_minmax(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                  ImageResult *out)
{
    const WsView v = carve(ws, cap);
    const unsigned nb = blocks_for(cap, kThreads * 16, kMinMaxBlocks);
    hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(kThreads), 0, s, x, res, n, cap, v.partial, v.counts,
                       out);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(kThreads), 0, s, x, res, n, cap, v.partial,
                       static_cast<int>(nb), v.limits, out);
}

void image_percent(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, float percent,
                   void *ws, ImageResult *out)
{
    const WsView v = carve(ws, cap);
    image_minmax(s, x, res, n, cap, ws, out);
    // few, fat workgroups: every one ends with up to 1000 same-address global atomics
    const unsigned nb = blocks_for(cap, kThreads * 64, 128);
    hipLaunchKernelGGL(k_histogram, dim3(nb), dim3(kThreads), 0, s, x, res, n, cap, v.limits, v.counts);
    hipLaunchKernelGGL(k_percent_final, dim3(1), dim3(kThreads), 0, s, res, n, cap, percent, v.counts,
                       v.limits, out);
}

void image_telemetry(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                     ImageResult *out, bool set_limits)
{
    const WsView v = carve(ws, cap);
    const uint64_t rows = cap / kPx;
    hipLaunchKernelGGL(k_telemetry_rows, dim3(blocks_for(rows, kThreads, 1u << 20)), dim3(kThreads), 0, s, x,
                       res, n, cap, v.mean_a, v.mean_b, v.variance, out);
    hipLaunchKernelGGL(k_telemetry_corr, dim3(blocks_for(rows, kThreads, 1u << 20)), dim3(kThreads), 0, s,
                       res, n, cap, v.mean_a, v.mean_b, v.variance, v.corr, v.quality);
    hipLaunchKernelGGL(k_telemetry_best, dim3(1), dim3(kThreads), 0, s, res, n, cap, v.mean_a, v.mean_b,
                       v.quality, v.limits, out, set_limits ? 1 : 0);
}

void image_set_limits(hipStream_t s, void *ws, uint64_t cap, float low, float high)
{
    const WsView v = carve(ws, cap);
    hipLaunchKernelGGL(k_set_limits, dim3(1), dim3(64), 0, s, v.limits, low, high);
}

void image_map_u8(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                  bool rotate, uint8_t *out, ImageResult *info)
{
    const WsView v = carve(ws, cap);
    const uint64_t quads = (cap + 3) / 4;
    hipLaunchKernelGGL(k_map_u8, dim3(blocks_for(quads, kThreads, 1u << 30)), dim3(kThreads), 0, s, x, res, n,
                       cap, v.limits, rotate ? 1 : 0, out, info);
}

}  // namespace apt::gpu
