// apt_capi_image.hip — extern "C" surface of include/aptgpu.h §4: the consumers of decode()'s
// pixel rows (contrast limits, u8 mapping, telemetry), host-buffer and device-resident forms.
#include <cmath>
#include <string>
#include <vector>

#include "apt_capi_util.hpp"

namespace {

using namespace apt::capi;
using apt::gpu::ImageResult;

static_assert(sizeof(ImageResult) == sizeof(aptgpu_image_result), "ImageResult must mirror aptgpu_image_result");

const char *kZeroMin = "Can't get minimum of a zero length vector";
const char *kZeroMax = "Can't get maximum of a zero length vector";
const char *kTelemetryShort = "Recording too short for telemetry decoding";
const char *kBadPercent = "Percent given should be between 0 and 1";
const char *kNoLowBucket = "percent: no bucket reaches the low threshold (the reference panics here)";
const char *kChannelNames[9] = {"1", "2", "3a", "4", "5", "3b", "Unknown", "Unknown", "Unknown"};

// Rust's `{}` for an f32: shortest decimal that round-trips, never in exponent form.
std::string rust_display_f32(float v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[64];
    int prec = 1;
    for (; prec <= 9; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, static_cast<double>(v));
        if (std::strtof(buf, nullptr) == v) break;
    }
    std::string digits;
    bool neg = false;
    const char *p = buf;
    if (*p == '-') {
        neg = true;
        ++p;
    }
    for (; *p && *p != 'e'; ++p)
        if (*p != '.') digits.push_back(*p);
    const int exp10 = *p == 'e' ? std::atoi(p + 1) : 0;
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    const int point = exp10 + 1;  // digits before the decimal point
    if (point <= 0) {
        out = "0." + std::string(static_cast<size_t>(-point), '0') + digits;
    } else if (static_cast<size_t>(point) >= digits.size()) {
        out = digits + std::string(static_cast<size_t>(point) - digits.size(), '0');
    } else {
        out = digits.substr(0, static_cast<size_t>(point)) + "." + digits.substr(static_cast<size_t>(point));
    }
    if (digits == "0") out = "0";
    return neg ? "-" + out : out;
}

void throw_for(const ImageResult &r, int contrast)
{
    if (r.status == 0) return;
    switch (r.reason) {
    case 1: throw Error{ErrorKind::Internal, kZeroMin};
    case 2: throw Error{ErrorKind::Internal, kTelemetryShort};
    case 3: throw Error{ErrorKind::Internal, kNoLowBucket};
    default: throw Error{ErrorKind::Internal, "image stage failed"};
    }
    (void)contrast;
}

// One host-buffer call: signal in HBM + scratch + record.
struct ImageCall {
    Scratch sc;
    apt::DeviceBuffer<float> d_x;
    apt::DeviceBuffer<char> ws;
    apt::DeviceBuffer<ImageResult> d_info;
    uint64_t n;
    ImageCall(const aptgpu_context *ctx, const float *signal, size_t n_) : sc(ctx), n(n_)
    {
        d_x = sc.upload(signal, n, 2080 + 16);
        ws.alloc(apt::gpu::image_ws_bytes(n));
        d_info.alloc(1);
        apt::gpu::image_begin(sc.stream, d_info.ptr);
    }
    ImageResult info()
    {
        ImageResult r{};
        apt::hip_check(hipMemcpyAsync(&r, d_info.ptr, sizeof r, hipMemcpyDeviceToHost, sc.stream),
                       "hipMemcpyAsync D2H");
        apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
        return r;
    }
    void limits(float *low, float *high)
    {
        float lim[2] = {0.f, 0.f};
        const auto p = apt::gpu::image_ws_pointers(ws.ptr, n);
        apt::hip_check(hipMemcpyAsync(lim, p.limits, sizeof lim, hipMemcpyDeviceToHost, sc.stream),
                       "hipMemcpyAsync D2H");
        apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
        *low = lim[0];
        *high = lim[1];
    }
    apt::Signal band(const float *d, size_t count)
    {
        apt::Signal h(count);
        if (count) {
            apt::hip_check(hipMemcpyAsync(h.data(), d, count * sizeof(float), hipMemcpyDeviceToHost, sc.stream),
                           "hipMemcpyAsync D2H");
            apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
        }
        return h;
    }
};

void copy_out(aptgpu_image_result *dst, const ImageResult &r)
{
    if (dst) std::memcpy(dst, &r, sizeof r);
}

// read_telemetry's step exports, telemetry.rs:234-238 (same order, same ids)
void telemetry_steps(const aptgpu_context *ctx, ImageCall &c, const ImageResult &r)
{
    if (!ctx || !ctx->step) return;
    const auto p = apt::gpu::image_ws_pointers(c.ws.ptr, c.n);
    const size_t rows = c.n / 2080;
    const size_t nc = rows >= 200 ? rows - 200 : 0;
    (void)r;
    const apt::Signal a = c.band(p.mean_a, rows), b = c.band(p.mean_b, rows), v = c.band(p.variance, rows);
    const apt::Signal co = c.band(p.corr, nc), q = c.band(p.quality, nc);
    step(ctx, true, "telemetry_a", 0, a.data(), a.size(), 0);
    step(ctx, true, "telemetry_b", 0, b.data(), b.size(), 0);
    step(ctx, true, "telemetry_correlation", 0, co.data(), co.size(), 0);
    step(ctx, true, "telemetry_variance", 0, v.data(), v.size(), 0);
    step(ctx, true, "telemetry_quality", 0, q.data(), q.size(), 0);
}

int extreme(const aptgpu_context *ctx, const float *signal, size_t n, float *out, bool want_max, char *err,
            size_t err_cap)
{
    if ((!signal && n) || !out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (n == 0) throw Error{ErrorKind::Internal, want_max ? kZeroMax : kZeroMin};
        ImageCall c(ctx, signal, n);
        apt::gpu::image_minmax(c.sc.stream, c.d_x.ptr, nullptr, n, n, c.ws.ptr, c.d_info.ptr);
        float lo, hi;
        c.limits(&lo, &hi);
        *out = want_max ? hi : lo;
        return APTGPU_OK;
    });
}

}  // namespace

extern "C" {

const char *aptgpu_channel_name(int index)
{
    return index >= 0 && index < 9 ? kChannelNames[index] : "";
}

int aptgpu_get_min(const aptgpu_context *ctx, const float *signal, size_t n, float *out, char *err,
                   size_t err_cap)
{
    return extreme(ctx, signal, n, out, false, err, err_cap);
}

int aptgpu_get_max(const aptgpu_context *ctx, const float *signal, size_t n, float *out, char *err,
                   size_t err_cap)
{
    return extreme(ctx, signal, n, out, true, err, err_cap);
}

int aptgpu_percent(const aptgpu_context *ctx, const float *signal, size_t n, float percent, float *low,
                   float *high, char *err, size_t err_cap)
{
    if ((!signal && n) || !low || !high) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (percent < 0.f || percent > 1.f) throw Error{ErrorKind::Internal, kBadPercent};  // misc.rs:120-124
        if (n == 0) throw Error{ErrorKind::Internal, kZeroMin};                            // misc.rs:135
        ImageCall c(ctx, signal, n);
        apt::gpu::image_percent(c.sc.stream, c.d_x.ptr, nullptr, n, n, percent, c.ws.ptr, c.d_info.ptr);
        throw_for(c.info(), APTGPU_CONTRAST_PERCENT);
        c.limits(low, high);
        return APTGPU_OK;
    });
}

int aptgpu_map_signal_u8(const aptgpu_context *ctx, const float *signal, size_t n, float low, float high,
                         uint8_t **out, char *err, size_t err_cap)
{
    if ((!signal && n) || !out) return APTGPU_ERR_INVALID;
    *out = nullptr;
    return guarded(err, err_cap, [&] {
        ImageCall c(ctx, signal, n);
        apt::DeviceBuffer<uint8_t> d_img;
        d_img.alloc(n + 16);
        apt::gpu::image_set_limits(c.sc.stream, c.ws.ptr, n, low, high);
        apt::gpu::image_map_u8(c.sc.stream, c.d_x.ptr, nullptr, n, n, c.ws.ptr, false, d_img.ptr, c.d_info.ptr);
        uint8_t *h = host_alloc<uint8_t>(n);
        if (n && (hipMemcpyAsync(h, d_img.ptr, n, hipMemcpyDeviceToHost, c.sc.stream) != hipSuccess ||
                  hipStreamSynchronize(c.sc.stream) != hipSuccess)) {
            std::free(h);
            throw Error{ErrorKind::Hip, "D2H copy failed"};
        }
        *out = h;
        return APTGPU_OK;
    });
}

int aptgpu_read_telemetry(const aptgpu_context *ctx, const float *signal, size_t n,
                          aptgpu_image_result *telemetry, char *err, size_t err_cap)
{
    if ((!signal && n) || !telemetry) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        ImageCall c(ctx, signal, n);
        apt::gpu::image_telemetry(c.sc.stream, c.d_x.ptr, nullptr, n, n, c.ws.ptr, c.d_info.ptr, false);
        const ImageResult r = c.info();
        copy_out(telemetry, r);
        throw_for(r, APTGPU_CONTRAST_TELEMETRY);
        telemetry_steps(ctx, c, r);
        return APTGPU_OK;
    });
}

int aptgpu_process_gray(const aptgpu_context *ctx, const float *signal, size_t n, int contrast, float percent,
                        int rotate, uint8_t **image_out, size_t *n_out, aptgpu_image_result *info, char *err,
                        size_t err_cap)
{
    if ((!signal && n) || !image_out || !n_out) return APTGPU_ERR_INVALID;
    *image_out = nullptr;
    *n_out = 0;
    return guarded(err, err_cap, [&] {
        if (contrast != APTGPU_CONTRAST_TELEMETRY && contrast != APTGPU_CONTRAST_PERCENT &&
            contrast != APTGPU_CONTRAST_MINMAX)
            throw Error{ErrorKind::Invalid, "unknown contrast adjustment"};
        if (rotate != APTGPU_ROTATE_NO && rotate != APTGPU_ROTATE_YES)
            throw Error{ErrorKind::Unsupported, "Rotate::Orbit needs orbit propagation (host side, out of scope)"};
        ImageCall c(ctx, signal, n);
        hipStream_t s = c.sc.stream;
        if (contrast == APTGPU_CONTRAST_TELEMETRY) {
            status(ctx, 0.1f, "Adjusting contrast from telemetry");  // noaa_apt.rs:142
            apt::gpu::image_telemetry(s, c.d_x.ptr, nullptr, n, n, c.ws.ptr, c.d_info.ptr, true);
            const ImageResult r = c.info();
            copy_out(info, r);
            throw_for(r, contrast);
            telemetry_steps(ctx, c, r);
        } else if (contrast == APTGPU_CONTRAST_PERCENT) {
            // noaa_apt.rs:152-155
            status(ctx, 0.1f, "Adjusting contrast using " + rust_display_f32(percent * 100.f) + " percent");
            if (percent < 0.f || percent > 1.f) throw Error{ErrorKind::Internal, kBadPercent};
            if (n == 0) throw Error{ErrorKind::Internal, kZeroMin};
            apt::gpu::image_percent(s, c.d_x.ptr, nullptr, n, n, percent, c.ws.ptr, c.d_info.ptr);
        } else {
            status(ctx, 0.1f, "Mapping values");  // noaa_apt.rs:159
            if (n == 0) throw Error{ErrorKind::Internal, kZeroMin};
            apt::gpu::image_minmax(s, c.d_x.ptr, nullptr, n, n, c.ws.ptr, c.d_info.ptr);
        }
        status(ctx, 0.3f, "Generating image");  // noaa_apt.rs:180
        apt::DeviceBuffer<uint8_t> d_img;
        d_img.alloc(n + 16);
        if (rotate == APTGPU_ROTATE_YES) status(ctx, 0.90f, "Rotating output image");  // noaa_apt.rs:229
        apt::gpu::image_map_u8(s, c.d_x.ptr, nullptr, n, n, c.ws.ptr, rotate == APTGPU_ROTATE_YES, d_img.ptr,
                               c.d_info.ptr);
        const ImageResult r = c.info();
        copy_out(info, r);
        throw_for(r, contrast);
        uint8_t *h = host_alloc<uint8_t>(n);
        if (n && (hipMemcpyAsync(h, d_img.ptr, n, hipMemcpyDeviceToHost, s) != hipSuccess ||
                  hipStreamSynchronize(s) != hipSuccess)) {
            std::free(h);
            throw Error{ErrorKind::Hip, "D2H copy failed"};
        }
        *image_out = h;
        *n_out = n;
        return APTGPU_OK;
    });
}

int aptgpu_plan_process_device(aptgpu_plan *plan, int count, const float *const *d_rows,
                               const size_t *rows_cap, int contrast, float percent, int rotate,
                               uint8_t *const *d_images, char *err, size_t err_cap)
{
    if (!plan || count < 0 || !d_rows || !rows_cap || !d_images) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (static_cast<size_t>(count) > plan->last_slots.size())
            throw Error{ErrorKind::Invalid, "count exceeds the recordings of the last decode call"};
        if (contrast < APTGPU_CONTRAST_TELEMETRY || contrast > APTGPU_CONTRAST_MINMAX)
            throw Error{ErrorKind::Invalid, "unknown contrast adjustment"};
        if (rotate != APTGPU_ROTATE_NO && rotate != APTGPU_ROTATE_YES)
            throw Error{ErrorKind::Unsupported, "Rotate::Orbit needs orbit propagation (host side, out of scope)"};
        if (contrast == APTGPU_CONTRAST_PERCENT && (percent < 0.f || percent > 1.f))
            throw Error{ErrorKind::Internal, kBadPercent};
        apt::hip_check(hipSetDevice(plan->device), "hipSetDevice");
        for (int i = 0; i < count; ++i) {
            if (!d_rows[i] || !d_images[i]) throw Error{ErrorKind::Invalid, "null device pointer"};
            plan->enqueue_image(i, d_rows[i], static_cast<uint64_t>(rows_cap[i]) * 2080u, contrast, percent, rotate == APTGPU_ROTATE_YES,
                                d_images[i]);
        }
        return APTGPU_OK;
    });
}

int aptgpu_plan_image_results(aptgpu_plan *plan, int count, aptgpu_image_result *results)
{
    if (!plan || count < 0 || (!results && count)) return APTGPU_ERR_INVALID;
    if (static_cast<size_t>(count) > plan->last_slots.size() || !plan->d_image_results.ptr)
        return APTGPU_ERR_INVALID;
    try {
        apt::hip_check(hipSetDevice(plan->device), "hipSetDevice");
        plan->sync_all();
        for (int i = 0; i < count; ++i)
            apt::hip_check(hipMemcpy(results + i,
                                     plan->d_image_results.ptr + plan->last_slots[static_cast<size_t>(i)],
                                     sizeof(aptgpu_image_result), hipMemcpyDeviceToHost),
                           "hipMemcpy");
    } catch (const apt::Error &) {
        return APTGPU_ERR_HIP;
    }
    return APTGPU_OK;
}

}  // extern "C"
