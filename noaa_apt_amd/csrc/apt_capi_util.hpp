// apt_capi_util.hpp — helpers shared by the extern "C" translation units (apt_capi.hip,
// apt_capi_image.hip): error mapping, malloc'd outputs, context callbacks, a per-call stream.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "apt_plan.hpp"

namespace apt::capi {

using apt::Error;
using apt::ErrorKind;

inline void put_err(char *err, size_t cap, const std::string &msg)
{
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
}

inline int fail(const Error &e, char *err, size_t cap)
{
    put_err(err, cap, e.message);
    return static_cast<int>(e.kind);
}

template <typename Fn>
int guarded(char *err, size_t cap, Fn &&fn)
{
    try {
        return fn();
    } catch (const Error &e) {
        return fail(e, err, cap);
    } catch (const std::bad_alloc &) {
        put_err(err, cap, "out of host memory");
        return APTGPU_ERR_INVALID;
    } catch (const std::exception &e) {
        put_err(err, cap, e.what());
        return APTGPU_ERR_INTERNAL;
    }
}

template <typename T>
T *host_alloc(size_t n)
{
    T *p = static_cast<T *>(std::malloc((n ? n : 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    return p;
}

inline void status(const aptgpu_context *ctx, float progress, const std::string &text)
{
    if (ctx && ctx->status) ctx->status(progress, text.c_str(), ctx->user);
}

// Context::step through the C callback; a nonzero return aborts like `?` in the reference.
inline void step(const aptgpu_context *ctx, bool on, const char *id, int variant, const float *data,
          size_t n, uint32_t rate)
{
    if (!on || !ctx || !ctx->step) return;
    if (ctx->step(id, variant, data, n, rate, ctx->user) != 0)
        throw Error{ErrorKind::Internal, std::string("step callback failed at \"") + id + "\""};
}

struct Scratch {
    int device;
    hipStream_t stream = nullptr;
    bool own = false;
    explicit Scratch(const aptgpu_context *ctx) : device(ctx ? ctx->device : 0)
    {
        apt::hip_check(hipSetDevice(device), "hipSetDevice");
        if (ctx && ctx->stream) {
            stream = static_cast<hipStream_t>(ctx->stream);
        } else {
            apt::hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking),
                           "hipStreamCreate");
            own = true;
        }
    }
    ~Scratch()
    {
        if (own && stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
    }
    apt::DeviceBuffer<float> upload(const float *h, size_t n, size_t pad = 16)
    {
        apt::DeviceBuffer<float> d;
        d.alloc(n + pad);
        if (n)
            apt::hip_check(hipMemcpyAsync(d.ptr, h, n * sizeof(float), hipMemcpyHostToDevice, stream),
                           "hipMemcpyAsync H2D");
        return d;
    }
    float *download_malloc(const float *d, size_t n)
    {
        float *p = host_alloc<float>(n);
        if (n) {
            if (hipMemcpyAsync(p, d, n * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                std::free(p);
                throw Error{ErrorKind::Hip, "D2H copy failed"};
            }
        }
        return p;
    }
};

// resample_with_filter (dsp.rs:62-126) on a signal already in HBM; returns the output length.
// steps_ctx (nullable; used when it has a step callback): the Context::step calls of dsp.rs:96-122 and of
// fast_resampling (:281-285) in the reference's order.  export_filtered: context.export_resample_filtered — the other
// decimation phase of fast_resampling (dsp.rs:265-273) and, with steps, the expanded signal.
inline uint64_t resample_device(Scratch &sc, const float *d_x, size_t n, uint32_t in_hz, uint32_t out_hz,
                                apt::Filter &filt, apt::DeviceBuffer<float> &d_y, bool export_filtered = false,
                                const aptgpu_context *steps_ctx = nullptr)
{
    if (out_hz == 0) throw Error{ErrorKind::Internal, "Can't resample to 0Hz"};  // dsp.rs:69-71
    const bool on = steps_ctx && steps_ctx->step;
    const apt::Rate in_rate = apt::Rate::hz(in_hz), out_rate = apt::Rate::hz(out_hz);
    const apt::LM lm = apt::interpolation_factors(in_rate, out_rate);
    auto export_signal = [&](const char *id, const float *d, uint64_t count, uint32_t rate) {
        float *h = sc.download_malloc(d, count);
        struct Free { float *p; ~Free() { std::free(p); } } guard{h};
        step(steps_ctx, on, id, 0, h, count, rate);
    };
    uint64_t w;
    if (lm.l > 1) {
        apt::Rate interpolated{};
        if (!in_rate.checked_mul(lm.l, &interpolated)) {
            char buf[512];
            std::snprintf(buf, sizeof buf,
                          "Can't resample, looks like the sample rates do not have a big\n"
                          "                divisor in common. input_rate: %u, output_rate: %u, "
                          "l: %u, m: %u",
                          in_hz, out_hz, lm.l, lm.m);
            throw Error{ErrorKind::RateOverflow, buf};
        }
        filt.resample(in_rate, interpolated);
        const apt::Signal coeff = filt.design();
        const uint32_t ntaps = static_cast<uint32_t>(coeff.size());
        step(steps_ctx, on, "resample_filter", 1, coeff.data(), coeff.size(), 0);  // dsp.rs:96
        auto d_c = sc.upload(coeff.data(), coeff.size());
        if (export_filtered) {
            const apt::ExportGeom g = apt::fast_resampling_export_geom(n, lm.l, lm.m, coeff.size());
            w = g.count;
            d_y.alloc(w + 16);
            apt::resample_at(sc.stream, d_x, n, d_c.ptr, ntaps, lm.l, g.d0, lm.m, d_y.ptr, w);
            if (on) {  // dsp.rs:269,281-285; where the n * l floats do not fit: an empty step, as dsp.rs:211-220 skips it
                bool done = false;
                try {
                    apt::DeviceBuffer<float> d_ex;
                    d_ex.alloc(g.expanded + 16);
                    apt::resample_at(sc.stream, d_x, n, d_c.ptr, ntaps, lm.l, 0, 1, d_ex.ptr, g.expanded);
                    export_signal("resample_filtered", d_ex.ptr, g.expanded, in_hz * lm.l);
                    done = true;
                } catch (const Error &e) {
                    if (e.kind != ErrorKind::Hip || e.hip_code != static_cast<int>(hipErrorOutOfMemory)) throw;
                    (void)hipGetLastError();
                } catch (const std::bad_alloc &) {
                }
                if (!done) {
                    std::fprintf(stderr, "aptgpu: expanded filtered signal (%llu samples) can't fit in memory, skipping step\n",
                                 static_cast<unsigned long long>(g.expanded));
                    step(steps_ctx, on, "resample_filtered", 0, nullptr, 0, in_hz * lm.l);
                }
            }
        } else {
            w = apt::fast_resampling_len(n, lm.l, lm.m, coeff.size());
            d_y.alloc(w + 16);
            apt::gpu::resample_generic(sc.stream, d_x, n, d_c.ptr, ntaps, lm.l, lm.m, d_y.ptr, w);
            step(steps_ctx, on, "resample_filtered", 0, nullptr, 0, in_hz * lm.l);  // empty unless exporting it
        }
        apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");  // d_c goes out of scope
    } else {
        const apt::Signal coeff = filt.design();
        step(steps_ctx, on, "resample_filter", 1, coeff.data(), coeff.size(), 0);  // dsp.rs:106
        w = n / lm.m;
        auto d_c = sc.upload(coeff.data(), coeff.size());
        d_y.alloc(w + 16);
        if (on) {  // dsp.rs:108-114: the filtered signal before the decimation
            apt::DeviceBuffer<float> d_f;
            d_f.alloc(n + 16);
            apt::gpu::fir_decimate(sc.stream, d_x, n, d_c.ptr, static_cast<uint32_t>(coeff.size()), 1, d_f.ptr, n);
            export_signal("resample_filtered", d_f.ptr, n, in_hz);
        }
        apt::gpu::fir_decimate(sc.stream, d_x, n, d_c.ptr, static_cast<uint32_t>(coeff.size()), lm.m, d_y.ptr, w);
        apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
    }
    if (on) export_signal("resample_decimated", d_y.ptr, w, out_hz);  // dsp.rs:100-104,118-122
    return w;
}

// apt_capi.hip
int decode_host(const aptgpu_context *ctx_in, const aptgpu_settings *settings, const float *signal,
                const uint8_t *wav_data, const apt::WavInfo *wav, size_t n, uint32_t input_rate_hz, int sync,
                float **rows_out, size_t *n_out, aptgpu_stats *stats, char *err, size_t err_cap);

}  // namespace apt::capi
