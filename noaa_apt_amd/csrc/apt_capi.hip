// apt_capi.hip — the extern "C" surface declared in include/aptgpu.h.
// No exception crosses the ABI; apt::Error maps to status code + message.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "apt_capi_util.hpp"
#include "apt_session.hpp"

namespace {

using namespace apt::capi;

using PlanPtr = std::unique_ptr<aptgpu_plan, apt::capi::PlanDeleter>;


apt::Signal download(const float *d, size_t n, hipStream_t s)
{
    apt::Signal h(n);
    if (n) {
        apt::hip_check(hipMemcpyAsync(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost, s),
                       "hipMemcpyAsync D2H");
        apt::hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    }
    return h;
}

const char *kTooShort = "Got less than 10 rows of samples, audio file is too short";
const char *kFewSync = "Found less than 5 sync frames, audio file is too short or too noisy";
const char *kNotMultiple = "work_rate is not multiple of FINAL_RATE";

aptgpu_context default_ctx()
{
    aptgpu_context c{};
    return c;
}

}  // namespace

extern "C" {

const char *aptgpu_version(void) { return "aptgpu 0.2.0 (gfx950)"; }
int aptgpu_abi_version(void) { return APTGPU_ABI_VERSION; }

int aptgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void aptgpu_free(void *p) { std::free(p); }

void aptgpu_cache_clear(void) { apt::capi::session_cache_clear(); }

void aptgpu_cache_info(int32_t *entries, uint64_t *device_bytes)
{
    int e = 0;
    uint64_t b = 0;
    apt::capi::session_cache_info(&e, &b);
    if (entries) *entries = e;
    if (device_bytes) *device_bytes = b;
}

// ------------------------------------------------------------------ plans
int aptgpu_plan_create(const aptgpu_context *ctx, const aptgpu_settings *settings,
                       uint32_t input_rate_hz, int sync, size_t max_samples, int max_batch,
                       aptgpu_plan **plan_out, char *err, size_t err_cap)
{
    if (!settings || !plan_out) {
        put_err(err, err_cap, "null argument");
        return APTGPU_ERR_INVALID;
    }
    return guarded(err, err_cap, [&] {
        *plan_out = apt::plan_create(ctx, *settings, input_rate_hz, sync != 0, max_samples, max_batch);
        return APTGPU_OK;
    });
}

void aptgpu_plan_destroy(aptgpu_plan *plan)
{
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    for (hipStream_t st : plan->streams) (void)hipStreamSynchronize(st);
    if (plan->ev_user) (void)hipEventDestroy(plan->ev_user);
    for (hipStream_t st : plan->streams) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : plan->ev_front) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : plan->ev_pre) (void)hipEventDestroy(ev);
    if (plan->front_stream) (void)hipStreamDestroy(plan->front_stream);
    delete plan;
}

int aptgpu_plan_get_info(const aptgpu_plan *plan, aptgpu_plan_info *info)
{
    if (!plan || !info) return APTGPU_ERR_INVALID;
    info->l = plan->l;
    info->m = plan->m;
    info->n_resample_taps = static_cast<uint32_t>(plan->taps_resample.size());
    info->n_lowpass_taps = static_cast<uint32_t>(plan->taps_lowpass.size());
    info->n_sync_taps = plan->n_sync_taps;
    info->samples_per_work_row = plan->spr;
    info->min_distance = plan->md;
    info->max_samples = plan->max_samples;
    info->max_work_len = plan->max_work_len;
    info->max_rows = plan->max_rows;
    info->fused = plan->fused;
    info->max_batch = plan->max_batch;
    return APTGPU_OK;
}

int aptgpu_plan_decode_device(aptgpu_plan *plan, int count, const float *const *d_signals,
                              const size_t *n, float *const *d_rows, const size_t *rows_cap,
                              char *err, size_t err_cap)
{
    if (!plan || !d_signals || !n || !d_rows || !rows_cap || count < 0 ||
        count > plan->max_batch) {
        put_err(err, err_cap, "bad argument to aptgpu_plan_decode_device");
        return APTGPU_ERR_INVALID;
    }
    return guarded(err, err_cap, [&] {
        apt::hip_check(hipSetDevice(plan->device), "hipSetDevice");
        for (int i = 0; i < count; ++i)
            if (n[i] > plan->max_samples)
                throw Error{ErrorKind::Invalid, "recording longer than the plan's max_samples"};
        std::vector<aptgpu_plan::Input> ins(static_cast<size_t>(count));
        std::vector<uint64_t> caps(static_cast<size_t>(count));
        for (int i = 0; i < count; ++i) {
            ins[static_cast<size_t>(i)].ptr = d_signals[i];
            ins[static_cast<size_t>(i)].n = n[i];
            caps[static_cast<size_t>(i)] = static_cast<uint64_t>(rows_cap[i]) * 2080u;
        }
        plan->run_call(count, ins.data(), d_rows, caps.data(), false);
        return APTGPU_OK;
    });
}

int aptgpu_plan_synchronize(aptgpu_plan *plan)
{
    if (!plan) return APTGPU_ERR_INVALID;
    (void)hipSetDevice(plan->device);
    try {
        plan->sync_all();
    } catch (const Error &) {
        return APTGPU_ERR_HIP;
    }
    return APTGPU_OK;
}

int aptgpu_plan_join(aptgpu_plan *plan)
{
    if (!plan) return APTGPU_ERR_INVALID;
    if (!plan->user_stream) return aptgpu_plan_synchronize(plan);
    (void)hipSetDevice(plan->device);
    // ctx.stream waits (on the device) for everything enqueued so far on the internal streams
    for (hipStream_t st : plan->streams) {
        if (hipEventRecord(plan->ev_user, st) != hipSuccess ||
            hipStreamWaitEvent(plan->user_stream, plan->ev_user, 0) != hipSuccess)
            return APTGPU_ERR_HIP;
    }
    return APTGPU_OK;
}

int aptgpu_plan_results(aptgpu_plan *plan, int count, aptgpu_result *results)
{
    if (!plan || !results || count < 0 || static_cast<size_t>(count) > plan->last_slots.size())
        return APTGPU_ERR_INVALID;
    static_assert(sizeof(aptgpu_result) == sizeof(apt::gpu::Result), "result layout");
    (void)hipSetDevice(plan->device);
    try {
        plan->sync_all();
    } catch (const Error &) {
        return APTGPU_ERR_HIP;
    }
    for (int i = 0; i < count; ++i)
        if (hipMemcpy(results + i, plan->result_of(i), sizeof(aptgpu_result), hipMemcpyDeviceToHost) !=
            hipSuccess)
            return APTGPU_ERR_HIP;
    return APTGPU_OK;
}

int aptgpu_plan_sync_positions(aptgpu_plan *plan, int i, uint64_t *pos, size_t cap, size_t *n_sync)
{
    if (!plan || i < 0 || static_cast<size_t>(i) >= plan->last_slots.size() || !n_sync)
        return APTGPU_ERR_INVALID;
    aptgpu_result r{};
    (void)hipSetDevice(plan->device);
    try {
        plan->sync_all();
    } catch (const Error &) {
        return APTGPU_ERR_HIP;
    }
    if (hipMemcpy(&r, plan->result_of(i), sizeof r, hipMemcpyDeviceToHost) != hipSuccess)
        return APTGPU_ERR_HIP;
    *n_sync = r.n_sync;
    size_t take = r.n_sync < cap ? r.n_sync : cap;
    if (!plan->sync || !pos || take == 0) return APTGPU_OK;
    if (take > plan->slot_of(i).peaks.count) take = plan->slot_of(i).peaks.count;
    std::vector<uint32_t> tmp(take);
    if (hipMemcpy(tmp.data(), plan->slot_of(i).peaks.ptr, take * sizeof(uint32_t),
                  hipMemcpyDeviceToHost) != hipSuccess)
        return APTGPU_ERR_HIP;
    for (size_t k = 0; k < take; ++k) pos[k] = tmp[k];
    return APTGPU_OK;
}

int aptgpu_plan_enable_timing(aptgpu_plan *plan, int on)
{
    if (!plan) return APTGPU_ERR_INVALID;
    plan->timer.enable(on < 0 ? 0 : (on > 2 ? 2 : on));
    return APTGPU_OK;
}

int aptgpu_plan_collect_timing(aptgpu_plan *plan, aptgpu_kernel_time *out, size_t cap, size_t *n_out)
{
    if (!plan || !n_out) return APTGPU_ERR_INVALID;
    try {
        (void)hipSetDevice(plan->device);
        plan->sync_all();
        plan->sync_all();  // the event pairs live on both streams
        auto v = plan->timer.collect(plan->stream);
        *n_out = v.size();
        for (size_t i = 0; i < v.size() && i < cap && out; ++i) out[i] = v[i];
        return APTGPU_OK;
    } catch (const Error &) {
        return APTGPU_ERR_HIP;
    }
}

int aptgpu_plan_read_internal(aptgpu_plan *plan, int i, const char *name, void *host_out,
                              size_t bytes, size_t *size_out)
{
    if (!plan || !name || i < 0 || static_cast<size_t>(i) >= plan->last_slots.size())
        return APTGPU_ERR_INVALID;
    aptgpu_plan::Slot &sl = plan->slot_of(i);
    const void *src = nullptr;
    size_t size = 0;
    const std::string n(name);
    if (n == "inv_sinphi") {  // host-side constant: RN(1/sin(phi)) when the fast exact divide is on, else 0
        if (size_out) *size_out = sizeof(float);
        if (host_out && bytes >= sizeof(float)) std::memcpy(host_out, &plan->inv_sinphi, sizeof(float));
        return APTGPU_OK;
    }
    if (n == "filtered") { src = sl.filtered.ptr; size = sl.filtered.count * sizeof(float); }
    else if (n == "correlation") { src = sl.correlation.ptr; size = sl.correlation.count * sizeof(float); }
    else if (n == "group_max") { src = sl.gm.ptr; size = sl.gm.count * sizeof(apt::gpu::GroupMax); }
    else if (n == "terminal_words") { src = sl.words.ptr; size = sl.words.count * sizeof(uint64_t); }
    else if (n == "peaks") { src = sl.peaks.ptr; size = sl.peaks.count * sizeof(uint32_t); }
    else if (n == "picker_flags") { src = sl.flags.ptr; size = sl.flags.count * sizeof(uint32_t); }
    else return APTGPU_ERR_INVALID;
    if (size_out) *size_out = size;
    (void)hipSetDevice(plan->device);
    try {
        plan->sync_all();
    } catch (const Error &) {
        return APTGPU_ERR_HIP;
    }
    const size_t take = bytes < size ? bytes : size;
    if (take && host_out && src &&
        hipMemcpy(host_out, src, take, hipMemcpyDeviceToHost) != hipSuccess)
        return APTGPU_ERR_HIP;
    return APTGPU_OK;
}

// ------------------------------------------------------------------ decode()
}  // extern "C"

namespace apt::capi {

// decode() on a host buffer: the f32 Signal, or (wav != nullptr) the payload of a WAV data chunk
// that is uploaded as it is and converted on the device (wav.rs:30-51).
int decode_host(const aptgpu_context *ctx_in, const aptgpu_settings *settings, const float *signal,
                const uint8_t *wav_data, const apt::WavInfo *wav, size_t n, uint32_t input_rate_hz, int sync,
                float **rows_out, size_t *n_out, aptgpu_stats *stats, char *err, size_t err_cap)
{
    *rows_out = nullptr;
    *n_out = 0;
    aptgpu_context ctx = ctx_in ? *ctx_in : default_ctx();
    const bool steps = settings->export_wav != 0 && ctx.step != nullptr;

    return guarded(err, err_cap, [&]() -> int {
        const uint32_t work = settings->work_rate;
        // decode.rs:59 (a WAV input is exported after its conversion, below)
        if (!wav) step(&ctx, steps, "input", 0, signal, n, input_rate_hz);
        // decode.rs:63
        if (!wav || !steps) status(&ctx, 0.1f, "Resampling to " + std::to_string(work));

        // The plan (designed taps, HBM workspace, streams) and the device input / output buffers come from the
        // process-wide session cache (apt_session.hpp; SURVEY.md §8(b), threading row) unless the caller wants the
        // step export (unfused kernels, every intermediate kept) or runs on a stream of their own: building and
        // tearing them down cost 2 ms per call, as much as the PCIe time of a ten-minute recording.
        const bool cached = !steps && ctx.stream == nullptr;
        SessionLease lease;
        struct PoisonOnThrow {  // a session that an exception left half-way through a call is not reused
            SessionLease &l;
            bool ok = false;
            ~PoisonOnThrow() { if (!ok && l) l.poison(); }
        } guard{lease};
        hipStream_t s_for_errors = nullptr;
        // decode()'s own per-recording errors (decode.rs:79-83,112-118,172-176) leave the GPU state perfectly valid:
        // wait for what the call enqueued and hand the session back to the cache — only a HIP failure or an unknown
        // exception poisons it
        auto decode_error = [&](const char *msg) -> Error {
            if (s_for_errors == nullptr || hipStreamSynchronize(s_for_errors) == hipSuccess) guard.ok = true;
            return Error{ErrorKind::Internal, msg};
        };
        PlanPtr own_plan;
        aptgpu_plan *plan = nullptr;
        if (cached) {
            SessionKey key;
            key.device = ctx.device;
            key.mode = ctx.mode;
            key.rate = input_rate_hz;
            key.sync = sync != 0;
            key.per_call = 1;
            key.settings = *settings;
            key.settings.export_wav = 0;
            key.settings.export_resample_filtered = settings->export_resample_filtered ? 1 : 0;  // another plan (dsp.rs:265-273)
            lease = session_acquire(key, n);
            plan = lease->plan.get();
        } else {
            own_plan.reset(apt::plan_create(&ctx, *settings, input_rate_hz, sync != 0, n, 1, /*depth*/ 1));
            plan = own_plan.get();
        }
        if (plan->spr == 0) throw Error{ErrorKind::Invalid, "work_rate too small"};
        hipStream_t s = plan->stream;  // a single-stream plan: every call runs on streams[0] == stream
        s_for_errors = s;
        const uint64_t w = plan->work_len_for(n);
        const uint64_t out_cap =
            sync ? (plan->spr ? w / plan->spr + 2 : 2) * 2080u : plan->out_len_nosync(w) + 16;

        apt::DeviceBuffer<float> own_in, own_rows;
        apt::DeviceBuffer<uint8_t> own_wav;
        const size_t in_bytes = wav ? wav->data_len : n * sizeof(float);
        void *d_in_ptr = nullptr;
        float *d_rows_ptr = nullptr;
        if (cached) {
            lease->ensure_set(0, in_bytes + 64, out_cap);
            d_in_ptr = lease->sets[0].in[0].ptr;
            d_rows_ptr = lease->sets[0].out[0].ptr;
        } else if (wav) {
            own_wav.alloc(in_bytes + 16);
            d_in_ptr = own_wav.ptr;
        } else {
            own_in.alloc(n + 16);
            d_in_ptr = own_in.ptr;
        }
        if (!cached) {
            own_rows.alloc(out_cap);
            d_rows_ptr = own_rows.ptr;
        }
        struct { float *ptr; } d_rows{d_rows_ptr};
        aptgpu_plan::Input in;
        in.n = n;
        in.ptr = d_in_ptr;
        if (in_bytes)
            apt::hip_check(hipMemcpyAsync(d_in_ptr, wav ? static_cast<const void *>(wav_data) : static_cast<const void *>(signal),
                                          in_bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D");
        if (wav) {
            in.channels = wav->channels;
            in.bytes_per_sample = wav->bytes_per_sample;
            in.codec = static_cast<int>(wav->codec);
        }

        float *rows_ptr = d_rows.ptr;
        const uint64_t cap64 = out_cap;
        plan->run_call(1, &in, &rows_ptr, &cap64, steps);
        aptgpu_plan::Slot &sl = plan->slot_of(0);
        if (steps) plan->sync_all();  // the step exports below read the slot's buffers
        if (wav && steps) {
            apt::Signal x = download(sl.ingest.ptr, n, s);
            step(&ctx, steps, "input", 0, x.data(), x.size(), input_rate_hz);
            status(&ctx, 0.1f, "Resampling to " + std::to_string(work));
        }

        // The expanded signal of export_resample_filtered (n * l floats: gigabytes for a whole pass) on the device and on
        // the host.  Where either allocation fails the step is delivered empty and the decode goes on — the reference logs
        // "Expanded filtered signal can't fit in memory, skipping step" and does the same (dsp.rs:211-220).
        auto export_expanded = [&](uint64_t cnt, uint32_t rate, auto &&fill) {
            try {
                apt::DeviceBuffer<float> d_ex;
                d_ex.alloc(cnt + 16);
                fill(d_ex.ptr);
                apt::Signal ex = download(d_ex.ptr, cnt, s);
                step(&ctx, steps, "resample_filtered", 0, ex.data(), ex.size(), rate);
                return;
            } catch (const Error &e) {
                if (e.kind != ErrorKind::Hip || e.hip_code != static_cast<int>(hipErrorOutOfMemory)) throw;
                (void)hipGetLastError();
            } catch (const std::bad_alloc &) {
            }
            std::fprintf(stderr, "aptgpu: expanded filtered signal (%llu samples) can't fit in memory, skipping step\n",
                         static_cast<unsigned long long>(cnt));
            step(&ctx, steps, "resample_filtered", 0, nullptr, 0, rate);
        };
        // dsp.rs:96 / :106 — the resample filter, then the resample steps
        step(&ctx, steps, "resample_filter", 1, plan->taps_resample.data(),
             plan->taps_resample.size(), 0);
        if (w < 10ull * plan->spr) throw decode_error(kTooShort);  // decode.rs:79-83
        if (steps) {
            if (plan->l > 1 && plan->export_filtered) {
                // dsp.rs:269,281-285: every sum of the interpolated axis (gigabytes for a whole pass, as the
                // reference's documentation warns)
                const uint64_t cnt = apt::fast_resampling_export_geom(n, plan->l, plan->m, plan->taps_resample.size()).expanded;
                export_expanded(cnt, input_rate_hz * plan->l, [&](float *d_ex) {
                    plan->expanded_filtered(s, wav ? sl.ingest.ptr : static_cast<const float *>(d_in_ptr), n, false, d_ex, cnt);
                });
            } else if (plan->l > 1) {
                // dsp.rs:281-285: expanded signal is empty unless export_resample_filtered
                step(&ctx, steps, "resample_filtered", 0, nullptr, 0, input_rate_hz * plan->l);
            } else {
                // l == 1 (dsp.rs:106-116): filter() then decimate(); the step carries the filtered signal, at the input rate
                apt::DeviceBuffer<float> d_fl;
                d_fl.alloc(n + 16);
                apt::gpu::fir_decimate(s, wav ? sl.ingest.ptr : static_cast<const float *>(d_in_ptr), n, plan->d_taps_resample.ptr,
                                       static_cast<uint32_t>(plan->taps_resample.size()), 1, d_fl.ptr, n);
                apt::Signal fl = download(d_fl.ptr, n, s);
                step(&ctx, steps, "resample_filtered", 0, fl.data(), fl.size(), input_rate_hz);
            }
            apt::Signal r = download(sl.resampled.ptr, w, s);
            step(&ctx, steps, "resample_decimated", 0, r.data(), r.size(), work);
        }

        status(&ctx, 0.4f, "Demodulating");  // decode.rs:87
        if (steps) {
            apt::Signal d = download(sl.demodulated.ptr, w, s);
            step(&ctx, steps, "demodulation_result", 0, d.data(), d.size(), 0);
        }
        status(&ctx, 0.42f, "Filtering");  // decode.rs:93
        apt::Signal filtered_host;
        if (steps) {
            step(&ctx, steps, "filter_filter", 1, plan->taps_lowpass.data(),
                 plan->taps_lowpass.size(), 0);
            filtered_host = download(sl.filtered.ptr, w, s);
            step(&ctx, steps, "filter_result", 0, filtered_host.data(), filtered_host.size(), 0);
        }

        aptgpu_result res{};
        if (sync) {
            status(&ctx, 0.5f, "Syncing");  // decode.rs:107
            if (!plan->work_is_multiple) throw decode_error(kNotMultiple);
            if (steps) {
                apt::Signal c = download(sl.correlation.ptr, w - plan->n_sync_taps, s);
                step(&ctx, steps, "sync_correlation", 0, c.data(), c.size(), 0);
            }
            if (aptgpu_plan_results(plan, 1, &res) != APTGPU_OK)
                throw Error{ErrorKind::Hip, "could not read the result record"};
            if (res.n_sync < 5) throw decode_error(kFewSync);  // decode.rs:112-118
            if (steps) {
                // "sync_result": the aligned rows at work_rate (decode.rs:150)
                apt::DeviceBuffer<float> d_al;
                d_al.alloc(static_cast<uint64_t>(res.n_rows) * plan->spr + 16);
                apt::gpu::gather_rows(s, sl.filtered.ptr, sl.peaks.ptr, plan->result_of(0), plan->spr,
                                      1, true, d_al.ptr, res.n_rows);
                apt::Signal al = download(d_al.ptr, static_cast<uint64_t>(res.n_rows) * plan->spr, s);
                step(&ctx, steps, "sync_result", 0, al.data(), al.size(), work);
                status(&ctx, 0.90f, "Resampling to 4160");
                // final resample_with_filter(NoFilter), l == 1 branch (dsp.rs:106-122)
                const float one = 1.f;
                step(&ctx, steps, "resample_filter", 1, &one, 1, 0);
                apt::gpu::gather_rows(s, sl.filtered.ptr, sl.peaks.ptr, plan->result_of(0), plan->spr,
                                      1, false, d_al.ptr, res.n_rows);
                al = download(d_al.ptr, static_cast<uint64_t>(res.n_rows) * plan->spr, s);
                step(&ctx, steps, "filter_filter", 1, &one, 1, 0);
                step(&ctx, steps, "filter_result", 0, al.data(), al.size(), 0);
                step(&ctx, steps, "resample_filtered", 0, al.data(), al.size(), work);
            } else {
                status(&ctx, 0.90f, "Resampling to 4160");  // decode.rs:154
            }
        } else {
            status(&ctx, 0.5f, "Skipping Syncing");  // decode.rs:136
            step(&ctx, steps, "sync_correlation", 0, nullptr, 0, work);  // decode.rs:139
            if (aptgpu_plan_results(plan, 1, &res) != APTGPU_OK)
                throw Error{ErrorKind::Hip, "could not read the result record"};
            if (steps) {
                const uint64_t aligned = w / plan->spr * plan->spr;
                step(&ctx, steps, "sync_result", 0, filtered_host.data(), aligned, work);
                status(&ctx, 0.90f, "Resampling to 4160");
                // final resample_with_filter(NoFilter) (decode.rs:158-159): the steps of dsp.rs:96-122
                const float one = 1.f;
                step(&ctx, steps, "resample_filter", 1, &one, 1, 0);
                if (plan->l2 > 1 && plan->export_filtered) {
                    const uint64_t cnt = apt::fast_resampling_export_geom(aligned, plan->l2, plan->m2, 1).expanded;
                    export_expanded(cnt, work * plan->l2, [&](float *d_ex) { plan->expanded_filtered(s, sl.filtered.ptr, aligned, true, d_ex, cnt); });
                } else if (plan->l2 > 1) {
                    // fast_resampling: the expanded signal is empty unless export_resample_filtered
                    step(&ctx, steps, "resample_filtered", 0, nullptr, 0, work * plan->l2);
                } else {
                    // filter([1.]): 0.0 + x*1.0, and nothing at all for sample 0 (the `i > j` guard)
                    apt::DeviceBuffer<float> d_fl;
                    d_fl.alloc(aligned + 16);
                    apt::gpu::fir_decimate(s, sl.filtered.ptr, aligned, plan->d_one.ptr, 1, 1, d_fl.ptr, aligned);
                    apt::Signal fl = download(d_fl.ptr, aligned, s);
                    step(&ctx, steps, "filter_filter", 1, &one, 1, 0);
                    step(&ctx, steps, "filter_result", 0, fl.data(), fl.size(), 0);
                    step(&ctx, steps, "resample_filtered", 0, fl.data(), fl.size(), work);
                }
            } else {
                status(&ctx, 0.90f, "Resampling to 4160");
            }
        }
        if (res.status != APTGPU_OK)
            throw decode_error(res.reason == 2 ? kFewSync : res.reason == 3 ? kNotMultiple : kTooShort);

        float *rows = host_alloc<float>(res.n_out);
        if (res.n_out) {
            if (hipMemcpyAsync(rows, d_rows.ptr, res.n_out * sizeof(float), hipMemcpyDeviceToHost,
                               s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) {
                std::free(rows);
                throw Error{ErrorKind::Hip, "D2H copy of the image rows failed"};
            }
        }
        if (steps) step(&ctx, steps, "resample_decimated", 0, rows, res.n_out, apt::FINAL_RATE);
        *rows_out = rows;
        *n_out = res.n_out;
        guard.ok = true;
        if (stats) {
            stats->work_len = w;
            stats->n_sync = res.n_sync;
            stats->n_rows = res.n_rows;
            stats->l = plan->l;
            stats->m = plan->m;
            stats->n_resample_taps = static_cast<uint32_t>(plan->taps_resample.size());
            stats->n_lowpass_taps = static_cast<uint32_t>(plan->taps_lowpass.size());
            stats->fused = plan->fused;
            stats->orbit_path = 1;
        }
        return APTGPU_OK;
    });
}

}  // namespace apt::capi

extern "C" {

int aptgpu_decode(const aptgpu_context *ctx_in, const aptgpu_settings *settings,
                  const float *signal, size_t n, uint32_t input_rate_hz, int sync,
                  float **rows_out, size_t *n_out, aptgpu_stats *stats, char *err, size_t err_cap)
{
    if (!settings || (!signal && n) || !rows_out || !n_out) {
        put_err(err, err_cap, "null argument");
        return APTGPU_ERR_INVALID;
    }
    return apt::capi::decode_host(ctx_in, settings, signal, nullptr, nullptr, n, input_rate_hz, sync, rows_out,
                                  n_out, stats, err, err_cap);
}

// ------------------------------------------------------------------ building blocks
int aptgpu_filter_design(const aptgpu_filter *f, float **coeff_out, size_t *n_out)
{
    if (!f || !coeff_out || !n_out) return APTGPU_ERR_INVALID;
    auto flt = apt::make_filter(f->kind, f->cutout_pi_rad, f->atten, f->delta_w_pi_rad);
    if (!flt) return APTGPU_ERR_INVALID;
    try {
        apt::Signal c = flt->design();
        float *p = host_alloc<float>(c.size());
        std::memcpy(p, c.data(), c.size() * sizeof(float));
        *coeff_out = p;
        *n_out = c.size();
        return APTGPU_OK;
    } catch (...) {
        return APTGPU_ERR_INVALID;
    }
}

void aptgpu_filter_resample(aptgpu_filter *f, uint32_t input_rate_hz, uint32_t output_rate_hz)
{
    if (!f || f->kind == APTGPU_FILTER_NOFILTER) return;
    auto flt = apt::make_filter(f->kind, f->cutout_pi_rad, f->atten, f->delta_w_pi_rad);
    if (!flt) return;
    flt->resample(apt::Rate::hz(input_rate_hz), apt::Rate::hz(output_rate_hz));
    if (auto *lp = dynamic_cast<apt::Lowpass *>(flt.get())) {
        f->cutout_pi_rad = lp->cutout.get_pi_rad();
        f->delta_w_pi_rad = lp->delta_w.get_pi_rad();
    } else if (auto *dc = dynamic_cast<apt::LowpassDcRemoval *>(flt.get())) {
        f->cutout_pi_rad = dc->cutout.get_pi_rad();
        f->delta_w_pi_rad = dc->delta_w.get_pi_rad();
    }
}

int aptgpu_generate_sync_frame(uint32_t work_rate_hz, int8_t **frame_out, size_t *n_out, char *err,
                               size_t err_cap)
{
    if (!frame_out || !n_out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        std::vector<int8_t> g;
        std::string msg;
        if (!apt::generate_sync_frame(apt::Rate::hz(work_rate_hz), &g, &msg))
            throw Error{ErrorKind::Internal, msg};
        int8_t *p = host_alloc<int8_t>(g.size());
        std::memcpy(p, g.data(), g.size());
        *frame_out = p;
        *n_out = g.size();
        return APTGPU_OK;
    });
}

namespace {

// resample_with_filter, dsp.rs:62-126, on host buffers.
int resample_with_filter_impl(const aptgpu_context *ctx, const float *signal, size_t n,
                              uint32_t in_hz, uint32_t out_hz, apt::Filter &filt, float **out,
                              size_t *n_out)
{
    if (out_hz == 0) throw Error{ErrorKind::Internal, "Can't resample to 0Hz"};
    if (in_hz == 0) throw Error{ErrorKind::Invalid, "input_rate is 0"};
    Scratch sc(ctx);
    auto d_x = sc.upload(signal, n);
    apt::DeviceBuffer<float> d_y;
    const uint64_t w = resample_device(sc, d_x.ptr, n, in_hz, out_hz, filt, d_y);
    *out = sc.download_malloc(d_y.ptr, w);
    *n_out = w;
    return APTGPU_OK;
}

}  // namespace

int aptgpu_resample_with_filter(const aptgpu_context *ctx, const float *signal, size_t n,
                                uint32_t input_rate_hz, uint32_t output_rate_hz,
                                aptgpu_filter filt, float **out, size_t *n_out, char *err,
                                size_t err_cap)
{
    if ((!signal && n) || !out || !n_out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        auto f = apt::make_filter(filt.kind, filt.cutout_pi_rad, filt.atten, filt.delta_w_pi_rad);
        if (!f) throw Error{ErrorKind::Invalid, "unknown filter kind"};
        return resample_with_filter_impl(ctx, signal, n, input_rate_hz, output_rate_hz, *f, out,
                                         n_out);
    });
}

int aptgpu_resample(const aptgpu_context *ctx, const float *signal, size_t n,
                    uint32_t input_rate_hz, uint32_t output_rate_hz, float atten,
                    float delta_w_pi_rad, float **out, size_t *n_out, char *err, size_t err_cap)
{
    if ((!signal && n) || !out || !n_out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (input_rate_hz == 0) throw Error{ErrorKind::Invalid, "input_rate is 0"};
        const apt::Rate in_rate = apt::Rate::hz(input_rate_hz);
        // dsp.rs:140-150
        const apt::Freq cutout =
            output_rate_hz > input_rate_hz
                ? apt::Freq::hz(static_cast<float>(input_rate_hz) / 2.f, in_rate)
                : apt::Freq::hz(static_cast<float>(output_rate_hz) / 2.f, in_rate);
        apt::Lowpass f(cutout, atten, apt::Freq::pi_rad(delta_w_pi_rad));
        return resample_with_filter_impl(ctx, signal, n, input_rate_hz, output_rate_hz, f, out, n_out);
    });
}

int aptgpu_demodulate(const aptgpu_context *ctx, const float *signal, size_t n,
                      float carrier_pi_rad, float **out, char *err, size_t err_cap)
{
    if ((!signal && n) || !out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (n == 0) throw Error{ErrorKind::Invalid, "empty signal (the reference panics)"};
        Scratch sc(ctx);
        const float phi = 2.f * apt::Freq::pi_rad(carrier_pi_rad).get_rad();  // dsp.rs:360
        auto d_x = sc.upload(signal, n);
        apt::DeviceBuffer<float> d_y;
        d_y.alloc(n + 16);
        apt::gpu::demodulate(sc.stream, d_x.ptr, n, cosf(phi) * 2.f, sinf(phi), d_y.ptr);
        *out = sc.download_malloc(d_y.ptr, n);
        return APTGPU_OK;
    });
}

int aptgpu_filter_signal(const aptgpu_context *ctx, const float *signal, size_t n,
                         aptgpu_filter filt, float **out, char *err, size_t err_cap)
{
    if ((!signal && n) || !out) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        auto f = apt::make_filter(filt.kind, filt.cutout_pi_rad, filt.atten, filt.delta_w_pi_rad);
        if (!f) throw Error{ErrorKind::Invalid, "unknown filter kind"};
        const apt::Signal coeff = f->design();
        Scratch sc(ctx);
        auto d_x = sc.upload(signal, n);
        auto d_c = sc.upload(coeff.data(), coeff.size());
        apt::DeviceBuffer<float> d_y;
        d_y.alloc(n + 16);
        apt::gpu::fir_decimate(sc.stream, d_x.ptr, n, d_c.ptr, static_cast<uint32_t>(coeff.size()), 1,
                               d_y.ptr, n);
        *out = sc.download_malloc(d_y.ptr, n);
        return APTGPU_OK;
    });
}

int aptgpu_find_sync(const aptgpu_context *ctx, const float *signal, size_t n,
                     uint32_t work_rate_hz, uint64_t **pos_out, size_t *n_pos,
                     float **correlation_out, size_t *n_corr_out, char *err, size_t err_cap)
{
    if ((!signal && n) || !pos_out || !n_pos) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        if (work_rate_hz % apt::FINAL_RATE != 0) throw Error{ErrorKind::Internal, kNotMultiple};
        const uint32_t pw = work_rate_hz / apt::FINAL_RATE;
        const uint64_t spr64 = static_cast<uint64_t>(apt::PX_PER_ROW) * work_rate_hz / apt::FINAL_RATE;
        const uint32_t spr = static_cast<uint32_t>(spr64);
        const uint32_t md = static_cast<uint32_t>(static_cast<uint64_t>(spr) * 8 / 10);
        const uint32_t g = 38 * pw;
        if (pw == 0 || n < g)
            throw Error{ErrorKind::Invalid, "signal shorter than the sync frame (the reference panics)"};
        if (static_cast<uint64_t>(md) * 8 > 160u * 1024u)
            throw Error{ErrorKind::Unsupported, "work_rate too large for the sync search (LDS)"};
        const uint64_t n_corr = n - g;
        Scratch sc(ctx);
        auto d_f = sc.upload(signal, n);
        apt::DeviceBuffer<float> d_c;
        d_c.alloc(n_corr + 64);
        apt::DeviceBuffer<uint64_t> d_bits;
        d_bits.alloc(n_corr / 64 + md / 64 + 4);
        const uint32_t cap = static_cast<uint32_t>(n / spr + 4);
        apt::DeviceBuffer<uint32_t> d_peaks;
        d_peaks.alloc(cap);
        apt::DeviceBuffer<apt::gpu::Result> d_res;
        d_res.alloc(1);
        apt::gpu::correlate(sc.stream, d_f.ptr, n_corr, pw, d_c.ptr);
        if (ctx && ctx->mode == APTGPU_MODE_GENERIC) {
            apt::gpu::terminals(sc.stream, d_c.ptr, n_corr, md, d_bits.ptr);
            apt::gpu::orbit_walk(sc.stream, d_bits.ptr, d_c.ptr, n_corr, n, spr, md, d_peaks.ptr, cap, d_res.ptr);
        } else {
            const uint64_t ng = n_corr / apt::gpu::sync_group_size() + 2;
            const uint64_t chunks = ng / apt::gpu::sync_chunk_groups() + 2;
            apt::DeviceBuffer<apt::gpu::GroupMax> d_gm;
            d_gm.alloc(ng + 64);
            apt::DeviceBuffer<uint64_t> d_words, d_nanw;
            d_words.alloc(ng + 64);
            d_nanw.alloc(ng + 64);
            apt::DeviceBuffer<uint32_t> d_slot, d_cnt, d_flags;
            d_slot.alloc(chunks * apt::gpu::sync_slot_cap());
            d_cnt.alloc(chunks);
            d_flags.alloc(32);
            apt::hip_check(hipMemsetAsync(d_flags.ptr, 0, 32 * sizeof(uint32_t), sc.stream), "hipMemsetAsync");
            const char *fw = std::getenv("APTGPU_FORCE_WALK");
            apt::DeviceBuffer<uint32_t> d_ws;
            d_ws.alloc(apt::gpu::sync_orbit_ws_words(n, spr));
            // a one-recording "call" over a one-entry slot table
            apt::gpu::SlotPtrs sp{};
            sp.f = d_f.ptr;
            sp.gm = d_gm.ptr;
            sp.corr = d_c.ptr;
            sp.words = d_words.ptr;
            sp.nanw = d_nanw.ptr;
            sp.slot_nt = d_slot.ptr;
            sp.slot_cnt = d_cnt.ptr;
            sp.flags = d_flags.ptr;
            sp.orbit_ws = d_ws.ptr;
            sp.peaks = d_peaks.ptr;
            sp.res = d_res.ptr;
            sp.peaks_cap = cap;
            apt::DeviceBuffer<apt::gpu::SlotPtrs> d_sp;
            d_sp.alloc(1);
            apt::hip_check(hipMemcpyAsync(d_sp.ptr, &sp, sizeof sp, hipMemcpyHostToDevice, sc.stream), "hipMemcpyAsync");
            apt::gpu::CallArgs call{};
            call.count = 1;
            call.rec[0].x = nullptr;
            call.rec[0].n = 0;
            call.rec[0].w = n;
            call.rec[0].rows = nullptr;
            call.rec[0].rows_cap = 0;
            call.rec[0].slot = 0;
            apt::gpu::group_max(sc.stream, d_c.ptr, n_corr, d_gm.ptr);
            const apt::gpu::LaunchSwitches sw = apt::gpu::read_launch_switches();  // (once per API call)
            apt::gpu::sync_nodes(sc.stream, call, d_sp.ptr, n, pw, spr, md, false, true, sw);
            apt::gpu::sync_orbit(sc.stream, call, d_sp.ptr, spr, md, pw, (fw && fw[0] == '1') ? 1 : 0, sw);
            apt::hip_check(hipGetLastError(), "kernel launch (find_sync)");
            apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
        }
        apt::gpu::Result r{};
        apt::hip_check(hipMemcpyAsync(&r, d_res.ptr, sizeof r, hipMemcpyDeviceToHost, sc.stream),
                       "hipMemcpyAsync");
        apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
        std::vector<uint32_t> tmp(r.n_sync);
        if (r.n_sync)
            apt::hip_check(hipMemcpy(tmp.data(), d_peaks.ptr, r.n_sync * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost),
                           "hipMemcpy peaks");
        uint64_t *pos = host_alloc<uint64_t>(r.n_sync);
        for (size_t k = 0; k < r.n_sync; ++k) pos[k] = tmp[k];
        *pos_out = pos;
        *n_pos = r.n_sync;
        if (correlation_out) {
            *correlation_out = sc.download_malloc(d_c.ptr, n_corr);
            if (n_corr_out) *n_corr_out = n_corr;
        }
        return APTGPU_OK;
    });
}

}  // extern "C"
