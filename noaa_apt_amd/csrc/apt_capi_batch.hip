// apt_capi_batch.hip — host-fed batch decode over one or more GPUs (include/aptgpu.h §2b).
//
// What a batch driver over independent recordings binds: the loop `for file in files { load();
// decode(); }` of the reference's CLI (src/main.rs:102-104), with the recordings sharded over the
// GPUs of the node.  Recordings never talk to each other (decode() touches only its arguments), so
// there is no collective of any kind: every device entry gets a host thread, a plan and its share of
// the recordings (longest first onto the least loaded entry — the same rule as
// noaa_apt_amd/shard.py), and works through it in calls of `recordings_per_call`.
//
// Per worker, chunk k+1's inputs go up (copy stream, its own device buffers) while chunk k decodes
// (the plan's streams): a worker owns TWO sets of input / output device buffers and alternates.  WAV
// file images are uploaded as their data-chunk payload (2 bytes per sample for PCM16) and converted
// on the device.  Host buffers that are pinned (aptgpu_host_alloc, or hipHostMalloc/hipHostRegister by
// the caller) are DMA'd directly; pageable ones go through the runtime's staging path.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "apt_capi_util.hpp"

namespace {

using namespace apt::capi;

struct Item {
    int index;            // position in the caller's arrays
    const uint8_t *data;  // f32 samples, or the payload of the WAV data chunk
    uint64_t bytes;       // bytes to upload
    uint64_t n;           // samples (frames)
    apt::WavInfo wav;     // when is_wav
    bool is_wav = false;
};

struct Shared {
    const aptgpu_settings *settings;
    uint32_t rate;
    bool sync;
    int mode;
    int per_call;
    float **rows_out;
    size_t *n_out;
    int32_t *status;
    aptgpu_result *results;  // nullable
    std::mutex mu;
    std::string first_error;
    int first_error_code = APTGPU_OK;
    double h2d_seconds = 0, d2h_seconds = 0;  // summed over workers (host wall time inside the copies)
    uint64_t h2d_bytes = 0, d2h_bytes = 0;
};

struct PlanDeleter {
    void operator()(aptgpu_plan *p) const { aptgpu_plan_destroy(p); }
};

void fail_item(Shared &sh, const Item &it, int code)
{
    sh.status[it.index] = code;
    sh.rows_out[it.index] = nullptr;
    sh.n_out[it.index] = 0;
}

// one device entry: decodes `items` (already ordered) on `device`
void worker(Shared &sh, int device, std::vector<Item> items)
{
    using clock = std::chrono::steady_clock;
    auto seconds = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        apt::hip_check(hipSetDevice(device), "hipSetDevice");
        uint64_t max_n = 0, max_bytes = 0;
        for (const Item &it : items) {
            max_n = std::max(max_n, it.n);
            max_bytes = std::max(max_bytes, it.bytes);
        }
        const int B = std::max(1, std::min<int>(sh.per_call, static_cast<int>(items.size())));
        aptgpu_context ctx{};
        ctx.device = device;
        ctx.mode = sh.mode;
        // two calls in flight: the plan's stream of call k+1 picks up where the copy stream left its inputs
        std::unique_ptr<aptgpu_plan, PlanDeleter> plan(
            apt::plan_create(&ctx, *sh.settings, sh.rate, sh.sync, max_n, B, /*depth*/ 2));
        const uint64_t out_cap = sh.sync ? static_cast<uint64_t>(plan->max_rows) * 2080u
                                         : plan->out_len_nosync(plan->work_len_for(max_n)) + 16;
        hipStream_t copy = nullptr;
        apt::hip_check(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking), "hipStreamCreate");
        struct Set {
            std::vector<apt::DeviceBuffer<uint8_t>> in;
            std::vector<apt::DeviceBuffer<float>> out;
            hipEvent_t uploaded = nullptr, decoded = nullptr;
        } sets[2];
        for (Set &s : sets) {
            s.in.resize(static_cast<size_t>(B));
            s.out.resize(static_cast<size_t>(B));
            for (int b = 0; b < B; ++b) {
                s.in[static_cast<size_t>(b)].alloc(max_bytes + 64);
                s.out[static_cast<size_t>(b)].alloc(out_cap);
            }
            apt::hip_check(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming), "hipEventCreate");
            apt::hip_check(hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming), "hipEventCreate");
        }
        double t_h2d = 0, t_d2h = 0;
        uint64_t b_h2d = 0, b_d2h = 0;

        const size_t n_chunks = (items.size() + static_cast<size_t>(B) - 1) / static_cast<size_t>(B);
        auto chunk_of = [&](size_t c, size_t *from, size_t *to) {
            *from = c * static_cast<size_t>(B);
            *to = std::min(items.size(), *from + static_cast<size_t>(B));
        };
        // upload chunk c into set c % 2 (its previous user, chunk c - 2, was collected before)
        auto upload = [&](size_t c) {
            Set &s = sets[c % 2];
            size_t from, to;
            chunk_of(c, &from, &to);
            const auto a = clock::now();
            for (size_t k = from; k < to; ++k) {
                const Item &it = items[k];
                if (it.bytes)
                    apt::hip_check(hipMemcpyAsync(s.in[k - from].ptr, it.data, it.bytes, hipMemcpyHostToDevice, copy),
                                   "hipMemcpyAsync H2D");
                b_h2d += it.bytes;
            }
            apt::hip_check(hipEventRecord(s.uploaded, copy), "hipEventRecord");
            t_h2d += seconds(a, clock::now());
        };
        // enqueue the decode of chunk c behind its upload
        std::vector<std::vector<int>> chunk_slots(n_chunks);
        auto decode = [&](size_t c) {
            Set &s = sets[c % 2];
            size_t from, to;
            chunk_of(c, &from, &to);
            const int cnt = static_cast<int>(to - from);
            std::vector<aptgpu_plan::Input> ins(static_cast<size_t>(cnt));
            std::vector<float *> rows(static_cast<size_t>(cnt));
            std::vector<uint64_t> caps(static_cast<size_t>(cnt), out_cap);
            for (int b = 0; b < cnt; ++b) {
                const Item &it = items[from + static_cast<size_t>(b)];
                aptgpu_plan::Input &in = ins[static_cast<size_t>(b)];
                in.ptr = s.in[static_cast<size_t>(b)].ptr;
                in.n = it.n;
                if (it.is_wav) {
                    in.channels = it.wav.channels;
                    in.bytes_per_sample = it.wav.bytes_per_sample;
                    in.codec = static_cast<int>(it.wav.codec);
                }
                rows[static_cast<size_t>(b)] = s.out[static_cast<size_t>(b)].ptr;
            }
            // the call's stream (plan->streams[calls % 2]) must not start before the inputs have landed
            hipStream_t next = plan->streams[static_cast<size_t>(plan->calls % plan->streams.size())];
            apt::hip_check(hipStreamWaitEvent(next, s.uploaded, 0), "hipStreamWaitEvent");
            plan->run_call(cnt, ins.data(), rows.data(), caps.data(), false);
            apt::hip_check(hipEventRecord(s.decoded, next), "hipEventRecord");
            chunk_slots[c] = plan->last_slots;
        };
        // wait for chunk c, read its records, copy the rows out
        auto collect = [&](size_t c) {
            Set &s = sets[c % 2];
            size_t from, to;
            chunk_of(c, &from, &to);
            apt::hip_check(hipEventSynchronize(s.decoded), "hipEventSynchronize");
            for (size_t k = from; k < to; ++k) {
                const Item &it = items[k];
                aptgpu_result r{};
                apt::hip_check(hipMemcpy(&r, plan->d_results.ptr + chunk_slots[c][k - from], sizeof r, hipMemcpyDeviceToHost),
                               "hipMemcpy result");
                if (sh.results) sh.results[it.index] = r;
                if (r.status != APTGPU_OK) {
                    fail_item(sh, it, r.status);
                    continue;
                }
                float *rows = static_cast<float *>(std::malloc((r.n_out ? r.n_out : 1) * sizeof(float)));
                if (!rows) throw std::bad_alloc();
                const auto a = clock::now();
                if (r.n_out)
                    apt::hip_check(hipMemcpy(rows, s.out[k - from].ptr, r.n_out * sizeof(float), hipMemcpyDeviceToHost),
                                   "hipMemcpy D2H rows");
                t_d2h += seconds(a, clock::now());
                b_d2h += r.n_out * sizeof(float);
                sh.rows_out[it.index] = rows;
                sh.n_out[it.index] = r.n_out;
                sh.status[it.index] = APTGPU_OK;
            }
        };

        // software pipeline: upload(c+1) is issued before chunk c is collected, so the copy engine works
        // on the next inputs while the kernels of chunk c run
        if (n_chunks > 0) upload(0);
        for (size_t c = 0; c < n_chunks; ++c) {
            decode(c);
            if (c + 1 < n_chunks) {
                if (c >= 1) collect(c - 1);  // frees set (c + 1) % 2
                upload(c + 1);
            }
        }
        if (n_chunks >= 2) collect(n_chunks - 2);
        if (n_chunks >= 1) collect(n_chunks - 1);
        plan->sync_all();
        for (Set &s : sets) {
            (void)hipEventDestroy(s.uploaded);
            (void)hipEventDestroy(s.decoded);
        }
        (void)hipStreamDestroy(copy);
        std::lock_guard<std::mutex> lock(sh.mu);
        sh.h2d_seconds += t_h2d;
        sh.d2h_seconds += t_d2h;
        sh.h2d_bytes += b_h2d;
        sh.d2h_bytes += b_d2h;
    } catch (const Error &e) {
        std::lock_guard<std::mutex> lock(sh.mu);
        if (sh.first_error_code == APTGPU_OK) {
            sh.first_error_code = static_cast<int>(e.kind);
            sh.first_error = e.message;
        }
        for (const Item &it : items)
            if (sh.status[it.index] == -1) fail_item(sh, it, static_cast<int>(e.kind));
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lock(sh.mu);
        if (sh.first_error_code == APTGPU_OK) {
            sh.first_error_code = APTGPU_ERR_INTERNAL;
            sh.first_error = e.what();
        }
        for (const Item &it : items)
            if (sh.status[it.index] == -1) fail_item(sh, it, APTGPU_ERR_INTERNAL);
    }
}

int decode_batch_impl(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                      int count, const void *const *inputs, const size_t *n, bool wav_images, const int32_t *devices,
                      int n_devices, int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                      aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    if (!settings || count < 0 || (count && (!inputs || !n || !rows_out || !n_out || !status)) || n_devices < 0 ||
        (n_devices && !devices)) {
        put_err(err, err_cap, "bad argument to aptgpu_decode_batch");
        return APTGPU_ERR_INVALID;
    }
    return guarded(err, err_cap, [&]() -> int {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<int32_t> devs(devices, devices + n_devices);
        if (devs.empty()) devs.push_back(ctx ? ctx->device : 0);
        int visible = 0;
        apt::hip_check(hipGetDeviceCount(&visible), "hipGetDeviceCount");
        for (int32_t d : devs)
            if (d < 0 || d >= visible) throw Error{ErrorKind::Invalid, "device ordinal out of range"};

        Shared sh{};
        sh.settings = settings;
        sh.rate = input_rate_hz;
        sh.sync = sync != 0;
        sh.mode = ctx ? ctx->mode : APTGPU_MODE_STRICT;
        sh.per_call = std::max(1, std::min(recordings_per_call <= 0 ? 8 : recordings_per_call, apt::gpu::kMaxCall));
        sh.rows_out = rows_out;
        sh.n_out = n_out;
        sh.status = status;
        sh.results = results;
        std::vector<Item> items;
        uint64_t total_samples = 0;
        for (int i = 0; i < count; ++i) {
            status[i] = -1;  // pending
            rows_out[i] = nullptr;
            n_out[i] = 0;
            if (results) results[i] = aptgpu_result{};
            Item it;
            it.index = i;
            if (!inputs[i] && n[i]) {
                status[i] = APTGPU_ERR_INVALID;
                continue;
            }
            if (wav_images) {
                try {
                    it.wav = apt::parse_wav(static_cast<const uint8_t *>(inputs[i]), n[i]);
                } catch (const Error &e) {
                    status[i] = static_cast<int>(e.kind);  // the reference's error for this file (err.rs:72-83)
                    continue;
                }
                if (it.wav.sample_rate != input_rate_hz) {
                    status[i] = APTGPU_ERR_INVALID;  // one (settings, rate) per batch: the plans are built for it
                    continue;
                }
                it.is_wav = true;
                it.data = static_cast<const uint8_t *>(inputs[i]) + it.wav.data_offset;
                it.bytes = it.wav.data_len;
                it.n = it.wav.n_frames;
            } else {
                it.data = static_cast<const uint8_t *>(inputs[i]);
                it.bytes = static_cast<uint64_t>(n[i]) * sizeof(float);
                it.n = n[i];
            }
            total_samples += it.n;
            items.push_back(it);
        }
        // longest first onto the least loaded entry (LPT; noaa_apt_amd/shard.py uses the same rule across ranks)
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.n > b.n; });
        std::vector<std::vector<Item>> share(devs.size());
        std::vector<uint64_t> load(devs.size(), 0);
        for (const Item &it : items) {
            const size_t k = static_cast<size_t>(std::min_element(load.begin(), load.end()) - load.begin());
            share[k].push_back(it);
            load[k] += it.n;
        }
        std::vector<std::thread> threads;
        for (size_t k = 0; k < devs.size(); ++k)
            if (!share[k].empty()) threads.emplace_back(worker, std::ref(sh), static_cast<int>(devs[k]), std::move(share[k]));
        for (auto &t : threads) t.join();
        if (stats) {
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            stats->samples = total_samples;
            stats->h2d_bytes = sh.h2d_bytes;
            stats->d2h_bytes = sh.d2h_bytes;
            stats->h2d_seconds = sh.h2d_seconds;
            stats->d2h_seconds = sh.d2h_seconds;
            stats->workers = static_cast<int32_t>(threads.size());
            stats->recordings_per_call = sh.per_call;
        }
        if (sh.first_error_code != APTGPU_OK) {
            put_err(err, err_cap, sh.first_error);
            return sh.first_error_code;
        }
        return APTGPU_OK;
    });
}

}  // namespace

extern "C" {

int aptgpu_decode_batch(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                        int count, const float *const *signals, const size_t *n, const int32_t *devices, int n_devices,
                        int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                        aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    return decode_batch_impl(ctx, settings, input_rate_hz, sync, count, reinterpret_cast<const void *const *>(signals), n,
                             false, devices, n_devices, recordings_per_call, rows_out, n_out, status, results, stats, err,
                             err_cap);
}

int aptgpu_decode_batch_wav(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                            int count, const void *const *wav_images, const size_t *wav_bytes, const int32_t *devices,
                            int n_devices, int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                            aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    return decode_batch_impl(ctx, settings, input_rate_hz, sync, count, wav_images, wav_bytes, true, devices, n_devices,
                             recordings_per_call, rows_out, n_out, status, results, stats, err, err_cap);
}

// Pinned host memory for inputs that should go over PCIe by direct DMA (a WAV reader can read a file
// straight into it).  Plain malloc'd buffers work everywhere in this API, at the runtime's staged rate.
void *aptgpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void aptgpu_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
