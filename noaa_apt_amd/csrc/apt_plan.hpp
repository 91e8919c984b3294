// apt_plan.hpp — the decode() plan: designed taps, HBM workspace, stream, pipeline.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/aptgpu.h"
#include "apt_host.hpp"
#include "apt_kernels.hpp"
#include "apt_wav.hpp"

namespace apt {

// Throws apt::Error on failure.
void hip_check(hipError_t e, const char *what);

template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : ptr(o.ptr), count(o.count) { o.ptr = nullptr; o.count = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
    {
        if (this != &o) {
            release();
            ptr = o.ptr;
            count = o.count;
            o.ptr = nullptr;
            o.count = 0;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void alloc(size_t n)
    {
        release();
        count = n;
        if (n) hip_check(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)), "hipMalloc");
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

// Event-pair timing of kernel launches on the plan's stream.
class KernelTimer {
public:
    ~KernelTimer();
    // 0 = off, 1 = every 8th launch flagged `dominant`, 2 = every launch
    void enable(int mode);
    int mode() const { return mode_; }
    void begin(hipStream_t s, const char *name, bool dominant);
    void end(hipStream_t s);
    // synchronises the stream, folds all recorded pairs into per-name averages
    std::vector<aptgpu_kernel_time> collect(hipStream_t s);

private:
    struct Pair {
        const char *name;
        hipEvent_t a, b;
    };
    int mode_ = 0;
    bool open_ = false;
    uint64_t sampled_ = 0;
    std::vector<Pair> pairs_;
    std::vector<hipEvent_t> pool_;
    hipEvent_t take();
};

}  // namespace apt

namespace apt {
// fast_resampling (dsp.rs:186-289) evaluated at t = off + d0 + k*step of the interpolated axis, k < w (apt_plan.hip):
// the export_resample_filtered branch (dsp.rs:265-273) — d0 from fast_resampling_export_geom and step = m for the output,
// d0 = 0 and step = 1 for the expanded signal
void resample_at(hipStream_t s, const float *x, uint64_t n, const float *coeff, uint32_t ntaps, uint32_t l, uint64_t d0,
                 uint32_t step, float *out, uint64_t w);
}  // namespace apt

// The opaque C-ABI plan.
struct aptgpu_plan {
    int device = 0;
    int mode = APTGPU_MODE_STRICT;
    // Pipeline over consecutive calls: the recordings of ONE decode_device call go through ONE launch
    // per stage (front end -> k_sync_words -> k_sync_slots -> k_sync_orbit -> k_gather_rows; blockIdx.y / .x picks
    // the recording, per-recording arguments travel by value in the kernel arguments), in order on
    // one of `depth` streams; consecutive calls go round-robin over the streams, so the front end
    // of call j+1 overlaps the latency-bound picker of call j.  Stream k owns the workspace slots
    // [k*max_batch, (k+1)*max_batch): a slot is only ever reused by a later call on its own stream
    // (in-order => no hazards, no cross-stream events, nothing for the host to wait on).
    std::vector<hipStream_t> streams;
    hipStream_t stream = nullptr;       // = streams[0] (host-API helpers, plan-creation uploads)
    // Batched plans serialise the fused front ends of consecutive calls: the front end of call j+1 (on its own
    // stream) waits for an event recorded behind the front end of call j (on another), so front ends run one
    // after the other, each with the whole GPU, and the picker / gather of call j overlap the front end of
    // call j+1.  Left alone, the front ends of `depth` consecutive calls start together, share the GPU,
    // finish together — and their latency-bound chains then run with nothing to overlap (a kernel trace showed
    // three calls marching in step: 2.4 ms of front ends, then 0.8 ms of chains).  Worth 7 % in a 20-step run
    // (477 against 445 Gsamples/s), nothing in a long one (the chains cost their own time either way); one
    // extra low-priority stream for all front ends was tried as well and is no better.
    bool front_serial = false;
    std::vector<hipEvent_t> ev_front;   // [depth]: behind the front end of the latest call on that stream
    int prev_front = -1;                // stream index whose ev_front the next front end waits for
    // Alternative (APTGPU_FRONT_STREAM=1, an A/B switch): every front end on ONE extra in-order stream — consecutive
    // front ends then follow each other inside one hardware queue instead of through a cross-queue event — and the
    // chain of call j on stream j % depth behind an event recorded after its front end.  The front-end stream
    // waits for an event recorded on the call's stream first (whatever the caller ordered the call behind, and the
    // previous chain on those workspace slots).
    hipStream_t front_stream = nullptr;
    std::vector<hipEvent_t> ev_pre;     // [depth]: the call's stream just before its front end is enqueued
    int chain_cus = 0;                  // APTGPU_CHAIN_CUS=n: the per-call (chain) streams run on n CUs only
    hipStream_t user_stream = nullptr;  // ctx.stream: inputs are ordered after it (may be null)
    hipEvent_t ev_user = nullptr;
    uint64_t calls = 0;             // calls enqueued so far (stream = calls % depth)
    int last_stream = 0;            // stream index of the most recent call
    std::vector<int> last_slots;    // slot of recording i of the most recent decode_device call

    aptgpu_settings settings{};
    uint32_t input_rate = 0;
    bool sync = true;

    // first resample (decode.rs:65-77)
    uint32_t l = 1, m = 1;
    apt::Signal taps_resample;
    // demodulation (decode.rs:89; dsp.rs:360-363)
    float cosphi2 = 0.f, sinphi = 0.f;
    // low-pass (decode.rs:95-102)
    apt::Signal taps_lowpass;
    // sync (decode.rs:204-216)
    uint32_t spr = 0, md = 0, pw = 0, n_sync_taps = 0;
    bool work_is_multiple = false;
    // final resample to 4160 (decode.rs:158-159)
    uint32_t l2 = 1, m2 = 1;

    size_t max_samples = 0;
    uint64_t max_work_len = 0;
    uint32_t max_rows = 0;
    int max_batch = 1;
    // strict: RN(1/sinphi) if the exactly rounded fast divide verified for it, else 0 (apt_envelope.hpp);
    // fast mode: RN(1/sinphi)
    float inv_sinphi = 0.f;
    bool fused_f16 = false;   // APTGPU_MODE_FP16_TAPS served by the specialised fused kernel (fp16 stage 1)
    bool fused_fast = false;  // APTGPU_MODE_FAST served by the specialised fused kernel
    bool fused_mfma = false;  // ... by its matrix-core form (kModeMfma: tuned tap counts, or APTGPU_FAST_MFMA=1)
    uint32_t fused_pad_t1 = 0;  // != 0: the strict kernel compiled for this tap-count bound serves the plan (kModeStrictPad)
    uint32_t fused_pad_t2 = 0;  // != 0: ... and for this low-pass bound (kModeStrictPad2: d_taps_lowpass_pad, FusedParams::t2)
    // 0 unfused generic kernels, 1 compile-time specialised k_fused, 2 run-time k_fused_any,
    // 3 k_fused with the table-driven stage 1 (run-time l / m / taps, specialised work-rate stages)
    int fused = 0;
    apt::gpu::TableGeom table_geom{};
    apt::gpu::LaunchSwitches sw{};  // the A/B switches of the launch wrappers as the environment had them at plan creation
    int picker_force = 0;  // 0 parallel picker; APTGPU_FORCE_WALK=1 -> 1 (the sequential fallback)

    apt::DeviceBuffer<float> d_taps_resample, d_taps_lowpass, d_one, d_taps_branch, d_taps_lowpass_pairs, d_taps_any, d_taps_lowpass_pad;
    apt::DeviceBuffer<uint16_t> d_taps_f16;  // APTGPU_MODE_FP16_TAPS
    float f16_unscale = 1.f;
    struct Slot {
        apt::DeviceBuffer<float> resampled, demodulated, filtered;
        apt::DeviceBuffer<float> correlation;  // unfused kernels / step export only (the fused front ends never write it)
        apt::DeviceBuffer<uint64_t> bits;     // 64-bit terminal words (generic-mode picker)
        apt::DeviceBuffer<uint32_t> peaks;
        apt::DeviceBuffer<apt::gpu::GroupMax> gm;  // per-group maxima of the correlation
        apt::DeviceBuffer<uint64_t> words;    // 52-bit terminal words
        apt::DeviceBuffer<uint64_t> nanw;     // 52-bit words of NaN correlation positions
        apt::DeviceBuffer<uint32_t> slot_nt, slot_cnt, flags, orbit_ws;
        apt::DeviceBuffer<char> image_ws;  // scratch of the image stage, allocated on first use
        apt::DeviceBuffer<float> ingest;   // WAV -> f32 staging when the fused PCM16 path does not apply
    };
    std::vector<Slot> slots;
    apt::DeviceBuffer<apt::gpu::SlotPtrs> d_slots;  // the slots' pointers, for the per-call launches
    apt::DeviceBuffer<apt::gpu::FusedParams> d_fused_params;  // parameters of the specialised front end
    void upload_slot_table();                       // (re)writes d_slots + d_fused_params; synchronises plan->stream
    apt::DeviceBuffer<apt::gpu::Result> d_results;
    apt::DeviceBuffer<apt::gpu::ImageResult> d_image_results;  // one per slot, on first use
    // image stage (contrast limits -> u8, telemetry) of recording i of the last call, enqueued
    // behind its decode on the same stream
    void enqueue_image(int i, const float *d_rows, uint64_t rows_cap_floats, int contrast, float percent,
                       bool rotate, uint8_t *d_image);
    hipStream_t stream_of(int) { return streams[static_cast<size_t>(last_stream)]; }

    apt::KernelTimer timer;

    // geometry for an n-sample recording
    uint64_t work_len_for(uint64_t n) const;
    uint64_t out_len_nosync(uint64_t work_len) const;

    // Settings.export_resample_filtered (config.rs:83 -> context.rs:113): fast_resampling then walks every t of the
    // interpolated axis and decimates at (t + 1) % m == 0 (dsp.rs:265-273) — another phase than the t = off + k*m of
    // the normal branch, whether or not anything is exported.  Such a plan runs the unfused kernels.
    bool export_filtered = false;
    // the "resample_filtered" step of that mode (dsp.rs:269,281-285): every sum of the interpolated axis from t = off
    // on, `count` of them into d_out; final_stage: the NoFilter resample to 4160 Hz (decode.rs:158-159) instead of
    // the first one
    void expanded_filtered(hipStream_t s, const float *d_x, uint64_t n, bool final_stage, float *d_out, uint64_t count);

    // what a recording looks like in HBM: the f32 Signal (codec < 0), or the payload of a WAV
    // data chunk (codec = apt::WavCodec) that is converted on the device — inside the fused
    // front end for mono PCM16, through the slot's staging buffer otherwise
    struct Input {
        const void *ptr = nullptr;
        uint64_t n = 0;  // samples of the Signal == WAV frames
        uint32_t channels = 1, bytes_per_sample = 4;
        int codec = -1;
    };
    // enqueue the whole decode() of the `count` device-resident recordings of one call (count <=
    // max_batch) on the next stream; never synchronises with the host.  keep_steps: unfused kernels,
    // every intermediate stays in the slot (Context::step export).
    void run_call(int count, const Input *ins, float *const *d_rows, const uint64_t *rows_cap_floats,
                  bool keep_steps);
    void sync_all();                // waits for every internal stream
    Slot &slot_of(int i) { return slots[static_cast<size_t>(last_slots[static_cast<size_t>(i)])]; }
    apt::gpu::Result *result_of(int i) { return d_results.ptr + last_slots[static_cast<size_t>(i)]; }
};

namespace apt {

// Builds a plan; throws apt::Error with the reference's messages.
aptgpu_plan *plan_create(const aptgpu_context *ctx, const aptgpu_settings &settings,
                         uint32_t input_rate, bool sync, size_t max_samples, int max_batch,
                         int depth = 0 /* calls in flight; 0 = default (APTGPU_STREAMS overrides) */);

}  // namespace apt
