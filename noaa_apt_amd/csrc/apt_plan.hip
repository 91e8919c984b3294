// apt_plan.hip — plan construction (host-side design, HBM workspace) and the
// kernel pipeline of decode() (reference: src/decode.rs:43-162).
#include "apt_plan.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace apt {

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        throw Error{ErrorKind::Hip, std::string(what) + ": " + hipGetErrorString(e)};
    }
}

// ---------------------------------------------------------------- KernelTimer
KernelTimer::~KernelTimer()
{
    for (auto &p : pairs_) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : pool_) (void)hipEventDestroy(e);
}

void KernelTimer::enable(int mode) { mode_ = mode; }

hipEvent_t KernelTimer::take()
{
    if (!pool_.empty()) {
        hipEvent_t e = pool_.back();
        pool_.pop_back();
        return e;
    }
    hipEvent_t e;
    hip_check(hipEventCreate(&e), "hipEventCreate");
    return e;
}

void KernelTimer::begin(hipStream_t s, const char *name, bool dominant)
{
    // mode 1 samples: every 8th dominant launch is bracketed, so the markers (which serialise
    // the stream for ~6 us each) cost ~1.5 us per recording instead of ~12
    open_ = mode_ == 2 || (mode_ == 1 && dominant && (sampled_++ % 8) == 0);
    if (!open_) return;
    Pair p{name, take(), take()};
    hip_check(hipEventRecord(p.a, s), "hipEventRecord");
    pairs_.push_back(p);
}

void KernelTimer::end(hipStream_t s)
{
    if (!open_) return;
    open_ = false;
    hip_check(hipEventRecord(pairs_.back().b, s), "hipEventRecord");
}

std::vector<aptgpu_kernel_time> KernelTimer::collect(hipStream_t s)
{
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    std::vector<aptgpu_kernel_time> out;
    for (auto &p : pairs_) {
        float ms = 0.f;
        hip_check(hipEventElapsedTime(&ms, p.a, p.b), "hipEventElapsedTime");
        aptgpu_kernel_time *slot = nullptr;
        for (auto &o : out)
            if (std::strcmp(o.name, p.name) == 0) slot = &o;
        if (!slot) {
            aptgpu_kernel_time t{};
            std::snprintf(t.name, sizeof t.name, "%s", p.name);
            out.push_back(t);
            slot = &out.back();
        }
        slot->avg_ms += ms;
        slot->launches += 1;
        pool_.push_back(p.a);
        pool_.push_back(p.b);
    }
    pairs_.clear();
    for (auto &o : out)
        if (o.launches) o.avg_ms /= static_cast<double>(o.launches);
    return out;
}

// ---------------------------------------------------------------- plan_create
aptgpu_plan *plan_create(const aptgpu_context *ctx, const aptgpu_settings &settings,
                         uint32_t input_rate, bool sync, size_t max_samples, int max_batch, int depth)
{
    if (settings.export_resample_filtered)
        throw Error{ErrorKind::Unsupported,
                    "export_resample_filtered is not available on the GPU path"};
    if (input_rate == 0) throw Error{ErrorKind::Invalid, "input_rate is 0"};
    if (max_batch < 1) max_batch = 1;

    auto plan = std::make_unique<aptgpu_plan>();
    plan->device = ctx ? ctx->device : 0;
    plan->mode = ctx ? ctx->mode : APTGPU_MODE_STRICT;
    plan->settings = settings;
    plan->input_rate = input_rate;
    plan->sync = sync;
    plan->max_samples = max_samples;
    plan->max_batch = max_batch;
    if (const char *e = std::getenv("APTGPU_FORCE_WALK")) plan->picker_force = e[0] == '1' ? 1 : 0;
    if (const char *e = std::getenv("APTGPU_PICKER_LDS")) if (e[0] == '1') plan->picker_force = 4;

    // decode.rs:55 — u32 arithmetic; the reference would panic on overflow
    const uint64_t spr64 = static_cast<uint64_t>(PX_PER_ROW) * settings.work_rate;
    if (spr64 > 0xFFFFFFFFull) throw Error{ErrorKind::Invalid, "work_rate too large"};
    plan->spr = static_cast<uint32_t>(spr64 / FINAL_RATE);

    const Rate in_rate = Rate::hz(input_rate);
    const Rate work_rate = Rate::hz(settings.work_rate);

    // ---- first resample: decode.rs:65-77 -> dsp.rs:62-126
    if (work_rate.get_hz() == 0) throw Error{ErrorKind::Internal, "Can't resample to 0Hz"};
    LowpassDcRemoval f1(Freq::hz(settings.resample_cutout, in_rate), settings.resample_atten,
                        Freq::hz(settings.resample_delta_freq, in_rate));
    const LM lm = interpolation_factors(in_rate, work_rate);
    plan->l = lm.l;
    plan->m = lm.m;
    if (lm.l > 1) {
        Rate interpolated{};
        if (!in_rate.checked_mul(lm.l, &interpolated)) {
            char buf[512];
            std::snprintf(buf, sizeof buf,
                          "Can't resample, looks like the sample rates do not have a big\n"
                          "                divisor in common. input_rate: %u, output_rate: %u, "
                          "l: %u, m: %u",
                          in_rate.get_hz(), work_rate.get_hz(), lm.l, lm.m);
            throw Error{ErrorKind::RateOverflow, buf};
        }
        f1.resample(in_rate, interpolated);
    }
    plan->taps_resample = f1.design();

    // ---- demodulation constants: decode.rs:89, dsp.rs:360-363 (phi = 2*get_rad())
    const Freq carrier = Freq::hz(static_cast<float>(CARRIER_FREQ), work_rate);
    const float phi = 2.f * carrier.get_rad();
    plan->cosphi2 = cosf(phi) * 2.f;
    plan->sinphi = sinf(phi);

    // ---- low-pass: decode.rs:95-100
    const Freq cutout = Freq::pi_rad(static_cast<float>(FINAL_RATE) /
                                     static_cast<float>(work_rate.get_hz()));
    Lowpass f2(cutout, settings.demodulation_atten, cutout / 5.f);
    plan->taps_lowpass = f2.design();

    // ---- sync geometry: decode.rs:204-216
    plan->work_is_multiple = (work_rate.get_hz() % FINAL_RATE) == 0;
    plan->pw = work_rate.get_hz() / FINAL_RATE;
    plan->n_sync_taps = 38 * plan->pw;
    plan->md = static_cast<uint32_t>(static_cast<uint64_t>(plan->spr) * 8 / 10);
    if (sync && plan->work_is_multiple && static_cast<uint64_t>(plan->md) * 8 > 160u * 1024u)
        throw Error{ErrorKind::Unsupported, "work_rate too large for the sync search (LDS)"};

    // ---- final resample to 4160 Hz: decode.rs:158-159
    const LM lm2 = interpolation_factors(work_rate, Rate::hz(FINAL_RATE));
    plan->l2 = lm2.l;
    plan->m2 = lm2.m;
    if (lm2.l > 1) {
        Rate tmp{};
        if (!work_rate.checked_mul(lm2.l, &tmp)) {
            char buf[512];
            std::snprintf(buf, sizeof buf,
                          "Can't resample, looks like the sample rates do not have a big\n"
                          "                divisor in common. input_rate: %u, output_rate: %u, "
                          "l: %u, m: %u",
                          work_rate.get_hz(), FINAL_RATE, lm2.l, lm2.m);
            throw Error{ErrorKind::RateOverflow, buf};
        }
    }

    // ---- device side
    hip_check(hipSetDevice(plan->device), "hipSetDevice");
    if (ctx && ctx->stream) {
        plan->user_stream = static_cast<hipStream_t>(ctx->stream);
        hip_check(hipEventCreateWithFlags(&plan->ev_user, hipEventDisableTiming), "hipEventCreate");
    }

    plan->max_work_len = plan->work_len_for(max_samples);
    const uint64_t rows = plan->spr ? plan->max_work_len / plan->spr + 2 : 2;
    plan->max_rows = static_cast<uint32_t>(rows);

    auto upload = [&](DeviceBuffer<float> &dst, const Signal &src) {
        dst.alloc(src.size());
        hip_check(hipMemcpy(dst.ptr, src.data(), src.size() * sizeof(float),
                            hipMemcpyHostToDevice),
                  "hipMemcpy taps");
    };
    upload(plan->d_taps_resample, plan->taps_resample);
    upload(plan->d_taps_lowpass, plan->taps_lowpass);
    upload(plan->d_one, Signal{1.f});
    {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        const uint32_t t2 = static_cast<uint32_t>(plan->taps_lowpass.size());
        const bool eligible = plan->mode == APTGPU_MODE_STRICT && plan->l > 1 && plan->work_is_multiple;
        const char *force_any = std::getenv("APTGPU_FUSED_ANY");  // tests: run-time kernel even if specialised
        plan->fused = 0;
        if (eligible && gpu::fused_supported(plan->l, plan->m, t1, t2, plan->pw) && !(force_any && force_any[0] == '1'))
            plan->fused = 1;
        else if (eligible && gpu::fused_any_supported(plan->l, plan->m, t1, t2, plan->pw))
            plan->fused = 2;
        // fp16-tap mode inside the specialised fused kernel where one exists (else the generic kernel)
        if (plan->mode == APTGPU_MODE_FP16_TAPS && plan->l > 1 && plan->work_is_multiple &&
            gpu::fused_f16_supported(plan->l, plan->m, t1, t2, plan->pw)) {
            plan->fused = 1;
            plan->fused_f16 = true;
        }
    }
    if (plan->mode == APTGPU_MODE_FP16_TAPS && plan->l > 1) {  // (also the unfused step-export path of the fused fp16 mode)
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        std::vector<uint16_t> tab(static_cast<size_t>(plan->l) * gpu::f16taps_pairs_per_phase(plan->l, t1) * 2 + 8, 0);
        plan->f16_unscale = gpu::f16taps_pack(plan->l, plan->taps_resample.data(), t1, tab.data());
        plan->d_taps_f16.alloc(tab.size());
        hip_check(hipMemcpy(plan->d_taps_f16.ptr, tab.data(), tab.size() * 2, hipMemcpyHostToDevice), "hipMemcpy f16 taps");
    }
    if (plan->fused != 0) {
        // division by sin(phi) through its reciprocal: only if it is exactly rounded for every x
        const float rc = 1.0f / plan->sinphi;
        const char *off = std::getenv("APTGPU_GENERAL_ENVELOPE");  // tests: force the general code
        if (!(off && off[0] == '1') && gpu::verify_fast_divide(nullptr, plan->sinphi, rc)) plan->inv_sinphi = rc;
    }
    if (plan->fused == 2) {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        Signal tab(static_cast<size_t>(gpu::fused_any_table_floats(plan->l, t1)) + 16, 0.f);
        gpu::fused_any_table(plan->l, plan->taps_resample.data(), t1, tab.data());
        upload(plan->d_taps_any, tab);
        Signal h2p(2 * (plan->taps_lowpass.size() + 1) + 16, 0.f);
        gpu::fused_lowpass_pairs(plan->taps_lowpass.data(), static_cast<uint32_t>(plan->taps_lowpass.size()),
                                 h2p.data());
        upload(plan->d_taps_lowpass_pairs, h2p);
    }
    if (plan->fused == 1) {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        Signal hs(static_cast<size_t>(gpu::fused_tap_table_floats(plan->l, plan->m, t1)) + 16, 0.f);
        if (plan->fused_f16) {
            // same buffer, different content: half2 tap pairs as raw dwords
            std::vector<uint32_t> tab(gpu::fused_f16_table_dwords(plan->l, plan->m, t1) + 16, 0u);
            plan->f16_unscale = gpu::fused_f16_branch_taps(plan->l, plan->m, plan->taps_resample.data(), t1, tab.data());
            hs.assign(tab.size(), 0.f);
            std::memcpy(hs.data(), tab.data(), tab.size() * sizeof(uint32_t));
        } else {
            gpu::fused_branch_taps(plan->l, plan->m, plan->taps_resample.data(), t1, hs.data());
        }
        upload(plan->d_taps_branch, hs);
        Signal h2p(2 * (plan->taps_lowpass.size() + 1) + 16, 0.f);
        gpu::fused_lowpass_pairs(plan->taps_lowpass.data(), static_cast<uint32_t>(plan->taps_lowpass.size()),
                                 h2p.data());
        upload(plan->d_taps_lowpass_pairs, h2p);
    }

    // one extra slot so that consecutive calls (and consecutive recordings of a call) overlap
    // Recordings in flight = streams (slot k always runs on stream k % depth).  Measured on
    // MI355X at config 2 (ms per recording, HBM-cold inputs): 1: 0.143, 2: 0.117, 4: 0.119,
    // 5: 0.105, 6: 0.103, 7: 0.117, 8: 0.113, 10: 0.103 — six keeps enough front-end launches
    // queued that the tail of one is always filled by the head of the next, whatever the
    // latency of the picker chain behind it.
    if (depth <= 0) {
        // batch-capable plans run ONE front-end launch per call on a stream of its own and fan the
        // per-recording chains out over 3 streams (front + 3 = the 4 hardware queues; measured at
        // config 4's per-GPU share, 32 x 15 min: 1 chain stream 0.178, 2-4: 0.143, 6: 0.171 ms per
        // recording; the recording-by-recording pipeline with 6 streams: 0.154 ms)
        const char *e = std::getenv("APTGPU_STREAMS");
        depth = e ? std::atoi(e) : (max_batch >= 2 ? 3 : 6);
    }
    depth = std::max(1, std::min(depth, 16));
    plan->slots.resize(std::max<size_t>(max_batch >= 2 ? 2 * static_cast<size_t>(max_batch)
                                                       : static_cast<size_t>(max_batch) + (depth > 1 ? 1 : 0),
                                        static_cast<size_t>(depth)));
    plan->streams.resize(static_cast<size_t>(depth));
    for (auto &st : plan->streams)
        hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    plan->stream = plan->streams[0];
    if (max_batch >= 2) {
        // lowest priority: HIP keeps a separate hardware queue per priority level, so the batched
        // launch never sits in an in-order queue in front of (or behind) a chain stream's kernels,
        // and the small chain kernels are dispatched ahead of its thousands of workgroups
        int prio_least = 0, prio_greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest), "hipDeviceGetStreamPriorityRange");
        hip_check(hipStreamCreateWithPriority(&plan->stream_front, hipStreamNonBlocking, prio_least),
                  "hipStreamCreateWithPriority");
        hip_check(hipEventCreateWithFlags(&plan->ev_front, hipEventDisableTiming), "hipEventCreate");
        plan->d_batch.alloc(static_cast<size_t>(max_batch));
        for (auto &sl : plan->slots)
            hip_check(hipEventCreateWithFlags(&sl.ev_free, hipEventDisableTiming), "hipEventCreate");
    }
    const uint64_t w = plan->max_work_len;
    for (auto &sl : plan->slots) {
        // (resampled / demodulated are only needed by the unfused kernels: allocated on first use)
        sl.filtered.alloc(w + 64);
        if (sync) {
            sl.correlation.alloc(w + 64);
            sl.bits.alloc(w / 64 + plan->md / 64 + 4);
            sl.peaks.alloc(plan->max_rows + 2);
            const uint64_t ng = w / gpu::sync_group_size() + 2;
            const uint64_t chunks = ng / gpu::sync_chunk_groups() + 2;
            sl.gm.alloc(ng + 64);
            sl.words.alloc(ng + 64);
            sl.slot_nt.alloc(chunks * gpu::sync_slot_cap());
            sl.slot_cnt.alloc(chunks);
            sl.orbit_ws.alloc(gpu::sync_orbit_ws_words(w, plan->spr));
            sl.flags.alloc(32);
            hip_check(hipMemset(sl.flags.ptr, 0, 32 * sizeof(uint32_t)), "hipMemset flags");
        }
    }
    plan->d_results.alloc(plan->slots.size());
    hip_check(hipMemset(plan->d_results.ptr, 0, sizeof(gpu::Result) * plan->slots.size()), "hipMemset");
    // the uploads and memsets above went through the null stream, which the plan's non-blocking
    // streams do not wait for: make them land before the first decode can be enqueued
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    return plan.release();
}

}  // namespace apt

// ------------------------------------------------------------------ geometry
uint64_t aptgpu_plan::work_len_for(uint64_t n) const
{
    if (l > 1) return apt::fast_resampling_len(n, l, m, taps_resample.size());
    return n / m;  // decimate, dsp.rs:299-303
}

uint64_t aptgpu_plan::out_len_nosync(uint64_t work_len) const
{
    const uint64_t aligned = spr ? work_len / spr * spr : 0;  // decode.rs:142-147
    if (l2 > 1) return apt::fast_resampling_len(aligned, l2, m2, 1);
    return aligned / m2;
}

// ------------------------------------------------------------------ pipeline
// The kernel sequence of decode() for one recording already in HBM.
void aptgpu_plan::begin_call(int count)
{
    last_slots.assign(static_cast<size_t>(count), 0);
    // the streams this call uses wait for ctx.stream in enqueue()
    if (user_stream) apt::hip_check(hipEventRecord(ev_user, user_stream), "hipEventRecord");
}

void aptgpu_plan::sync_all()
{
    if (stream_front) apt::hip_check(hipStreamSynchronize(stream_front), "hipStreamSynchronize");
    for (hipStream_t st : streams) apt::hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
}

int aptgpu_plan::enqueue(int i, const Input &in, float *d_rows, uint64_t rows_cap_floats, bool keep_steps,
                         int forced_slot, bool front_done)
{
    using namespace apt::gpu;
    const uint64_t n = in.n;
    const int slot = forced_slot >= 0 ? forced_slot : static_cast<int>(seq++ % slots.size());
    if (static_cast<size_t>(i) < last_slots.size()) last_slots[static_cast<size_t>(i)] = slot;
    Slot &sl = slots[static_cast<size_t>(slot)];
    Result *res = d_results.ptr + slot;
    const uint64_t w = work_len_for(n);
    // everything of this recording runs in order on the slot's own stream; the recording that
    // reuses the slot is enqueued on the same stream, so no hand-over events are needed
    hipStream_t cur = streams[static_cast<size_t>(slot) % streams.size()];
    if (user_stream && !front_done) apt::hip_check(hipStreamWaitEvent(cur, ev_user, 0), "hipStreamWaitEvent");
    // batch-capable plans: tell the batched front end when this slot's chain is over
    struct FreeMark {
        aptgpu_plan *p;
        Slot &sl;
        hipStream_t st;
        ~FreeMark()
        {
            if (p->stream_front && sl.ev_free && hipEventRecord(sl.ev_free, st) == hipSuccess) sl.ev_free_recorded = true;
        }
    } free_mark{this, sl, cur};
    auto timed = [&](const char *name, auto &&launch) {
        const bool dominant = !std::strcmp(name, "fused_front_end") || !std::strcmp(name, "resample_generic") ||
                              !std::strcmp(name, "resample_f16taps");
        timer.begin(cur, name, dominant);
        launch();
        timer.end(cur);
    };

    // decode.rs:79-83 — fewer than 10 rows of samples
    if (w < 10ull * spr) {
        set_result(cur, res, Result{APTGPU_ERR_INTERNAL, 1, 0, 0, w, 0});
        return slot;
    }

    const bool use_fused = fused != 0 && !keep_steps;
    // WAV ingest (wav.rs:30-51): mono PCM16 goes straight into the fused front end, anything
    // else is converted into the slot's f32 staging buffer first
    const float *d_signal = static_cast<const float *>(in.ptr);
    const bool pcm16 = in.codec == static_cast<int>(apt::WavCodec::I16) && in.channels == 1 && use_fused &&
                       (reinterpret_cast<uintptr_t>(in.ptr) & (fused == 1 ? 3u : 1u)) == 0;
    if (in.codec >= 0 && !pcm16) {
        if (!sl.ingest.ptr) sl.ingest.alloc(max_samples + 16);
        timed("wav_to_signal", [&] {
            wav_to_signal(cur, in.ptr, n, in.channels, in.bytes_per_sample, in.codec, sl.ingest.ptr);
        });
        d_signal = sl.ingest.ptr;
    }
    if (use_fused && front_done) {
        // the batched launch of enqueue_batch() has produced F, C and GM for this slot already
    } else if (use_fused) {
        // 1-3 fused: resample -> envelope -> low-pass in one launch (apt_kernels_fused.hip)
        timed("fused_front_end", [&] {
            const uint32_t t1 = static_cast<uint32_t>(taps_resample.size());
            const uint32_t t2 = static_cast<uint32_t>(taps_lowpass.size());
            const void *xin = pcm16 ? in.ptr : static_cast<const void *>(d_signal);
            float *c_out = (sync && work_is_multiple) ? sl.correlation.ptr : nullptr;
            float *gm_out = (sync && work_is_multiple) ? sl.gm.ptr : nullptr;
            if (fused == 1)
                fused_front_end(cur, l, m, t1, t2, pw, xin, pcm16, n, d_taps_branch.ptr, d_taps_lowpass.ptr,
                                d_taps_lowpass_pairs.ptr, cosphi2, sinphi, inv_sinphi,
                                fused_f16 ? f16_unscale : 0.f, sl.filtered.ptr, c_out, gm_out, w,
                                w - n_sync_taps);
            else
                fused_any_front_end(cur, l, m, t1, t2, pw, xin, pcm16, n, d_taps_any.ptr, d_taps_lowpass.ptr,
                                    d_taps_lowpass_pairs.ptr, cosphi2, sinphi, inv_sinphi, sl.filtered.ptr, c_out,
                                    gm_out, w, w - n_sync_taps);
        });
    } else {
    if (!sl.resampled.ptr) {
        sl.resampled.alloc(max_work_len + 64);
        sl.demodulated.alloc(max_work_len + 64);
    }
    // 1. resample to work_rate (dsp.rs:62-126)
    if (l > 1 && mode == APTGPU_MODE_FP16_TAPS) {
        timed("resample_f16taps", [&] {
            resample_f16taps(cur, d_signal, n, d_taps_f16.ptr, static_cast<uint32_t>(taps_resample.size()),
                             l, m, f16_unscale, sl.resampled.ptr, w);
        });
    } else if (l > 1) {
        timed("resample_generic", [&] {
            resample_generic(cur, d_signal, n, d_taps_resample.ptr,
                             static_cast<uint32_t>(taps_resample.size()), l, m, sl.resampled.ptr, w);
        });
    } else {
        timed("fir_decimate", [&] {
            fir_decimate(cur, d_signal, n, d_taps_resample.ptr,
                         static_cast<uint32_t>(taps_resample.size()), m, sl.resampled.ptr, w);
        });
    }
    // 2. AM envelope (dsp.rs:350-383)
    timed("demodulate",
          [&] { demodulate(cur, sl.resampled.ptr, w, cosphi2, sinphi, sl.demodulated.ptr); });
    // 3. low-pass (dsp.rs:386-410)
    timed("lowpass", [&] {
        fir_decimate(cur, sl.demodulated.ptr, w, d_taps_lowpass.ptr,
                     static_cast<uint32_t>(taps_lowpass.size()), 1, sl.filtered.ptr, w);
    });
    }

    if (sync && !work_is_multiple) {
        // generate_sync_frame, decode.rs:172-176
        set_result(cur, res, Result{APTGPU_ERR_INTERNAL, 3, 0, 0, w, 0});
    } else if (sync) {
        // 4. find_sync (decode.rs:204-263): correlation, terminal flags, orbit
        const uint64_t n_corr = w - n_sync_taps;  // w >= 10*spr > 38*pw
        if (!use_fused)
            timed("correlate", [&] { correlate(cur, sl.filtered.ptr, n_corr, pw, sl.correlation.ptr); });
        if (mode == APTGPU_MODE_GENERIC) {
            // reference-shaped picker: full sliding-window terminals + sequential orbit
            timed("terminals", [&] { terminals(cur, sl.correlation.ptr, n_corr, md, sl.bits.ptr); });
            timed("orbit_walk", [&] {
                orbit_walk(cur, sl.bits.ptr, n_corr, w, spr, md, sl.peaks.ptr,
                           static_cast<uint32_t>(sl.peaks.count), res);
            });
        } else {
            if (!use_fused)
                timed("group_max", [&] { group_max(cur, sl.correlation.ptr, n_corr, sl.gm.ptr); });
            timed("sync_nodes", [&] {
                sync_nodes(cur, sl.gm.ptr, sl.correlation.ptr, n_corr, spr, md, sl.words.ptr,
                           sl.slot_nt.ptr, sl.slot_cnt.ptr, sl.flags.ptr);
            });
            timed("sync_orbit", [&] {
                sync_orbit(cur, sl.words.ptr, sl.slot_nt.ptr, sl.slot_cnt.ptr, sl.flags.ptr, n_corr,
                           w, spr, md, sl.orbit_ws.ptr, sl.peaks.ptr,
                           static_cast<uint32_t>(sl.peaks.count), res, picker_force);
            });
        }
        // 5. aligned rows + final /pw (decode.rs:120-134,158-159)
        uint64_t rows_cap = rows_cap_floats / 2080u;
        if (rows_cap > max_rows) rows_cap = max_rows;
        timed("gather_rows", [&] {
            gather_rows(cur, sl.filtered.ptr, sl.peaks.ptr, res, spr, pw, false, d_rows,
                        static_cast<uint32_t>(rows_cap));
        });
    } else {
        // decode.rs:135-159 — crop to whole rows, resample_with_filter(NoFilter)
        const uint64_t aligned = w / spr * spr;
        uint64_t n_out = out_len_nosync(w);
        if (n_out > rows_cap_floats) n_out = rows_cap_floats;
        if (l2 > 1) {
            timed("final_resample", [&] {
                resample_generic(cur, sl.filtered.ptr, aligned, d_one.ptr, 1, l2, m2, d_rows, n_out);
            });
        } else {
            timed("final_decimate", [&] {
                fir_decimate(cur, sl.filtered.ptr, aligned, d_one.ptr, 1, m2, d_rows, n_out);
            });
        }
        set_result(cur, res,
                   Result{APTGPU_OK, 0, static_cast<uint32_t>(n_out / 2080u), 0, w, n_out});
    }
    return slot;
}

// ------------------------------------------------------------------ image stage
void aptgpu_plan::enqueue_image(int i, const float *d_rows, uint64_t rows_cap_floats, int contrast,
                                float percent, bool rotate, uint8_t *d_image)
{
    using namespace apt::gpu;
    const size_t slot = static_cast<size_t>(last_slots[static_cast<size_t>(i)]);
    Slot &sl = slots[slot];
    uint64_t cap = static_cast<uint64_t>(max_rows) * 2080u;
    if (!sync) cap = out_len_nosync(work_len_for(max_samples)) + 16;
    if (rows_cap_floats < cap) cap = rows_cap_floats;
    const uint64_t ws_cap = std::max<uint64_t>(static_cast<uint64_t>(max_rows) * 2080u,
                                               out_len_nosync(work_len_for(max_samples)) + 16);
    if (!d_image_results.ptr) {
        // (no memset: a null-stream memset is not ordered with the plan's non-blocking streams and
        // could land after the kernels below; the first kernel of every variant resets its record)
        d_image_results.alloc(slots.size());
    }
    if (!sl.image_ws.ptr) sl.image_ws.alloc(image_ws_bytes(ws_cap));
    hipStream_t cur = stream_of(i);
    ImageResult *out = d_image_results.ptr + slot;
    const Result *res = d_results.ptr + slot;
    void *ws = sl.image_ws.ptr;
    auto timed = [&](const char *name, auto &&launch) {
        timer.begin(cur, name, false);
        launch();
        timer.end(cur);
    };
    // (the first kernel of every variant resets the record)
    if (contrast == APTGPU_CONTRAST_TELEMETRY)
        timed("image_telemetry", [&] { image_telemetry(cur, d_rows, res, 0, cap, ws, out, true); });
    else if (contrast == APTGPU_CONTRAST_PERCENT)
        timed("image_percent", [&] { image_percent(cur, d_rows, res, 0, cap, percent, ws, out); });
    else
        timed("image_minmax", [&] { image_minmax(cur, d_rows, res, 0, cap, ws, out); });
    timed("image_map_u8", [&] { image_map_u8(cur, d_rows, res, 0, cap, ws, rotate, d_image, out); });
}

// ------------------------------------------------------------------ batched front end
bool aptgpu_plan::enqueue_batch(int count, const Input *ins, float *const *d_rows, const uint64_t *rows_cap_floats)
{
    using namespace apt::gpu;
    if (fused != 1 || fused_f16 || !stream_front || count < 2 || count > max_batch) return false;
    if (!fused_batch_supported(l, m, static_cast<uint32_t>(taps_resample.size()),
                               static_cast<uint32_t>(taps_lowpass.size()), pw))
        return false;
    const bool pcm16 = ins[0].codec == static_cast<int>(apt::WavCodec::I16);
    for (int i = 0; i < count; ++i) {
        const Input &in = ins[i];
        const bool is_pcm = in.codec == static_cast<int>(apt::WavCodec::I16) && in.channels == 1 &&
                            (reinterpret_cast<uintptr_t>(in.ptr) & 3u) == 0;
        if (pcm16 ? !is_pcm : in.codec >= 0) return false;          // one input kind per launch
        if (work_len_for(in.n) < 10ull * spr) return false;         // error paths stay per recording
    }
    if (!h_batch)
        apt::hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_batch),
                                     4 * static_cast<size_t>(max_batch) * sizeof(FusedRec), hipHostMallocDefault),
                       "hipHostMalloc");
    FusedRec *recs = h_batch + (batch_calls++ % 4) * static_cast<size_t>(max_batch);
    std::vector<int> slot(static_cast<size_t>(count));
    uint64_t max_w = 0;
    const bool want_sync = sync && work_is_multiple;
    for (int i = 0; i < count; ++i) {
        slot[static_cast<size_t>(i)] = static_cast<int>(seq++ % slots.size());
        last_slots[static_cast<size_t>(i)] = slot[static_cast<size_t>(i)];
        Slot &sl = slots[static_cast<size_t>(slot[static_cast<size_t>(i)])];
        // the launch overwrites this slot: its previous chain must be over
        if (sl.ev_free_recorded) apt::hip_check(hipStreamWaitEvent(stream_front, sl.ev_free, 0), "hipStreamWaitEvent");
        const uint64_t w = work_len_for(ins[i].n);
        max_w = std::max(max_w, w);
        recs[i] = FusedRec{ins[i].ptr, ins[i].n, sl.filtered.ptr, want_sync ? sl.correlation.ptr : nullptr,
                           want_sync ? sl.gm.ptr : nullptr, w, w - n_sync_taps};
    }
    if (user_stream) apt::hip_check(hipStreamWaitEvent(stream_front, ev_user, 0), "hipStreamWaitEvent");
    apt::hip_check(hipMemcpyAsync(d_batch.ptr, recs, static_cast<size_t>(count) * sizeof(FusedRec),
                                  hipMemcpyHostToDevice, stream_front),
                   "hipMemcpyAsync batch records");
    timer.begin(stream_front, "fused_front_end", true);
    const bool ok = fused_front_end_batch(stream_front, l, m, static_cast<uint32_t>(taps_resample.size()),
                                          static_cast<uint32_t>(taps_lowpass.size()), pw, pcm16, d_batch.ptr, count,
                                          max_w, d_taps_branch.ptr, d_taps_lowpass.ptr, d_taps_lowpass_pairs.ptr,
                                          cosphi2, sinphi, inv_sinphi, fused_f16 ? f16_unscale : 0.f);
    timer.end(stream_front);
    if (!ok) throw apt::Error{apt::ErrorKind::Internal, "batched front end: no kernel for this geometry"};
    apt::hip_check(hipEventRecord(ev_front, stream_front), "hipEventRecord");
    for (int i = 0; i < count; ++i) {
        const int sidx = slot[static_cast<size_t>(i)];
        hipStream_t cur = streams[static_cast<size_t>(sidx) % streams.size()];
        apt::hip_check(hipStreamWaitEvent(cur, ev_front, 0), "hipStreamWaitEvent");
        enqueue(i, ins[i], d_rows[i], rows_cap_floats[i], false, sidx, true);
    }
    return true;
}
