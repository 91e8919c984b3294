// apt_plan.hip — plan construction (host-side design, HBM workspace) and the
// kernel pipeline of decode() (reference: src/decode.rs:43-162).
#include "apt_plan.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace apt {

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        throw Error{ErrorKind::Hip, std::string(what) + ": " + hipGetErrorString(e), static_cast<int>(e)};
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
    plan->sw = gpu::read_launch_switches();
    plan->export_filtered = settings.export_resample_filtered != 0;
    if (const char *e = std::getenv("APTGPU_FORCE_WALK")) plan->picker_force = e[0] == '1' ? 1 : 0;
    if (const char *e = std::getenv("APTGPU_PICKER_LDS")) if (e[0] == '1') plan->picker_force = 4;

    // decode.rs:55 — u32 arithmetic; the reference would panic on overflow
    const uint64_t spr64 = static_cast<uint64_t>(PX_PER_ROW) * settings.work_rate;
    if (spr64 > 0xFFFFFFFFull) throw Error{ErrorKind::Invalid, "work_rate too large"};
    plan->spr = static_cast<uint32_t>(spr64 / FINAL_RATE);
    // (the reference would divide by zero at decode.rs:142 / find nothing to sync on)
    if (plan->spr == 0) throw Error{ErrorKind::Invalid, "work_rate too small"};

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
    // Calls in flight = streams (call j runs on stream j % depth and owns that stream's slots).
    if (depth <= 0) {
        const char *e = std::getenv("APTGPU_STREAMS");
        depth = e ? std::atoi(e) : (max_batch >= 4 ? 3 : 6);
    }
    depth = std::max(1, std::min(depth, 16));
    plan->streams.resize(static_cast<size_t>(depth));
    // A/B switches of the pipeline's shape (tools/sweep.py; read at plan creation):
    //   APTGPU_FRONT_STREAM=1  all front ends on one extra stream (apt_plan.hpp)
    //   APTGPU_CHAIN_CUS=n     with it: the per-call streams (words, slots, orbit, gather) are created with a CU mask of
    //                          n CUs, APTGPU_FRONT_EXCL=1: and the front-end stream with the mask of the others
    int cu_count = 0;
    (void)hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, plan->device);
    const char *e_fs = std::getenv("APTGPU_FRONT_STREAM");
    const bool want_front_stream = e_fs && e_fs[0] == '1' && !plan->user_stream && depth > 1 && max_batch >= 4;
    const char *e_cc = std::getenv("APTGPU_CHAIN_CUS");
    plan->chain_cus = (want_front_stream && e_cc) ? std::max(0, std::min(std::atoi(e_cc), cu_count - 8)) : 0;
    auto masked_stream = [&](hipStream_t *st, int first, int count) {
        // bit i of the mask = CU i in the runtime's numbering (consecutive bits go round the XCDs)
        std::vector<uint32_t> mask(static_cast<size_t>((cu_count + 31) / 32), 0u);
        for (int c = first; c < first + count && c < cu_count; ++c) mask[static_cast<size_t>(c / 32)] |= 1u << (c % 32);
        hip_check(hipExtStreamCreateWithCUMask(st, static_cast<uint32_t>(mask.size()), mask.data()), "hipExtStreamCreateWithCUMask");
    };
    for (auto &st : plan->streams) {
        if (plan->chain_cus > 0) masked_stream(&st, 0, plan->chain_cus);
        else hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    }
    plan->stream = plan->streams[0];
    if (want_front_stream) {
        const char *e_ex = std::getenv("APTGPU_FRONT_EXCL");
        if (plan->chain_cus > 0 && e_ex && e_ex[0] == '1') masked_stream(&plan->front_stream, plan->chain_cus, cu_count - plan->chain_cus);
        else hip_check(hipStreamCreateWithFlags(&plan->front_stream, hipStreamNonBlocking), "hipStreamCreate");
        plan->ev_pre.resize(static_cast<size_t>(depth));
        for (auto &ev : plan->ev_pre) hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    }
    {
        // (not with a user stream: such plans may be captured into a HIP graph, call by call, and an event
        // recorded before the capture cannot be waited for inside it)
        const char *e = std::getenv("APTGPU_FRONT_SERIAL");  // A/B switch
        plan->front_serial = ((e ? e[0] != '0' : max_batch >= 4) && !plan->user_stream && depth > 1) || plan->front_stream;
        if (plan->front_serial) {
            plan->ev_front.resize(static_cast<size_t>(depth));
            for (auto &ev : plan->ev_front) hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        }
    }

    plan->max_work_len = plan->work_len_for(max_samples);
    const uint64_t rows = plan->spr ? plan->max_work_len / plan->spr + 2 : 2;
    plan->max_rows = static_cast<uint32_t>(rows);

    auto upload = [&](DeviceBuffer<float> &dst, const Signal &src) {
        dst.alloc(src.size());
        // on the plan's own stream (the null stream would serialise against other plans' work);
        // the source is a temporary of the caller, so wait for the copy before returning
        hip_check(hipMemcpyAsync(dst.ptr, src.data(), src.size() * sizeof(float),
                                 hipMemcpyHostToDevice, plan->stream),
                  "hipMemcpy taps");
        hip_check(hipStreamSynchronize(plan->stream), "hipStreamSynchronize");
    };
    upload(plan->d_taps_resample, plan->taps_resample);
    upload(plan->d_taps_lowpass, plan->taps_lowpass);
    upload(plan->d_one, Signal{1.f});
    {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        const uint32_t t2 = static_cast<uint32_t>(plan->taps_lowpass.size());
        // APTGPU_MODE_FAST permits deviations within the stated tolerance; where no fast kernel exists
        // (other rates / profiles) the strict kernels serve it
        const bool eligible = (plan->mode == APTGPU_MODE_STRICT || plan->mode == APTGPU_MODE_FAST) && plan->l > 1 &&
                              plan->work_is_multiple && !plan->export_filtered;
        const char *force_any = std::getenv("APTGPU_FUSED_ANY");  // tests: run-time kernel even if specialised
        plan->fused = 0;
        const bool no_spec = force_any && force_any[0] == '1';
        const char *phase_first = std::getenv("APTGPU_PHASE_FIRST");  // tests: 0 = the table-driven stage 1 wherever it exists
        // APTGPU_MODE_FAST on the matrix cores (kModeMfma: 48 / 96 kHz at the standard profile, ANY tap count its K holds).
        // Measured at parity with / a few percent behind the VALU fast kernels at the stock tap counts (fast
        // mode is bound by the tile's HBM round trip, not by the FIRs), so it serves the plans those kernels cannot — a tuned
        // resample_atten / resample_delta_freq, which changes the tap count — and, APTGPU_FAST_MFMA=1 (A/B switch, tests), all
        // it has an instantiation for; APTGPU_FAST_MFMA=0 switches it off.
        const char *mfma_env = std::getenv("APTGPU_FAST_MFMA");
        const bool mfma_want = mfma_env ? mfma_env[0] == '1' : !gpu::fused_supported(plan->l, plan->m, t1, t2, plan->pw);
        const bool mfma_ok = eligible && !no_spec && plan->mode == APTGPU_MODE_FAST && mfma_want &&
                             gpu::fused_mfma_supported(plan->l, plan->m, t1, t2, plan->pw);
        // a tap count no specialised kernel is compiled for (a tuned resample_atten / resample_delta_freq): the strict kernel
        // compiled for a tap-count BOUND, its table zero-padded (kModeStrictPad; APTGPU_FUSED_PAD=1 forces it for the stock
        // counts too, =0 switches it off: A/B, tests)
        const char *pad_env = std::getenv("APTGPU_FUSED_PAD");
        const bool exact = gpu::fused_supported(plan->l, plan->m, t1, t2, plan->pw);
        const uint32_t pad_t1 = (eligible && !no_spec && !mfma_ok && (pad_env ? pad_env[0] == '1' : !exact))
                                    ? gpu::fused_pad_t1(plan->l, plan->m, t1, t2, plan->pw) : 0u;
        plan->fused_pad_t1 = 0;
        plan->fused_pad_t2 = 0;
        if (eligible && (exact || mfma_ok || pad_t1 != 0) && !no_spec) {
            plan->fused = 1;
            plan->fused_pad_t1 = pad_t1;
            plan->fused_pad_t2 = pad_t1 ? gpu::fused_pad_t2(plan->l, plan->m, t2, plan->pw) : 0u;  // (a tuned demodulation_atten)
        }
        // (where both the table-driven and the phase-resident stage 1 exist, the latter: since round 5 — thread assignment
        // lists, interior tile loads, pipelined taps, two / four branches per thread — it is the faster one at every rate
        // measured (8 / 11.025 / 16 / 32 kHz: 0.60 / 0.64 / 0.71 / 0.95 against 0.70 / 0.68 / 0.94 / 1.56 ms per 16 recordings,
        // profiles/r05_sweeps.txt); APTGPU_PHASE_FIRST=0 (tests) puts the table-driven form first again)
        else if (eligible && !no_spec && !(phase_first && phase_first[0] == '0') &&
                 gpu::fused_phase_supported(plan->l, plan->m, t1, t2, plan->pw, &plan->table_geom))
            plan->fused = 4;
        else if (eligible && !no_spec && gpu::fused_table_supported(plan->l, plan->m, t1, t2, plan->pw, &plan->table_geom))
            plan->fused = 3;  // table-driven stage 1 + the specialised work-rate stages (11 025 Hz)
        else if (eligible && !no_spec && gpu::fused_phase_supported(plan->l, plan->m, t1, t2, plan->pw, &plan->table_geom))
            plan->fused = 4;  // phase-resident taps in stage 1 + the specialised work-rate stages (44 100 Hz)
        else if (eligible && gpu::fused_any_supported(plan->l, plan->m, t1, t2, plan->pw))
            plan->fused = 2;
        // A plan whose resampling factors have a specialised kernel but whose tap COUNT differs (a tuned
        // resample_atten / resample_delta_freq) lands on a slower kernel — the count fixes which window samples each
        // polyphase branch uses, a compile-time pattern.  Same results; say so once instead of leaving it to
        // stats.fused (APTGPU_QUIET=1 silences it).
        if (eligible && !no_spec && plan->fused != 1 && plan->l == 13 && (plan->m == 50 || plan->m == 100) && plan->pw == 3) {
            static std::atomic<bool> told{false};
            const char *q = std::getenv("APTGPU_QUIET");
            if (!(q && q[0] == '1') && !told.exchange(true))
                std::fprintf(stderr,
                             "aptgpu: %u -> %u Hz with %u resample / %u low-pass taps has no compile-time specialised front end "
                             "(those exist for up to 1079 / 2145 resample taps and up to 45 low-pass taps at 48 / 96 kHz); using kernel path %d "
                             "(same results, lower throughput)\n",
                             plan->input_rate, plan->settings.work_rate, t1, t2, plan->fused);
        }
        plan->fused_mfma = mfma_ok && plan->fused == 1;
        plan->fused_fast = plan->mode == APTGPU_MODE_FAST &&
                           ((plan->fused == 1 && plan->fused_pad_t1 == 0 &&
                             (plan->fused_mfma || gpu::fused_fast_supported(plan->l, plan->m, t1, t2, plan->pw))) ||
                            plan->fused == 3 ||
                            // (a tuned low-pass on the PHASE kernels: strict kModeStrictPad2 instantiations only)
                            (plan->fused == 4 && gpu::fused_phase_pad_t2(t2, plan->pw) == 0));
        // fp16-tap mode inside the specialised fused kernel where one exists (else the generic kernel)
        // (not with export_resample_filtered: the fused kernels decimate at t = off + k m, the flag moves the phase —
        // dsp.rs:265-273 — and work_len_for() follows the flag; the unfused k_resample_at path serves such a plan)
        if (plan->mode == APTGPU_MODE_FP16_TAPS && plan->l > 1 && plan->work_is_multiple && !plan->export_filtered &&
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
        hip_check(hipMemcpyAsync(plan->d_taps_f16.ptr, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, plan->stream),
                  "hipMemcpy f16 taps");
        hip_check(hipStreamSynchronize(plan->stream), "hipStreamSynchronize");  // `tab` dies here
    }
    if (plan->fused != 0) {
        // division by sin(phi) through its reciprocal: only if it is exactly rounded for every x
        const float rc = 1.0f / plan->sinphi;
        const char *off = std::getenv("APTGPU_GENERAL_ENVELOPE");  // tests: force the general code
        if (plan->fused_fast) plan->inv_sinphi = rc;  // a plain multiplication by the rounded reciprocal
        else if (!(off && off[0] == '1') && gpu::verify_fast_divide(plan->device, plan->sinphi, rc)) plan->inv_sinphi = rc;
    }
    if (plan->fused == 2 || plan->fused == 3 || plan->fused == 4) {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        Signal tab(static_cast<size_t>(plan->fused == 4 ? gpu::fused_phase_table_floats(plan->table_geom)
                                                        : gpu::fused_any_table_floats(plan->l, t1)) + 16, 0.f);
        if (plan->fused == 4)
            gpu::fused_phase_table(plan->table_geom, static_cast<uint32_t>(plan->taps_lowpass.size()), plan->pw,
                                   plan->taps_resample.data(), t1, tab.data());
        else gpu::fused_any_table(plan->l, plan->taps_resample.data(), t1, tab.data());
        upload(plan->d_taps_any, tab);
        // (PHASE kernels with a tuned low-pass — kModeStrictPad2: the low-pass tables laid out for the kernel's bound)
        Signal lp(plan->taps_lowpass);
        plan->fused_pad_t2 = plan->fused == 4 ? gpu::fused_phase_pad_t2(static_cast<uint32_t>(lp.size()), plan->pw) : 0u;
        if (plan->fused_pad_t2) {
            lp.resize(plan->fused_pad_t2, 0.f);
            Signal padded(lp);
            padded.resize(lp.size() + 16, 0.f);
            upload(plan->d_taps_lowpass_pad, padded);
        }
        Signal h2p(2 * (lp.size() + 1) + 16, 0.f);
        gpu::fused_lowpass_pairs(lp.data(), static_cast<uint32_t>(lp.size()), h2p.data());
        upload(plan->d_taps_lowpass_pairs, h2p);
    }
    if (plan->fused == 1) {
        const uint32_t t1 = static_cast<uint32_t>(plan->taps_resample.size());
        // (the kernel that will read it: fast plans run the fast instantiation, whose chunks may differ)
        const int ch = gpu::fused_chunk_of(plan->m, plan->fused_fast);
        const uint32_t t1_layout = plan->fused_pad_t1 ? plan->fused_pad_t1 : t1;
        Signal hs(static_cast<size_t>(gpu::fused_tap_table_floats(plan->l, plan->m, t1_layout, ch)) + 16, 0.f);
        if (plan->fused_mfma) {
            // same buffer, different content: the bf16 fragments of the resampler's Toeplitz matrix
            std::vector<uint32_t> tab(gpu::fused_mfma_table_dwords(plan->l, plan->m) + 16, 0u);
            gpu::fused_mfma_table(plan->l, plan->m, plan->taps_resample.data(), t1, tab.data());
            hs.assign(tab.size(), 0.f);
            std::memcpy(hs.data(), tab.data(), tab.size() * sizeof(uint32_t));
        } else if (plan->fused_f16) {
            // same buffer, different content: half2 tap pairs as raw dwords
            std::vector<uint32_t> tab(gpu::fused_f16_table_dwords(plan->l, plan->m, t1) + 16, 0u);
            plan->f16_unscale = gpu::fused_f16_branch_taps(plan->l, plan->m, plan->taps_resample.data(), t1, tab.data());
            hs.assign(tab.size(), 0.f);
            std::memcpy(hs.data(), tab.data(), tab.size() * sizeof(uint32_t));
        } else {
            gpu::fused_branch_taps(plan->l, plan->m, plan->taps_resample.data(), t1, ch, hs.data(), t1_layout);
        }
        upload(plan->d_taps_branch, hs);
        // (kModeStrictPad2: the low-pass tables laid out for the kernel's bound, zeros behind the filter's last tap)
        Signal lp(plan->taps_lowpass);
        if (plan->fused_pad_t2) {
            lp.resize(plan->fused_pad_t2, 0.f);
            Signal padded(lp);
            padded.resize(lp.size() + 16, 0.f);
            upload(plan->d_taps_lowpass_pad, padded);
        }
        Signal h2p(2 * (lp.size() + 1) + 16, 0.f);
        gpu::fused_lowpass_pairs(lp.data(), static_cast<uint32_t>(lp.size()), h2p.data());
        upload(plan->d_taps_lowpass_pairs, h2p);
    }

    // workspace: one set of max_batch slots per stream
    plan->slots.resize(static_cast<size_t>(depth) * static_cast<size_t>(max_batch));
    const uint64_t w = plan->max_work_len;
    for (auto &sl : plan->slots) {
        // (resampled / demodulated / correlation are only needed by the unfused kernels: allocated on first use)
        sl.filtered.alloc(w + 64);
        if (sync) {
            sl.peaks.alloc(plan->max_rows + 2);
            const uint64_t ng = w / gpu::sync_group_size() + 2;
            const uint64_t chunks = ng / gpu::sync_chunk_groups() + 2;
            sl.gm.alloc(ng + 64);
            sl.words.alloc(ng + 64);
            sl.nanw.alloc(ng + 64);
            sl.slot_nt.alloc(chunks * gpu::sync_slot_cap());
            sl.slot_cnt.alloc(chunks);
            sl.orbit_ws.alloc(gpu::sync_orbit_ws_words(w, plan->spr));
            sl.flags.alloc(32);
            hip_check(hipMemsetAsync(sl.flags.ptr, 0, 32 * sizeof(uint32_t), plan->stream), "hipMemset flags");
            if (plan->fused == 0) sl.correlation.alloc(w + 64);
            if (plan->mode == APTGPU_MODE_GENERIC) sl.bits.alloc(w / 64 + plan->md / 64 + 4);
        }
    }
    plan->d_results.alloc(plan->slots.size());
    hip_check(hipMemsetAsync(plan->d_results.ptr, 0, sizeof(gpu::Result) * plan->slots.size(), plan->stream),
              "hipMemset");
    plan->d_slots.alloc(plan->slots.size());
    // the uploads and memsets above ran on the plan's first stream: make them land before a decode can
    // be enqueued on any of the others (a stream sync, not a device sync — other plans of this process,
    // e.g. concurrent aptgpu_decode calls on other host threads, keep running)
    plan->upload_slot_table();
    return plan.release();
}

}  // namespace apt

void aptgpu_plan::upload_slot_table()
{
    std::vector<apt::gpu::SlotPtrs> tab(slots.size());
    for (size_t k = 0; k < slots.size(); ++k) {
        Slot &sl = slots[k];
        apt::gpu::SlotPtrs &t = tab[k];
        t.f = sl.filtered.ptr;
        t.gm = sl.gm.ptr;
        t.corr = sl.correlation.ptr;
        t.words = sl.words.ptr;
        t.nanw = sl.nanw.ptr;
        t.slot_nt = sl.slot_nt.ptr;
        t.slot_cnt = sl.slot_cnt.ptr;
        t.flags = sl.flags.ptr;
        t.orbit_ws = sl.orbit_ws.ptr;
        t.peaks = sl.peaks.ptr;
        t.res = d_results.ptr + k;
        t.peaks_cap = static_cast<uint32_t>(sl.peaks.count);
        t.reserved = 0;
    }
    apt::hip_check(hipMemcpyAsync(d_slots.ptr, tab.data(), tab.size() * sizeof(apt::gpu::SlotPtrs),
                                  hipMemcpyHostToDevice, stream),
                   "hipMemcpy slot table");
    apt::gpu::FusedParams prm{};
    if (fused == 1 || fused == 3 || fused == 4) {
        prm.hs = d_taps_branch.ptr;
        prm.table = d_taps_any.ptr;
        prm.tab = table_geom;
        prm.h2 = fused_pad_t2 ? d_taps_lowpass_pad.ptr : d_taps_lowpass.ptr;
        prm.h2p = d_taps_lowpass_pairs.ptr;
        prm.t2 = static_cast<uint32_t>(taps_lowpass.size());
        prm.slots = d_slots.ptr;
        prm.cosphi2 = cosphi2;
        prm.sinphi = sinphi;
        prm.inv_sinphi = inv_sinphi;
        prm.f16_unscale = fused_f16 ? f16_unscale : 0.f;
        prm.coeff = d_taps_resample.ptr;
        prm.t1 = static_cast<uint32_t>(taps_resample.size());
        prm.want_gm = (sync && work_is_multiple) ? 1 : 0;
        prm.gm_slack = apt::gpu::fused_gm_slack(pw, sw.gm_slack_scale);
        if (!d_fused_params.ptr) d_fused_params.alloc(1);
        apt::hip_check(hipMemcpyAsync(d_fused_params.ptr, &prm, sizeof prm, hipMemcpyHostToDevice, stream),
                       "hipMemcpy fused params");
    }
    apt::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
}

// ------------------------------------------------------------------ geometry
uint64_t aptgpu_plan::work_len_for(uint64_t n) const
{
    if (l > 1)
        return export_filtered ? apt::fast_resampling_export_geom(n, l, m, taps_resample.size()).count
                               : apt::fast_resampling_len(n, l, m, taps_resample.size());
    return n / m;  // decimate, dsp.rs:299-303
}

uint64_t aptgpu_plan::out_len_nosync(uint64_t work_len) const
{
    const uint64_t aligned = spr ? work_len / spr * spr : 0;  // decode.rs:142-147
    if (l2 > 1)
        return export_filtered ? apt::fast_resampling_export_geom(aligned, l2, m2, 1).count
                               : apt::fast_resampling_len(aligned, l2, m2, 1);
    return aligned / m2;
}

// ------------------------------------------------------- export_resample_filtered
// fast_resampling (dsp.rs:186-289) evaluated at t = off + d0 + k*step of the interpolated axis: the window of
// such a t starts at the first multiple of l at or after t - off = d (:237-248), input x0 = ceil(d / l), tap
// p = x0*l - d, then taps p + i*l while they stay <= 2*off (:254); inputs at or beyond n are skipped (:257).
// step == m, d0 == 0 is the normal branch (k_resample_generic); the export branch (:265-273) needs d0 =
// fast_resampling_export_geom().d0 for the output and step == 1 for the expanded signal.  Only this mode uses the
// kernel, which is why it lives beside the plan and not with the hot kernels.
namespace {
__global__ void __launch_bounds__(256)
k_resample_at(const float *__restrict__ x, uint64_t n, const float *__restrict__ coeff, uint32_t jlim, uint32_t l,
              uint64_t d0, uint32_t step, float *__restrict__ out, uint64_t w)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < w; k += stride) {
        const uint64_t d = d0 + k * step;
        uint64_t xi = (d + l - 1) / l;
        const uint32_t p = static_cast<uint32_t>(xi * l - d);
        float sum = 0.f;
        for (uint32_t j = p; j < jlim; j += l, ++xi)
            if (xi < n) sum = __fadd_rn(sum, __fmul_rn(coeff[j], x[xi]));
        out[k] = sum;
    }
}

}  // namespace

namespace apt {
void resample_at(hipStream_t s, const float *x, uint64_t n, const float *coeff, uint32_t ntaps, uint32_t l, uint64_t d0,
                 uint32_t step, float *out, uint64_t w)
{
    if (w == 0) return;
    const uint32_t jlim = 2 * ((ntaps - 1) / 2) + 1;  // n <= t + offset  <=>  j <= 2*offset
    const uint64_t blocks = std::min<uint64_t>((w + 255) / 256, 256u * 64u);
    hipLaunchKernelGGL(k_resample_at, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, x, n, coeff, jlim, l, d0, step,
                       out, w);
}
}  // namespace apt
using apt::resample_at;

void aptgpu_plan::expanded_filtered(hipStream_t s, const float *d_x, uint64_t n, bool final_stage, float *d_out,
                                    uint64_t count)
{
    if (final_stage) resample_at(s, d_x, n, d_one.ptr, 1, l2, 0, 1, d_out, count);
    else resample_at(s, d_x, n, d_taps_resample.ptr, static_cast<uint32_t>(taps_resample.size()), l, 0, 1, d_out, count);
    apt::hip_check(hipGetLastError(), "kernel launch (expanded signal)");
}

void aptgpu_plan::sync_all()
{
    if (front_stream) apt::hip_check(hipStreamSynchronize(front_stream), "hipStreamSynchronize");
    for (hipStream_t st : streams) apt::hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
}

// ------------------------------------------------------------------ pipeline
// The kernel sequence of decode() for the recordings of one call, all already in HBM.
void aptgpu_plan::run_call(int count, const Input *ins, float *const *d_rows, const uint64_t *rows_cap_floats,
                           bool keep_steps)
{
    using namespace apt::gpu;
    if (count < 0 || count > max_batch) throw apt::Error{apt::ErrorKind::Invalid, "more recordings than max_batch"};
    last_stream = static_cast<int>(calls++ % streams.size());
    hipStream_t cur = streams[static_cast<size_t>(last_stream)];
    const int slot0 = last_stream * max_batch;
    last_slots.assign(static_cast<size_t>(count), 0);
    for (int i = 0; i < count; ++i) last_slots[static_cast<size_t>(i)] = slot0 + i;
    // the call's stream waits for what is already enqueued on ctx.stream (inputs may be produced there)
    if (user_stream) {
        apt::hip_check(hipEventRecord(ev_user, user_stream), "hipEventRecord");
        apt::hip_check(hipStreamWaitEvent(cur, ev_user, 0), "hipStreamWaitEvent");
    }
    auto timed_on = [&](hipStream_t st, const char *name, auto &&launch) {
        const bool dominant = !std::strcmp(name, "fused_front_end") || !std::strcmp(name, "resample_generic") ||
                              !std::strcmp(name, "resample_f16taps");
        timer.begin(st, name, dominant);
        launch();
        timer.end(st);
    };
    auto timed = [&](const char *name, auto &&launch) { timed_on(cur, name, launch); };

    const bool use_fused = fused != 0 && !keep_steps;
    const bool want_sync = sync && work_is_multiple;
    if (!use_fused && sync) {
        // the unfused kernels keep the full correlation; a fused plan that exports steps gets the
        // buffers (and their entry in the slot table) on first use
        bool grew = false;
        for (int i = 0; i < count; ++i) {
            Slot &sl = slots[static_cast<size_t>(slot0 + i)];
            if (!sl.correlation.ptr) {
                sl.correlation.alloc(max_work_len + 64);
                grew = true;
            }
        }
        if (grew) {
            sync_all();  // (no launch in flight may be reading the table while it is rewritten)
            upload_slot_table();
        }
    }

    // recordings that go through the kernels; the error paths of decode() are settled per recording
    std::vector<int> live;
    std::vector<uint64_t> wlen(static_cast<size_t>(count));
    std::vector<const void *> xin(static_cast<size_t>(count));
    std::vector<char> is_pcm(static_cast<size_t>(count), 0);
    for (int i = 0; i < count; ++i) {
        const Input &in = ins[i];
        Slot &sl = slots[static_cast<size_t>(slot0 + i)];
        Result *res = d_results.ptr + slot0 + i;
        const uint64_t w = work_len_for(in.n);
        wlen[static_cast<size_t>(i)] = w;
        // decode.rs:79-83 — fewer than 10 rows of samples
        if (w < 10ull * spr) {
            set_result(cur, res, Result{APTGPU_ERR_INTERNAL, 1, 0, 0, w, 0});
            continue;
        }
        // WAV ingest (wav.rs:30-51): mono PCM16 goes straight into the fused front end, anything
        // else is converted into the slot's f32 staging buffer first
        const bool pcm16 = in.codec == static_cast<int>(apt::WavCodec::I16) && in.channels == 1 && use_fused &&
                           (reinterpret_cast<uintptr_t>(in.ptr) & (fused == 1 ? 3u : 1u)) == 0 &&
                           (fused != 1 || apt::gpu::fused_takes_pcm16(l, m));
        xin[static_cast<size_t>(i)] = in.ptr;
        is_pcm[static_cast<size_t>(i)] = pcm16 ? 1 : 0;
        if (in.codec >= 0 && !pcm16) {
            if (!sl.ingest.ptr) sl.ingest.alloc(max_samples + 16);
            timed("wav_to_signal", [&] {
                wav_to_signal(cur, in.ptr, in.n, in.channels, in.bytes_per_sample, in.codec, sl.ingest.ptr);
            });
            xin[static_cast<size_t>(i)] = sl.ingest.ptr;
        }
        live.push_back(i);
    }

    // per-stage launches cover up to kMaxCall recordings each
    auto make_call = [&](const std::vector<int> &idx, size_t from, size_t to, uint64_t *max_w, uint32_t *max_cap) {
        CallArgs c{};
        c.count = static_cast<uint32_t>(to - from);
        *max_w = 0;
        *max_cap = 0;
        for (size_t k = from; k < to; ++k) {
            const int i = idx[k];
            RecArgs &r = c.rec[k - from];
            r.x = xin[static_cast<size_t>(i)];
            r.n = ins[i].n;
            r.w = wlen[static_cast<size_t>(i)];
            r.rows = d_rows[i];
            uint64_t rc = rows_cap_floats[i] / 2080u;
            if (rc > max_rows) rc = max_rows;
            r.rows_cap = static_cast<uint32_t>(rc);
            r.slot = static_cast<uint32_t>(slot0 + i);
            *max_w = std::max(*max_w, r.w);
            *max_cap = std::max(*max_cap, r.rows_cap);
        }
        return c;
    };
    auto for_chunks = [&](const std::vector<int> &idx, auto &&fn) {
        for (size_t from = 0; from < idx.size(); from += static_cast<size_t>(kMaxCall)) {
            const size_t to = std::min(idx.size(), from + static_cast<size_t>(kMaxCall));
            uint64_t max_w = 0;
            uint32_t max_cap = 0;
            const CallArgs c = make_call(idx, from, to, &max_w, &max_cap);
            fn(c, max_w, max_cap);
        }
    };

    const uint32_t t1 = static_cast<uint32_t>(taps_resample.size());
    const uint32_t t2 = static_cast<uint32_t>(taps_lowpass.size());
    if (use_fused && (fused == 1 || fused == 3 || fused == 4)) {
        // 1-3 fused: resample -> envelope -> low-pass (-> correlation maxima) in one launch per input
        // kind (apt_kernels_fused.hip), behind the previous call's front end (see front_serial, apt_plan.hpp)
        hipStream_t fs = cur;
        if (front_stream && !live.empty()) {
            // the front end runs on the front-end stream, behind everything the call's stream holds so far
            fs = front_stream;
            apt::hip_check(hipEventRecord(ev_pre[static_cast<size_t>(last_stream)], cur), "hipEventRecord");
            apt::hip_check(hipStreamWaitEvent(fs, ev_pre[static_cast<size_t>(last_stream)], 0), "hipStreamWaitEvent");
        } else if (front_serial && !live.empty() && prev_front >= 0 && prev_front != last_stream)
            apt::hip_check(hipStreamWaitEvent(cur, ev_front[static_cast<size_t>(prev_front)], 0), "hipStreamWaitEvent");
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<int> idx;
            for (int i : live)
                if (is_pcm[static_cast<size_t>(i)] == kind) idx.push_back(i);
#if defined(APT_WITH_PROBES) || defined(APT_DEBUG_SKIP)
            const int rep_f = [] { const char *e = std::getenv("APTGPU_DEBUG_REPEAT_FRONT"); return e ? std::max(1, std::atoi(e)) : 1; }();
#else
            constexpr int rep_f = 1;
#endif
            for (int rf = 0; rf < rep_f; ++rf)
            for_chunks(idx, [&](const CallArgs &c, uint64_t max_w, uint32_t) {
                timed_on(fs, "fused_front_end", [&] {
                    const int kmode = fused_f16 ? 1 : (fused_mfma ? 3 : (fused_pad_t1 ? 4 : (fused_fast ? 2 : 0)));
                    const bool ok = fused == 4 ? fused_phase_front_end(fs, table_geom, t2, pw, kmode, kind == 1, c,
                                                                       d_fused_params.ptr, max_w)
                                  : fused == 3 ? fused_table_front_end(fs, table_geom, kmode, kind == 1, c,
                                                                       d_fused_params.ptr, max_w)
                                               : fused_front_end(fs, l, m, t1, t2, pw, kmode, kind == 1, c,
                                                                 d_fused_params.ptr, max_w, sw.fused_lds_pad);
                    if (!ok)
                        throw apt::Error{apt::ErrorKind::Internal, "fused front end: no kernel for this geometry"};
                });
            });
        }
        if (front_serial && !live.empty()) {
            apt::hip_check(hipEventRecord(ev_front[static_cast<size_t>(last_stream)], fs), "hipEventRecord");
            prev_front = last_stream;
            // (the chain follows on the call's stream)
            if (front_stream) apt::hip_check(hipStreamWaitEvent(cur, ev_front[static_cast<size_t>(last_stream)], 0), "hipStreamWaitEvent");
        }
    } else if (use_fused) {
        // the run-time front end (k_fused_any: fast / slow profiles, odd rates): one launch per input kind too
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<int> idx;
            for (int i : live)
                if (is_pcm[static_cast<size_t>(i)] == kind) idx.push_back(i);
            for_chunks(idx, [&](const CallArgs &c, uint64_t max_w, uint32_t) {
                timed("fused_front_end", [&] {
                    if (!fused_any_front_end(cur, l, m, t1, t2, pw, kind == 1, c, d_slots.ptr, max_w, d_taps_any.ptr,
                                             d_taps_lowpass.ptr, d_taps_lowpass_pairs.ptr, cosphi2, sinphi, inv_sinphi,
                                             want_sync, sw.gm_slack_scale))
                        throw apt::Error{apt::ErrorKind::Internal, "fused front end: no kernel for this geometry"};
                });
            });
        }
    } else {
        for (int i : live) {
            Slot &sl = slots[static_cast<size_t>(slot0 + i)];
            const uint64_t w = wlen[static_cast<size_t>(i)];
            const uint64_t n = ins[i].n;
            const float *d_signal = static_cast<const float *>(xin[static_cast<size_t>(i)]);
            if (!sl.resampled.ptr) {
                sl.resampled.alloc(max_work_len + 64);
                sl.demodulated.alloc(max_work_len + 64);
            }
            // 1. resample to work_rate (dsp.rs:62-126)
            if (l > 1 && export_filtered) {
                const uint64_t d0 = apt::fast_resampling_export_geom(n, l, m, t1).d0;
                timed("resample_generic", [&] { resample_at(cur, d_signal, n, d_taps_resample.ptr, t1, l, d0, m, sl.resampled.ptr, w); });
            } else if (l > 1 && mode == APTGPU_MODE_FP16_TAPS) {
                timed("resample_f16taps", [&] {
                    resample_f16taps(cur, d_signal, n, d_taps_f16.ptr, t1, l, m, f16_unscale, sl.resampled.ptr, w);
                });
            } else if (l > 1) {
                timed("resample_generic", [&] {
                    resample_generic(cur, d_signal, n, d_taps_resample.ptr, t1, l, m, sl.resampled.ptr, w);
                });
            } else {
                timed("fir_decimate", [&] {
                    fir_decimate(cur, d_signal, n, d_taps_resample.ptr, t1, m, sl.resampled.ptr, w);
                });
            }
            // 2. AM envelope (dsp.rs:350-383)
            timed("demodulate", [&] { demodulate(cur, sl.resampled.ptr, w, cosphi2, sinphi, sl.demodulated.ptr); });
            // 3. low-pass (dsp.rs:386-410)
            timed("lowpass", [&] {
                fir_decimate(cur, sl.demodulated.ptr, w, d_taps_lowpass.ptr, t2, 1, sl.filtered.ptr, w);
            });
            // 4a. the full correlation of find_sync (decode.rs:225-233) and its group maxima
            if (want_sync) {
                const uint64_t n_corr = w - n_sync_taps;  // w >= 10*spr > 38*pw
                timed("correlate", [&] { correlate(cur, sl.filtered.ptr, n_corr, pw, sl.correlation.ptr); });
                if (mode != APTGPU_MODE_GENERIC)
                    timed("group_max", [&] { group_max(cur, sl.correlation.ptr, n_corr, sl.gm.ptr); });
            }
        }
    }

    if (sync && !work_is_multiple) {
        // generate_sync_frame, decode.rs:172-176
        for (int i : live)
            set_result(cur, d_results.ptr + slot0 + i, Result{APTGPU_ERR_INTERNAL, 3, 0, 0, wlen[static_cast<size_t>(i)], 0});
    } else if (sync) {
        bool gather_wanted = true;
        int gather_reps = 1;
        // 4. find_sync (decode.rs:204-263): terminal flags, orbit
        if (mode == APTGPU_MODE_GENERIC) {
            // reference-shaped picker: full sliding-window terminals + sequential orbit
            for (int i : live) {
                Slot &sl = slots[static_cast<size_t>(slot0 + i)];
                const uint64_t w = wlen[static_cast<size_t>(i)];
                const uint64_t n_corr = w - n_sync_taps;
                timed("terminals", [&] { terminals(cur, sl.correlation.ptr, n_corr, md, sl.bits.ptr); });
                timed("orbit_walk", [&] {
                    orbit_walk(cur, sl.bits.ptr, sl.correlation.ptr, n_corr, w, spr, md, sl.peaks.ptr,
                               static_cast<uint32_t>(sl.peaks.count),
                               d_results.ptr + slot0 + i);
                });
            }
        } else {
            // (probe builds only — make PROBES=1, or libaptgpu_probe.so of `make probe-lib` — APTGPU_DEBUG_SKIP, a bit mask: 1 words + slots, 2 orbit,
            // 4 gather.  Timing experiments: what each kernel behind the front end costs a pipelined step; the results
            // are then garbage, so a product build does not even read the variable.)
#if defined(APT_WITH_PROBES) || defined(APT_DEBUG_SKIP)
            const int skip = [] {
                const char *e = std::getenv("APTGPU_DEBUG_SKIP");
                return e ? std::atoi(e) : 0;
            }();
            // APTGPU_DEBUG_REPEAT_WORDS / _ORBIT / _GATHER = n: the kernel is launched n times instead of once (each is
            // idempotent: same inputs, same outputs) — what ONE more launch of it costs a pipelined step, with every
            // result still valid
            const int rep_w = [] { const char *e = std::getenv("APTGPU_DEBUG_REPEAT_WORDS"); return e ? std::max(1, std::atoi(e)) : 1; }();
            const int rep_o = [] { const char *e = std::getenv("APTGPU_DEBUG_REPEAT_ORBIT"); return e ? std::max(1, std::atoi(e)) : 1; }();
            const int rep_g = [] { const char *e = std::getenv("APTGPU_DEBUG_REPEAT_GATHER"); return e ? std::max(1, std::atoi(e)) : 1; }();
#else
            constexpr int skip = 0, rep_w = 1, rep_o = 1, rep_g = 1;
#endif
            gather_reps = rep_g;
            for_chunks(live, [&](const CallArgs &c, uint64_t max_w, uint32_t) {
                if (!(skip & 1))
                    for (int r = 0; r < rep_w; ++r)
                        timed("sync_nodes", [&] { sync_nodes(cur, c, d_slots.ptr, max_w, pw, spr, md, use_fused && fused_fast, !use_fused, sw); });
                if (!(skip & 2))
                    for (int r = 0; r < rep_o; ++r)
                        timed("sync_orbit", [&] { sync_orbit(cur, c, d_slots.ptr, spr, md, pw, picker_force, sw); });
            });
            gather_wanted = !(skip & 4);
        }
        // 5. aligned rows + final /pw (decode.rs:120-134,158-159)
        if (gather_wanted)
            for (int r = 0; r < gather_reps; ++r)
                for_chunks(live, [&](const CallArgs &c, uint64_t, uint32_t max_cap) {
                    timed("gather_rows", [&] { gather_rows_call(cur, c, d_slots.ptr, spr, pw, max_cap, sw); });
                });
    } else {
        // decode.rs:135-159 — crop to whole rows, resample_with_filter(NoFilter).  Where that is a pure decimation
        // (every stock profile: l2 == 1) the recordings of the call share ONE launch, result records included
        if (l2 == 1 && !keep_steps) {
            for (size_t from = 0; from < live.size(); from += static_cast<size_t>(kMaxCall)) {
                const size_t to = std::min(live.size(), from + static_cast<size_t>(kMaxCall));
                uint64_t max_w = 0;
                uint32_t max_cap = 0;
                CallArgs c = make_call(live, from, to, &max_w, &max_cap);
                for (size_t k = from; k < to; ++k)  // (floats, not rows, in this launch)
                    c.rec[k - from].rows_cap = static_cast<uint32_t>(std::min<uint64_t>(rows_cap_floats[live[k]], 0xFFFFFFFFull));
                timed("final_decimate", [&] { nosync_rows_call(cur, c, d_slots.ptr, spr, m2, max_w); });
            }
        } else
        for (int i : live) {
            Slot &sl = slots[static_cast<size_t>(slot0 + i)];
            const uint64_t w = wlen[static_cast<size_t>(i)];
            const uint64_t aligned = w / spr * spr;
            uint64_t n_out = out_len_nosync(w);
            if (n_out > rows_cap_floats[i]) n_out = rows_cap_floats[i];
            if (l2 > 1 && export_filtered) {
                const uint64_t d0 = apt::fast_resampling_export_geom(aligned, l2, m2, 1).d0;
                timed("final_resample", [&] { resample_at(cur, sl.filtered.ptr, aligned, d_one.ptr, 1, l2, d0, m2, d_rows[i], n_out); });
            } else if (l2 > 1) {
                timed("final_resample", [&] {
                    resample_generic(cur, sl.filtered.ptr, aligned, d_one.ptr, 1, l2, m2, d_rows[i], n_out);
                });
            } else {
                timed("final_decimate", [&] {
                    fir_decimate(cur, sl.filtered.ptr, aligned, d_one.ptr, 1, m2, d_rows[i], n_out);
                });
            }
            set_result(cur, d_results.ptr + slot0 + i,
                       Result{APTGPU_OK, 0, static_cast<uint32_t>(n_out / 2080u), 0, w, n_out});
        }
    }
    // a launch that failed (bad grid / LDS size, an earlier fault) would otherwise leave a stale or
    // zeroed — i.e. "OK" — result record behind
    apt::hip_check(hipGetLastError(), "kernel launch (decode chain)");
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
        // (no memset: the first kernel of every variant resets its record)
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
    apt::hip_check(hipGetLastError(), "kernel launch (image stage)");
}
