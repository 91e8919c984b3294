/*
 * aptgpu.h — C ABI of libaptgpu.so: the MI355X (gfx950) implementation of
 * martinber/noaa-apt's signal-to-image hot path, i.e. the body of
 * noaa_apt::decode()  (reference: src/decode.rs:43-162, re-exported at
 * src/noaa_apt.rs:5) and the dsp.rs / filters.rs functions it calls.
 *
 * This is the drop-in boundary: plain C types, plain pointers and sizes, no
 * C++/torch types.  A Rust `extern "C"` block binds exactly these symbols
 * (see INTEGRATION.md for the shim that gives them the reference's Rust
 * signatures).  Every entry point names the reference interface it replaces.
 *
 * Conventions
 *  - all functions return an APTGPU_* status code; on error `err` (if given)
 *    receives the same message string the reference puts in its err::Error.
 *  - "host" pointers are ordinary process memory; "d_" pointers are device
 *    (HBM) memory on the plan's GPU.
 *  - thread-safe and re-entrant; a plan must not be used from two threads at
 *    once (make one plan per thread / per stream).  The only process-global
 *    mutable state is the mutex-guarded session cache behind the host-array
 *    entry points (aptgpu_cache_clear / aptgpu_cache_info) and per-device
 *    caches of kernel attributes and of the envelope's divide check.
 *  - numerics: f32 throughout, every product and sum rounded separately and
 *    accumulated in the reference's order, so outputs are bit-identical to
 *    the reference's scalar loops (APTGPU_MODE_STRICT, the default).
 */
#ifndef APTGPU_H
#define APTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define APTGPU_OK 0
#define APTGPU_ERR_INTERNAL 1      /* err::Error::Internal(String)      src/err.rs:27   */
#define APTGPU_ERR_RATE_OVERFLOW 2 /* err::Error::RateOverflow(String)  src/err.rs:31   */
#define APTGPU_ERR_HIP 3           /* HIP runtime / device failure (Rust shim: Internal) */
#define APTGPU_ERR_INVALID 4       /* FFI misuse: null pointer, capacity too small, ...  */
#define APTGPU_ERR_UNSUPPORTED 5   /* reference feature not offered on the GPU path      */
#define APTGPU_ERR_WAV_OPEN 6      /* err::Error::WavOpen(String)       src/err.rs:14   */
#define APTGPU_ERR_IO 7            /* err::Error::Io(std::io::Error)    src/err.rs:11   */

/* ---- filters (src/filters.rs:22-46) ------------------------------------ */
#define APTGPU_FILTER_NOFILTER 0          /* filters::NoFilter          */
#define APTGPU_FILTER_LOWPASS 1           /* filters::Lowpass           */
#define APTGPU_FILTER_LOWPASS_DC_REMOVAL 2 /* filters::LowpassDcRemoval */

/* A `impl filters::Filter` value: frequencies are Freq.pi_rad (fractions of
 * pi rad/sample, src/frequency.rs:30-32), atten in positive dB. */
typedef struct aptgpu_filter {
    int32_t kind;
    float cutout_pi_rad;
    float atten;
    float delta_w_pi_rad;
} aptgpu_filter;

/* ---- config::Settings, the fields decode() reads (src/config.rs:76-106;
 *      read at src/decode.rs:55,57,68,70,75,98) ---------------------------- */
typedef struct aptgpu_settings {
    uint32_t work_rate;         /* Hz, intermediate processing rate            */
    float resample_atten;       /* dB                                          */
    float resample_delta_freq;  /* Hz                                          */
    float resample_cutout;      /* Hz                                          */
    float demodulation_atten;   /* dB                                          */
    int32_t export_wav;         /* Settings.export_wav: deliver steps via step */
    int32_t export_resample_filtered; /* Settings.export_resample_filtered (src/config.rs:83 ->
                                   context.rs:113).  As in the reference it moves the decimation
                                   phase of fast_resampling (src/dsp.rs:265-273) -- other rows,
                                   exported or not -- and with export_wav the "resample_filtered"
                                   step carries the expanded signal (n*l floats).  Served by the
                                   unfused kernels: a debugging aid, not a fast path            */
} aptgpu_settings;

/* ---- context::Context (src/context.rs:100-133) -------------------------- */
/* Context::status(progress, description)  src/context.rs:127-129 */
typedef void (*aptgpu_status_fn)(float progress, const char *description, void *user);
/* Context::step(Step{id, variant, data, rate})  src/context.rs:132-211.
 * variant: 0 = Variant::Signal, 1 = Variant::Filter; rate_hz 0 = None.
 * `data` is host memory valid only during the call.  Return nonzero to abort
 * the decode with APTGPU_ERR_INTERNAL (the reference propagates step errors). */
typedef int (*aptgpu_step_fn)(const char *id, int variant, const float *data, size_t n,
                              uint32_t rate_hz, void *user);

typedef struct aptgpu_context {
    aptgpu_status_fn status; /* nullable */
    aptgpu_step_fn step;     /* nullable; only called when settings.export_wav != 0 */
    void *user;
    int32_t device;          /* HIP device ordinal */
    int32_t mode;            /* APTGPU_MODE_* */
    void *stream;            /* hipStream_t to run on, NULL = the plan's own stream */
} aptgpu_context;

#define APTGPU_MODE_STRICT 0 /* bit-exact with the reference's f32 loops; fused kernels when
                                the (L, M, taps) combination has a specialisation          */
#define APTGPU_MODE_GENERIC 1 /* force the unfused generic kernels (any rate combination)   */
#define APTGPU_MODE_FP16_TAPS 2 /* first resample with fp16 taps + fp16 samples through
                                   v_dot2_f32_f16, f32 accumulate (BASELINE.json config 5).  NOT
                                   bit-exact: pixels within ~2e-3 of the row peak, sync positions
                                   unchanged on APT data; every other stage as in STRICT        */
#define APTGPU_MODE_FAST 3 /* f32 throughout with the same taps in the same order, but fused
                              multiply-adds in the two FIR stages, the native square root and a
                              reciprocal multiplication in the envelope, and the +-1 sync correlation
                              evaluated from pulse sums: about a third of the strict arithmetic.  NOT
                              bit-exact, deterministic; tolerance (enforced by the tests):
                              same row count, sync positions identical for >= 99.9 % of the rows and
                              never off by more than one work-rate sample, |d px| <= 1e-4 * max|px| on
                              rows with identical position.  Rates / profiles without a fast kernel
                              are served by the strict kernels.  With a user-tuned resample filter (a
                              tap count other than the stock profiles') at 48 / 96 kHz the resampler
                              runs on the matrix cores from bf16 pieces of the f32 taps (three: exact)
                              and samples (two: exact for 16-bit data), f32 accumulation: the same
                              tolerance (measured 5e-7 of full scale). */

/* What find_sync()/decode() learned; the reference only logs it
 * (`info!("Found {} sync frames")` src/decode.rs:260). */
typedef struct aptgpu_stats {
    uint64_t work_len;      /* samples after the first resample                 */
    uint64_t n_sync;        /* peaks.len() of find_sync (0 when sync == 0)      */
    uint64_t n_rows;        /* image rows returned                              */
    uint32_t l, m;          /* interpolation / decimation factors dsp.rs:73-75  */
    uint32_t n_resample_taps, n_lowpass_taps;
    int32_t fused;          /* front end used: 0 unfused generic kernels, 1 compile-time
                               specialised fused kernel, 2 run-time fused kernel, 3 table-driven
                               stage 1 in front of the specialised work-rate stages (11 025 Hz),
                               4 phase-resident taps in stage 1, same work-rate stages (44 100 Hz) */
    int32_t orbit_path;     /* peak-picker path: 0 doubling (LDS), 1 bitmask walk */
} aptgpu_stats;

/* ====================================================================== */
/* 1. decode()                                                             */
/* ====================================================================== */

/* Replaces  pub fn decode(context: &mut Context, settings: &config::Settings,
 *                         signal: &Signal, input_rate: Rate, sync: bool)
 *                         -> err::Result<Signal>        src/decode.rs:43-49
 * signal: host f32 samples exactly as wav::load_wav produces them (unscaled,
 * first channel; src/wav.rs:30-51).  On success *rows_out is a malloc'd
 * buffer of *n_out = rows*2080 floats (release with aptgpu_free).
 * Errors and their messages are the reference's (src/decode.rs:79-83,112-118,
 * 172-176; src/dsp.rs:69-71,82-91). */
int aptgpu_decode(const aptgpu_context *ctx, const aptgpu_settings *settings,
                  const float *signal, size_t n, uint32_t input_rate_hz, int sync,
                  float **rows_out, size_t *n_out, aptgpu_stats *stats /* nullable */,
                  char *err, size_t err_cap);

/* Releases any buffer this library returned (Vec<f32> drop). */
void aptgpu_free(void *p);

/* The host-array entry points (aptgpu_decode, aptgpu_decode_wav, aptgpu_decode_batch[_wav]) keep what a call
 * needs on the device between calls — the plan (designed taps, HBM workspace, streams), input / output buffers,
 * pinned staging — in a process-wide, mutex-guarded, least-recently-used cache keyed by (device, the five
 * settings decode() reads, input rate, sync, mode, recordings per call): SURVEY.md section 8(b), threading
 * row.  A cached session serves one call at a time (concurrent callers with the same key each get their own);
 * at most 8 idle sessions / APTGPU_SESSION_CACHE_MB (default: a quarter of the device's memory; 0 = no caching)
 * of device memory are kept, and a session that cannot be built for lack of device memory empties the cache and
 * tries once more.  Calls that export steps or run on a caller's stream do not use it.
 * aptgpu_cache_clear() releases every idle session; aptgpu_cache_info() reports what is idle. */
void aptgpu_cache_clear(void);
void aptgpu_cache_info(int32_t *entries /* nullable */, uint64_t *device_bytes /* nullable */);

/* ====================================================================== */
/* 2. plans: device-resident and batched decode                            */
/* ====================================================================== */

/* A plan owns the designed taps, the HBM workspace and a stream for decode()
 * calls of one (settings, input_rate, sync) combination on one GPU.  It is
 * what a long-running caller (GUI worker thread src/gui/work.rs:174-197, or a
 * batch driver over independent recordings) keeps between calls. */
typedef struct aptgpu_plan aptgpu_plan;

typedef struct aptgpu_plan_info {
    uint32_t l, m;
    uint32_t n_resample_taps, n_lowpass_taps, n_sync_taps;
    uint32_t samples_per_work_row; /* PX_PER_ROW * work_rate / FINAL_RATE  decode.rs:55 */
    uint32_t min_distance;         /* samples_per_work_row * 8 / 10        decode.rs:216 */
    uint64_t max_samples;          /* capacity the plan was created for                  */
    uint64_t max_work_len;         /* work-rate samples at max_samples                   */
    uint64_t max_rows;             /* upper bound on rows for max_samples                */
    int32_t fused;                 /* front end: 0 unfused, 1 specialised fused, 2 run-time fused,
                                      3 table-driven / 4 phase-resident stage 1 + specialised
                                      work-rate stages */
    int32_t max_batch;
} aptgpu_plan_info;

/* Result record the device fills per recording (also readable from the host
 * with aptgpu_plan_results). */
typedef struct aptgpu_result {
    int32_t status;    /* APTGPU_OK, or APTGPU_ERR_INTERNAL (<10 rows / <5 sync frames) */
    int32_t reason;    /* 0 ok, 1 "<10 rows", 2 "<5 sync frames", 3 work_rate not a multiple of
                          4160, 4 ok but rows truncated to the caller's rows_cap; -1 while the
                          recording's kernels have not finished                              */
    uint32_t n_rows;   /* image rows written (n_out / 2080 when sync != 0; never more than rows_cap) */
    uint32_t n_sync;   /* peaks.len() of find_sync, 0 when sync == 0                         */
    uint64_t work_len; /* samples after the first resample                                   */
    uint64_t n_out;    /* floats written to d_rows                                           */
} aptgpu_result;

int aptgpu_plan_create(const aptgpu_context *ctx, const aptgpu_settings *settings,
                       uint32_t input_rate_hz, int sync, size_t max_samples, int max_batch,
                       aptgpu_plan **plan_out, char *err, size_t err_cap);
void aptgpu_plan_destroy(aptgpu_plan *plan);
int aptgpu_plan_get_info(const aptgpu_plan *plan, aptgpu_plan_info *info);

/* Enqueue decode() of `count` independent recordings already resident in HBM.
 * d_signals[i] points to n[i] device floats; d_rows[i] receives up to
 * rows_cap[i]*2080 device floats.  Asynchronous, no host synchronisation.
 * The recordings of one call go through ONE launch per stage (front end, sync words, sync slots, orbit,
 * row gather), in order on one of the plan's `depth` internal streams; consecutive calls go round-robin
 * over those streams (depth = 3 for plans created with max_batch >= 4, else 6; APTGPU_STREAMS overrides),
 * so the front end of call j+1 overlaps the latency-bound peak picker and the row gather of call j.
 * Stream k owns the workspace slots [k*max_batch, (k+1)*max_batch) — depth*max_batch slots in all — and a
 * slot is only reused by a later call on its own stream.  Plans with max_batch >= 4 (and no ctx.stream)
 * additionally order the front end of call j+1 behind the front end of call j with an event, so that each
 * front end has the whole GPU.  If
 * ctx.stream was given at plan creation the work is ordered AFTER what is
 * already enqueued on ctx.stream (inputs may be produced there); to order
 * ctx.stream after the decode, call aptgpu_plan_join().  Recordings of different
 * calls run on different internal streams: a call that reuses the output buffers of
 * an earlier call is only ordered after it if aptgpu_plan_join() / _synchronize() /
 * _results() came in between.  Outcome per recording lands in the plan's result
 * records. */
int aptgpu_plan_decode_device(aptgpu_plan *plan, int count, const float *const *d_signals,
                              const size_t *n, float *const *d_rows, const size_t *rows_cap,
                              char *err, size_t err_cap);
/* Waits for the stream and copies the `count` result records to the host. */
int aptgpu_plan_results(aptgpu_plan *plan, int count, aptgpu_result *results);
/* Sync-frame positions found by the last decode of recording i (find_sync()'s
 * return value, src/decode.rs:262); writes min(cap, n_sync) entries. */
int aptgpu_plan_sync_positions(aptgpu_plan *plan, int i, uint64_t *pos, size_t cap,
                               size_t *n_sync);
int aptgpu_plan_synchronize(aptgpu_plan *plan);
/* Makes ctx.stream wait, on the device, for everything the plan has enqueued
 * so far (host does not block).  Without a ctx.stream: same as synchronize. */
int aptgpu_plan_join(aptgpu_plan *plan);

/* Kernel timing with HIP events recorded on the plan's stream.  on = 0: off;
 * 1: the dominant (first-stage) kernel of every 8th decode is bracketed (sampling keeps the
 * markers, which serialise the stream, cheap); 2: every
 * kernel launch is.  Enable, run decodes, then collect: averages are over all
 * bracketed launches since the last collect. */
typedef struct aptgpu_kernel_time {
    char name[48];
    double avg_ms;
    uint64_t launches;
} aptgpu_kernel_time;
int aptgpu_plan_enable_timing(aptgpu_plan *plan, int on);
int aptgpu_plan_collect_timing(aptgpu_plan *plan, aptgpu_kernel_time *out, size_t cap,
                               size_t *n_out);

/* Introspection: copies one of the plan's internal HBM buffers of recording slot i to the
 * host after synchronising — the device-side counterpart of the intermediate signals the
 * reference exports through Context::step (src/context.rs:132-211).  Names: "filtered"
 * (f32, "filter_result"), "correlation" (f32, "sync_correlation"), "group_max" (f32, maxima
 * of the correlation over groups of 52 positions), "terminal_words" (u64), "peaks" (u32,
 * find_sync positions), "picker_flags" (u32[32]: [0] list overflow, [1] 1 = sequential
 * fallback ran, [8..] cycle stamps of the picker kernels).  Writes min(bytes, size) bytes
 * and returns the buffer's size in *size_out. */
int aptgpu_plan_read_internal(aptgpu_plan *plan, int i, const char *name, void *host_out,
                              size_t bytes, size_t *size_out);

/* ====================================================================== */
/* 2b. host-fed batch decode over one or more GPUs                          */
/* ====================================================================== */
/* The batch variant of decode(): what a driver that loops `load(); decode();` over many recordings
 * (the reference's CLI does it for one, src/main.rs:102-104) binds instead of the loop.  Recordings are
 * independent, so they are sharded over `devices` (ordinals, repeats allowed: {0, 0} = two workers on
 * GPU 0; n_devices == 0 = ctx->device) longest-first onto the least loaded entry, with NO collective of
 * any kind; every entry gets a host thread and a cached session (plan, device buffers, copy streams) and keeps
 * the uploads of calls k+1 / k+2, the kernels of call k and the download of call k-1 in flight together
 * (recordings_per_call recordings per call, <= 0 = 16); rows are DMA'd straight into the returned buffers.
 * Every worker thread pins itself to the CPUs of its GPU's NUMA node first (sysfs: the device's numa_node and
 * that node's cpulist; APTGPU_NUMA_PIN=0 turns it off), so the buffers it allocates and the staging copies it
 * drives stay on the socket the GPU hangs off.  (Throughput figures: DESIGN.md section 7, profiles/.)  All
 * recordings of a batch share (settings, input_rate_hz, sync); ctx supplies mode (and the device when
 * n_devices == 0), callbacks are not used.
 *   rows_out[i] / n_out[i]: malloc'd rows of recording i (aptgpu_free), NULL / 0 when status[i] != 0;
 *   status[i]: APTGPU_OK, or the error decode() would have returned for that recording
 *              (APTGPU_ERR_INTERNAL: too short / too few sync frames; for WAV images also the
 *              reference's WAV errors, and APTGPU_ERR_INVALID for a rate other than input_rate_hz);
 *   results[i] (nullable): the device-side record (rows, sync frames found, work length).
 * Returns APTGPU_OK when every worker ran (per-recording failures are in status[]), else the first
 * worker-level error (HIP failure, bad argument).  Host buffers may be pageable; pinned ones
 * (aptgpu_host_alloc) are DMA'd directly. */
/* ABI note (0.2.0): `struct_size` is the caller's sizeof(aptgpu_batch_stats), set BEFORE the call; the library fills
 * at most that many bytes, so a caller built against this header keeps working when fields are appended.  A
 * struct_size below 8 (a zero-initialised struct; the bytes a 0.1.0 caller's `double seconds` starts with) is refused
 * with APTGPU_ERR_INVALID before anything runs.  The struct had no such field in 0.1.0 and grew twice: a binding
 * checks aptgpu_abi_version() at load time. */
typedef struct aptgpu_batch_stats {
    uint32_t struct_size;  /* in: sizeof(aptgpu_batch_stats) as the caller was compiled              */
    uint32_t reserved;
    double seconds;        /* wall time of the whole call                                   */
    uint64_t samples;      /* input samples of the recordings that were handed to a worker  */
    uint64_t h2d_bytes, d2h_bytes;
    double h2d_seconds;    /* host time inside the H2D copy calls, summed over the workers  */
    double d2h_seconds;
    int32_t workers;
    int32_t recordings_per_call;
    double gate_wait_seconds; /* summed over the workers: time spent waiting for the per-device upload gate    */
    double setup_seconds;     /* summed: leasing (or building) the session and sizing its buffers, before the
                                 first upload — a session built inside the call shows here                      */
    int32_t sessions_created; /* sessions that were not in the cache (0 in a warmed-up process)                 */
    int32_t workers_pinned;   /* workers that found their GPU's NUMA node and pinned themselves to its CPUs     */
} aptgpu_batch_stats;

int aptgpu_decode_batch(const aptgpu_context *ctx, const aptgpu_settings *settings,
                        uint32_t input_rate_hz, int sync, int count, const float *const *signals,
                        const size_t *n, const int32_t *devices, int n_devices, int recordings_per_call,
                        float **rows_out, size_t *n_out, int32_t *status,
                        aptgpu_result *results /* nullable */, aptgpu_batch_stats *stats /* nullable */,
                        char *err, size_t err_cap);
/* The same on WAV file images (noaa_apt::load + decode per file): the data chunk is uploaded as it
 * is — 2 bytes per sample for PCM16 — and converted on the device (wav.rs:30-51). */
int aptgpu_decode_batch_wav(const aptgpu_context *ctx, const aptgpu_settings *settings,
                            uint32_t input_rate_hz, int sync, int count, const void *const *wav_images,
                            const size_t *wav_bytes, const int32_t *devices, int n_devices,
                            int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                            aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap);
/* Pinned host memory (hipHostMalloc) for inputs that should cross PCIe by direct DMA; NULL on failure. */
void *aptgpu_host_alloc(size_t bytes);
void aptgpu_host_free(void *p);
/* Which host CPUs a worker of `device` pins itself to: the device's PCI address (hipDeviceGetPCIBusId), its NUMA
 * node and that node's CPU list as sysfs prints it ("0-63,128-191").  numa_node = -1 and an empty list when the
 * platform does not say (single-socket hosts, containers without sysfs): no pinning then.  The second form is the
 * same lookup on a given PCI address under a given sysfs root ("/sys" on a real host) and needs no GPU. */
int aptgpu_host_affinity(int device, char *pci_bdf /* >= 16 bytes, nullable */, int32_t *numa_node,
                         char *cpulist, size_t cpulist_cap);
int aptgpu_host_affinity_from_sysfs(const char *sysfs_root, const char *pci_bdf, int32_t *numa_node,
                                    char *cpulist, size_t cpulist_cap);

/* ====================================================================== */
/* 3. the dsp.rs / filters.rs / decode.rs building blocks (host buffers)    */
/* ====================================================================== */
/* These mirror the reference functions one-to-one so stage-level parity    */
/* tests read like the reference's own unit tests.  Outputs are malloc'd.   */

/* Filter::design()                     src/filters.rs:48-54,57-88,98-132 (host math) */
int aptgpu_filter_design(const aptgpu_filter *f, float **coeff_out, size_t *n_out);
/* Filter::resample(input_rate, output_rate)   src/filters.rs:90-94,134-138 */
void aptgpu_filter_resample(aptgpu_filter *f, uint32_t input_rate_hz, uint32_t output_rate_hz);
/* generate_sync_frame(work_rate)       src/decode.rs:171-199 */
int aptgpu_generate_sync_frame(uint32_t work_rate_hz, int8_t **frame_out, size_t *n_out,
                               char *err, size_t err_cap);
/* dsp::resample_with_filter(context, signal, input_rate, output_rate, filt)  src/dsp.rs:62-126 */
int aptgpu_resample_with_filter(const aptgpu_context *ctx, const float *signal, size_t n,
                                uint32_t input_rate_hz, uint32_t output_rate_hz,
                                aptgpu_filter filt, float **out, size_t *n_out, char *err,
                                size_t err_cap);
/* dsp::resample(context, signal, input_rate, output_rate, atten, delta_w)   src/dsp.rs:132-162
 * (the WAV->WAV tool path, src/resample.rs:36) */
int aptgpu_resample(const aptgpu_context *ctx, const float *signal, size_t n,
                    uint32_t input_rate_hz, uint32_t output_rate_hz, float atten,
                    float delta_w_pi_rad, float **out, size_t *n_out, char *err, size_t err_cap);
/* dsp::demodulate(context, signal, carrier_freq)   src/dsp.rs:350-383 */
int aptgpu_demodulate(const aptgpu_context *ctx, const float *signal, size_t n,
                      float carrier_pi_rad, float **out, char *err, size_t err_cap);
/* dsp::filter(context, signal, filter)             src/dsp.rs:386-410 */
int aptgpu_filter_signal(const aptgpu_context *ctx, const float *signal, size_t n,
                         aptgpu_filter filt, float **out, char *err, size_t err_cap);
/* find_sync(context, signal, work_rate)            src/decode.rs:204-263
 * correlation_out nullable (the "sync_correlation" step). */
int aptgpu_find_sync(const aptgpu_context *ctx, const float *signal, size_t n,
                     uint32_t work_rate_hz, uint64_t **pos_out, size_t *n_pos,
                     float **correlation_out, size_t *n_corr, char *err, size_t err_cap);

/* ====================================================================== */
/* 4. consumers of the pixel rows                 */
/* ====================================================================== */
/* The grayscale part of noaa_apt::process() (src/noaa_apt.rs:132-192): contrast limits   */
/* -> map_signal_u8, plus telemetry.rs and the 180-degree channel rotation.  False colour, */
/* histogram equalisation and the map overlay stay on the host (out of scope).            */

#define APTGPU_CONTRAST_TELEMETRY 0 /* Contrast::Telemetry   src/noaa_apt.rs:141-150 */
#define APTGPU_CONTRAST_PERCENT 1   /* Contrast::Percent(p)  src/noaa_apt.rs:151-157 */
#define APTGPU_CONTRAST_MINMAX 2    /* Contrast::MinMax      src/noaa_apt.rs:158-164 (Histogram
                                       takes the same limits before its equalisation) */
#define APTGPU_ROTATE_NO 0          /* Rotate::No  */
#define APTGPU_ROTATE_YES 1         /* Rotate::Yes  src/noaa_apt.rs:228-231, processing.rs:21-37 */

/* What the image stage found; also the telemetry::Telemetry values (src/telemetry.rs:19-23). */
typedef struct aptgpu_image_result {
    int32_t status;          /* APTGPU_OK or APTGPU_ERR_INTERNAL */
    int32_t reason;          /* 1 zero-length signal (dsp.rs:40-44), 2 too short for telemetry
                                (telemetry.rs:199-203), 3 no low bucket (misc.rs:172 panics),
                                4 the decode before it failed */
    uint32_t height;         /* rows of 2080 px */
    uint32_t telemetry_row;  /* best frame start, telemetry.rs:196,228-230 */
    float low, high;         /* the contrast limits used by map_signal_u8 */
    float telemetry_quality;
    int32_t channel_a, channel_b; /* index for aptgpu_channel_name(), -1 = not computed */
    uint32_t reserved;
    uint64_t n_px;           /* u8 pixels written */
    float values_a[16], values_b[16]; /* wedges 1-16 of each band */
} aptgpu_image_result;

/* dsp::get_min / dsp::get_max          src/dsp.rs:20-54 */
int aptgpu_get_min(const aptgpu_context *ctx, const float *signal, size_t n, float *out, char *err,
                   size_t err_cap);
int aptgpu_get_max(const aptgpu_context *ctx, const float *signal, size_t n, float *out, char *err,
                   size_t err_cap);
/* misc::percent(signal, percent)       src/misc.rs:119-175 */
int aptgpu_percent(const aptgpu_context *ctx, const float *signal, size_t n, float percent,
                   float *low, float *high, char *err, size_t err_cap);
/* map_signal_u8(signal, low, high)     src/noaa_apt.rs:249-259; *out malloc'd, n bytes */
int aptgpu_map_signal_u8(const aptgpu_context *ctx, const float *signal, size_t n, float low,
                         float high, uint8_t **out, char *err, size_t err_cap);
/* telemetry::read_telemetry(context, signal)   src/telemetry.rs:125-243; fills values_*,
 * telemetry_row/quality, channel_*.  With ctx->step set, the five "telemetry_*" steps are
 * exported in the reference's order (telemetry.rs:234-238). */
int aptgpu_read_telemetry(const aptgpu_context *ctx, const float *signal, size_t n,
                          aptgpu_image_result *telemetry, char *err, size_t err_cap);
/* Telemetry::get_channel_name table    src/telemetry.rs:104-106 */
const char *aptgpu_channel_name(int index);
/* process() up to the GrayImage (+ rotate): contrast limits, status callbacks at 0.1 / 0.3 /
 * 0.90 with the reference's texts, u8 image rows*2080 (malloc'd).  src/noaa_apt.rs:132-235 */
int aptgpu_process_gray(const aptgpu_context *ctx, const float *signal, size_t n, int contrast,
                        float percent, int rotate, uint8_t **image_out, size_t *n_out,
                        aptgpu_image_result *info, char *err, size_t err_cap);
/* Device-resident: the same stage chained behind the most recent aptgpu_plan_decode_device
 * call, recording i of that call, on the recording's own stream (no host round trip; the
 * pixel count comes from the decode result on the device).  d_rows[i] must be the rows
 * buffer given to the decode call (same rows_cap[i]); d_images[i] has room for
 * rows_cap[i]*2080 bytes. */
int aptgpu_plan_process_device(aptgpu_plan *plan, int count, const float *const *d_rows,
                               const size_t *rows_cap, int contrast, float percent, int rotate,
                               uint8_t *const *d_images, char *err, size_t err_cap);
/* Waits for the image stage of the last call and copies the records. */
int aptgpu_plan_image_results(aptgpu_plan *plan, int count, aptgpu_image_result *results);

/* ====================================================================== */
/* 5. WAV ingest in front of decode()                 */
/* ====================================================================== */
/* noaa_apt::load / wav::load_wav (src/noaa_apt.rs:114-130, src/wav.rs:11-57): the container is */
/* walked on the host exactly as hound 3.5.1's WavReader does (Cargo.toml:29), the samples are   */
/* converted on the GPU: first channel only, integers `as f32`, never scaled (wav.rs:30-51).     */

#define APTGPU_WAV_U8 0    /* 8 bit unsigned (minus 128)              */
#define APTGPU_WAV_I16 1   /* 16 bit                                  */
#define APTGPU_WAV_I24 2   /* 24 bit packed                           */
#define APTGPU_WAV_I24_4 3 /* 24 bit in a 4-byte container            */
#define APTGPU_WAV_I32 4   /* 32 bit                                  */
#define APTGPU_WAV_F32 5   /* IEEE float                              */

/* hound::WavSpec (+ where the samples are) */
typedef struct aptgpu_wav_spec {
    uint16_t channels;
    uint16_t bits_per_sample;
    uint16_t bytes_per_sample; /* block_align / channels */
    uint16_t sample_format;    /* 0 = hound::SampleFormat::Int, 1 = Float */
    uint32_t sample_rate;
    int32_t codec;             /* APTGPU_WAV_* */
    uint64_t data_offset;      /* payload of the data chunk inside the file image */
    uint64_t data_len;         /* bytes */
    uint64_t n_samples;        /* all channels (WavReader::len) */
    uint64_t n_frames;         /* == length of the Signal load_wav returns */
} aptgpu_wav_spec;

/* hound::WavReader::new + spec() on an in-memory image of the file (host only, no GPU).
 * Errors as the reference maps them (src/err.rs:72-83): APTGPU_ERR_WAV_OPEN for malformed or
 * unsupported files, APTGPU_ERR_IO for a short file, APTGPU_ERR_INTERNAL for too-wide samples. */
int aptgpu_wav_parse(const void *file_bytes, size_t n, aptgpu_wav_spec *spec, char *err,
                     size_t err_cap);
/* wav::load_wav: file image -> (Signal, rate).  *signal_out malloc'd; spec nullable. */
int aptgpu_load_wav(const aptgpu_context *ctx, const void *file_bytes, size_t n, float **signal_out,
                    size_t *n_out, uint32_t *sample_rate_hz, aptgpu_wav_spec *spec, char *err,
                    size_t err_cap);
/* noaa_apt::load(input_filename): reads the file, then as above. */
int aptgpu_load_wav_file(const aptgpu_context *ctx, const char *path, float **signal_out,
                         size_t *n_out, uint32_t *sample_rate_hz, aptgpu_wav_spec *spec, char *err,
                         size_t err_cap);
/* load + decode in one call: the data chunk is uploaded as it is (2 bytes per sample for PCM16
 * instead of 4) and converted in HBM — inside the fused front end for mono PCM16.  Output and
 * callbacks as aptgpu_decode; *sample_rate_hz (nullable) receives the WAV's rate. */
int aptgpu_decode_wav(const aptgpu_context *ctx, const aptgpu_settings *settings,
                      const void *file_bytes, size_t n, int sync, float **rows_out, size_t *n_out,
                      aptgpu_stats *stats, uint32_t *sample_rate_hz, char *err, size_t err_cap);
/* Device-resident batch: d_data[i] points at recording i's data-chunk payload in HBM, specs[i]
 * describes it (sample_rate must equal the plan's input rate).  Otherwise as
 * aptgpu_plan_decode_device. */
int aptgpu_plan_decode_device_wav(aptgpu_plan *plan, int count, const void *const *d_data,
                                  const aptgpu_wav_spec *specs, float *const *d_rows,
                                  const size_t *rows_cap, char *err, size_t err_cap);

/* wav::write_wav (src/wav.rs:59-98) for the {1 channel, 16 bit, Int} spec the resample tool
 * uses (src/resample.rs:53-58): normalise by dsp::get_max, `as i16`; returns the file image
 * hound's writer produces (44-byte PCM header + samples), malloc'd. */
int aptgpu_write_wav_i16(const aptgpu_context *ctx, const float *signal, size_t n,
                         uint32_t sample_rate_hz, void **wav_out, size_t *n_out, char *err,
                         size_t err_cap);
/* resample::resample (src/resample.rs:17-71) between file images: load_wav ->
 * dsp::resample(atten, delta_w) -> write_wav, all on the device; status callbacks at 0.0 / 0.2 /
 * 0.8 / 1.0 with the reference's texts (output_name, nullable, only appears in the 0.8 text),
 * "input" step when ctx->step is set.  atten / delta_w_pi_rad are settings.wav_resample_atten /
 * wav_resample_delta_freq (src/config.rs:100-106). */
int aptgpu_resample_wav(const aptgpu_context *ctx, const void *file_bytes, size_t n,
                        uint32_t output_rate_hz, float atten, float delta_w_pi_rad,
                        const char *output_name, void **wav_out, size_t *n_out, char *err,
                        size_t err_cap);
/* The same with file IO on both sides, copying the modification time (misc.rs:181-205). */
int aptgpu_resample_wav_file(const aptgpu_context *ctx, const char *input_path,
                             const char *output_path, uint32_t output_rate_hz, float atten,
                             float delta_w_pi_rad, char *err, size_t err_cap);
/* Both with Context::resample's second flag (src/context.rs:214-218; main.rs:125-130 passes
 * settings.export_resample_filtered): nonzero = fast_resampling's other decimation phase
 * (src/dsp.rs:265-273) and, when ctx->step is set, the expanded signal in "resample_filtered".
 * With ctx->step set every entry point of this section delivers the steps of Context::resample in the
 * reference's order: "input", "resample_filter", "resample_filtered", "resample_decimated". */
int aptgpu_resample_wav_ex(const aptgpu_context *ctx, const void *file_bytes, size_t n,
                           uint32_t output_rate_hz, float atten, float delta_w_pi_rad,
                           int export_resample_filtered, const char *output_name, void **wav_out,
                           size_t *n_out, char *err, size_t err_cap);
int aptgpu_resample_wav_file_ex(const aptgpu_context *ctx, const char *input_path,
                                const char *output_path, uint32_t output_rate_hz, float atten,
                                float delta_w_pi_rad, int export_resample_filtered, char *err,
                                size_t err_cap);

/* ====================================================================== */
/* 6. misc                                                                 */
/* ====================================================================== */
const char *aptgpu_version(void);
/* Incremented whenever a struct of this header changes layout or a function its signature (0.2.0: 2). */
int aptgpu_abi_version(void);
#define APTGPU_ABI_VERSION 2
int aptgpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* APTGPU_H */
