/*
 * apt_oracle.h — CPU parity oracle for the noaa-apt decode() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline.  The
 * product (noaa_apt_amd/csrc, libaptgpu.so) never links or calls it.
 *
 * What it is: a plain-C, scalar, single-threaded f32 restatement of the
 * reference's Rust loops (martinber/noaa-apt v1.4.1), same loop order, same
 * f32/u32/u64 types, compiled with -ffp-contract=off -fno-fast-math so every
 * product and sum is rounded exactly as rustc's (rustc never contracts).
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/).
 *
 * PINNING STATUS:
 *   pinned by the reference's own tests, re-run in tests/test_oracle_reference_kats.py:
 *     - generate_sync_frame: exact vectors            src/decode.rs:271-319
 *     - bessel_i0: 15 values, rel 1e-3                src/misc.rs:493-513
 *     - Lowpass / LowpassDcRemoval frequency-response bounds
 *                                                     src/filters.rs:243-366
 *     - NoFilter == [1.], Filter::resample rescaling  src/filters.rs:368-423
 *     - Freq unit conversions, 10 ULP                 src/frequency.rs:325-416
 *     - RateOverflow for 99371->93911; is_ok() cases  src/dsp.rs:420-468
 *   PARITY UNPINNED for the numeric output of decode() itself: the reference
 *   holds no golden rows/positions/taps for it (test/test.sh writes images
 *   for manual viewing), and no Rust toolchain exists in this environment to
 *   run the reference.  decode() goldens under tests/golden/ are produced by
 *   THIS restatement and cross-checked by an independent numpy-f32
 *   re-derivation (tests/test_oracle_numpy_crosscheck.py).
 */
#ifndef APT_ORACLE_H
#define APT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (mirror err::Error variants that the path can return,
 * src/err.rs:9-44) */
#define APT_ORACLE_OK 0
#define APT_ORACLE_ERR_INTERNAL 1      /* err::Error::Internal(String) */
#define APT_ORACLE_ERR_RATE_OVERFLOW 2 /* err::Error::RateOverflow(String) */
#define APT_ORACLE_ERR_WAV_OPEN 6      /* err::Error::WavOpen(String)      */
#define APT_ORACLE_ERR_IO 7            /* err::Error::Io(std::io::Error)   */

/* filter kinds (src/filters.rs:22-46) */
#define APT_FILTER_NOFILTER 0
#define APT_FILTER_LOWPASS 1
#define APT_FILTER_LOWPASS_DC_REMOVAL 2

/* The five config::Settings fields decode() reads (src/config.rs:76-106,
 * read at src/decode.rs:55,57,68,70,75,98). */
typedef struct {
    uint32_t work_rate;
    float resample_atten;
    float resample_delta_freq;
    float resample_cutout;
    float demodulation_atten;
} apt_oracle_settings;

/* A filter description: frequencies as fractions of pi rad/sample
 * (Freq.pi_rad, src/frequency.rs:30-32). */
typedef struct {
    int kind;
    float cutout_pi_rad;
    float atten;
    float delta_w_pi_rad;
} apt_oracle_filter_spec;

/* Optional per-stage outputs of decode(); every pointer is malloc'd by the
 * oracle and released with apt_oracle_free_steps().  These are the signals
 * the reference would export through Context::step (src/context.rs:132-211). */
typedef struct {
    float *resample_filter;   size_t n_resample_filter;   /* "resample_filter"     */
    float *resampled;         size_t n_resampled;         /* "resample_decimated"  */
    float *demodulated;       size_t n_demodulated;       /* "demodulation_result" */
    float *filter_filter;     size_t n_filter_filter;     /* "filter_filter"       */
    float *filtered;          size_t n_filtered;          /* "filter_result"       */
    float *correlation;       size_t n_correlation;       /* "sync_correlation"    */
    uint64_t *sync_pos;       size_t n_sync_pos;          /* find_sync() result    */
    float *aligned;           size_t n_aligned;           /* "sync_result"         */
    double t_resample, t_demod, t_filter, t_sync, t_gather; /* seconds */
} apt_oracle_steps;

/* --- frequency.rs ------------------------------------------------------- */
float apt_oracle_freq_hz(float f, uint32_t rate);          /* Freq::hz -> pi_rad   */
float apt_oracle_freq_rad(float f);                        /* Freq::rad -> pi_rad  */
float apt_oracle_freq_get_rad(float pi_rad);               /* Freq::get_rad        */
float apt_oracle_freq_get_hz(float pi_rad, uint32_t rate); /* Freq::get_hz         */

/* --- misc.rs ------------------------------------------------------------ */
float apt_oracle_bessel_i0(float x);

/* --- filters.rs --------------------------------------------------------- */
/* returns malloc'd coefficients, *n_out = length */
float *apt_oracle_kaiser(float atten, float delta_w_pi_rad, size_t *n_out);
float *apt_oracle_filter_design(const apt_oracle_filter_spec *f, size_t *n_out);
void apt_oracle_filter_resample(apt_oracle_filter_spec *f, uint32_t in_rate, uint32_t out_rate);

/* --- dsp.rs ------------------------------------------------------------- */
float *apt_oracle_fast_resampling(const float *x, size_t n, uint32_t l, uint32_t m,
                                  const float *coeff, size_t ncoeff, size_t *n_out);
/* the same with context.export_resample_filtered set (dsp.rs:265-273): other decimation phase,
 * expanded_out (nullable) = the "resample_filtered" step */
float *apt_oracle_fast_resampling_export(const float *x, size_t n, uint32_t l, uint32_t m,
                                         const float *coeff, size_t ncoeff, size_t *n_out,
                                         float **expanded_out, size_t *n_expanded);
float *apt_oracle_decimate(const float *x, size_t n, uint32_t m, size_t *n_out);
float *apt_oracle_demodulate(const float *x, size_t n, float carrier_pi_rad);
float *apt_oracle_fir(const float *x, size_t n, const float *coeff, size_t ncoeff);
int apt_oracle_resample_with_filter(const float *x, size_t n, uint32_t in_rate,
                                    uint32_t out_rate, apt_oracle_filter_spec filt,
                                    float **out, size_t *n_out,
                                    float **coeff_out, size_t *ncoeff_out,
                                    char *err, size_t err_cap);
int apt_oracle_resample_with_filter_ex(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate,
                                       apt_oracle_filter_spec filt, int export_resample_filtered,
                                       float **out, size_t *n_out, float **coeff_out, size_t *ncoeff_out,
                                       float **expanded_out, size_t *n_expanded, char *err, size_t err_cap);
/* dsp::resample (the WAV->WAV tool path, src/dsp.rs:132-162) */
int apt_oracle_resample(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate,
                        float atten, float delta_w_pi_rad, float **out, size_t *n_out,
                        char *err, size_t err_cap);

int apt_oracle_resample_ex(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate, float atten,
                           float delta_w_pi_rad, int export_resample_filtered, float **out, size_t *n_out,
                           float **coeff_out, size_t *ncoeff_out, float **expanded_out, size_t *n_expanded,
                           char *err, size_t err_cap);

/* --- decode.rs ---------------------------------------------------------- */
int apt_oracle_generate_sync_frame(uint32_t work_rate, int8_t **out, size_t *n_out,
                                   char *err, size_t err_cap);
/* correlation_out may be NULL; if not, receives malloc'd corr[0..n-guard) */
int apt_oracle_find_sync(const float *x, size_t n, uint32_t work_rate,
                         uint64_t **pos_out, size_t *npos_out,
                         float **correlation_out, size_t *ncorr_out,
                         char *err, size_t err_cap);
int apt_oracle_decode(const apt_oracle_settings *s, const float *x, size_t n,
                      uint32_t input_rate, int sync, float **out, size_t *n_out,
                      apt_oracle_steps *steps /* nullable */, char *err, size_t err_cap);

int apt_oracle_decode_ex(const apt_oracle_settings *s, const float *x, size_t n, uint32_t input_rate,
                         int sync, int export_resample_filtered, float **out, size_t *n_out,
                         apt_oracle_steps *steps /* nullable */, float **expanded1, size_t *n_expanded1,
                         float **expanded2, size_t *n_expanded2, char *err, size_t err_cap);

/* --- consumers of decode()'s rows (apt_oracle_image.c) ---------- */
/* dsp::get_max / get_min, src/dsp.rs:20-54 */
int apt_oracle_get_max(const float *x, size_t n, float *out, char *err, size_t err_cap);
int apt_oracle_get_min(const float *x, size_t n, float *out, char *err, size_t err_cap);
/* misc::percent, src/misc.rs:119-175; buckets_out nullable (1000 counts) */
int apt_oracle_percent(const float *x, size_t n, float percent, float *low, float *high,
                       uint32_t *buckets_out, char *err, size_t err_cap);
/* map_signal_u8, src/noaa_apt.rs:249-259 */
void apt_oracle_map_signal_u8(const float *x, size_t n, float low, float high, uint8_t *out);
/* telemetry.rs:30-121; channel: 0 = A, 1 = B, -1 = None (average) */
void apt_oracle_telemetry_from_bands(const float *means_a, const float *means_b, size_t n,
                                     size_t row, float values_a[16], float values_b[16]);
float apt_oracle_telemetry_wedge_value(const float values_a[16], const float values_b[16],
                                       uint32_t wedge, int channel);
int apt_oracle_telemetry_channel_index(const float values_a[16], const float values_b[16], int channel);
/* read_telemetry, src/telemetry.rs:125-243; optional malloc'd step outputs */
int apt_oracle_read_telemetry(const float *signal, size_t n, float values_a[16], float values_b[16],
                              uint64_t *best_row, float *best_quality, float **mean_a_out,
                              float **mean_b_out, float **variance_out, float **corr_out,
                              float **quality_out, size_t *rows_out, char *err, size_t err_cap);
/* grayscale part of process(), src/noaa_apt.rs:132-192; contrast 0 Telemetry, 1 Percent, 2 MinMax */
int apt_oracle_process_gray(const float *signal, size_t n, int contrast, float percent,
                            uint8_t **image_out, size_t *n_out, float *low_out, float *high_out,
                            char *err, size_t err_cap);

/* --- WAV ingest (apt_oracle_wav.c) ---------------------------------- */
typedef struct {
    uint16_t channels, bits_per_sample, bytes_per_sample, sample_format; /* 0 Int, 1 Float */
    uint32_t sample_rate;
    uint64_t data_offset, data_len, n_samples;
} apt_oracle_wav_spec;
/* wav::load_wav on an in-memory file image, src/wav.rs:11-57 (+ hound 3.5.1's WavReader) */
int apt_oracle_load_wav(const uint8_t *bytes, size_t n, float **signal_out, size_t *n_out,
                        apt_oracle_wav_spec *spec, char *err, size_t err_cap);

/* wav::write_wav for the 16-bit Int spec, src/wav.rs:59-98 */
int apt_oracle_write_wav_i16(const float *signal, size_t n, uint32_t rate, uint8_t **out, size_t *n_out,
                             char *err, size_t err_cap);

void apt_oracle_free(void *p);
void apt_oracle_free_steps(apt_oracle_steps *s);

#ifdef __cplusplus
}
#endif
#endif
