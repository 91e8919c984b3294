/*
 * apt_oracle.c — CPU parity oracle (TEST INFRASTRUCTURE ONLY; see apt_oracle.h).
 *
 * Scalar f32 restatement of martinber/noaa-apt v1.4.1's decode() path.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Every function cites the reference lines it follows.  PARITY UNPINNED for
 * decode()'s numeric output (no reference golden exists, no Rust toolchain
 * here) — see the header of apt_oracle.h for what IS pinned.
 */
#include "apt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* std::f32::consts::PI */
static const float PI_F = 3.14159265358979323846f;

/* src/decode.rs:14-38 */
#define FINAL_RATE 4160u
#define PX_PER_ROW 2080u
#define CARRIER_FREQ 2400u

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void set_err(char *err, size_t cap, const char *msg)
{
    if (err && cap) {
        snprintf(err, cap, "%s", msg);
    }
}

void apt_oracle_free(void *p) { free(p); }

void apt_oracle_free_steps(apt_oracle_steps *s)
{
    if (!s) return;
    free(s->resample_filter);
    free(s->resampled);
    free(s->demodulated);
    free(s->filter_filter);
    free(s->filtered);
    free(s->correlation);
    free(s->sync_pos);
    free(s->aligned);
    memset(s, 0, sizeof(*s));
}

/* ---------------------------------------------------------------------- */
/* frequency.rs                                                            */
/* ---------------------------------------------------------------------- */

/* Freq::hz, src/frequency.rs:68-72: pi_rad = 2. * f / rate as f32 */
float apt_oracle_freq_hz(float f, uint32_t rate) { return 2.f * f / (float)rate; }

/* Freq::rad, src/frequency.rs:58-60 */
float apt_oracle_freq_rad(float f) { return f / PI_F; }

/* Freq::get_rad, src/frequency.rs:75-77 */
float apt_oracle_freq_get_rad(float pi_rad) { return pi_rad * PI_F; }

/* Freq::get_hz, src/frequency.rs:85-87 */
float apt_oracle_freq_get_hz(float pi_rad, uint32_t rate)
{
    return pi_rad * (float)rate / 2.f;
}

/* ---------------------------------------------------------------------- */
/* misc.rs                                                                 */
/* ---------------------------------------------------------------------- */

/* BESSEL_TABLE, src/misc.rs:20-41: 1 / (n! * 2^n)^2 as f32 literals */
static const float BESSEL_TABLE[20] = {
    1.0f,
    0.25f,
    0.015625f,
    0.00043402777777777775f,
    6.781684027777777e-06f,
    6.781684027777778e-08f,
    4.709502797067901e-10f,
    2.4028075495244395e-12f,
    9.385966990329842e-15f,
    2.896903392077112e-17f,
    7.242258480192779e-20f,
    1.4963343967340453e-22f,
    2.5978027721077174e-25f,
    3.842903509035085e-28f,
    4.9016626390753635e-31f,
    5.4462918211948485e-34f,
    5.318644356635594e-37f,
    4.60090342269515e-40f,
    3.5500798014623073e-43f,
    2.458504017633177e-46f, /* rounds to 0 in f32, as in Rust */
};

/* bessel_i0, src/misc.rs:47-57 */
float apt_oracle_bessel_i0(float x)
{
    float result = 0.f;
    for (int k = 8; k >= 1; k--) {
        result += BESSEL_TABLE[k];
        result *= x * x; /* x.powi(2) */
    }
    return result + 1.f;
}

/* ---------------------------------------------------------------------- */
/* filters.rs                                                              */
/* ---------------------------------------------------------------------- */

/* Rust `f32 as i32`: saturating, NaN -> 0 */
static int32_t f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.f) return INT32_MAX;
    if (v <= -2147483648.f) return INT32_MIN;
    return (int32_t)v;
}

/* kaiser, src/filters.rs:144-183 */
float *apt_oracle_kaiser(float atten, float delta_w_pi_rad, size_t *n_out)
{
    float beta;
    if (atten > 50.f) {
        beta = 0.1102f * (atten - 8.7f);
    } else if (atten < 21.f) {
        beta = 0.f;
    } else {
        beta = 0.5842f * powf(atten - 21.f, 0.4f) + 0.07886f * (atten - 21.f);
    }

    /* :164-167 */
    int32_t length =
        f32_as_i32(ceilf((atten - 8.f) / (2.285f * apt_oracle_freq_get_rad(delta_w_pi_rad)))) + 1;
    if (length % 2 == 0) {
        length += 1;
    }

    float *window = (float *)malloc(sizeof(float) * (size_t)(length > 0 ? length : 1));
    size_t cnt = 0;
    /* :171-175  for n in -(length-1)/2 ..= (length-1)/2 */
    for (int32_t ni = -(length - 1) / 2; ni <= (length - 1) / 2; ni++) {
        float n = (float)ni;
        float m = (float)length;
        float q = n / (m / 2.f);
        window[cnt++] = apt_oracle_bessel_i0(beta * sqrtf(1.f - q * q)) / apt_oracle_bessel_i0(beta);
    }
    *n_out = cnt;
    return window;
}

/* Filter::design for NoFilter / Lowpass / LowpassDcRemoval,
 * src/filters.rs:48-54, 57-88, 98-132; product() :186-196 */
float *apt_oracle_filter_design(const apt_oracle_filter_spec *f, size_t *n_out)
{
    if (f->kind == APT_FILTER_NOFILTER) {
        float *c = (float *)malloc(sizeof(float));
        c[0] = 1.f;
        *n_out = 1;
        return c;
    }

    size_t wlen = 0;
    float *window = apt_oracle_kaiser(f->atten, f->delta_w_pi_rad, &wlen);
    /* even length would panic in the reference (:68-70, :109-111); kaiser()
     * forces odd so this cannot happen */
    float *filter = (float *)malloc(sizeof(float) * (wlen ? wlen : 1));
    int32_t m = (int32_t)wlen;
    size_t cnt = 0;
    float cutout = f->cutout_pi_rad;
    float half_dw = f->delta_w_pi_rad / 2.f; /* (self.delta_w / 2.).get_pi_rad() */

    for (int32_t ni = -(m - 1) / 2; ni <= (m - 1) / 2; ni++) {
        if (f->kind == APT_FILTER_LOWPASS) {
            if (ni == 0) {
                filter[cnt++] = cutout; /* :77 */
            } else {
                float n = (float)ni;
                filter[cnt++] = sinf(n * PI_F * cutout) / (n * PI_F); /* :80 */
            }
        } else {
            if (ni == 0) {
                filter[cnt++] = cutout - half_dw; /* :118 */
            } else {
                float n = (float)ni;
                /* :121-124 */
                filter[cnt++] = sinf(n * PI_F * cutout) / (n * PI_F) -
                                sinf(n * PI_F * half_dw) / (n * PI_F);
            }
        }
    }
    /* product(filter, &window) :186-196 */
    for (size_t i = 0; i < wlen; i++) {
        filter[i] *= window[i];
    }
    free(window);
    *n_out = wlen;
    return filter;
}

/* Filter::resample, src/filters.rs:90-94, 134-138 (NoFilter: no-op :53) */
void apt_oracle_filter_resample(apt_oracle_filter_spec *f, uint32_t in_rate, uint32_t out_rate)
{
    if (f->kind == APT_FILTER_NOFILTER) return;
    float ratio = (float)out_rate / (float)in_rate;
    f->cutout_pi_rad /= ratio;
    f->delta_w_pi_rad /= ratio;
}

/* ---------------------------------------------------------------------- */
/* dsp.rs                                                                  */
/* ---------------------------------------------------------------------- */

/* fast_resampling, src/dsp.rs:186-289 (export_resample_filtered == false) */
float *apt_oracle_fast_resampling(const float *signal, size_t len, uint32_t l32, uint32_t m32,
                                  const float *coeff, size_t ncoeff, size_t *n_out)
{
    uint64_t l = l32, m = m32;
    uint64_t interpolated_len = (uint64_t)len * l; /* :203 */
    uint64_t output_len = interpolated_len / m;    /* :206 (capacity only) */
    size_t cap = (size_t)output_len + 2;
    float *output = (float *)malloc(sizeof(float) * cap);
    size_t cnt = 0;

    uint64_t offset = ((uint64_t)ncoeff - 1) / 2; /* :226 */
    uint64_t n;
    uint64_t t = offset; /* :230 */

    while (t < interpolated_len) { /* :234 */
        if (t > offset) {          /* :237-248 */
            n = t - offset;
            uint64_t rem = n % l;
            if (rem != 0) n += l - rem;
        } else {
            n = 0;
        }

        float sum = 0.f; /* :252 */
        uint64_t x = n / l;
        while (n <= t + offset) { /* :254 */
            if (x < (uint64_t)len) {  /* signal.get(x) */
                sum += coeff[n + offset - t] * signal[x]; /* :259 */
            }
            x += 1;
            n += l;
        }
        if (cnt == cap) {
            cap *= 2;
            output = (float *)realloc(output, sizeof(float) * cap);
        }
        output[cnt++] = sum; /* :276 */
        t += m;              /* :277 */
    }
    *n_out = cnt;
    return output;
}

/* decimate, src/dsp.rs:294-307 */
float *apt_oracle_decimate(const float *x, size_t n, uint32_t m, size_t *n_out)
{
    size_t cnt = n / m;
    float *out = (float *)malloc(sizeof(float) * (cnt ? cnt : 1));
    for (size_t i = 0; i < cnt; i++) out[i] = x[i * m];
    *n_out = cnt;
    return out;
}

/* demodulate, src/dsp.rs:350-383; carrier given as Freq.pi_rad.
 * NOTE phi = 2 * get_rad() as the code says (:360), not as its comment says. */
float *apt_oracle_demodulate(const float *signal, size_t n, float carrier_pi_rad)
{
    float *output = (float *)calloc(n ? n : 1, sizeof(float)); /* vec![0; len] :357 */
    if (n == 0) return output; /* the reference would panic on signal[0] (:367) */
    float phi = 2.f * apt_oracle_freq_get_rad(carrier_pi_rad); /* :360 */
    float cosphi2 = cosf(phi) * 2.f;                           /* :362 */
    float sinphi = sinf(phi);                                  /* :363 */

    float curr, curr_sq;
    float prev = signal[0];
    float prev_sq = signal[0] * signal[0];
    for (size_t i = 1; i < n; i++) { /* :369-377 */
        curr = signal[i];
        curr_sq = signal[i] * signal[i];
        output[i] = sqrtf(prev_sq + curr_sq - (prev * curr * cosphi2)) / sinphi;
        prev = curr;
        prev_sq = curr_sq;
    }
    return output;
}

/* filter, src/dsp.rs:386-410 — causal FIR with the `i > j` guard */
float *apt_oracle_fir(const float *signal, size_t n, const float *coeff, size_t ncoeff)
{
    float *output = (float *)calloc(n ? n : 1, sizeof(float));
    for (size_t i = 0; i < n; i++) { /* :396-404 */
        float sum = 0.f;
        size_t jmax = ncoeff < i ? ncoeff : i; /* j < ncoeff && i > j */
        for (size_t j = 0; j < jmax; j++) {
            sum += signal[i - j] * coeff[j];
        }
        output[i] = sum;
    }
    return output;
}

static uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    /* gcd crate 2.3.0 (Cargo.lock:532-533); any correct gcd is identical */
    while (b) {
        uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

/* fast_resampling with context.export_resample_filtered set, src/dsp.rs:186-289: every t of the
 * interpolated axis is evaluated (:265-273), `expanded` receives all of them (:269) and the output
 * the ones with (t + 1) % m == 0 (:270-273) -- NOT the t = offset + k*m of the other branch. */
float *apt_oracle_fast_resampling_export(const float *signal, size_t len, uint32_t l32, uint32_t m32,
                                         const float *coeff, size_t ncoeff, size_t *n_out,
                                         float **expanded_out, size_t *n_expanded)
{
    uint64_t l = l32, m = m32;
    uint64_t interpolated_len = (uint64_t)len * l; /* :203 */
    uint64_t offset = ((uint64_t)ncoeff - 1) / 2;  /* :226 */
    size_t cap = (size_t)(interpolated_len / m) + 2;
    float *output = (float *)malloc(sizeof(float) * cap);
    size_t ecap = interpolated_len > offset ? (size_t)(interpolated_len - offset) : 1;
    float *expanded = (float *)malloc(sizeof(float) * ecap);
    size_t cnt = 0, ecnt = 0;
    uint64_t n;
    uint64_t t = offset; /* :230 */

    while (t < interpolated_len) { /* :234 */
        if (t > offset) {          /* :237-248 */
            n = t - offset;
            uint64_t rem = n % l;
            if (rem != 0) n += l - rem;
        } else {
            n = 0;
        }
        float sum = 0.f; /* :252 */
        uint64_t x = n / l;
        while (n <= t + offset) { /* :254 */
            if (x < (uint64_t)len) {
                sum += coeff[n + offset - t] * signal[x]; /* :259 */
            }
            x += 1;
            n += l;
        }
        expanded[ecnt++] = sum; /* :269 */
        t += 1;                 /* :270 */
        if (t % m == 0) {       /* :271 */
            if (cnt == cap) {
                cap *= 2;
                output = (float *)realloc(output, sizeof(float) * cap);
            }
            output[cnt++] = sum; /* :272 */
        }
    }
    *n_out = cnt;
    if (expanded_out) {
        *expanded_out = expanded;
        if (n_expanded) *n_expanded = ecnt;
    } else {
        free(expanded);
    }
    return output;
}

/* resample_with_filter, src/dsp.rs:62-126 */
int apt_oracle_resample_with_filter(const float *x, size_t n, uint32_t in_rate,
                                    uint32_t out_rate, apt_oracle_filter_spec filt, float **out,
                                    size_t *n_out, float **coeff_out, size_t *ncoeff_out,
                                    char *err, size_t err_cap)
{
    return apt_oracle_resample_with_filter_ex(x, n, in_rate, out_rate, filt, 0, out, n_out, coeff_out,
                                              ncoeff_out, NULL, NULL, err, err_cap);
}

/* the same with Context.export_resample_filtered (src/context.rs:113) as an argument; `expanded_out`
 * (nullable) receives what the "resample_filtered" step would carry (dsp.rs:281-285; on the l == 1 branch
 * the step carries the filtered signal, :110-114) */
int apt_oracle_resample_with_filter_ex(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate,
                                       apt_oracle_filter_spec filt, int export_resample_filtered,
                                       float **out, size_t *n_out, float **coeff_out, size_t *ncoeff_out,
                                       float **expanded_out, size_t *n_expanded, char *err, size_t err_cap)
{
    if (expanded_out) *expanded_out = NULL;
    if (n_expanded) *n_expanded = 0;
    if (out_rate == 0) { /* :69-71 */
        set_err(err, err_cap, "Can't resample to 0Hz");
        return APT_ORACLE_ERR_INTERNAL;
    }
    uint32_t g = gcd_u32(in_rate, out_rate); /* :73 */
    uint32_t l = out_rate / g;               /* :74 */
    uint32_t m = in_rate / g;                /* :75 */

    size_t ncoeff = 0;
    float *coeff = NULL;
    if (l > 1) { /* :79 */
        uint64_t prod = (uint64_t)in_rate * (uint64_t)l; /* checked_mul :82 */
        if (prod > 0xFFFFFFFFull) {
            char buf[512];
            snprintf(buf, sizeof buf,
                     "Can't resample, looks like the sample rates do not have a big\n"
                     "                divisor in common. input_rate: %u, output_rate: %u, l: %u, m: %u",
                     in_rate, out_rate, l, m);
            set_err(err, err_cap, buf);
            return APT_ORACLE_ERR_RATE_OVERFLOW;
        }
        apt_oracle_filter_resample(&filt, in_rate, (uint32_t)prod); /* :93 */
        coeff = apt_oracle_filter_design(&filt, &ncoeff);           /* :94 */
        if (export_resample_filtered)
            *out = apt_oracle_fast_resampling_export(x, n, l, m, coeff, ncoeff, n_out, expanded_out, n_expanded);
        else
            *out = apt_oracle_fast_resampling(x, n, l, m, coeff, ncoeff, n_out); /* :98 */
    } else {
        coeff = apt_oracle_filter_design(&filt, &ncoeff);
        float *filtered = apt_oracle_fir(x, n, coeff, ncoeff); /* :108 */
        *out = apt_oracle_decimate(filtered, n, m, n_out);        /* :116 */
        if (expanded_out) { /* :110-114 */
            *expanded_out = filtered;
            if (n_expanded) *n_expanded = n;
        } else {
            free(filtered);
        }
    }
    if (coeff_out) {
        *coeff_out = coeff;
        if (ncoeff_out) *ncoeff_out = ncoeff;
    } else {
        free(coeff);
    }
    return APT_ORACLE_OK;
}

/* dsp::resample, src/dsp.rs:132-162 */
int apt_oracle_resample(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate,
                        float atten, float delta_w_pi_rad, float **out, size_t *n_out,
                        char *err, size_t err_cap)
{
    float cutout;
    if (out_rate > in_rate) {
        cutout = apt_oracle_freq_hz((float)in_rate / 2.f, in_rate); /* :144 */
    } else {
        cutout = apt_oracle_freq_hz((float)out_rate / 2.f, in_rate); /* :148 */
    }
    apt_oracle_filter_spec f = {APT_FILTER_LOWPASS, cutout, atten, delta_w_pi_rad};
    return apt_oracle_resample_with_filter(x, n, in_rate, out_rate, f, out, n_out, NULL, NULL,
                                           err, err_cap);
}

/* the same with Context.export_resample_filtered and the steps of Context::resample (context.rs:214-256):
 * coeff_out = "resample_filter", expanded_out = "resample_filtered" (all nullable) */
int apt_oracle_resample_ex(const float *x, size_t n, uint32_t in_rate, uint32_t out_rate, float atten,
                           float delta_w_pi_rad, int export_resample_filtered, float **out, size_t *n_out,
                           float **coeff_out, size_t *ncoeff_out, float **expanded_out, size_t *n_expanded,
                           char *err, size_t err_cap)
{
    float cutout = out_rate > in_rate ? apt_oracle_freq_hz((float)in_rate / 2.f, in_rate)   /* :144 */
                                      : apt_oracle_freq_hz((float)out_rate / 2.f, in_rate); /* :148 */
    apt_oracle_filter_spec f = {APT_FILTER_LOWPASS, cutout, atten, delta_w_pi_rad};
    return apt_oracle_resample_with_filter_ex(x, n, in_rate, out_rate, f, export_resample_filtered, out, n_out,
                                              coeff_out, ncoeff_out, expanded_out, n_expanded, err, err_cap);
}

/* ---------------------------------------------------------------------- */
/* decode.rs                                                               */
/* ---------------------------------------------------------------------- */

/* generate_sync_frame, src/decode.rs:171-199 */
int apt_oracle_generate_sync_frame(uint32_t work_rate, int8_t **out, size_t *n_out, char *err,
                                   size_t err_cap)
{
    if (work_rate % FINAL_RATE != 0) { /* :172-176 */
        set_err(err, err_cap, "work_rate is not multiple of FINAL_RATE");
        return APT_ORACLE_ERR_INTERNAL;
    }
    size_t pixel_width = work_rate / FINAL_RATE;
    size_t spw = pixel_width * 2; /* sync_pulse_width */
    size_t total = spw + 7 * 2 * spw + 8 * pixel_width;
    int8_t *g = (int8_t *)malloc(total ? total : 1);
    size_t c = 0;
    for (size_t i = 0; i < spw; i++) g[c++] = -1;            /* :188-189 */
    for (size_t i = 0; i < 7 * 2 * spw; i++) {               /* :191-196 cycle */
        size_t ph = i % (2 * spw);
        g[c++] = (ph < spw) ? -1 : 1;
    }
    for (size_t i = 0; i < 8 * pixel_width; i++) g[c++] = -1; /* :197 */
    *out = g;
    *n_out = c;
    return APT_ORACLE_OK;
}

/* find_sync, src/decode.rs:204-263 */
int apt_oracle_find_sync(const float *signal, size_t len, uint32_t work_rate,
                         uint64_t **pos_out, size_t *npos_out, float **correlation_out,
                         size_t *ncorr_out, char *err, size_t err_cap)
{
    int8_t *guard = NULL;
    size_t glen = 0;
    int rc = apt_oracle_generate_sync_frame(work_rate, &guard, &glen, err, err_cap);
    if (rc) return rc;

    /* peaks: Vec<(usize, f32)>, starts with (0, 0.) :207-209 */
    size_t pcap = 1024, plen = 0;
    uint64_t *pidx = (uint64_t *)malloc(sizeof(uint64_t) * pcap);
    float *pval = (float *)malloc(sizeof(float) * pcap);
    pidx[0] = 0;
    pval[0] = 0.f;
    plen = 1;

    uint32_t spr32 = PX_PER_ROW * work_rate / FINAL_RATE; /* :212 (u32 arithmetic) */
    size_t spr = spr32;
    size_t min_distance = spr * 8 / 10; /* :216 */

    size_t ncorr = (len >= glen) ? len - glen : 0; /* reference panics if len < glen */
    float *correlation = NULL;
    if (correlation_out) correlation = (float *)malloc(sizeof(float) * (ncorr ? ncorr : 1));

    for (size_t i = 0; i < ncorr; i++) { /* :225 */
        float corr = 0.f;
        for (size_t j = 0; j < glen; j++) { /* :227-233 */
            if (guard[j] == 1)
                corr += signal[i + j];
            else
                corr -= signal[i + j];
        }
        if (correlation) correlation[i] = corr;

        if (i - (size_t)pidx[plen - 1] > min_distance) { /* :241 */
            while (i / spr > plen) {                     /* :244 */
                if (plen == pcap) {
                    pcap *= 2;
                    pidx = (uint64_t *)realloc(pidx, sizeof(uint64_t) * pcap);
                    pval = (float *)realloc(pval, sizeof(float) * pcap);
                }
                pidx[plen] = i;
                pval[plen] = corr;
                plen++;
            }
        } else if (corr > pval[plen - 1]) { /* :250-253 */
            pidx[plen - 1] = i;
            pval[plen - 1] = corr;
        }
    }
    free(guard);
    free(pval);
    *pos_out = pidx;
    *npos_out = plen;
    if (correlation_out) {
        *correlation_out = correlation;
        if (ncorr_out) *ncorr_out = ncorr;
    }
    return APT_ORACLE_OK;
}

/* decode, src/decode.rs:43-162 */
int apt_oracle_decode(const apt_oracle_settings *s, const float *x, size_t n,
                      uint32_t input_rate, int sync, float **out, size_t *n_out,
                      apt_oracle_steps *steps, char *err, size_t err_cap)
{
    return apt_oracle_decode_ex(s, x, n, input_rate, sync, 0, out, n_out, steps, NULL, NULL, NULL, NULL,
                                err, err_cap);
}

/* decode with Context.export_resample_filtered as an argument (it moves the decimation phase of every
 * fast_resampling call, dsp.rs:265-273, whether or not anything is exported); expanded1 / expanded2
 * (nullable) receive the "resample_filtered" steps of the first and of the final resample */
int apt_oracle_decode_ex(const apt_oracle_settings *s, const float *x, size_t n, uint32_t input_rate,
                         int sync, int export_resample_filtered, float **out, size_t *n_out,
                         apt_oracle_steps *steps, float **expanded1, size_t *n_expanded1,
                         float **expanded2, size_t *n_expanded2, char *err, size_t err_cap)
{
    int rc;
    if (expanded1) *expanded1 = NULL;
    if (n_expanded1) *n_expanded1 = 0;
    if (expanded2) *expanded2 = NULL;
    if (n_expanded2) *n_expanded2 = 0;
    double t0, t1;
    if (steps) memset(steps, 0, sizeof(*steps));
    *out = NULL;
    *n_out = 0;

    uint32_t spr32 = PX_PER_ROW * s->work_rate / FINAL_RATE; /* :55 */
    size_t spr = spr32;
    uint32_t work_rate = s->work_rate;

    /* :65-77 */
    apt_oracle_filter_spec f1 = {APT_FILTER_LOWPASS_DC_REMOVAL,
                            apt_oracle_freq_hz(s->resample_cutout, input_rate), s->resample_atten,
                            apt_oracle_freq_hz(s->resample_delta_freq, input_rate)};
    float *sig = NULL, *coeff1 = NULL;
    size_t nsig = 0, ncoeff1 = 0;
    t0 = now_s();
    rc = apt_oracle_resample_with_filter_ex(x, n, input_rate, work_rate, f1, export_resample_filtered, &sig,
                                            &nsig, &coeff1, &ncoeff1, expanded1, n_expanded1, err, err_cap);
    t1 = now_s();
    if (rc) return rc;
    if (steps) {
        steps->t_resample = t1 - t0;
        steps->resample_filter = coeff1;
        steps->n_resample_filter = ncoeff1;
    } else {
        free(coeff1);
    }

    if (nsig < 10 * spr) { /* :79-83 */
        free(sig);
        set_err(err, err_cap, "Got less than 10 rows of samples, audio file is too short");
        return APT_ORACLE_ERR_INTERNAL;
    }

    /* :89 */
    t0 = now_s();
    float *dem = apt_oracle_demodulate(sig, nsig, apt_oracle_freq_hz((float)CARRIER_FREQ, work_rate));
    t1 = now_s();
    if (steps) {
        steps->t_demod = t1 - t0;
        steps->resampled = sig;
        steps->n_resampled = nsig;
    } else {
        free(sig);
    }

    /* :95-102 */
    float cutout = (float)FINAL_RATE / (float)work_rate; /* Freq::pi_rad(...) */
    apt_oracle_filter_spec f2 = {APT_FILTER_LOWPASS, cutout, s->demodulation_atten, cutout / 5.f};
    size_t ncoeff2 = 0;
    float *coeff2 = apt_oracle_filter_design(&f2, &ncoeff2);
    t0 = now_s();
    float *fil = apt_oracle_fir(dem, nsig, coeff2, ncoeff2);
    t1 = now_s();
    if (steps) {
        steps->t_filter = t1 - t0;
        steps->demodulated = dem;
        steps->n_demodulated = nsig;
        steps->filter_filter = coeff2;
        steps->n_filter_filter = ncoeff2;
    } else {
        free(dem);
        free(coeff2);
    }

    float *aligned = NULL;
    size_t naligned = 0;
    if (sync) { /* :106-134 */
        uint64_t *pos = NULL;
        size_t npos = 0;
        float *corr = NULL;
        size_t ncorr = 0;
        t0 = now_s();
        rc = apt_oracle_find_sync(fil, nsig, work_rate, &pos, &npos, steps ? &corr : NULL,
                                  &ncorr, err, err_cap);
        t1 = now_s();
        if (rc) {
            if (!steps) free(fil); else { steps->filtered = fil; steps->n_filtered = nsig; }
            return rc;
        }
        if (steps) {
            steps->t_sync = t1 - t0;
            steps->correlation = corr;
            steps->n_correlation = ncorr;
        }
        if (npos < 5) { /* :112-118 */
            free(pos);
            if (!steps) free(fil); else { steps->filtered = fil; steps->n_filtered = nsig; }
            set_err(err, err_cap,
                    "Found less than 5 sync frames, audio file is too short or too noisy");
            return APT_ORACLE_ERR_INTERNAL;
        }
        t0 = now_s();
        aligned = (float *)malloc(sizeof(float) * ((npos * spr) != 0 ? npos * spr : 1));
        for (size_t i = 0; i + 1 < npos; i++) { /* :125 for i in 0..len-1 */
            if ((size_t)pos[i] + spr < nsig) {  /* :127 strict */
                memcpy(aligned + naligned, fil + pos[i], sizeof(float) * spr);
                naligned += spr;
            }
        }
        t1 = now_s();
        if (steps) {
            steps->t_gather = t1 - t0;
            steps->sync_pos = pos;
            steps->n_sync_pos = npos;
        } else {
            free(pos);
        }
    } else { /* :135-148 */
        naligned = nsig / spr * spr;
        aligned = (float *)malloc(sizeof(float) * (naligned ? naligned : 1));
        memcpy(aligned, fil, sizeof(float) * naligned);
    }
    if (steps) {
        steps->filtered = fil;
        steps->n_filtered = nsig;
    } else {
        free(fil);
    }

    /* :158-159: resample_with_filter(aligned, work_rate, 4160, NoFilter)
     * -> l == 1 branch: filter([1.]) then decimate(m) */
    apt_oracle_filter_spec nf = {APT_FILTER_NOFILTER, 0.f, 0.f, 0.f};
    t0 = now_s();
    rc = apt_oracle_resample_with_filter_ex(aligned, naligned, work_rate, FINAL_RATE, nf,
                                            export_resample_filtered, out, n_out, NULL, NULL, expanded2,
                                            n_expanded2, err, err_cap);
    t1 = now_s();
    if (steps) {
        steps->t_gather += t1 - t0;
        steps->aligned = aligned;
        steps->n_aligned = naligned;
    } else {
        free(aligned);
    }
    return rc;
}
