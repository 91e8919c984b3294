/*
 * apt_oracle_image.c — CPU parity oracle for the consumers of decode()'s pixel rows:
 * contrast limits, u8 mapping and telemetry.
 *
 * TEST INFRASTRUCTURE ONLY (see apt_oracle.h).  Plain-C scalar restatement of the
 * reference's Rust loops, same order, f32 everywhere the Rust is f32.  Paths below are
 * relative to /root/reference/.
 *
 * Rust semantics restated here on purpose:
 *   - `x as usize` / `x as u8` on a float saturates and maps NaN to 0;
 *   - f32::max / f32::min return the non-NaN operand;
 *   - f32::round rounds half away from zero (roundf);
 *   - Iterator::sum::<f32>() folds left to right starting from 0.0;
 *   - powi(2) is x*x.
 *
 * PINNING: map_signal_u8 by src/noaa_apt.rs:266-281 (exact vector); percent by
 * src/misc.rs:515-543 (1 % bounds); Telemetry::from_bands / get_channel_name by
 * src/telemetry.rs:255-348 (10 ULP / exact names).  read_telemetry()'s numeric output is
 * "parity unpinned" by the reference (no golden), like decode().
 */
#include "apt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void img_err(char *err, size_t cap, const char *msg)
{
    if (err && cap) snprintf(err, cap, "%s", msg);
}

/* dsp::get_max, src/dsp.rs:20-35 — strict `>` keeps the FIRST of equal values */
int apt_oracle_get_max(const float *x, size_t n, float *out, char *err, size_t err_cap)
{
    if (n == 0) {
        img_err(err, err_cap, "Can't get maximum of a zero length vector");
        return APT_ORACLE_ERR_INTERNAL;
    }
    float best = x[0];
    for (size_t i = 0; i < n; i++)
        if (x[i] > best) best = x[i];
    *out = best;
    return APT_ORACLE_OK;
}

/* dsp::get_min, src/dsp.rs:38-54 */
int apt_oracle_get_min(const float *x, size_t n, float *out, char *err, size_t err_cap)
{
    if (n == 0) {
        img_err(err, err_cap, "Can't get minimum of a zero length vector");
        return APT_ORACLE_ERR_INTERNAL;
    }
    float best = x[0];
    for (size_t i = 0; i < n; i++)
        if (x[i] < best) best = x[i];
    *out = best;
    return APT_ORACLE_OK;
}

/* Rust `f as usize`: saturating, NaN -> 0 */
static size_t f32_as_usize(float f)
{
    if (!(f > 0.0f)) return 0; /* negatives, -0, +0 and NaN */
    if (f >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)f;
}

/* misc::percent, src/misc.rs:119-175.  buckets_out (nullable) receives the 1000 counts. */
int apt_oracle_percent(const float *x, size_t n, float percent, float *low, float *high,
                       uint32_t *buckets_out, char *err, size_t err_cap)
{
    if (percent < 0.f || percent > 1.f) { /* :120-124 */
        img_err(err, err_cap, "Percent given should be between 0 and 1");
        return APT_ORACLE_ERR_INTERNAL;
    }
    const float remainder = (1.f - percent) / 2.f; /* :126 */
    enum { NB = 1000 };                             /* :129 */
    uint32_t buckets[NB];
    memset(buckets, 0, sizeof buckets);
    float min, max;
    int rc = apt_oracle_get_min(x, n, &min, err, err_cap); /* :135 */
    if (rc) return rc;
    rc = apt_oracle_get_max(x, n, &max, err, err_cap); /* :136 */
    if (rc) return rc;
    const float total_range = max - min; /* :137 */
    for (size_t i = 0; i < n; i++) {     /* :147-149 with get_bucket :140-144 */
        size_t b = f32_as_usize(truncf((x[i] - min) / total_range * (float)NB));
        if (b > NB - 1) b = NB - 1;
        buckets[b] += 1;
    }
    uint32_t accum = 0; /* :152-163 */
    long low_bucket = -1, high_bucket = -1;
    for (size_t b = 0; b < NB; b++) {
        accum += buckets[b];
        const float frac = (float)accum / (float)n;
        if (low_bucket < 0 && frac > remainder)
            low_bucket = (long)b;
        else if (high_bucket < 0 && frac > 1.f - remainder)
            high_bucket = (long)b;
    }
    if (high_bucket < 0) high_bucket = NB - 1; /* :165-169 */
    if (low_bucket < 0) {
        /* `low_bucket.unwrap()` panics in the reference (:172); only reachable with NaN input */
        img_err(err, err_cap, "percent: no low bucket (reference would panic)");
        return APT_ORACLE_ERR_INTERNAL;
    }
    *low = (float)low_bucket / (float)NB * total_range + min; /* :171-174 */
    *high = (float)high_bucket / (float)NB * total_range + min;
    if (buckets_out) memcpy(buckets_out, buckets, sizeof buckets);
    return APT_ORACLE_OK;
}

/* map_signal_u8, src/noaa_apt.rs:249-259 */
void apt_oracle_map_signal_u8(const float *x, size_t n, float low, float high, uint8_t *out)
{
    const float range = high - low;
    for (size_t i = 0; i < n; i++) {
        float v = (x[i] - low) / range * 255.f;
        v = fmaxf(v, 0.f);   /* .max(0.)  : NaN -> 0 */
        v = fminf(v, 255.f); /* .min(255.) */
        v = roundf(v);       /* .round()  : half away from zero */
        out[i] = (uint8_t)v; /* 0 <= v <= 255 here */
    }
}

/* ------------------------------------------------------------------ telemetry.rs */
#define APT_TELEMETRY_SAMPLE_LEN 200 /* 25 wedges x 8 rows, src/telemetry.rs:134-141 */

static void telemetry_sample(float *s)
{
    static const float wedges[25] = {31.f, 63.f, 95.f, 127.f, 159.f, 191.f, 224.f, 255.f, 0.f,
                                     0.f,  0.f,  0.f,  0.f,   0.f,   0.f,   0.f,
                                     31.f, 63.f, 95.f, 127.f, 159.f, 191.f, 224.f, 255.f, 0.f};
    for (int w = 0; w < 25; w++)
        for (int r = 0; r < 8; r++) s[w * 8 + r] = wedges[w];
}

/* Telemetry::from_bands, src/telemetry.rs:30-72 */
void apt_oracle_telemetry_from_bands(const float *means_a, const float *means_b, size_t n,
                                     size_t row, float values_a[16], float values_b[16])
{
    float wa[25], wb[25];
    for (int w = 0; w < 25; w++) { /* chunks_exact(8).map(sum/8).take(25), :34-43 */
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < 8; r++) {
            size_t k = row + (size_t)w * 8 + (size_t)r;
            sa += k < n ? means_a[k] : 0.f; /* callers guarantee k < n */
            sb += k < n ? means_b[k] : 0.f;
        }
        wa[w] = sa / 8.f;
        wb[w] = sb / 8.f;
    }
    for (int wedge = 1; wedge <= 16; wedge++) { /* :46-63 */
        values_a[wedge - 1] = wedge <= 9 ? (wa[wedge - 1] + wa[wedge + 16 - 1]) / 2.f : wa[wedge - 1];
        values_b[wedge - 1] = wedge <= 9 ? (wb[wedge - 1] + wb[wedge + 16 - 1]) / 2.f : wb[wedge - 1];
    }
}

/* Telemetry::get_wedge_value, src/telemetry.rs:78-90; channel: 0 = A, 1 = B, -1 = None */
float apt_oracle_telemetry_wedge_value(const float values_a[16], const float values_b[16],
                                       uint32_t wedge, int channel)
{
    if (channel == 0) return values_a[wedge - 1];
    if (channel == 1) return values_b[wedge - 1];
    return (values_a[wedge - 1] + values_b[wedge - 1]) / 2.f;
}

/* Telemetry::get_channel_name, src/telemetry.rs:93-121 — index into
 * ["1","2","3a","4","5","3b","Unknown","Unknown","Unknown"]; Iterator::min_by keeps the
 * FIRST minimum.  Returns -1 where the reference would panic ("Can't compare values"). */
int apt_oracle_telemetry_channel_index(const float values_a[16], const float values_b[16], int channel)
{
    const float value = apt_oracle_telemetry_wedge_value(values_a, values_b, 16, channel);
    int best = 0;
    float best_d = 0.f;
    for (int i = 1; i <= 9; i++) {
        const float d = fabsf(apt_oracle_telemetry_wedge_value(values_a, values_b, (uint32_t)i, -1) - value);
        if (isnan(d)) return -1;
        if (i == 1 || d < best_d) {
            best = i - 1;
            best_d = d;
        }
    }
    return best;
}

/* read_telemetry, src/telemetry.rs:125-243.  `signal` is decode()'s output (rows x 2080).
 * Optional outputs (nullable, malloc'd here, rows or rows-200 long): mean_a, mean_b,
 * variance, corr, quality — the "telemetry_*" steps of :234-238. */
int apt_oracle_read_telemetry(const float *signal, size_t n, float values_a[16], float values_b[16],
                              uint64_t *best_row, float *best_quality, float **mean_a_out,
                              float **mean_b_out, float **variance_out, float **corr_out,
                              float **quality_out, size_t *rows_out, char *err, size_t err_cap)
{
    enum { PX = 2080, TS = APT_TELEMETRY_SAMPLE_LEN };
    float sample[TS];
    telemetry_sample(sample);
    const size_t rows = n / PX; /* chunks_exact(PX_PER_ROW), :154 */
    float *mean_a = malloc(sizeof(float) * (rows ? rows : 1));
    float *mean_b = malloc(sizeof(float) * (rows ? rows : 1));
    float *variance = malloc(sizeof(float) * (rows ? rows : 1));
    float *corr = NULL, *quality = NULL;
    for (size_t r = 0; r < rows; r++) {
        const float *a = signal + r * PX + 994;  /* :156 */
        const float *b = signal + r * PX + 2034; /* :157 */
        float sa = 0.f, sb = 0.f;
        for (int i = 0; i < 44; i++) sa += a[i];
        for (int i = 0; i < 44; i++) sb += b[i];
        const float ma = sa / 44.f, mb = sb / 44.f; /* :160-161 */
        mean_a[r] = ma;
        mean_b[r] = mb;
        float va = 0.f, vb = 0.f; /* :166-176 */
        for (int i = 0; i < 44; i++) {
            const float d = a[i] - ma;
            va += d * d;
        }
        for (int i = 0; i < 44; i++) {
            const float d = b[i] - mb;
            vb += d * d;
        }
        variance[r] = (va + vb) / 88.f;
    }
    int rc = APT_ORACLE_OK;
    if (rows < TS) { /* :199-203 */
        img_err(err, err_cap, "Recording too short for telemetry decoding");
        rc = APT_ORACLE_ERR_INTERNAL;
    } else {
        const size_t nc = rows - TS; /* :210 */
        corr = malloc(sizeof(float) * (nc ? nc : 1));
        quality = malloc(sizeof(float) * (nc ? nc : 1));
        size_t best_i = 0; /* :196 */
        float best_q = 0.f;
        for (size_t i = 0; i < nc; i++) {
            float sum = 0.f;
            for (int j = 0; j < TS; j++) { /* :212-215 */
                sum += sample[j] * mean_a[i + (size_t)j];
                sum += sample[j] * mean_b[i + (size_t)j];
            }
            float sd = 0.f; /* :222-226 */
            for (int j = 0; j < TS; j++) sd += sqrtf(variance[i + (size_t)j]);
            const float q = sum / sd;
            if (q > best_q) { /* :228-230 */
                best_i = i;
                best_q = q;
            }
            corr[i] = sum;
            quality[i] = q;
        }
        apt_oracle_telemetry_from_bands(mean_a, mean_b, rows, best_i, values_a, values_b); /* :233 */
        if (best_row) *best_row = best_i;
        if (best_quality) *best_quality = best_q;
    }
    if (rows_out) *rows_out = rows;
    if (mean_a_out) *mean_a_out = mean_a; else free(mean_a);
    if (mean_b_out) *mean_b_out = mean_b; else free(mean_b);
    if (variance_out) *variance_out = variance; else free(variance);
    if (corr_out) *corr_out = corr; else free(corr);
    if (quality_out) *quality_out = quality; else free(quality);
    return rc;
}

/* The grayscale part of noaa_apt::process(), src/noaa_apt.rs:132-192: contrast limits then
 * map_signal_u8.  contrast: 0 = Telemetry (:141-150: low = wedge 9, high = wedge 8, both
 * channels averaged), 1 = Percent(p) (:151-157), 2 = MinMax (:158-164; Histogram takes the
 * same limits, its equalisation happens after this point and is out of scope). */
int apt_oracle_process_gray(const float *signal, size_t n, int contrast, float percent,
                            uint8_t **image_out, size_t *n_out, float *low_out, float *high_out,
                            char *err, size_t err_cap)
{
    float low = 0.f, high = 0.f;
    int rc;
    if (contrast == 0) {
        float va[16], vb[16];
        rc = apt_oracle_read_telemetry(signal, n, va, vb, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                                       NULL, err, err_cap);
        if (rc) return rc;
        low = apt_oracle_telemetry_wedge_value(va, vb, 9, -1);
        high = apt_oracle_telemetry_wedge_value(va, vb, 8, -1);
    } else if (contrast == 1) {
        rc = apt_oracle_percent(signal, n, percent, &low, &high, NULL, err, err_cap);
        if (rc) return rc;
    } else {
        rc = apt_oracle_get_min(signal, n, &low, err, err_cap);
        if (rc) return rc;
        rc = apt_oracle_get_max(signal, n, &high, err, err_cap);
        if (rc) return rc;
    }
    /* height = len / 2080 (:182); GrayImage::from_vec needs len >= 2080*height and takes the
     * whole mapped vector; decode() always returns whole rows */
    uint8_t *img = malloc(n ? n : 1);
    apt_oracle_map_signal_u8(signal, n, low, high, img);
    *image_out = img;
    *n_out = n;
    if (low_out) *low_out = low;
    if (high_out) *high_out = high;
    return APT_ORACLE_OK;
}
