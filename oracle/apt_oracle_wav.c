/*
 * apt_oracle_wav.c — CPU parity oracle for the WAV ingest in front of decode()
 * (SURVEY.md §8(f) N1).  TEST INFRASTRUCTURE ONLY (see apt_oracle.h).
 *
 * Restates wav::load_wav (src/wav.rs:11-57) + noaa_apt::load (src/noaa_apt.rs:114-130).
 * The container parsing lives in a third-party crate that is NOT in the reference tree:
 * hound 3.5.1 (Cargo.toml:29, Cargo.lock:811-812).  Its published algorithm is restated
 * here from the crate's source as remembered (WavReader::new -> read_wave_header,
 * read_until_data, read_fmt_chunk, read_wave_format_{pcm,ieee_float,extensible},
 * Sample::read for i32 and f32); it cannot be re-verified offline.  PINNING: the reference's
 * own fixture test/noise_48000hz.wav (tests/test_wav_ingest.py, when /root/reference is
 * present) and Python's independent `wave` / scipy.io.wavfile readers on generated files.
 *
 * Structure follows the reference: a sequential reader, samples pulled one by one and
 * collected, then every channels-th value kept and converted with `as f32` (wav.rs:31-51).
 */
#include "apt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *p;
    size_t n, at;
    int eof; /* set by a short read: io::ErrorKind::UnexpectedEof "Failed to read enough bytes." */
} rd;

static int rd_bytes(rd *r, void *out, size_t k)
{
    if (r->eof || k > r->n - r->at) {
        r->eof = 1;
        return 0;
    }
    if (out) memcpy(out, r->p + r->at, k);
    r->at += k;
    return 1;
}
static uint16_t rd_u16(rd *r)
{
    uint8_t b[2] = {0, 0};
    rd_bytes(r, b, 2);
    return (uint16_t)(b[0] | (b[1] << 8));
}
static uint32_t rd_u32(rd *r)
{
    uint8_t b[4] = {0, 0, 0, 0};
    rd_bytes(r, b, 4);
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}

static int fail(char *err, size_t cap, int code, const char *prefix, const char *msg)
{
    if (err && cap) snprintf(err, cap, "%s%s", prefix, msg);
    return code;
}
#define FORMAT_ERROR(msg) return fail(err, err_cap, APT_ORACLE_ERR_WAV_OPEN, "Ill-formed WAVE file: ", msg)
#define UNSUPPORTED() return fail(err, err_cap, APT_ORACLE_ERR_WAV_OPEN, "", "The wave format of the file is not supported.")
#define SHORT_READ() return fail(err, err_cap, APT_ORACLE_ERR_IO, "", "Failed to read enough bytes.")
#define TOO_WIDE() return fail(err, err_cap, APT_ORACLE_ERR_INTERNAL, "", "The sample has more bits than the destination type.")

int apt_oracle_load_wav(const uint8_t *bytes, size_t n, float **signal_out, size_t *n_out,
                        apt_oracle_wav_spec *spec, char *err, size_t err_cap)
{
    rd r = {bytes, n, 0, 0};
    uint8_t tag[4];
    *signal_out = NULL;
    *n_out = 0;

    /* read_wave_header */
    if (!rd_bytes(&r, tag, 4)) SHORT_READ();
    if (memcmp(tag, "RIFF", 4) != 0) FORMAT_ERROR("no RIFF tag found");
    (void)rd_u32(&r);
    if (!rd_bytes(&r, tag, 4)) SHORT_READ();
    if (memcmp(tag, "WAVE", 4) != 0) FORMAT_ERROR("no WAVE tag found");

    /* read_until_data */
    int have_fmt = 0, is_float = 0;
    uint16_t channels = 0, bits = 0, bytes_per_sample = 0;
    uint32_t rate = 0, data_len = 0;
    for (;;) {
        if (!rd_bytes(&r, tag, 4)) SHORT_READ();
        uint32_t len = rd_u32(&r);
        if (r.eof) SHORT_READ();
        if (memcmp(tag, "fmt ", 4) == 0) {
            /* read_fmt_chunk */
            if (len < 16) FORMAT_ERROR("invalid fmt chunk size");
            uint16_t format_tag = rd_u16(&r);
            uint16_t n_channels = rd_u16(&r);
            uint32_t n_samples_per_sec = rd_u32(&r);
            uint32_t n_bytes_per_sec = rd_u32(&r);
            uint16_t block_align = rd_u16(&r);
            uint16_t bits_per_sample = rd_u16(&r);
            if (r.eof) SHORT_READ();
            if (n_channels == 0) FORMAT_ERROR("file contains zero channels");
            bytes_per_sample = (uint16_t)(block_align / n_channels);
            if ((uint32_t)bytes_per_sample * 8u > 0xffffu || bits_per_sample > (uint32_t)bytes_per_sample * 8u)
                FORMAT_ERROR("sample bits exceeds size of sample");
            if ((uint64_t)block_align * n_samples_per_sec > 0xffffffffull ||
                (uint64_t)block_align * n_samples_per_sec != n_bytes_per_sec)
                FORMAT_ERROR("inconsistent fmt chunk");
            if (bits_per_sample % 8 != 0) FORMAT_ERROR("bits per sample is not a multiple of 8");
            if (bits_per_sample == 0) FORMAT_ERROR("bits per sample is 0");
            channels = n_channels;
            rate = n_samples_per_sec;
            bits = bits_per_sample;
            is_float = 0;
            if (format_tag == 1) { /* PCM */
                if (len != 16 && len != 18 && len != 40) FORMAT_ERROR("unexpected fmt chunk size");
                if (len != 16) {
                    (void)rd_u16(&r);
                    if (r.eof) SHORT_READ();
                    if (bits != 8 && bits != 16 && bits != 24) UNSUPPORTED();
                }
                if (len == 40 && !rd_bytes(&r, NULL, 22)) SHORT_READ();
            } else if (format_tag == 3) { /* IEEE float */
                if (len != 16 && len != 18) FORMAT_ERROR("unexpected fmt chunk size");
                if (len == 18) {
                    uint16_t cb = rd_u16(&r);
                    if (r.eof) SHORT_READ();
                    if (cb != 0) FORMAT_ERROR("unexpected WAVEFORMATEX size");
                }
                if (bits != 32) FORMAT_ERROR("bits per sample is not 32");
                is_float = 1;
            } else if (format_tag == 0xfffe) { /* extensible */
                static const uint8_t pcm[16] = {0x01, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71};
                static const uint8_t flt[16] = {0x03, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71};
                uint8_t guid[16];
                if (len < 40) FORMAT_ERROR("unexpected fmt chunk size");
                uint16_t cb = rd_u16(&r);
                if (r.eof) SHORT_READ();
                if (cb != 22) FORMAT_ERROR("unexpected WAVEFORMATEXTENSIBLE size");
                uint16_t valid = rd_u16(&r);
                (void)rd_u32(&r);
                if (!rd_bytes(&r, guid, 16)) SHORT_READ();
                if (memcmp(guid, pcm, 16) == 0) is_float = 0;
                else if (memcmp(guid, flt, 16) == 0) is_float = 1;
                else UNSUPPORTED();
                if (valid > 0) bits = valid;
            } else {
                UNSUPPORTED(); /* ADPCM and everything else */
            }
            have_fmt = 1;
        } else if (memcmp(tag, "fact", 4) == 0) {
            (void)rd_u32(&r); /* result ignored by hound */
            r.eof = 0;
            if (r.n - r.at < 4 && r.at != r.n) r.at = r.n;
        } else if (memcmp(tag, "data", 4) == 0) {
            if (!have_fmt) FORMAT_ERROR("missing fmt chunk");
            data_len = len;
            break;
        } else {
            if (!rd_bytes(&r, NULL, len)) SHORT_READ();
        }
    }

    /* WavReader::new */
    uint32_t num_samples = data_len / bytes_per_sample;
    if (num_samples * (uint32_t)bytes_per_sample != data_len)
        FORMAT_ERROR("data chunk length is not a multiple of sample size");
    if (num_samples % channels != 0) FORMAT_ERROR("invalid data chunk length");
    if (spec) {
        spec->channels = channels;
        spec->bits_per_sample = bits;
        spec->bytes_per_sample = bytes_per_sample;
        spec->sample_format = (uint16_t)is_float;
        spec->sample_rate = rate;
        spec->data_offset = r.at;
        spec->data_len = data_len;
        spec->n_samples = num_samples;
    }

    /* wav.rs:30-51: collect every sample (any error aborts), keep i % channels == 0 */
    size_t frames = num_samples / channels;
    float *out = malloc(sizeof(float) * (frames ? frames : 1));
    size_t k = 0;
    for (uint32_t i = 0; i < num_samples; i++) {
        float v;
        uint8_t b[4] = {0, 0, 0, 0};
        if (is_float) { /* samples::<f32>() */
            if (!(bytes_per_sample == 4 && bits == 32)) {
                free(out);
                if (bytes_per_sample > 4) TOO_WIDE();
                UNSUPPORTED();
            }
            if (!rd_bytes(&r, b, 4)) { free(out); SHORT_READ(); }
            uint32_t u = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
            memcpy(&v, &u, 4);
        } else { /* samples::<i32>() then `*x as f32` */
            int32_t s;
            if (bytes_per_sample == 1 && bits == 8) {
                if (!rd_bytes(&r, b, 1)) { free(out); SHORT_READ(); }
                s = (int32_t)(int8_t)(uint8_t)((int)b[0] - 128);
            } else if (bytes_per_sample == 2 && bits == 16) {
                if (!rd_bytes(&r, b, 2)) { free(out); SHORT_READ(); }
                s = (int16_t)(uint16_t)(b[0] | (b[1] << 8));
            } else if ((bytes_per_sample == 3 || bytes_per_sample == 4) && bits == 24) {
                if (!rd_bytes(&r, b, bytes_per_sample)) { free(out); SHORT_READ(); }
                uint32_t u = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
                if (u & 0x800000u) u |= 0xff000000u;
                s = (int32_t)u;
            } else if (bytes_per_sample == 4 && bits == 32) {
                if (!rd_bytes(&r, b, 4)) { free(out); SHORT_READ(); }
                s = (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
            } else {
                free(out);
                if (bytes_per_sample > 4) TOO_WIDE();
                UNSUPPORTED();
            }
            v = (float)s;
        }
        if (i % channels == 0) out[k++] = v;
    }
    *signal_out = out;
    *n_out = k;
    return APT_ORACLE_OK;
}

/* wav::write_wav, src/wav.rs:59-98, for the {channels: 1, bits_per_sample: 16, Int} spec that
 * resample::resample uses (src/resample.rs:53-58): `(sample / max * 32767.) as i16` with
 * max = dsp::get_max(signal) (wav.rs:70,83-86; Rust's float->int `as` truncates, saturates, NaN->0),
 * behind the header hound 3.5.1's WavWriter emits for such a spec (PCMWAVEFORMAT, 16-byte fmt
 * chunk: the canonical 44-byte header). */
int apt_oracle_write_wav_i16(const float *signal, size_t n, uint32_t rate, uint8_t **out, size_t *n_out,
                             char *err, size_t err_cap)
{
    float max;
    int rc = apt_oracle_get_max(signal, n, &max, err, err_cap);
    if (rc) return rc;
    uint8_t *f = malloc(44 + 2 * n);
    uint32_t data_len = (uint32_t)(2 * n), riff_len = 36 + data_len, byte_rate = rate * 2;
    memcpy(f, "RIFF", 4);
    memcpy(f + 4, &riff_len, 4); /* little-endian host */
    memcpy(f + 8, "WAVEfmt ", 8);
    uint32_t sixteen = 16;
    uint16_t pcm = 1, ch = 1, align = 2, bits = 16;
    memcpy(f + 16, &sixteen, 4);
    memcpy(f + 20, &pcm, 2);
    memcpy(f + 22, &ch, 2);
    memcpy(f + 24, &rate, 4);
    memcpy(f + 28, &byte_rate, 4);
    memcpy(f + 32, &align, 2);
    memcpy(f + 34, &bits, 2);
    memcpy(f + 36, "data", 4);
    memcpy(f + 40, &data_len, 4);
    for (size_t i = 0; i < n; i++) {
        const float v = signal[i] / max * 32767.f;
        int16_t q;
        if (v != v) q = 0;
        else if (v >= 32767.f) q = 32767;
        else if (v <= -32768.f) q = -32768;
        else q = (int16_t)v;
        memcpy(f + 44 + 2 * i, &q, 2);
    }
    *out = f;
    *n_out = 44 + 2 * n;
    return APT_ORACLE_OK;
}
