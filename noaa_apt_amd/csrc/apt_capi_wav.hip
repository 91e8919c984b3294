// apt_capi_wav.hip — extern "C" surface of include/aptgpu.h §5: WAV ingest in front of decode().
#include <cstdio>
#include <vector>

#include "apt_capi_util.hpp"

namespace {

using namespace apt::capi;

void fill_spec(aptgpu_wav_spec *spec, const apt::WavInfo &w)
{
    if (!spec) return;
    spec->channels = w.channels;
    spec->bits_per_sample = w.bits_per_sample;
    spec->bytes_per_sample = w.bytes_per_sample;
    spec->sample_format = w.is_float ? 1 : 0;
    spec->sample_rate = w.sample_rate;
    spec->codec = static_cast<int32_t>(w.codec);
    spec->data_offset = w.data_offset;
    spec->data_len = w.data_len;
    spec->n_samples = w.n_samples;
    spec->n_frames = w.n_frames;
}

std::vector<uint8_t> read_file(const char *path)
{
    std::FILE *f = std::fopen(path, "rb");
    // hound::WavReader::open -> io::Error -> err::Error::Io (err.rs:75)
    if (!f) throw Error{ErrorKind::Io, std::string("could not open ") + path};
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    const bool bad = std::ferror(f) != 0;
    std::fclose(f);
    if (bad) throw Error{ErrorKind::Io, std::string("read error on ") + path};
    return bytes;
}

int load_impl(const aptgpu_context *ctx, const uint8_t *bytes, size_t n, float **signal_out, size_t *n_out,
              uint32_t *rate, aptgpu_wav_spec *spec)
{
    const apt::WavInfo w = apt::parse_wav(bytes, n);
    fill_spec(spec, w);
    Scratch sc(ctx);
    apt::DeviceBuffer<uint8_t> d_raw;
    d_raw.alloc(w.data_len + 16);
    if (w.data_len)
        apt::hip_check(hipMemcpyAsync(d_raw.ptr, bytes + w.data_offset, w.data_len, hipMemcpyHostToDevice,
                                      sc.stream),
                       "hipMemcpyAsync H2D");
    apt::DeviceBuffer<float> d_sig;
    d_sig.alloc(w.n_frames + 16);
    apt::gpu::wav_to_signal(sc.stream, d_raw.ptr, w.n_frames, w.channels, w.bytes_per_sample,
                            static_cast<int>(w.codec), d_sig.ptr);
    *signal_out = sc.download_malloc(d_sig.ptr, w.n_frames);
    *n_out = w.n_frames;
    if (rate) *rate = w.sample_rate;
    return APTGPU_OK;
}

}  // namespace

extern "C" {

int aptgpu_wav_parse(const void *file_bytes, size_t n, aptgpu_wav_spec *spec, char *err, size_t err_cap)
{
    if ((!file_bytes && n) || !spec) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        fill_spec(spec, apt::parse_wav(static_cast<const uint8_t *>(file_bytes), n));
        return APTGPU_OK;
    });
}

int aptgpu_load_wav(const aptgpu_context *ctx, const void *file_bytes, size_t n, float **signal_out,
                    size_t *n_out, uint32_t *sample_rate_hz, aptgpu_wav_spec *spec, char *err, size_t err_cap)
{
    if ((!file_bytes && n) || !signal_out || !n_out) return APTGPU_ERR_INVALID;
    *signal_out = nullptr;
    *n_out = 0;
    return guarded(err, err_cap, [&] {
        return load_impl(ctx, static_cast<const uint8_t *>(file_bytes), n, signal_out, n_out, sample_rate_hz,
                         spec);
    });
}

int aptgpu_load_wav_file(const aptgpu_context *ctx, const char *path, float **signal_out, size_t *n_out,
                         uint32_t *sample_rate_hz, aptgpu_wav_spec *spec, char *err, size_t err_cap)
{
    if (!path || !signal_out || !n_out) return APTGPU_ERR_INVALID;
    *signal_out = nullptr;
    *n_out = 0;
    return guarded(err, err_cap, [&] {
        const std::vector<uint8_t> bytes = read_file(path);
        return load_impl(ctx, bytes.data(), bytes.size(), signal_out, n_out, sample_rate_hz, spec);
    });
}

int aptgpu_decode_wav(const aptgpu_context *ctx, const aptgpu_settings *settings, const void *file_bytes,
                      size_t n, int sync, float **rows_out, size_t *n_out, aptgpu_stats *stats,
                      uint32_t *sample_rate_hz, char *err, size_t err_cap)
{
    if (!settings || (!file_bytes && n) || !rows_out || !n_out) {
        put_err(err, err_cap, "null argument");
        return APTGPU_ERR_INVALID;
    }
    *rows_out = nullptr;
    *n_out = 0;
    apt::WavInfo w;
    const uint8_t *bytes = static_cast<const uint8_t *>(file_bytes);
    const int rc = guarded(err, err_cap, [&] {
        w = apt::parse_wav(bytes, n);
        return APTGPU_OK;
    });
    if (rc != APTGPU_OK) return rc;
    if (sample_rate_hz) *sample_rate_hz = w.sample_rate;
    return decode_host(ctx, settings, nullptr, bytes + w.data_offset, &w, w.n_frames, w.sample_rate, sync,
                       rows_out, n_out, stats, err, err_cap);
}

int aptgpu_plan_decode_device_wav(aptgpu_plan *plan, int count, const void *const *d_data,
                                  const aptgpu_wav_spec *specs, float *const *d_rows, const size_t *rows_cap,
                                  char *err, size_t err_cap)
{
    if (!plan || !d_data || !specs || !d_rows || !rows_cap || count < 0 || count > plan->max_batch) {
        put_err(err, err_cap, "null argument or count > max_batch");
        return APTGPU_ERR_INVALID;
    }
    return guarded(err, err_cap, [&] {
        for (int i = 0; i < count; ++i) {
            const aptgpu_wav_spec &sp = specs[i];
            if (sp.n_frames > plan->max_samples) throw Error{ErrorKind::Invalid, "recording longer than max_samples"};
            if (sp.sample_rate != plan->input_rate)
                throw Error{ErrorKind::Invalid, "WAV sample rate differs from the plan's input rate"};
            if (sp.codec < APTGPU_WAV_U8 || sp.codec > APTGPU_WAV_F32 || sp.channels == 0 ||
                sp.bytes_per_sample == 0 || sp.bytes_per_sample > 4)
                throw Error{ErrorKind::Invalid, "bad WAV spec"};
            if (!d_data[i] || !d_rows[i]) throw Error{ErrorKind::Invalid, "null device pointer"};
        }
        apt::hip_check(hipSetDevice(plan->device), "hipSetDevice");
        plan->begin_call(count);
        for (int i = 0; i < count; ++i) {
            aptgpu_plan::Input in;
            in.ptr = d_data[i];
            in.n = specs[i].n_frames;
            in.channels = specs[i].channels;
            in.bytes_per_sample = specs[i].bytes_per_sample;
            in.codec = specs[i].codec;
            plan->enqueue(i, in, d_rows[i], static_cast<uint64_t>(rows_cap[i]) * 2080u, false);
        }
        return APTGPU_OK;
    });
}

}  // extern "C"
