// apt_capi_wav.hip — extern "C" surface of include/aptgpu.h §5: WAV ingest in front of decode().
#include <fcntl.h>
#include <sys/stat.h>

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

// wav::write_wav (wav.rs:59-98) for the 16-bit Int spec on a signal in HBM: the file image hound's
// WavWriter produces for {channels: 1, bits_per_sample: 16, Int} — a 16-byte PCMWAVEFORMAT fmt
// chunk, i.e. the canonical 44-byte header — followed by the normalised samples.
std::vector<uint8_t> write_wav_i16_device(Scratch &sc, const float *d_signal, uint64_t n, uint32_t rate)
{
    if (n == 0) throw Error{ErrorKind::Internal, "Can't get maximum of a zero length vector"};  // wav.rs:70
    if (2 * n + 36 > 0xFFFFFFFFull) throw Error{ErrorKind::Unsupported, "WAV data chunk would exceed 4 GiB"};
    apt::DeviceBuffer<char> ws;
    ws.alloc(apt::gpu::image_ws_bytes(n));
    apt::DeviceBuffer<apt::gpu::ImageResult> d_info;
    d_info.alloc(1);
    apt::gpu::image_begin(sc.stream, d_info.ptr);
    apt::gpu::image_minmax(sc.stream, d_signal, nullptr, n, n, ws.ptr, d_info.ptr);  // dsp::get_max
    apt::DeviceBuffer<int16_t> d_q;
    d_q.alloc(n + 16);
    apt::gpu::quantize_i16(sc.stream, d_signal, n, apt::gpu::image_ws_pointers(ws.ptr, n).limits, d_q.ptr);
    std::vector<uint8_t> file(44 + 2 * n);
    auto put32 = [&](size_t at, uint32_t v) {
        for (int k = 0; k < 4; ++k) file[at + k] = static_cast<uint8_t>(v >> (8 * k));
    };
    auto put16 = [&](size_t at, uint16_t v) {
        file[at] = static_cast<uint8_t>(v);
        file[at + 1] = static_cast<uint8_t>(v >> 8);
    };
    std::memcpy(&file[0], "RIFF", 4);
    put32(4, static_cast<uint32_t>(36 + 2 * n));
    std::memcpy(&file[8], "WAVEfmt ", 8);
    put32(16, 16);
    put16(20, 1);  // WAVE_FORMAT_PCM
    put16(22, 1);  // channels
    put32(24, rate);
    put32(28, rate * 2);  // bytes per second
    put16(32, 2);         // block align
    put16(34, 16);        // bits per sample
    std::memcpy(&file[36], "data", 4);
    put32(40, static_cast<uint32_t>(2 * n));
    apt::hip_check(hipMemcpyAsync(&file[44], d_q.ptr, 2 * n, hipMemcpyDeviceToHost, sc.stream), "hipMemcpyAsync D2H");
    apt::hip_check(hipStreamSynchronize(sc.stream), "hipStreamSynchronize");
    return file;
}

void *to_malloc(const std::vector<uint8_t> &v)
{
    uint8_t *p = host_alloc<uint8_t>(v.size());
    std::memcpy(p, v.data(), v.size());
    return p;
}

// resample::resample (resample.rs:17-71) between file images
std::vector<uint8_t> resample_wav_impl(const aptgpu_context *ctx, const uint8_t *bytes, size_t n,
                                       uint32_t output_rate, float atten, float delta_w_pi_rad,
                                       const char *output_name, bool reading_announced, bool export_filtered)
{
    if (!reading_announced) status(ctx, 0.0f, "Reading WAV file");  // resample.rs:25
    const apt::WavInfo w = apt::parse_wav(bytes, n);
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
    if (ctx && ctx->step) {  // resample.rs:31
        float *x = sc.download_malloc(d_sig.ptr, w.n_frames);
        const int rc = ctx->step("input", 0, x, w.n_frames, w.sample_rate, ctx->user);
        std::free(x);
        if (rc != 0) throw Error{ErrorKind::Internal, "step callback failed at \"input\""};
    }
    status(ctx, 0.2f, "Resampling to " + std::to_string(output_rate));  // resample.rs:34
    if (w.sample_rate == 0) throw Error{ErrorKind::Invalid, "input_rate is 0"};
    // dsp::resample, dsp.rs:132-162
    const apt::Rate in_rate = apt::Rate::hz(w.sample_rate);
    const apt::Freq cutout = output_rate > w.sample_rate
                                 ? apt::Freq::hz(static_cast<float>(w.sample_rate) / 2.f, in_rate)
                                 : apt::Freq::hz(static_cast<float>(output_rate) / 2.f, in_rate);
    apt::Lowpass f(cutout, atten, apt::Freq::pi_rad(delta_w_pi_rad));
    apt::DeviceBuffer<float> d_y;
    const uint64_t n_res = resample_device(sc, d_sig.ptr, w.n_frames, w.sample_rate, output_rate, f, d_y, export_filtered, ctx);
    if (n_res == 0)  // resample.rs:45-51
        throw Error{ErrorKind::Internal,
                    "Got zero samples after resampling, audio file too short or output sampling frequency too low"};
    status(ctx, 0.8f, std::string("Writing WAV to '") + (output_name ? output_name : "") + "'");  // resample.rs:61
    std::vector<uint8_t> out = write_wav_i16_device(sc, d_y.ptr, n_res, output_rate);
    return out;
}

}  // namespace

extern "C" {

int aptgpu_write_wav_i16(const aptgpu_context *ctx, const float *signal, size_t n, uint32_t sample_rate_hz,
                         void **wav_out, size_t *n_out, char *err, size_t err_cap)
{
    if ((!signal && n) || !wav_out || !n_out) return APTGPU_ERR_INVALID;
    *wav_out = nullptr;
    *n_out = 0;
    return guarded(err, err_cap, [&] {
        Scratch sc(ctx);
        auto d_x = sc.upload(signal, n);
        const std::vector<uint8_t> file = write_wav_i16_device(sc, d_x.ptr, n, sample_rate_hz);
        *wav_out = to_malloc(file);
        *n_out = file.size();
        return APTGPU_OK;
    });
}

int aptgpu_resample_wav(const aptgpu_context *ctx, const void *file_bytes, size_t n, uint32_t output_rate_hz,
                        float atten, float delta_w_pi_rad, const char *output_name, void **wav_out,
                        size_t *n_out, char *err, size_t err_cap)
{
    return aptgpu_resample_wav_ex(ctx, file_bytes, n, output_rate_hz, atten, delta_w_pi_rad, 0, output_name, wav_out,
                                  n_out, err, err_cap);
}

int aptgpu_resample_wav_ex(const aptgpu_context *ctx, const void *file_bytes, size_t n, uint32_t output_rate_hz,
                           float atten, float delta_w_pi_rad, int export_resample_filtered, const char *output_name,
                           void **wav_out, size_t *n_out, char *err, size_t err_cap)
{
    if ((!file_bytes && n) || !wav_out || !n_out) return APTGPU_ERR_INVALID;
    *wav_out = nullptr;
    *n_out = 0;
    return guarded(err, err_cap, [&] {
        const std::vector<uint8_t> file = resample_wav_impl(ctx, static_cast<const uint8_t *>(file_bytes), n,
                                                            output_rate_hz, atten, delta_w_pi_rad, output_name, false,
                                                            export_resample_filtered != 0);
        *wav_out = to_malloc(file);
        *n_out = file.size();
        status(ctx, 1.f, "Finished");  // resample.rs:69
        return APTGPU_OK;
    });
}

int aptgpu_resample_wav_file(const aptgpu_context *ctx, const char *input_path, const char *output_path,
                             uint32_t output_rate_hz, float atten, float delta_w_pi_rad, char *err,
                             size_t err_cap)
{
    return aptgpu_resample_wav_file_ex(ctx, input_path, output_path, output_rate_hz, atten, delta_w_pi_rad, 0, err, err_cap);
}

int aptgpu_resample_wav_file_ex(const aptgpu_context *ctx, const char *input_path, const char *output_path,
                                uint32_t output_rate_hz, float atten, float delta_w_pi_rad,
                                int export_resample_filtered, char *err, size_t err_cap)
{
    if (!input_path || !output_path) return APTGPU_ERR_INVALID;
    return guarded(err, err_cap, [&] {
        status(ctx, 0.0f, "Reading WAV file");
        const std::vector<uint8_t> in = read_file(input_path);
        struct stat st {};
        if (::stat(input_path, &st) != 0)  // misc::read_timestamp, misc.rs:181-194
            throw Error{ErrorKind::Internal, "Could not read metadata from input file: stat failed"};
        const std::vector<uint8_t> out = resample_wav_impl(ctx, in.data(), in.size(), output_rate_hz, atten,
                                                           delta_w_pi_rad, output_path, true, export_resample_filtered != 0);
        std::FILE *f = std::fopen(output_path, "wb");
        if (!f) throw Error{ErrorKind::Io, std::string("could not create ") + output_path};
        const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
        if (std::fclose(f) != 0 || !ok) throw Error{ErrorKind::Io, std::string("write error on ") + output_path};
        // misc::write_timestamp, misc.rs:200-205: modification time, whole seconds
        struct timespec ts[2];
        ts[0].tv_sec = st.st_atime;
        ts[0].tv_nsec = 0;
        ts[1].tv_sec = st.st_mtime;
        ts[1].tv_nsec = 0;
        if (::utimensat(AT_FDCWD, output_path, ts, 0) != 0)
            throw Error{ErrorKind::Internal, "Could not write timestamp to file"};
        status(ctx, 1.f, "Finished");  // resample.rs:69
        return APTGPU_OK;
    });
}

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
        std::vector<aptgpu_plan::Input> ins(static_cast<size_t>(count));
        std::vector<uint64_t> caps(static_cast<size_t>(count));
        for (int i = 0; i < count; ++i) {
            aptgpu_plan::Input &in = ins[static_cast<size_t>(i)];
            in.ptr = d_data[i];
            in.n = specs[i].n_frames;
            in.channels = specs[i].channels;
            in.bytes_per_sample = specs[i].bytes_per_sample;
            in.codec = specs[i].codec;
            caps[static_cast<size_t>(i)] = static_cast<uint64_t>(rows_cap[i]) * 2080u;
        }
        plan->run_call(count, ins.data(), d_rows, caps.data(), false);
        return APTGPU_OK;
    });
}

}  // extern "C"
