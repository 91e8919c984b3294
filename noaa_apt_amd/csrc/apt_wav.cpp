// apt_wav.cpp — see apt_wav.hpp.
#include "apt_wav.hpp"

#include <cstring>
#include <string>

namespace apt {

namespace {

[[noreturn]] void format_error(const char *reason)
{
    // hound::Error::FormatError's Display, mapped to err::Error::WavOpen (err.rs:76)
    throw Error{ErrorKind::WavOpen, std::string("Ill-formed WAVE file: ") + reason};
}

[[noreturn]] void unsupported()
{
    // hound::Error::Unsupported -> err::Error::WavOpen (err.rs:79)
    throw Error{ErrorKind::WavOpen, "The wave format of the file is not supported."};
}

// A forward-only byte cursor with hound's short-read error.
struct Cursor {
    const uint8_t *p;
    size_t n, at = 0;
    void need(size_t k) const
    {
        // io::ErrorKind::UnexpectedEof -> hound::Error::IoError -> err::Error::Io (err.rs:75)
        if (k > n - at) throw Error{ErrorKind::Io, "Failed to read enough bytes."};
    }
    uint8_t u8()
    {
        need(1);
        return p[at++];
    }
    uint16_t u16()
    {
        need(2);
        const uint16_t v = static_cast<uint16_t>(p[at] | (p[at + 1] << 8));
        at += 2;
        return v;
    }
    uint32_t u32()
    {
        need(4);
        const uint32_t v = static_cast<uint32_t>(p[at]) | (static_cast<uint32_t>(p[at + 1]) << 8) |
                           (static_cast<uint32_t>(p[at + 2]) << 16) | (static_cast<uint32_t>(p[at + 3]) << 24);
        at += 4;
        return v;
    }
    void tag(char out[4])
    {
        need(4);
        std::memcpy(out, p + at, 4);
        at += 4;
    }
    void skip(size_t k)
    {
        need(k);
        at += k;
    }
};

const uint8_t kGuidTail[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71};

// hound's read_fmt_chunk and its three per-format tails
void read_fmt(Cursor &c, uint32_t chunk_len, WavInfo &w)
{
    if (chunk_len < 16) format_error("invalid fmt chunk size");
    const uint16_t format_tag = c.u16();
    const uint16_t n_channels = c.u16();
    const uint32_t n_samples_per_sec = c.u32();
    const uint32_t n_bytes_per_sec = c.u32();
    const uint16_t block_align = c.u16();
    const uint16_t bits_per_sample = c.u16();
    if (n_channels == 0) format_error("file contains zero channels");
    const uint16_t bytes_per_sample = static_cast<uint16_t>(block_align / n_channels);
    // `Some(bits) > bytes.checked_mul(8)`: an overflowing product is None, and Some(x) > None
    const uint32_t container_bits = static_cast<uint32_t>(bytes_per_sample) * 8u;
    if (container_bits > 0xFFFFu || bits_per_sample > container_bits) format_error("sample bits exceeds size of sample");
    const uint64_t expect_bps = static_cast<uint64_t>(block_align) * n_samples_per_sec;
    if (expect_bps > 0xFFFFFFFFull || n_bytes_per_sec != expect_bps) format_error("inconsistent fmt chunk");
    if (bits_per_sample % 8 != 0) format_error("bits per sample is not a multiple of 8");
    if (bits_per_sample == 0) format_error("bits per sample is 0");
    w.channels = n_channels;
    w.sample_rate = n_samples_per_sec;
    w.bits_per_sample = bits_per_sample;
    w.bytes_per_sample = bytes_per_sample;
    w.is_float = false;
    switch (format_tag) {
    case 0x0001: {  // WAVE_FORMAT_PCM
        bool ex;
        if (chunk_len == 16) ex = false;
        else if (chunk_len == 18 || chunk_len == 40) ex = true;
        else format_error("unexpected fmt chunk size");
        if (ex) {
            (void)c.u16();  // cbSize, ignored for PCM
            if (bits_per_sample != 8 && bits_per_sample != 16 && bits_per_sample != 24) unsupported();
        }
        if (chunk_len == 40) c.skip(22);
        break;
    }
    case 0x0002: unsupported();  // ADPCM
    case 0x0003: {               // WAVE_FORMAT_IEEE_FLOAT
        const bool ex = chunk_len == 18;
        if (!ex && chunk_len != 16) format_error("unexpected fmt chunk size");
        if (ex && c.u16() != 0) format_error("unexpected WAVEFORMATEX size");
        if (bits_per_sample != 32) format_error("bits per sample is not 32");
        w.is_float = true;
        break;
    }
    case 0xfffe: {  // WAVE_FORMAT_EXTENSIBLE
        if (chunk_len < 40) format_error("unexpected fmt chunk size");
        if (c.u16() != 22) format_error("unexpected WAVEFORMATEXTENSIBLE size");
        const uint16_t valid_bits = c.u16();
        (void)c.u32();  // channel mask
        c.need(16);
        const uint8_t *guid = c.p + c.at;
        c.at += 16;
        if (std::memcmp(guid + 2, kGuidTail, 14) != 0 || guid[1] != 0) unsupported();
        if (guid[0] == 0x01) w.is_float = false;       // KSDATAFORMAT_SUBTYPE_PCM
        else if (guid[0] == 0x03) w.is_float = true;   // KSDATAFORMAT_SUBTYPE_IEEE_FLOAT
        else unsupported();
        if (valid_bits > 0) w.bits_per_sample = valid_bits;
        break;
    }
    default: unsupported();
    }
}

// hound's Sample::read for i32 (Int files) / f32 (Float files), which wav.rs:30-51 selects by
// spec.sample_format
WavCodec pick_codec(const WavInfo &w)
{
    const uint16_t bytes = w.bytes_per_sample, bits = w.bits_per_sample;
    if (w.is_float) {
        if (bytes == 4 && bits == 32) return WavCodec::F32;
        if (bytes > 4) throw Error{ErrorKind::Internal, "The sample has more bits than the destination type."};
        unsupported();
    }
    if (bytes == 1 && bits == 8) return WavCodec::U8;
    if (bytes == 2 && bits == 16) return WavCodec::I16;
    if (bytes == 3 && bits == 24) return WavCodec::I24;
    if (bytes == 4 && bits == 24) return WavCodec::I24_4;
    if (bytes == 4 && bits == 32) return WavCodec::I32;
    if (bytes > 4) throw Error{ErrorKind::Internal, "The sample has more bits than the destination type."};
    unsupported();
}

}  // namespace

WavInfo parse_wav(const uint8_t *bytes, size_t n)
{
    Cursor c{bytes, n};
    char t[4];
    c.tag(t);
    if (std::memcmp(t, "RIFF", 4) != 0) format_error("no RIFF tag found");
    (void)c.u32();  // file length, not trusted
    c.tag(t);
    if (std::memcmp(t, "WAVE", 4) != 0) format_error("no WAVE tag found");
    WavInfo w;
    bool have_fmt = false;
    for (;;) {  // read_until_data
        c.tag(t);
        const uint32_t len = c.u32();
        if (std::memcmp(t, "fmt ", 4) == 0) {
            read_fmt(c, len, w);
            have_fmt = true;
        } else if (std::memcmp(t, "fact", 4) == 0) {
            // hound reads the one u32 and ignores a failure to do so
            if (n - c.at >= 4) c.at += 4; else c.at = n;
        } else if (std::memcmp(t, "data", 4) == 0) {
            if (!have_fmt) format_error("missing fmt chunk");
            w.data_offset = c.at;
            w.data_len = len;
            break;
        } else {
            c.skip(len);  // unknown chunk (no pad-byte handling in hound 3.x)
        }
    }
    const uint32_t len32 = static_cast<uint32_t>(w.data_len);
    const uint32_t num_samples = len32 / w.bytes_per_sample;
    if (num_samples * static_cast<uint32_t>(w.bytes_per_sample) != len32)
        format_error("data chunk length is not a multiple of sample size");
    if (num_samples % w.channels != 0) format_error("invalid data chunk length");
    w.n_samples = num_samples;
    w.n_frames = num_samples / w.channels;
    // the first samples::<T>().next() raises these; an empty data chunk never gets there
    if (w.n_samples) w.codec = pick_codec(w);
    // wav.rs:33,43 collect every sample: a file shorter than its data chunk fails as a whole
    if (w.data_len > n - w.data_offset) throw Error{ErrorKind::Io, "Failed to read enough bytes."};
    return w;
}

}  // namespace apt
