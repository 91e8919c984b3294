// apt_wav.hpp — WAV container parsing for the ingest step in front of decode()
// (SURVEY.md §8(f) N1): what wav::load_wav (wav.rs:11-57) gets from hound 3.5.1's WavReader,
// restated from hound's published source (the crate is not vendored in the reference tree:
// Cargo.toml:29, Cargo.lock:811).  Host-side header walk only; the sample conversion runs on
// the GPU (apt_kernels_ingest.hip).
#pragma once

#include <cstddef>
#include <cstdint>

#include "apt_host.hpp"

namespace apt {

// How one sample is stored; selects the device conversion (hound's Sample::read for i32 / f32).
enum class WavCodec : int32_t {
    U8 = 0,     // (1 byte, 8 bit):  unsigned, minus 128
    I16 = 1,    // (2, 16)
    I24 = 2,    // (3, 24)
    I24_4 = 3,  // (4, 24): low three bytes of a 4-byte container, sign-extended
    I32 = 4,    // (4, 32)
    F32 = 5,    // IEEE float, 32 bit
};

struct WavInfo {
    uint16_t channels = 0;
    uint16_t bits_per_sample = 0;
    uint16_t bytes_per_sample = 0;  // block_align / channels
    uint32_t sample_rate = 0;
    bool is_float = false;          // hound::SampleFormat::Float
    WavCodec codec = WavCodec::I16;
    uint64_t data_offset = 0;       // first byte of the data chunk's payload
    uint64_t data_len = 0;          // bytes, from the chunk header
    uint64_t n_samples = 0;         // all channels (hound's WavReader::len)
    uint64_t n_frames = 0;          // n_samples / channels == length of the Signal load_wav returns
};

// Walks RIFF/WAVE chunks up to the data chunk.  Throws apt::Error with the kinds the
// reference's From<hound::Error> (err.rs:72-83) produces: WavOpen for FormatError/Unsupported,
// Io for a short read, Internal for TooWide / InvalidSampleFormat.  `n` is the number of bytes
// available (the whole file); a data chunk reaching past it is the short-read error the
// reference hits while collecting samples.
WavInfo parse_wav(const uint8_t *bytes, size_t n);

}  // namespace apt
