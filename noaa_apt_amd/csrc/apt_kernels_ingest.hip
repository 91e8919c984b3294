// apt_kernels_ingest.hip — WAV sample ingest on the GPU: the data chunk's
// bytes -> the f32 Signal decode() takes, i.e. wav::load_wav's conversion loop (wav.rs:30-51):
// first channel only, integers converted with `as f32` (round to nearest even), never scaled.
// HBM-bound: reads bytes_per_sample*channels bytes and writes 4 per frame.
#include "apt_kernels.hpp"

namespace apt::gpu {

namespace {

constexpr int kThreads = 256;

// one sample at byte address p (any alignment), hound's Sample::read
__device__ inline float wav_sample(const uint8_t *p, int codec)
{
    switch (codec) {
    case 0:  // u8 - 128
        return static_cast<float>(static_cast<int>(p[0]) - 128);
    case 1:  // i16 LE
        return static_cast<float>(static_cast<int16_t>(static_cast<uint16_t>(p[0] | (p[1] << 8))));
    case 2:    // i24 LE
    case 3: {  // i24 in the low bytes of a 4-byte container
        uint32_t v = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
                     (static_cast<uint32_t>(p[2]) << 16);
        if (v & 0x800000u) v |= 0xff000000u;
        return static_cast<float>(static_cast<int32_t>(v));
    }
    case 4: {  // i32 LE; `as f32` rounds to nearest even
        const uint32_t v = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
                           (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24);
        return static_cast<float>(static_cast<int32_t>(v));
    }
    default: {  // f32 LE, bits preserved
        const uint32_t v = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
                           (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24);
        return __uint_as_float(v);
    }
    }
}

__global__ __launch_bounds__(kThreads) void k_wav_generic(const uint8_t *__restrict__ data, uint64_t n_frames,
                                                          uint32_t frame_bytes, int codec,
                                                          float *__restrict__ out)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n_frames; i += stride)
        out[i] = wav_sample(data + i * frame_bytes, codec);
}

// mono PCM16 at a 4-byte aligned address: 8 samples per thread, 16-byte loads when aligned
__global__ __launch_bounds__(kThreads) void k_wav_pcm16_mono(const uint32_t *__restrict__ pairs, uint64_t n_frames,
                                                             float *__restrict__ out)
{
    const uint64_t q = (static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x) * 8;
    if (q >= n_frames) return;
    if (q + 8 <= n_frames) {
        uint32_t w[4];
        for (int k = 0; k < 4; k++) w[k] = pairs[q / 2 + k];
        float v[8];
        for (int k = 0; k < 4; k++) {
            v[2 * k] = static_cast<float>(static_cast<int16_t>(w[k] & 0xffffu));
            v[2 * k + 1] = static_cast<float>(static_cast<int16_t>(w[k] >> 16));
        }
        for (int k = 0; k < 8; k++) out[q + k] = v[k];
        return;
    }
    const uint16_t *h = reinterpret_cast<const uint16_t *>(pairs);
    for (uint64_t i = q; i < n_frames; i++) out[i] = static_cast<float>(static_cast<int16_t>(h[i]));
}

// wav::write_wav's 16-bit branch (wav.rs:83-86): (sample / max * 32767.) as i16 — Rust's
// float -> int cast truncates toward zero, saturates, and maps NaN to 0.  max = limits[1].
__global__ __launch_bounds__(kThreads) void k_quantize_i16(const float *__restrict__ x, uint64_t n,
                                                           const float *limits, int16_t *__restrict__ out)
{
    const float mx = limits[1];
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        const float v = x[i] / mx * 32767.f;
        int q;
        if (v != v) q = 0;
        else if (v >= 32767.f) q = 32767;
        else if (v <= -32768.f) q = -32768;
        else q = static_cast<int>(v);  // truncation
        out[i] = static_cast<int16_t>(q);
    }
}

}  // namespace

void quantize_i16(hipStream_t s, const float *d_x, uint64_t n, const float *d_limits, int16_t *d_out)
{
    if (n == 0) return;
    uint64_t blocks = (n + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > (1u << 20)) blocks = 1u << 20;
    hipLaunchKernelGGL(k_quantize_i16, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, d_x, n, d_limits,
                       d_out);
}

void wav_to_signal(hipStream_t s, const void *d_data, uint64_t n_frames, uint32_t channels,
                   uint32_t bytes_per_sample, int codec, float *d_signal)
{
    if (n_frames == 0) return;
    if (codec == 1 && channels == 1 && (reinterpret_cast<uintptr_t>(d_data) & 3u) == 0) {
        const uint64_t blocks = (n_frames + kThreads * 8 - 1) / (kThreads * 8);
        hipLaunchKernelGGL(k_wav_pcm16_mono, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s,
                           static_cast<const uint32_t *>(d_data), n_frames, d_signal);
        return;
    }
    uint64_t blocks = (n_frames + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > (1u << 20)) blocks = 1u << 20;
    hipLaunchKernelGGL(k_wav_generic, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s,
                       static_cast<const uint8_t *>(d_data), n_frames, channels * bytes_per_sample, codec,
                       d_signal);
}

}  // namespace apt::gpu
