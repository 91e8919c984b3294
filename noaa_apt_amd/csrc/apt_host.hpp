// apt_host.hpp — host-side (CPU, once per plan) pieces of the decode() path:
// unit types, Kaiser-windowed FIR design, L/M factors, the sync template.
//
// These mirror the reference's Rust interfaces by name so the C ABI and the
// tests read like the reference:
//   Freq / Rate              /root/reference/src/frequency.rs:30-117
//   Filter, Lowpass, LowpassDcRemoval, NoFilter, kaiser
//                            /root/reference/src/filters.rs:10-196
//   bessel_i0                /root/reference/src/misc.rs:16-57
//   generate_sync_frame      /root/reference/src/decode.rs:171-199
// All arithmetic is f32 with every operation rounded separately (the library is
// built with -ffp-contract=off) and the transcendental calls go to the platform
// libm (sinf/cosf/powf), which is what rustc's f32::{sin,cos,powf} lower to.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace apt {

using Signal = std::vector<float>;  // dsp::Signal, dsp.rs:16

constexpr uint32_t FINAL_RATE = 4160;   // decode.rs:14
constexpr uint32_t PX_PER_ROW = 2080;   // decode.rs:35
constexpr uint32_t CARRIER_FREQ = 2400; // decode.rs:38
constexpr float PI_F32 = 3.14159265358979323846f;

enum class ErrorKind { Internal = 1, RateOverflow = 2, Hip = 3, Invalid = 4, Unsupported = 5, WavOpen = 6, Io = 7 };

// err::Error restricted to the variants the path can produce (err.rs:9-44).
struct Error {
    ErrorKind kind;
    std::string message;
    int hip_code = 0;  // ErrorKind::Hip: the hipError_t behind it (0 where the error is not a runtime status)
};

// Sample rate in Hz (frequency.rs:98-117).
struct Rate {
    uint32_t hz_;
    static Rate hz(uint32_t r) { return Rate{r}; }
    uint32_t get_hz() const { return hz_; }
    // checked_mul: false on u32 overflow
    bool checked_mul(uint32_t other, Rate *out) const
    {
        uint64_t p = static_cast<uint64_t>(hz_) * other;
        if (p > 0xFFFFFFFFull) return false;
        *out = Rate{static_cast<uint32_t>(p)};
        return true;
    }
};

// Discrete-time frequency stored as a fraction of pi rad/sample (frequency.rs:30-88).
struct Freq {
    float pi_rad_;
    static Freq rad(float f) { return Freq{f / PI_F32}; }
    static Freq pi_rad(float f) { return Freq{f}; }
    static Freq hz(float f, Rate rate) { return Freq{2.f * f / static_cast<float>(rate.get_hz())}; }
    float get_rad() const { return pi_rad_ * PI_F32; }
    float get_pi_rad() const { return pi_rad_; }
    float get_hz(Rate rate) const { return pi_rad_ * static_cast<float>(rate.get_hz()) / 2.f; }
    Freq operator/(float o) const { return Freq{pi_rad_ / o}; }
    Freq &operator/=(float o)
    {
        pi_rad_ /= o;
        return *this;
    }
};

float bessel_i0(float x);
Signal kaiser(float atten, Freq delta_w);

// trait Filter (filters.rs:10-16)
struct Filter {
    virtual ~Filter() = default;
    virtual Signal design() const = 0;
    virtual void resample(Rate input_rate, Rate output_rate) = 0;
};

struct NoFilter final : Filter {
    Signal design() const override { return Signal{1.f}; }
    void resample(Rate, Rate) override {}
};

struct Lowpass final : Filter {
    Freq cutout;
    float atten;
    Freq delta_w;
    Lowpass(Freq c, float a, Freq d) : cutout(c), atten(a), delta_w(d) {}
    Signal design() const override;
    void resample(Rate input_rate, Rate output_rate) override;
};

struct LowpassDcRemoval final : Filter {
    Freq cutout;
    float atten;
    Freq delta_w;
    LowpassDcRemoval(Freq c, float a, Freq d) : cutout(c), atten(a), delta_w(d) {}
    Signal design() const override;
    void resample(Rate input_rate, Rate output_rate) override;
};

// Builds the Filter a C-ABI aptgpu_filter describes.
std::unique_ptr<Filter> make_filter(int kind, float cutout_pi_rad, float atten,
                                    float delta_w_pi_rad);

// ±1 sync-A template; false + message when work_rate is not a multiple of 4160.
bool generate_sync_frame(Rate work_rate, std::vector<int8_t> *out, std::string *msg);

// gcd → (l, m) of resample_with_filter (dsp.rs:73-75).
struct LM {
    uint32_t l, m;
};
LM interpolation_factors(Rate input_rate, Rate output_rate);

// Number of outputs fast_resampling produces (dsp.rs:226-277):
// t = off, off+m, ... while t < n*l.
uint64_t fast_resampling_len(uint64_t n, uint32_t l, uint32_t m, uint64_t ntaps);
// The other branch of the same loop (context.export_resample_filtered, dsp.rs:265-273): the output is taken at
// t = off + d0 + k*m, k < count, and `expanded` sums (every t in [off, n*l)) go to the "resample_filtered" step.
struct ExportGeom {
    uint64_t d0, count, expanded;
};
ExportGeom fast_resampling_export_geom(uint64_t n, uint32_t l, uint32_t m, uint64_t ntaps);

}  // namespace apt
