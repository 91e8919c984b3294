// apt_host.cpp — FIR design and unit helpers on the host (see apt_host.hpp).
#include "apt_host.hpp"

#include <cmath>
#include <limits>
#include <numeric>

namespace apt {

namespace {

// 1 / (n! * 2^n)^2, n = 0..8 — the first nine entries of BESSEL_TABLE
// (misc.rs:20-41); bessel_i0 only ever reads [1..=8].
constexpr float kBesselTable[9] = {
    1.0f,
    0.25f,
    0.015625f,
    0.00043402777777777775f,
    6.781684027777777e-06f,
    6.781684027777778e-08f,
    4.709502797067901e-10f,
    2.4028075495244395e-12f,
    9.385966990329842e-15f,
};

// Rust's saturating `f32 as i32`.
int32_t to_i32_saturating(float v)
{
    if (std::isnan(v)) return 0;
    if (v >= 2147483648.f) return std::numeric_limits<int32_t>::max();
    if (v <= -2147483648.f) return std::numeric_limits<int32_t>::min();
    return static_cast<int32_t>(v);
}

// Windowed-sinc body shared by Lowpass and LowpassDcRemoval: `tap(n)` gives the
// ideal response at integer offset n from the centre (filters.rs:76-83,117-127);
// the result is multiplied element-wise by the Kaiser window (product(), :186-196).
template <typename TapFn>
Signal windowed(float atten, Freq delta_w, TapFn tap)
{
    Signal taps = kaiser(atten, delta_w);
    const int32_t half = (static_cast<int32_t>(taps.size()) - 1) / 2;
    for (int32_t n = -half; n <= half; ++n) {
        float &w = taps[static_cast<size_t>(n + half)];
        const float ideal = tap(n);
        w = ideal * w;
    }
    return taps;
}

}  // namespace

// misc.rs:47-57 — Horner evaluation of the 8-term series in x^2.
float bessel_i0(float x)
{
    const float x2 = x * x;
    float acc = 0.f;
    for (int k = 8; k >= 1; --k) {
        acc += kBesselTable[k];
        acc *= x2;
    }
    return acc + 1.f;
}

// filters.rs:144-183
Signal kaiser(float atten, Freq delta_w)
{
    float beta;
    if (atten > 50.f) {
        beta = 0.1102f * (atten - 8.7f);
    } else if (atten < 21.f) {
        beta = 0.f;
    } else {
        beta = 0.5842f * powf(atten - 21.f, 0.4f) + 0.07886f * (atten - 21.f);
    }

    int32_t length = to_i32_saturating(ceilf((atten - 8.f) / (2.285f * delta_w.get_rad()))) + 1;
    if (length % 2 == 0) length += 1;

    Signal window;
    window.reserve(static_cast<size_t>(length > 0 ? length : 0));
    const int32_t half = (length - 1) / 2;
    const float m = static_cast<float>(length);
    const float i0_beta = bessel_i0(beta);
    for (int32_t k = -half; k <= half; ++k) {
        const float r = static_cast<float>(k) / (m / 2.f);
        window.push_back(bessel_i0(beta * sqrtf(1.f - r * r)) / i0_beta);
    }
    return window;
}

// filters.rs:57-88
Signal Lowpass::design() const
{
    const float c = cutout.get_pi_rad();
    return windowed(atten, delta_w, [c](int32_t n) -> float {
        if (n == 0) return c;
        const float nf = static_cast<float>(n);
        return sinf(nf * PI_F32 * c) / (nf * PI_F32);
    });
}

// filters.rs:90-94
void Lowpass::resample(Rate input_rate, Rate output_rate)
{
    const float ratio =
        static_cast<float>(output_rate.get_hz()) / static_cast<float>(input_rate.get_hz());
    cutout /= ratio;
    delta_w /= ratio;
}

// filters.rs:98-132 — band-pass as the difference of two sincs.
Signal LowpassDcRemoval::design() const
{
    const float c = cutout.get_pi_rad();
    const float d = (delta_w / 2.f).get_pi_rad();
    return windowed(atten, delta_w, [c, d](int32_t n) -> float {
        if (n == 0) return c - d;
        const float nf = static_cast<float>(n);
        return sinf(nf * PI_F32 * c) / (nf * PI_F32) - sinf(nf * PI_F32 * d) / (nf * PI_F32);
    });
}

// filters.rs:134-138
void LowpassDcRemoval::resample(Rate input_rate, Rate output_rate)
{
    const float ratio =
        static_cast<float>(output_rate.get_hz()) / static_cast<float>(input_rate.get_hz());
    cutout /= ratio;
    delta_w /= ratio;
}

std::unique_ptr<Filter> make_filter(int kind, float cutout_pi_rad, float atten,
                                    float delta_w_pi_rad)
{
    switch (kind) {
        case 0: return std::make_unique<NoFilter>();
        case 1:
            return std::make_unique<Lowpass>(Freq::pi_rad(cutout_pi_rad), atten,
                                             Freq::pi_rad(delta_w_pi_rad));
        case 2:
            return std::make_unique<LowpassDcRemoval>(Freq::pi_rad(cutout_pi_rad), atten,
                                                      Freq::pi_rad(delta_w_pi_rad));
        default: return nullptr;
    }
}

// decode.rs:171-199: 2pw x (-1), 7 x [2pw x (-1), 2pw x (+1)], 8pw x (-1).
bool generate_sync_frame(Rate work_rate, std::vector<int8_t> *out, std::string *msg)
{
    if (work_rate.get_hz() % FINAL_RATE != 0) {
        if (msg) *msg = "work_rate is not multiple of FINAL_RATE";
        return false;
    }
    const size_t pixel_width = work_rate.get_hz() / FINAL_RATE;
    const size_t pulse = 2 * pixel_width;
    out->clear();
    out->insert(out->end(), pulse, int8_t(-1));
    for (int rep = 0; rep < 7; ++rep) {
        out->insert(out->end(), pulse, int8_t(-1));
        out->insert(out->end(), pulse, int8_t(1));
    }
    out->insert(out->end(), 8 * pixel_width, int8_t(-1));
    return true;
}

LM interpolation_factors(Rate input_rate, Rate output_rate)
{
    const uint32_t g = std::gcd(input_rate.get_hz(), output_rate.get_hz());
    return LM{output_rate.get_hz() / g, input_rate.get_hz() / g};
}

uint64_t fast_resampling_len(uint64_t n, uint32_t l, uint32_t m, uint64_t ntaps)
{
    const uint64_t off = (ntaps - 1) / 2;
    const uint64_t total = n * l;
    if (total <= off) return 0;
    return (total - off + m - 1) / m;
}

// fast_resampling with context.export_resample_filtered set (dsp.rs:265-273): t walks the interpolated axis one
// by one from `off`, and the output keeps the sums whose t + 1 is a multiple of m, i.e. t = j*m - 1 for
// ceil((off + 1) / m) <= j <= n*l / m.
ExportGeom fast_resampling_export_geom(uint64_t n, uint32_t l, uint32_t m, uint64_t ntaps)
{
    ExportGeom g{0, 0, 0};
    const uint64_t off = (ntaps - 1) / 2;
    const uint64_t total = n * l;
    if (total <= off) return g;
    g.expanded = total - off;
    const uint64_t j0 = (off + m) / m;
    const uint64_t j1 = total / m;
    g.d0 = j0 * m - 1 - off;
    g.count = j1 >= j0 ? j1 - j0 + 1 : 0;
    return g;
}

}  // namespace apt
