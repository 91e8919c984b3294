// apt_kernels_fused_any_launch.hpp — interface between the dispatcher (apt_kernels_fused_any.hip) and
// the per-shape translation units.
#pragma once

#include "apt_kernels.hpp"

namespace apt::gpu {

struct AnyGeom {
    uint32_t l, m, jlim;   // resampler: factors, taps the reference uses (2*off + 1)
    uint32_t tpp;          // row stride of the phase-major table (>= taps per phase, odd)
    uint32_t t2;           // low-pass taps
    uint32_t g, pulse;     // sync frame length 38*pw and pulse width 2*pw
    uint32_t kt, pre, own; // tile geometry (work samples)
    uint32_t xt;           // input tile, floats (multiple of 4)
    uint32_t off_x, off_a, off_b;  // LDS offsets in floats (table at 0)
    uint32_t step_q, step_r;       // (NTHR*m) / l and % l: x0 / phase update between a thread's outputs
    uint32_t jl_a, jl_b;           // jlim / l and % l: taps of phase p = jl_a + (p < jl_b)
    uint32_t table_in_global;      // the polyphase table does not fit LDS beside the tile: stage 1 reads its rows from HBM / L2
    uint64_t sign[4];              // bit j set <=> sync template[j] = +1 (decode.rs:188-198)
    float gm_slack_scale;          // APTGPU_GM_SLACK_SCALE as the plan read it (LaunchSwitches), host side only
};

#define APT_ANY_SHAPE_ARGS                                                                                          \
    hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, bool pcm16, const float *table,   \
        const float *h2, const float *h2p, float cosphi2, float sinphi, float inv_sinphi, bool want_gm,             \
        const AnyGeom &g, size_t lds, int prof
void fused_any_launch_256x8(APT_ANY_SHAPE_ARGS);
void fused_any_launch_1024x8(APT_ANY_SHAPE_ARGS);
void fused_any_launch_1024x4(APT_ANY_SHAPE_ARGS);
void fused_any_launch_256x4(APT_ANY_SHAPE_ARGS);

}  // namespace apt::gpu
