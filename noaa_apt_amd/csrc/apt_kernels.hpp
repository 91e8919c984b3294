// apt_kernels.hpp — launch wrappers of the gfx950 kernels (definitions in
// apt_kernels_generic.hip and apt_kernels_fused.hip).  All launches are
// asynchronous on the given stream and never synchronise with the host.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace apt::gpu {

// Device-side result record == aptgpu_result (include/aptgpu.h).
struct Result {
    int32_t status;
    int32_t reason;
    uint32_t n_rows;
    uint32_t n_sync;
    uint64_t work_len;
    uint64_t n_out;
};

// Device-side record of the image stage == aptgpu_image_result (include/aptgpu.h).
struct ImageResult {
    int32_t status;   // 0 ok, 1 Internal
    int32_t reason;   // 1 zero-length signal, 2 too short for telemetry, 3 no low bucket, 4 decode failed
    uint32_t height;  // rows of 2080 px
    uint32_t telemetry_row;
    float low, high;
    float telemetry_quality;
    int32_t channel_a, channel_b;  // index into the reference's channel-name table, -1 = none
    uint32_t reserved;
    uint64_t n_px;
    float values_a[16], values_b[16];
};

// ---- generic kernels (any l, m, tap count) --------------------------------------
// fast_resampling, dsp.rs:186-289: out[k], k < w
void resample_generic(hipStream_t s, const float *x, uint64_t n, const float *coeff,
                      uint32_t ntaps, uint32_t l, uint32_t m, float *out, uint64_t w);
// fp16-tap variant (APTGPU_MODE_FP16_TAPS, BASELINE config 5): tolerance-based, not bit-exact
uint32_t f16taps_pairs_per_phase(uint32_t l, uint32_t ntaps);
float f16taps_pack(uint32_t l, const float *coeff, uint32_t ntaps, uint16_t *table);  // returns 2^-s
void resample_f16taps(hipStream_t s, const float *x, uint64_t n, const uint16_t *table, uint32_t ntaps,
                      uint32_t l, uint32_t m, float unscale, float *out, uint64_t w);
// filter() followed by decimate(m), dsp.rs:386-410 + 294-307: out[k] = filter(x)[k*m]
void fir_decimate(hipStream_t s, const float *x, uint64_t n, const float *coeff, uint32_t ntaps,
                  uint32_t m, float *out, uint64_t n_out);
// demodulate, dsp.rs:350-383
void demodulate(hipStream_t s, const float *x, uint64_t n, float cosphi2, float sinphi, float *out);
// the cross-correlation of find_sync, decode.rs:225-233 (pw = work_rate / 4160)
void correlate(hipStream_t s, const float *f, uint64_t n_corr, uint32_t pw, float *corr);
// terminal flags of the peak picker: bit i of `bits` set <=> no corr[j] > corr[i] for
// j in (i, i+md]; corr[0] is clamped to >= 0 (the initial (0, 0.) peak, decode.rs:208).
// md must be a multiple of 64; needs 2*md*4 bytes of LDS.
void terminals(hipStream_t s, const float *corr, uint64_t n_corr, uint32_t md, uint64_t *bits);
// the orbit of the peak picker over the terminal bitmask (one wave): writes the peak list
// (find_sync's return value) and fills the result record.
void orbit_walk(hipStream_t s, const uint64_t *bits, uint64_t n_corr, uint64_t work_len,
                uint32_t spr, uint32_t md, uint32_t *peaks, uint32_t peaks_cap, Result *res);
// row gather (decode.rs:120-134) taking every pw-th sample; raw = plain copy (the
// "sync_result" step), !raw = through the final NoFilter stage (decode.rs:158-159)
void gather_rows(hipStream_t s, const float *f, const uint32_t *peaks, const Result *res,
                 uint32_t spr, uint32_t pw, bool raw, float *rows, uint32_t rows_cap);

// ---- fused specialised front end (apt_kernels_fused.hip) --------------------------
// true when a <L, M, T1, T2, PW> specialisation exists
bool fused_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
uint32_t fused_group_size(uint32_t l);
// host: stage-1 tap-pair table [WIN][PS][2] (see apt_kernels_fused.hip) and its size in floats
uint32_t fused_tap_table_floats(uint32_t l, uint32_t m, uint32_t t1);
void fused_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, float *hs);
// host: stage-3 tap pairs h2p[k] = (h2[k-1], h2[k]), k = 0 .. t2  (2*(t2+1) floats)
void fused_lowpass_pairs(const float *h2, uint32_t t2, float *h2p);
// One recording of a batched front-end launch (device-resident array, blockIdx.y selects).
struct FusedRec {
    const void *x;
    uint64_t n;
    float *f_out, *c_out, *gm_out;
    uint64_t w, n_corr;
};
// fp16-tap stage 1 (APTGPU_MODE_FP16_TAPS): table size in dwords, host-side table builder (returns
// the power-of-two unscale factor), availability
uint32_t fused_f16_table_dwords(uint32_t l, uint32_t m, uint32_t t1);
float fused_f16_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, uint32_t *table);
bool fused_f16_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
// x -> F (filtered work-rate signal) and, if gm_out != nullptr, the per-group maxima of
// the sync cross-correlation.  Returns false if no specialisation matches.
// x is the f32 Signal, or (pcm16) mono int16 samples at a 4-byte aligned address.
bool fused_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw,
                     const void *x, bool pcm16, uint64_t n, const float *hs, const float *h2, const float *h2p,
                     float cosphi2, float sinphi, float inv_sinphi /* verified RN(1/sinphi) or 0, apt_envelope.hpp */,
                     float f16_unscale /* 0: strict; else hs is the fp16 table and this its 2^-s */, float *f_out, float *c_out, float *gm_out, uint64_t w, uint64_t n_corr);

// Batched form: ONE launch over `count` recordings described by d_batch (in HBM, written before the
// launch on the same stream); grid.x covers the longest recording (max_w work samples).
bool fused_batch_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
bool fused_front_end_batch(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw,
                           bool pcm16, const FusedRec *d_batch, int count, uint64_t max_w, const float *hs,
                           const float *h2, const float *h2p, float cosphi2, float sinphi, float inv_sinphi,
                           float f16_unscale);

// ---- fused front end for any rate / profile (apt_kernels_fused_any.hip) -------------
// run-time parameters, taps phase-major in LDS; same outputs as fused_front_end
bool fused_any_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
uint32_t fused_any_table_floats(uint32_t l, uint32_t t1);
void fused_any_table(uint32_t l, const float *coeff, uint32_t t1, float *table);  // host
bool fused_any_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw,
                         const void *x, bool pcm16, uint64_t n, const float *table, const float *h2,
                         const float *h2p /* fused_lowpass_pairs */, float cosphi2, float sinphi, float inv_sinphi, float *f_out,
                         float *c_out, float *gm_out, uint64_t w, uint64_t n_corr);

// ---- parallel peak picker (apt_kernels_sync.hip) ------------------------------------
uint32_t sync_group_size();    // correlation positions per group (52)
uint32_t sync_chunk_groups();  // groups per k_sync_nodes workgroup
uint32_t sync_slot_cap();      // node terminals kept per chunk
// corr -> per-group maxima (unfused path; the fused front end writes them itself)
void group_max(hipStream_t s, const float *corr, uint64_t n_corr, float *gm);
// coarse/fine terminal detection -> terminal words + ordered node-terminal lists
void sync_nodes(hipStream_t s, const float *gm, const float *corr, uint64_t n_corr, uint32_t spr,
                uint32_t md, uint64_t *words, uint32_t *slot_nt, uint32_t *slot_cnt, uint32_t *flags);
// orbit of the picker (LDS pointer doubling, or the sequential walk as fallback)
size_t sync_orbit_ws_words(uint64_t n_corr, uint32_t spr);  // uint32 words of scratch it needs
void sync_orbit(hipStream_t s, const uint64_t *words, const uint32_t *slot_nt,
                const uint32_t *slot_cnt, uint32_t *flags, uint64_t n_corr, uint64_t work_len,
                uint32_t spr, uint32_t md, uint32_t *ws, uint32_t *peaks, uint32_t peaks_cap,
                Result *res, int force /* 0 global-memory kernel, 1 sequential walk, 4 LDS kernel first */);

// ---- consumers of the pixel rows (apt_kernels_image.hip) ------
// All take the pixel count from `res` (device) when it is non-null, else `n`; `cap` bounds it
// and sizes the launch.  `ws` is image_ws_bytes(cap) bytes of scratch, 16-byte aligned.
size_t image_ws_bytes(uint64_t max_px);
struct ImageWsPointers {
    float *limits;     // [0] low, [1] high
    uint32_t *counts;  // 1000 histogram buckets
    float *mean_a, *mean_b, *variance, *corr, *quality;
};
ImageWsPointers image_ws_pointers(void *ws, uint64_t max_px);
// resets the record; image_minmax / image_percent / image_telemetry do it themselves, so this is
// only needed in front of a bare image_set_limits + image_map_u8
void image_begin(hipStream_t s, ImageResult *out);
// dsp::get_min / get_max, dsp.rs:20-54 -> limits
void image_minmax(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                  ImageResult *out);
// misc::percent, misc.rs:119-175 -> limits
void image_percent(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, float percent,
                   void *ws, ImageResult *out);
// telemetry::read_telemetry, telemetry.rs:125-243 -> out->values_*, rows; limits if set_limits
void image_telemetry(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                     ImageResult *out, bool set_limits);
void image_set_limits(hipStream_t s, void *ws, uint64_t cap, float low, float high);
// map_signal_u8, noaa_apt.rs:249-259 (+ processing::rotate, processing.rs:21-37)
void image_map_u8(hipStream_t s, const float *x, const Result *res, uint64_t n, uint64_t cap, void *ws,
                  bool rotate, uint8_t *out, ImageResult *info);

// ---- WAV ingest (apt_kernels_ingest.hip) --------------------------
// data chunk bytes -> f32 Signal, first channel only, unscaled (wav.rs:30-51); codec is
// apt::WavCodec as int
void wav_to_signal(hipStream_t s, const void *d_data, uint64_t n_frames, uint32_t channels,
                   uint32_t bytes_per_sample, int codec, float *d_signal);

// wav::write_wav, 16-bit branch (wav.rs:83-86): normalise by d_limits[1] (get_max) -> i16
void quantize_i16(hipStream_t s, const float *d_x, uint64_t n, const float *d_limits, int16_t *d_out);

// exhaustive device-side check that division by c through rc = RN(1/c) + one FMA correction is
// correctly rounded (apt_envelope.hpp); run once per plan
bool verify_fast_divide(hipStream_t s, float c, float rc);

// writes a result record from the host's knowledge (too-short recording, no-sync path)
void set_result(hipStream_t s, Result *res, Result value);

}  // namespace apt::gpu
