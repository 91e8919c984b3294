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

// Per-group record the front ends hand to the picker: an interval [lo, hi] that holds the maximum of
// the sync correlation over the group's positions, NaNs ignored (position 0 clamped to >= 0: the picker
// starts from the peak (0, 0.), decode.rs:208).  lo == hi where the front end evaluated the very
// arithmetic the picker re-evaluates (fast mode, the unfused kernels); the strict front ends — the
// specialised kernels and, since round 4, k_fused_any — bound the reference's 38 pw-term chain from pulse
// sums (apt_kernels_fused_impl.hpp stage 4, apt_kernels_fused_any_impl.hpp stage 4).  k_sync_words prunes a group only when a later group's lo exceeds its hi, and settles every
// comparison the bounds leave open with the exact chain: the intervals only decide how much is
// re-evaluated, never the result.
// [-inf, +inf] marks a group that must reach the exact test whatever its neighbours hold: one with a NaN
// position (never exceeded — `corr > NaN` is false, decode.rs:250 — so a terminal of the picker whatever
// the finite values around it are), or one whose F window is not finite.
struct GroupMax {
    float hi;
    float lo;
};

// A/B switches of the launch wrappers (APTGPU_* environment variables of tools/ and tests/): read ONCE, when a plan
// is created (read_launch_switches), and carried by the plan to every launch — not per launch: getenv is not safe
// against a concurrent setenv of another thread, and a plan must not change kernels in mid-life.
struct LaunchSwitches {
    int gather_iters = 1;        // APTGPU_GATHER_ITERS: quads per thread of k_gather_rows_flat (0: the eight-rows form)
    bool words_dpp = true;       // APTGPU_WORDS_DPP=0: k_sync_words without the DPP scans
    bool orbit_lds = true;       // APTGPU_ORBIT_LDS=0: the orbit kernel's tables in global memory
    int orbit_threads = 0;       // APTGPU_ORBIT_THREADS=256: the 256-thread orbit kernel
    int orbit_alg = 1;           // APTGPU_ORBIT_ALG=0: the breadth-first closure form
    float gm_slack_scale = 1.f;  // APTGPU_GM_SLACK_SCALE (tests): widens the strict front ends' bounds
    int fused_lds_pad = 0;       // APTGPU_FUSED_LDS_PAD: extra dynamic LDS bytes per front-end workgroup
};
LaunchSwitches read_launch_switches();

// ---- one decode_device call = one launch per stage over all its recordings -----------
// Per-recording arguments travel BY VALUE in the kernel-argument segment (no H2D copy, no pinned
// staging, nothing for the host to wait on); blockIdx.y (front end, k_sync_words, k_sync_slots, k_gather_rows) or
// blockIdx.x (k_sync_orbit) picks the recording.  The workspace of a recording is a slot of the
// plan; the slots' pointers sit in a device table written once at plan creation.
constexpr int kMaxCall = 32;  // recordings per launch; longer calls are split
struct RecArgs {
    const void *x;       // f32 Signal, or mono PCM16 payload (front end only)
    uint64_t n;          // input samples
    uint64_t w;          // work-rate samples (fast_resampling_len / decimation length)
    float *rows;         // output pixel rows
    uint32_t rows_cap;   // rows `rows` has room for
    uint32_t slot;       // index into the slot table
};
struct CallArgs {
    uint32_t count;
    uint32_t tiles_x;  // persistent front end only: tiles of the call's longest recording (else 0)
    RecArgs rec[kMaxCall];
};
struct SlotPtrs {
    float *f;               // filtered work-rate signal F
    GroupMax *gm;           // per-group correlation maxima
    const float *corr;      // full correlation (unfused path / step export only), else nullptr
    uint64_t *words;        // 52-bit terminal words
    uint64_t *nanw;         // 52-bit words: positions whose correlation is NaN
    uint32_t *slot_nt, *slot_cnt, *flags, *orbit_ws, *peaks;
    Result *res;
    uint32_t peaks_cap;
    uint32_t reserved;
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
// j in (i, i+md]; corr[0] is clamped to >= 0 (the initial (0, 0.) peak, decode.rs:208); NaNs
// read as -inf (never a record; orbit_walk handles a phase that starts on one).
// md must be a multiple of 64; needs 2*md*4 bytes of LDS.
void terminals(hipStream_t s, const float *corr, uint64_t n_corr, uint32_t md, uint64_t *bits);
// the orbit of the peak picker over the terminal bitmask (one wave): writes the peak list
// (find_sync's return value) and fills the result record.
void orbit_walk(hipStream_t s, const uint64_t *bits, const float *corr, uint64_t n_corr, uint64_t work_len,
                uint32_t spr, uint32_t md, uint32_t *peaks, uint32_t peaks_cap, Result *res);
// row gather (decode.rs:120-134) taking every pw-th sample; raw = plain copy (the
// "sync_result" step), !raw = through the final NoFilter stage (decode.rs:158-159)
void gather_rows(hipStream_t s, const float *f, const uint32_t *peaks, Result *res,
                 uint32_t spr, uint32_t pw, bool raw, float *rows, uint32_t rows_cap);
// the same for the recordings of one call: rows / rows_cap from the call, F / peaks / res from the slots
void gather_rows_call(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t pw,
                      uint32_t max_rows_cap, const LaunchSwitches &sw = LaunchSwitches{});

// the no-sync branch's tail (decode.rs:135-159, final decimation by m2 through NoFilter) and the result records of the
// recordings of one call in one launch; `rows_cap` of the call's records counts floats
void nosync_rows_call(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t m2, uint64_t max_w);

// ---- fused specialised front end (apt_kernels_fused.hip) --------------------------
// true when a <L, M, T1, T2, PW> specialisation exists
bool fused_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
uint32_t fused_group_size(uint32_t l);
// does the specialisation for these factors take mono PCM16 payloads as they are (else they are converted to f32 first)?
bool fused_takes_pcm16(uint32_t l, uint32_t m);
// host: stage-1 tap-pair table [WIN][PS][2] (see apt_kernels_fused.hip) and its size in floats
uint32_t fused_tap_table_floats(uint32_t l, uint32_t m, uint32_t t1, int ch);
void fused_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, int ch, float *hs, uint32_t t1_layout = 0);
// kModeStrictPad (mode 4 of fused_front_end): the tap-count bound of the padded strict kernel that serves (l, m, t1, t2, pw),
// or 0 where there is none (then t1 must match a kernel exactly: fused_supported)
uint32_t fused_pad_t1(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
// kModeStrictPad2: the low-pass bound of the padded kernel when t2 is not the profile's own (a tuned demodulation_atten:
// h2 / h2p must then be laid out for that many taps, zeros behind the filter's last), else 0
uint32_t fused_pad_t2(uint32_t l, uint32_t m, uint32_t t2, uint32_t pw);
// ... and of the PHASE kernels (fused_phase_supported accepts such a t2 at the standard profile's pixel width): 0 = t2 is
// the profile's own or beyond the bound
uint32_t fused_phase_pad_t2(uint32_t t2, uint32_t pw);
int fused_chunk_of(uint32_t m, bool fast);  // window samples per stage-1 chunk of the specialised kernel for (m, strict / fast)
// host: stage-3 tap pairs h2p[k] = (h2[k-1], h2[k]), k = 0 .. t2  (2*(t2+1) floats)
void fused_lowpass_pairs(const float *h2, uint32_t t2, float *h2p);
// fp16-tap stage 1 (APTGPU_MODE_FP16_TAPS): table size in dwords, host-side table builder (returns
// the power-of-two unscale factor), availability
uint32_t fused_f16_table_dwords(uint32_t l, uint32_t m, uint32_t t1);
float fused_f16_branch_taps(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, uint32_t *table);
bool fused_f16_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
// strict front ends: |pulse-sum correlation - sequential chain| <= fused_gm_slack(pw) * sum|F| (see stage 4)
float fused_gm_slack(uint32_t pw, float scale = 1.f);
// fast mode (APTGPU_MODE_FAST): availability (same tables as strict)
bool fused_fast_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
// Per-plan parameters of the specialised front end, resident in HBM (the kernel fetches each when
// the stage that needs it starts, instead of holding them in SGPRs from its first instruction on).
// Run-time geometry of the table-driven stage 1 (k_fused in TABLE mode; fused_table_geom fills it)
struct TableGeom {
    uint32_t l, m;             // interpolation / decimation factors
    uint32_t jlim;             // taps the reference uses (2*off + 1)
    uint32_t tpp;              // row stride of the phase-major table (>= taps per phase, odd)
    uint32_t xt;               // input tile, floats (multiple of 4)
    uint32_t off_x;            // LDS offset of the input tile in floats (the table sits at 0)
    uint32_t step_q, step_r;   // (NTHR*m) / l and % l: x0 / phase update between a thread's outputs
    uint32_t jl_a, jl_b;       // jlim / l and % l: taps of phase p = jl_a + (p < jl_b)
    // PHASE mode: the thread -> branch-slot assignment (fused_phase_table): `nperm` lists of `nthr` uint32 entries at
    // float offset `perm_off` of the table buffer, list (tile % nperm) for a tile; an entry >= step_r marks an idle thread
    uint32_t nthr, perm_off, nperm;
    uint32_t nq;               // PHASE mode: branches per thread (1, 2, 4, 8, 16)
    // PHASE mode: the slot stride — a thread with list entry u < sq holds the slots u + q sq (q < nq) that are < step_r.
    // nq == 1: sq = step_r.  nq == 4 (standard and slow profile), 8: sq = nthr (round 6: every thread has work and only the LAST slots of some threads
    // do not exist — l = 832: four slots for threads 0-63, three for the others, 13 wave-branches per tile instead of the 16
    // of sq = step_r / nq = 208, whose fourth wave ran every branch for 16 lanes: 2-3 % on the front end, not the 5 % the
    // instruction count promised — these kernels wait more than they issue, profiles/r06_phase_balanced_ab.txt).  Else
    // sq = step_r / nq (no gain / a loss measured).  APTGPU_PHASE_BALANCED=0 / 1 forces one form for every nq.
    uint32_t sq;
    uint32_t stream;           // PHASE mode: taps streamed from the table (filters too long for the registers), rows padded to 16
    // PHASE mode, exact != 0 (the tile phases repeat with a period nperm = 1 << perm_shift <= 8): list (tile % nperm) is
    // built for that tile's phase, and everything a tile derives from its index by division is tabulated — per phase r the
    // first input sample's quotient x0r[r] and remainder rbr[r] (tile = nperm t + r: X0 = x0r[r] + t xd), per thread the
    // window start and polyphase branch of each of its nq slots, packed (c | p << 16) at float offset cp_off
    // ([nperm][nthr][nq] uint32).  exact == 0: one list serves every tile and the kernel divides.
    uint32_t exact, perm_shift, cp_off, xd;
    // ... and the taps once more in THREAD order at float offset tt_off: [nperm][nq][tpp / 4][nthr] quads — quad e of
    // the branch of thread t's slot q — so that a wave's tap load is 1 KB of consecutive memory (from the phase-major
    // table every lane reads a row of its own: 64 cache lines per load)
    uint32_t tt_off;
    int32_t x0r[8];
    uint32_t rbr[8];
};
struct FusedParams {
    const float *hs;        // stage-1 table: tap pairs (fused_branch_taps), or the fp16 table
    const float *h2;        // low-pass taps [T2]
    const float *h2p;       // low-pass tap pairs (fused_lowpass_pairs)
    const SlotPtrs *slots;  // the plan's slot table
    float cosphi2, sinphi;
    float inv_sinphi;       // strict: verified RN(1/sinphi) or 0 (apt_envelope.hpp); fast: RN(1/sinphi)
    float f16_unscale;      // 2^-s of the fp16 tap prescale (fp16-tap mode)
    int32_t want_gm;        // sync search wanted: emit the per-group bounds of the correlation maxima
    float gm_slack;         // strict modes: half-width of a group's bounds per unit of sum|F| (fused_gm_slack)
    const float *table;     // TABLE mode: phase-major tap table [l][tpp] (fused_any_table)
    TableGeom tab;
    // kModeMfma: the plain resampler taps and their count (the scalar path of tiles that hold a non-finite sample)
    const float *coeff;
    uint32_t t1;
    uint32_t t2;            // kModeStrictPad2: the low-pass filter's own tap count (h2 / h2p are padded to kPadT2Max)
};
// One launch over the recordings of `call`: x -> F (slot's filtered buffer) and, if prm->want_gm, the
// per-group maxima of the sync cross-correlation.  Returns false if no specialisation matches.
// Inputs are f32 Signals, or (pcm16) mono int16 samples at 4-byte aligned addresses.
// mode: 0 strict, 1 fp16 taps, 2 fast.
bool fused_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, int mode,
                     bool pcm16, const CallArgs &call, const FusedParams *d_prm, uint64_t max_w, int lds_pad = 0);
// kModeMfma (mode 3 of fused_front_end): which geometries have a matrix-core instantiation, and its table —
// [3 pieces][K / 32][64 lanes][4 dwords] bf16 fragments of the banded Toeplitz matrix of the resampler (h = h0 + h1 + h2
// exactly)
bool fused_mfma_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
uint32_t fused_mfma_table_dwords(uint32_t l, uint32_t m);
void fused_mfma_table(uint32_t l, uint32_t m, const float *coeff, uint32_t t1, uint32_t *table);
// Table-driven stage 1 + the specialised work-rate stages (k_fused in TABLE mode): any (l, m, taps) whose
// phase-major table and input tile fit two 512-thread workgroups per CU, standard-profile work-rate
// stages (37-tap low-pass, pw = 3).  11 025 Hz (l = 832) is the rate this exists for.
bool fused_table_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, TableGeom *geom);
bool fused_table_front_end(hipStream_t s, const TableGeom &geom, int mode, bool pcm16, const CallArgs &call,
                           const FusedParams *d_prm, uint64_t max_w);

// Phase-resident stage 1 + the specialised work-rate stages (k_fused in PHASE mode): a thread holds nq slots u + q S'
// (S' = step_r / nq <= threads) of the step_r outputs after which the polyphase branches repeat, and computes the
// 16 / nq outputs of each that fall into the tile, with the taps of their common branch in registers (or, `stream`,
// fetched sixteen at a time).  256 threads with nq = 1, 2, 4 (l <= 256, 512, 1024) at the standard profile, nq = 1, 4, 8,
// 16 at the fast profile, nq = 1, 2, 4 with streamed taps at the slow profile; 512- / 1024-thread forms with nq = 1 as
// fallbacks.  Every rate a sound card records at is served this way (44 100 Hz: l = 208; 22 050: 416; 11 025: 832).
// TableGeom use: step_r = S, step_q = S*m/l, tpp = row stride of the table (multiple of 4; 16 when streamed), off_x = f2
// entries per region of the paired input tile, xt = floats of the whole tile; the rest: see the struct.
bool fused_phase_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, TableGeom *geom);
uint32_t fused_phase_table_floats(const TableGeom &geom);
// host: [l][tpp] taps, rows 16-byte aligned, then the thread assignment lists (geom.perm_off)
void fused_phase_table(const TableGeom &geom, uint32_t t2, uint32_t pw, const float *coeff, uint32_t t1, float *table);
bool fused_phase_front_end(hipStream_t s, const TableGeom &geom, uint32_t t2, uint32_t pw, int mode, bool pcm16,
                           const CallArgs &call, const FusedParams *d_prm, uint64_t max_w);

// ---- fused front end for any rate / profile (apt_kernels_fused_any.hip) -------------
// run-time parameters, taps phase-major in LDS; same outputs as fused_front_end
bool fused_any_supported(uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw);
uint32_t fused_any_table_floats(uint32_t l, uint32_t t1);
void fused_any_table(uint32_t l, const float *coeff, uint32_t t1, float *table);  // host
// One launch over the recordings of `call` (f32 Signals, or — pcm16 — mono int16 payloads at even addresses): F into
// the slots' filtered buffers and, if want_gm, the per-group maxima of the sync correlation into their gm buffers.
bool fused_any_front_end(hipStream_t s, uint32_t l, uint32_t m, uint32_t t1, uint32_t t2, uint32_t pw, bool pcm16,
                         const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, const float *table,
                         const float *h2, const float *h2p /* fused_lowpass_pairs */, float cosphi2, float sinphi,
                         float inv_sinphi, bool want_gm, float gm_slack_scale = 1.f);

// ---- parallel peak picker (apt_kernels_sync.hip) ------------------------------------
// Each launch covers the recordings of one call (CallArgs by value, slot table in HBM).
uint32_t sync_group_size();    // correlation positions per group (52)
uint32_t sync_chunk_groups();  // groups per k_sync_words / k_sync_slots workgroup
uint32_t sync_slot_cap();      // node terminals kept per chunk
// corr -> per-group maxima (unfused path; the fused front ends write them themselves)
void group_max(hipStream_t s, const float *corr, uint64_t n_corr, GroupMax *gm);
// coarse/fine terminal detection -> terminal words + ordered node-terminal lists.  use_corr: read
// the slots' full correlation (unfused kernels); else the correlation of the candidate groups is
// re-evaluated from F — strictly (the reference's chain) or, `fast`, from pulse sums, exactly as the
// front end of that mode did.
void sync_nodes(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint64_t max_w, uint32_t pw,
                uint32_t spr, uint32_t md, bool fast, bool use_corr, const LaunchSwitches &sw = LaunchSwitches{});
// orbit of the picker (direct / pointer doubling, or the sequential walk as fallback): peak
// list + result record of every recording
size_t sync_orbit_ws_words(uint64_t w, uint32_t spr);  // uint32 words of scratch it needs
void sync_orbit(hipStream_t s, const CallArgs &call, const SlotPtrs *d_slots, uint32_t spr, uint32_t md,
                uint32_t pw, int force /* 0 parallel picker, 1 sequential walk */, const LaunchSwitches &sw = LaunchSwitches{});

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
// correctly rounded (apt_envelope.hpp); run once per (device, divisor) and process, on a stream of
// its own (the current device must be `device`)
bool verify_fast_divide(int device, float c, float rc);

// writes a result record from the host's knowledge (too-short recording, no-sync path)
void set_result(hipStream_t s, Result *res, Result value);

}  // namespace apt::gpu
