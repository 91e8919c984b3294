// apt_session.hpp — what a host-fed decode keeps between calls, and the process-wide cache of it.
//
// SURVEY.md §8(b), threading row: "no process-global mutable state except a lazily-built, mutex-guarded
// per-device tap/plan cache".  The reference designs its filters inside every decode() (decode.rs:65-77,
// 95-102: microseconds on a CPU); here a plan also owns HBM workspace, streams and uploaded tap tables, and a
// host-fed call owns device input / output buffers and pinned staging besides — 2 ms of hipMalloc / hipFree /
// stream creation per call, against 2 ms of PCIe time for a ten-minute recording.  A Session bundles all of
// it for one (device, settings, input rate, sync, mode, recordings per call); aptgpu_decode, aptgpu_decode_wav
// and aptgpu_decode_batch[_wav] lease one from the cache for the duration of a call (one user at a time — a
// second concurrent caller with the same key gets a second Session) and hand it back; idle sessions are kept
// least-recently-used first up to kMaxIdle and APTGPU_SESSION_CACHE_MB of device memory (default: a quarter of
// the device's memory; 0 disables the cache; a session that cannot be built for lack of device memory empties
// the cache and tries once more), and aptgpu_cache_clear() drops them all.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "apt_plan.hpp"

namespace apt::capi {

struct SessionKey {
    int device = 0;
    int mode = 0;
    uint32_t rate = 0;
    bool sync = true;
    int per_call = 1;
    int depth = 1;  // calls in flight the plan is built for: 1 (aptgpu_decode: one stream, `plan->stream`), kSets (batch workers)
    aptgpu_settings settings{};  // the five fields decode() reads + export_resample_filtered (it changes the plan); export_wav zeroed
    bool operator==(const SessionKey &o) const;
};

struct PlanDeleter {
    void operator()(aptgpu_plan *p) const;
};

// One set of device / pinned buffers: the inputs, rows and result records of one call in flight.
struct IoSet {
    std::vector<apt::DeviceBuffer<uint8_t>> in;   // [per_call] input payloads (f32 Signal or WAV data chunk)
    std::vector<apt::DeviceBuffer<float>> out;    // [per_call] pixel rows
    float *h_rows = nullptr;                      // pinned, [per_call][out_cap]: D2H staging of rows whose destination could
                                                  // not be page-locked (allocated on first use)
    apt::gpu::Result *h_res = nullptr;            // pinned, [per_call]: the call's result records, one copy
    uint64_t in_bytes = 0, out_cap = 0;           // capacity of each `in` (bytes) / `out` (floats)
    hipEvent_t uploaded = nullptr, decoded = nullptr, downloaded = nullptr;
    bool in_use_before = false;                   // events recorded at least once
};

struct Session {
    SessionKey key;
    uint64_t max_n = 0;        // samples per recording the plan was built for
    std::unique_ptr<aptgpu_plan, PlanDeleter> plan;
    hipStream_t up = nullptr, down = nullptr;  // H2D and D2H copy streams (the link is full duplex)
    static constexpr int kSets = 3;            // call k decodes while k+1 (and k+2) upload and k-1 downloads
    IoSet sets[kSets];
    uint64_t device_bytes = 0;  // what the session holds in HBM (cache accounting): the plan's buffers + the sets'
    uint64_t plan_bytes = 0;    // ... of which the plan's (re-counted when the session returns to the cache)
    uint64_t last_used = 0;
    bool fresh = true;          // built for the current lease (false once it has been through the cache)

    ~Session();
    // (re)sizes set `k` for inputs of `in_bytes` bytes and rows of `out_cap` floats; no-op when large enough
    void ensure_set(int k, uint64_t in_bytes, uint64_t out_cap);
};

// Lease: returns the session to the cache on destruction (or destroys it when the call failed half-way:
// a session whose streams may hold failed work is not reused).
class SessionLease {
public:
    SessionLease() = default;
    explicit SessionLease(std::unique_ptr<Session> s) : s_(std::move(s)) {}
    SessionLease(SessionLease &&) = default;
    SessionLease &operator=(SessionLease &&) = default;
    ~SessionLease();
    Session *operator->() { return s_.get(); }
    Session &operator*() { return *s_; }
    explicit operator bool() const { return static_cast<bool>(s_); }
    void poison() { poisoned_ = true; }

private:
    std::unique_ptr<Session> s_;
    bool poisoned_ = false;
};

// A session for `key` whose plan takes recordings of up to `max_n` samples (an idle cached one if there is
// any, else a new one sized with 1/8 of headroom).  Throws apt::Error.
SessionLease session_acquire(const SessionKey &key, uint64_t max_n);
void session_cache_clear();
// (entries, device bytes) idle in the cache
void session_cache_info(int *entries, uint64_t *device_bytes);

}  // namespace apt::capi
