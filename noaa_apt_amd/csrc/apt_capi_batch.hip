// apt_capi_batch.hip — host-fed batch decode over one or more GPUs (include/aptgpu.h §2b).
//
// What a batch driver over independent recordings binds: the loop `for file in files { load();
// decode(); }` of the reference's CLI (src/main.rs:102-104), with the recordings sharded over the
// GPUs of the node.  Recordings never talk to each other (decode() touches only its arguments), so
// there is no collective of any kind: every device entry gets a host thread, a plan and its share of
// the recordings (longest first onto the least loaded entry — the same rule as
// noaa_apt_amd/shard.py), and works through it in calls of `recordings_per_call`.
//
// Per worker (one host thread per device entry): a Session leased from the process-wide cache (apt_session.hpp:
// plan, three sets of device input / output buffers, pinned staging for rows and result records, an upload and
// a download stream), and a three-stage software pipeline over calls of `recordings_per_call` recordings in
// which the HOST never waits for anything but finished rows:
//   upload(c+2)   H2D of the inputs, `up` stream, behind the decode that last read that set's buffers
//   decode(c)     the plan's five launches, behind upload(c) and the download that last read that set's rows
//   download(c)   the call's result records (ONE copy) and the rows, `down` stream, into pinned staging
//   collect(c-1)  host: wait for download(c-1), hand malloc'd rows to the caller
// so H2D of call k+1 / k+2, the kernels of call k and D2H of call k-1 are in flight together and the link is
// used in both directions at once.  WAV file images are uploaded as their data-chunk payload (2 bytes per sample
// for PCM16) and converted on the device.  Pageable host buffers are handed to hipMemcpyAsync as they are: the
// runtime's pinned staging moves them at 56 GB/s here, as fast as a caller-pinned buffer (57 GB/s;
// tools/ubench/hostpath.cpp — pinning 173 MB on the fly costs 4.7 ms, more than its transfer).
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <list>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "apt_capi_util.hpp"
#include "apt_session.hpp"

namespace apt::capi {

// ------------------------------------------------------------------ session cache
bool SessionKey::operator==(const SessionKey &o) const
{
    return device == o.device && mode == o.mode && rate == o.rate && sync == o.sync && per_call == o.per_call && depth == o.depth &&
           settings.work_rate == o.settings.work_rate &&
           std::memcmp(&settings.resample_atten, &o.settings.resample_atten, sizeof(float)) == 0 &&
           std::memcmp(&settings.resample_delta_freq, &o.settings.resample_delta_freq, sizeof(float)) == 0 &&
           std::memcmp(&settings.resample_cutout, &o.settings.resample_cutout, sizeof(float)) == 0 &&
           std::memcmp(&settings.demodulation_atten, &o.settings.demodulation_atten, sizeof(float)) == 0 &&
           (settings.export_resample_filtered != 0) == (o.settings.export_resample_filtered != 0);
}

void PlanDeleter::operator()(aptgpu_plan *p) const { aptgpu_plan_destroy(p); }

Session::~Session()
{
    (void)hipSetDevice(key.device);
    // nothing of this session may still be in flight when its buffers go
    if (up) (void)hipStreamSynchronize(up);
    if (down) (void)hipStreamSynchronize(down);
    if (plan) {
        try {
            plan->sync_all();
        } catch (...) {
        }
    }
    for (IoSet &s : sets) {
        if (s.h_rows) (void)hipHostFree(s.h_rows);
        if (s.h_res) (void)hipHostFree(s.h_res);
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        if (s.decoded) (void)hipEventDestroy(s.decoded);
        if (s.downloaded) (void)hipEventDestroy(s.downloaded);
    }
    if (up) (void)hipStreamDestroy(up);
    if (down) (void)hipStreamDestroy(down);
}

void Session::ensure_set(int k, uint64_t in_bytes, uint64_t out_cap)
{
    IoSet &s = sets[k];
    const size_t B = static_cast<size_t>(key.per_call);
    if (!s.uploaded) {
        apt::hip_check(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming), "hipEventCreate");
        apt::hip_check(hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming), "hipEventCreate");
        apt::hip_check(hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming), "hipEventCreate");
        s.in.resize(B);
        s.out.resize(B);
    }
    if (!s.h_res) apt::hip_check(hipHostMalloc(reinterpret_cast<void **>(&s.h_res), B * sizeof(apt::gpu::Result), hipHostMallocDefault), "hipHostMalloc");
    // (an allocation that fails for lack of device memory while idle sessions hold some: drop them, try once more)
    auto alloc = [&](auto &buf, uint64_t count) {
        try {
            buf.alloc(count);
        } catch (const Error &e) {
            buf.count = 0;
            if (e.kind != ErrorKind::Hip || e.hip_code != static_cast<int>(hipErrorOutOfMemory)) throw;  // (the status itself, not its wording)
            (void)hipGetLastError();
            session_cache_clear();
            try {
                buf.alloc(count);
            } catch (...) {
                buf.count = 0;
                throw;
            }
        }
    };
    if (in_bytes > s.in_bytes) {
        for (auto &b : s.in) {
            device_bytes -= b.count;
            b.release();
            alloc(b, in_bytes);
            device_bytes += in_bytes;
        }
        s.in_bytes = in_bytes;
    }
    if (out_cap > s.out_cap) {
        for (auto &b : s.out) {
            device_bytes -= b.count * sizeof(float);
            b.release();
            alloc(b, out_cap);
            device_bytes += out_cap * sizeof(float);
        }
        if (s.h_rows) (void)hipHostFree(s.h_rows);  // (re-created at the new size if it is ever needed)
        s.h_rows = nullptr;
        s.out_cap = out_cap;
    }
}

// the pinned staging for rows whose destination could not be page-locked: allocated on first use
float *staging_rows(IoSet &s, size_t per_call)
{
    if (!s.h_rows)
        apt::hip_check(hipHostMalloc(reinterpret_cast<void **>(&s.h_rows), per_call * s.out_cap * sizeof(float), hipHostMallocDefault),
                       "hipHostMalloc");
    return s.h_rows;
}

static_assert(sizeof(aptgpu_batch_stats) == 88, "aptgpu_batch_stats: layout of include/aptgpu.h (api.py mirrors it)");

namespace {

struct Cache {
    std::mutex mu;
    std::list<std::unique_ptr<Session>> idle;  // most recently used first
    uint64_t clock = 0;
    static constexpr size_t kMaxIdle = 8;
    // APTGPU_SESSION_CACHE_MB (0 disables the cache); unset: a quarter of the current device's memory — idle sessions
    // are invisible to whatever else allocates on the GPU in this process (torch, another library), so they may
    // not sit on most of it
    static uint64_t budget_bytes()
    {
        static const uint64_t v = []() -> uint64_t {
            if (const char *e = std::getenv("APTGPU_SESSION_CACHE_MB")) return static_cast<uint64_t>(std::strtoull(e, nullptr, 10)) << 20;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) {
                (void)hipGetLastError();
                return 16384ull << 20;
            }
            return static_cast<uint64_t>(total_b) / 4u;
        }();
        return v;
    }
};
Cache &cache()
{
    static Cache *c = new Cache;  // (never destroyed: sessions may not outlive the HIP runtime's own teardown)
    return *c;
}

uint64_t plan_device_bytes(const aptgpu_plan &p)
{
    // what plan_create allocated, buffer by buffer (buffers a slot only gets on first use — the unfused kernels'
    // intermediates, the WAV staging, the image scratch — are counted when the session comes back to the cache)
    uint64_t b = 0;
    auto add = [&](const auto &buf) { b += static_cast<uint64_t>(buf.count) * sizeof(*buf.ptr); };
    for (const auto &sl : p.slots) {
        add(sl.resampled); add(sl.demodulated); add(sl.filtered); add(sl.correlation); add(sl.bits); add(sl.peaks);
        add(sl.gm); add(sl.words); add(sl.nanw); add(sl.slot_nt); add(sl.slot_cnt); add(sl.flags); add(sl.orbit_ws);
        add(sl.image_ws); add(sl.ingest);
    }
    add(p.d_taps_resample); add(p.d_taps_lowpass); add(p.d_taps_branch); add(p.d_taps_lowpass_pairs); add(p.d_taps_any);
    add(p.d_taps_f16); add(p.d_slots); add(p.d_results); add(p.d_image_results);
    return b;
}

}  // namespace

SessionLease session_acquire(const SessionKey &key, uint64_t max_n)
{
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto it = c.idle.begin(); it != c.idle.end(); ++it) {
            if ((*it)->key == key && (*it)->max_n >= max_n) {
                std::unique_ptr<Session> s = std::move(*it);
                c.idle.erase(it);
                s->fresh = false;
                return SessionLease(std::move(s));
            }
        }
    }
    // a new one, with headroom so that the next, slightly longer recording still fits
    auto build = [&]() {
        auto s = std::make_unique<Session>();
        s->key = key;
        s->max_n = max_n + max_n / 8 + 4096;
        apt::hip_check(hipSetDevice(key.device), "hipSetDevice");
        aptgpu_context ctx{};
        ctx.device = key.device;
        ctx.mode = key.mode;
        // `depth` calls in flight: a batch worker pipelines Session::kSets calls whatever `per_call` is — with one
        // stream and one slot per recording, decode(c+1) would write the result record download(c) is still copying
        s->plan.reset(apt::plan_create(&ctx, key.settings, key.rate, key.sync, s->max_n, key.per_call, std::max(1, key.depth)));
        apt::hip_check(hipStreamCreateWithFlags(&s->up, hipStreamNonBlocking), "hipStreamCreate");
        apt::hip_check(hipStreamCreateWithFlags(&s->down, hipStreamNonBlocking), "hipStreamCreate");
        s->plan_bytes = plan_device_bytes(*s->plan);
        s->device_bytes = s->plan_bytes;
        return s;
    };
    try {
        return SessionLease(build());
    } catch (const Error &e) {
        // out of device memory while idle sessions hold some: drop them and try once more
        if (e.kind != ErrorKind::Hip || e.hip_code != static_cast<int>(hipErrorOutOfMemory)) throw;  // (the status itself, not its wording)
        (void)hipGetLastError();
        session_cache_clear();
        return SessionLease(build());
    }
}

SessionLease::~SessionLease()
{
    if (!s_) return;
    if (poisoned_ || Cache::budget_bytes() == 0) {
        s_.reset();
        return;
    }
    Cache &c = cache();
    std::list<std::unique_ptr<Session>> evicted;  // destroyed outside the lock (their destructors synchronise)
    // (slots grow buffers on first use: account what the plan holds now)
    s_->device_bytes += plan_device_bytes(*s_->plan) - s_->plan_bytes;
    s_->plan_bytes = plan_device_bytes(*s_->plan);
    {
        std::lock_guard<std::mutex> lock(c.mu);
        s_->last_used = ++c.clock;
        c.idle.push_front(std::move(s_));
        uint64_t bytes = 0;
        size_t n = 0;
        for (auto it = c.idle.begin(); it != c.idle.end();) {
            bytes += (*it)->device_bytes;
            ++n;
            if (n > Cache::kMaxIdle || (bytes > Cache::budget_bytes() && n > 1)) {
                evicted.push_back(std::move(*it));
                it = c.idle.erase(it);
            } else {
                ++it;
            }
        }
    }
}

void session_cache_clear()
{
    Cache &c = cache();
    std::list<std::unique_ptr<Session>> all;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        all.swap(c.idle);
    }
}

void session_cache_info(int *entries, uint64_t *device_bytes)
{
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    uint64_t b = 0;
    for (const auto &s : c.idle) b += s->device_bytes;
    if (entries) *entries = static_cast<int>(c.idle.size());
    if (device_bytes) *device_bytes = b;
}

}  // namespace apt::capi

namespace {

using namespace apt::capi;

// ------------------------------------------------------------------ NUMA placement of the workers
// A worker feeds its GPU from host memory at PCIe rate: with eight of them on a two-socket host, threads (and the
// rows buffers and staging copies they first touch) that sit on the other socket cross the inter-socket link for
// every byte.  sysfs says where a PCI device hangs: <root>/bus/pci/devices/<bdf>/numa_node, and the node's CPUs in
// <root>/devices/system/node/node<N>/cpulist (the device's own local_cpulist as a fallback).
struct HostAffinity {
    int node = -1;
    std::string cpulist;    // as sysfs prints it: "0-63,128-191"
    std::vector<int> cpus;  // parsed
};

std::string read_line(const std::string &path)
{
    std::ifstream f(path);
    std::string line;
    if (f) std::getline(f, line);
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r' || line.back() == ' ')) line.pop_back();
    return line;
}

std::vector<int> parse_cpulist(const std::string &s)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string tok = s.substr(i, j - i);
        i = j + 1;
        if (tok.empty()) continue;
        const size_t dash = tok.find('-');
        char *end = nullptr;
        const long a = std::strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str() || a < 0) return {};
        long b = a;
        if (dash != std::string::npos) {
            const char *bs = tok.c_str() + dash + 1;
            b = std::strtol(bs, &end, 10);
            if (end == bs || b < a) return {};
        }
        if (b - a > 65536) return {};
        for (long c = a; c <= b; ++c) out.push_back(static_cast<int>(c));
    }
    return out;
}

HostAffinity affinity_from_sysfs(const std::string &root, const std::string &bdf_in)
{
    HostAffinity h;
    std::string bdf = bdf_in;
    for (char &c : bdf) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));  // sysfs spells addresses in lower case
    const std::string dev = root + "/bus/pci/devices/" + bdf;
    const std::string node_s = read_line(dev + "/numa_node");
    if (node_s.empty()) return h;
    char *end = nullptr;
    const long node = std::strtol(node_s.c_str(), &end, 10);
    if (end == node_s.c_str() || node < 0) return h;  // -1: the platform does not say
    std::string list = read_line(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist");
    if (list.empty()) list = read_line(dev + "/local_cpulist");
    std::vector<int> cpus = parse_cpulist(list);
    if (cpus.empty()) return h;
    h.node = static_cast<int>(node);
    h.cpulist = list;
    h.cpus = std::move(cpus);
    return h;
}

HostAffinity affinity_of_device(int device, std::string *bdf_out)
{
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return {};
    }
    if (bdf_out) *bdf_out = bdf;
    const char *root = std::getenv("APTGPU_SYSFS_ROOT");  // tests: a mocked tree
    return affinity_from_sysfs(root ? root : "/sys", bdf);
}

// the calling thread onto the CPUs of `device`'s NUMA node (no-op when unknown, or APTGPU_NUMA_PIN=0)
bool pin_worker_to_device_node(int device)
{
    const char *e = std::getenv("APTGPU_NUMA_PIN");
    if (e && e[0] == '0') return false;
    const HostAffinity h = affinity_of_device(device, nullptr);
    if (h.node < 0) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (int c : h.cpus)
        if (c >= 0 && c < CPU_SETSIZE) {
            CPU_SET(c, &set);
            ++n;
        }
    // (only CPUs this process may run on anyway: a container's cpuset stays in force)
    cpu_set_t allowed;
    if (n && pthread_getaffinity_np(pthread_self(), sizeof allowed, &allowed) == 0) {
        cpu_set_t both;
        CPU_AND(&both, &set, &allowed);
        if (CPU_COUNT(&both) > 0) return pthread_setaffinity_np(pthread_self(), sizeof both, &both) == 0;
    }
    return false;
}

int put_affinity(const HostAffinity &h, int32_t *numa_node, char *cpulist, size_t cap)
{
    if (numa_node) *numa_node = h.node;
    if (cpulist && cap) {
        std::snprintf(cpulist, cap, "%s", h.cpulist.c_str());
    }
    return APTGPU_OK;
}

struct Item {
    int index;            // position in the caller's arrays
    const uint8_t *data;  // f32 samples, or the payload of the WAV data chunk
    uint64_t bytes;       // bytes to upload
    uint64_t n;           // samples (frames)
    apt::WavInfo wav;     // when is_wav
    bool is_wav = false;
};

struct Shared {
    const aptgpu_settings *settings;
    uint32_t rate;
    bool sync;
    int mode;
    int per_call;
    float **rows_out;
    size_t *n_out;
    int32_t *status;
    aptgpu_result *results;  // nullable
    std::mutex mu;
    std::string first_error;
    int first_error_code = APTGPU_OK;
    double h2d_seconds = 0, d2h_seconds = 0;  // summed over workers (host wall time inside the copy calls / collects)
    uint64_t h2d_bytes = 0, d2h_bytes = 0;
    double gate_wait_seconds = 0, setup_seconds = 0;
    int sessions_created = 0, workers_pinned = 0;
};

std::mutex &upload_gate(int device)
{
    static std::mutex gates[64];
    return gates[(device >= 0 && device < 64) ? device : 0];
}

void fail_item(Shared &sh, const Item &it, int code)
{
    sh.status[it.index] = code;
    if (sh.rows_out[it.index]) std::free(sh.rows_out[it.index]);
    sh.rows_out[it.index] = nullptr;
    sh.n_out[it.index] = 0;
}

// one device entry: decodes `items` (already ordered) on `device`
void worker(Shared &sh, int device, std::vector<Item> items)
{
    using clock = std::chrono::steady_clock;
    auto seconds = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    SessionLease lease;
    try {
        apt::hip_check(hipSetDevice(device), "hipSetDevice");
        const auto t_setup0 = clock::now();
        const bool pinned = pin_worker_to_device_node(device);  // before anything is allocated or first touched by this thread
        double t_gate = 0;
        uint64_t max_n = 0, max_bytes = 0;
        for (const Item &it : items) {
            max_n = std::max(max_n, it.n);
            max_bytes = std::max(max_bytes, it.bytes);
        }
        const int B = std::max(1, std::min<int>(sh.per_call, static_cast<int>(items.size())));
        SessionKey key;
        key.device = device;
        key.mode = sh.mode;
        key.rate = sh.rate;
        key.sync = sh.sync;
        key.per_call = B;
        key.depth = Session::kSets;
        key.settings = *sh.settings;
        key.settings.export_wav = 0;
        key.settings.export_resample_filtered = sh.settings->export_resample_filtered ? 1 : 0;  // another plan (dsp.rs:265-273)
        lease = session_acquire(key, max_n);
        Session &S = *lease;
        aptgpu_plan *plan = S.plan.get();
        auto rows_bound = [&](uint64_t n) -> uint64_t {  // floats the rows of an n-sample recording can take
            const uint64_t w = plan->work_len_for(n);
            return sh.sync ? (plan->spr ? w / plan->spr + 2 : 2) * 2080u : plan->out_len_nosync(w) + 16;
        };
        const uint64_t out_cap = rows_bound(max_n);
        const size_t n_chunks = (items.size() + static_cast<size_t>(B) - 1) / static_cast<size_t>(B);
        const int n_sets = static_cast<int>(std::min<size_t>(Session::kSets, n_chunks));
        for (int k = 0; k < n_sets; ++k) S.ensure_set(k, max_bytes + 64, out_cap);
        const double t_setup = seconds(t_setup0, clock::now());
        const bool created = S.fresh;
        double t_h2d = 0, t_d2h = 0;
        uint64_t b_h2d = 0, b_d2h = 0;

        auto chunk_of = [&](size_t c, size_t *from, size_t *to) {
            *from = c * static_cast<size_t>(B);
            *to = std::min(items.size(), *from + static_cast<size_t>(B));
        };
        std::vector<uint8_t> used(Session::kSets, 0);  // the set's events have been recorded in THIS call
        // H2D of chunk c into set c % 3, behind the decode that last read that set's input buffers
        auto upload = [&](size_t c) {
            IoSet &s = S.sets[c % Session::kSets];
            size_t from, to;
            chunk_of(c, &from, &to);
            const auto a = clock::now();
            if (used[c % Session::kSets]) apt::hip_check(hipStreamWaitEvent(S.up, s.decoded, 0), "hipStreamWaitEvent");
            // hipMemcpyAsync from PAGEABLE memory returns when the runtime has staged the data: the calling thread is
            // busy for the duration of the copy.  Two workers of one device doing that at once contend for the link
            // and for the runtime's staging path (measured: 34 GB/s together against 52 GB/s one after the other),
            // so uploads to one device take turns; the worker that waits has its decode / download in flight meanwhile.
            const auto g0 = clock::now();
            std::lock_guard<std::mutex> gate(upload_gate(device));
            t_gate += seconds(g0, clock::now());
            for (size_t k = from; k < to; ++k) {
                const Item &it = items[k];
                if (it.bytes)
                    apt::hip_check(hipMemcpyAsync(s.in[k - from].ptr, it.data, it.bytes, hipMemcpyHostToDevice, S.up),
                                   "hipMemcpyAsync H2D");
                b_h2d += it.bytes;
            }
            apt::hip_check(hipEventRecord(s.uploaded, S.up), "hipEventRecord");
            t_h2d += seconds(a, clock::now());
        };
        // the decode of chunk c, behind its upload and behind the download that last read that set's rows
        std::vector<std::vector<int>> chunk_slots(n_chunks);
        auto decode = [&](size_t c) {
            IoSet &s = S.sets[c % Session::kSets];
            size_t from, to;
            chunk_of(c, &from, &to);
            const int cnt = static_cast<int>(to - from);
            std::vector<aptgpu_plan::Input> ins(static_cast<size_t>(cnt));
            std::vector<float *> rows(static_cast<size_t>(cnt));
            std::vector<uint64_t> caps(static_cast<size_t>(cnt), s.out_cap);
            for (int b = 0; b < cnt; ++b) {
                const Item &it = items[from + static_cast<size_t>(b)];
                aptgpu_plan::Input &in = ins[static_cast<size_t>(b)];
                in.ptr = s.in[static_cast<size_t>(b)].ptr;
                in.n = it.n;
                if (it.is_wav) {
                    in.channels = it.wav.channels;
                    in.bytes_per_sample = it.wav.bytes_per_sample;
                    in.codec = static_cast<int>(it.wav.codec);
                }
                rows[static_cast<size_t>(b)] = s.out[static_cast<size_t>(b)].ptr;
            }
            hipStream_t next = plan->streams[static_cast<size_t>(plan->calls % plan->streams.size())];
            apt::hip_check(hipStreamWaitEvent(next, s.uploaded, 0), "hipStreamWaitEvent");
            if (used[c % Session::kSets]) apt::hip_check(hipStreamWaitEvent(next, s.downloaded, 0), "hipStreamWaitEvent");
            plan->run_call(cnt, ins.data(), rows.data(), caps.data(), false);
            apt::hip_check(hipEventRecord(s.decoded, next), "hipEventRecord");
            chunk_slots[c] = plan->last_slots;
        };
        // D2H of chunk c: the call's result records in ONE copy (its slots are consecutive) into pinned memory, and the
        // rows STRAIGHT into the buffers the caller will own: malloc'd for as many floats as a recording of that
        // length can produce (the record says how many of them count), page-locked for the duration of the copy.
        // (Through pinned staging and a host memcpy the rows cost the worker thread 3 ms per recording — first-touch
        // page faults included — and that copy, not the link, set the pace of PCM16 batches.)
        struct Dest {
            float *rows = nullptr;
            bool locked = false;
        };
        std::vector<std::vector<Dest>> dests(n_chunks);
        auto release_dests = [&](size_t c, bool free_rows) {
            for (Dest &d : dests[c]) {
                if (d.locked) (void)hipHostUnregister(d.rows);
                d.locked = false;
                if (free_rows && d.rows) std::free(d.rows);
                if (free_rows) d.rows = nullptr;
            }
        };
        auto download = [&](size_t c) {
            IoSet &s = S.sets[c % Session::kSets];
            size_t from, to;
            chunk_of(c, &from, &to);
            const int cnt = static_cast<int>(to - from);
            const std::vector<int> &sl = chunk_slots[c];
            const auto a = clock::now();
            dests[c].resize(static_cast<size_t>(cnt));
            for (int b = 0; b < cnt; ++b) {
                const uint64_t fl = std::min<uint64_t>(rows_bound(items[from + static_cast<size_t>(b)].n), s.out_cap);
                Dest &d = dests[c][static_cast<size_t>(b)];
                d.rows = static_cast<float *>(std::malloc((fl ? fl : 1) * sizeof(float)));
                if (!d.rows) throw std::bad_alloc();
                d.locked = fl && hipHostRegister(d.rows, fl * sizeof(float), hipHostRegisterDefault) == hipSuccess;
                if (!d.locked) (void)hipGetLastError();  // (falls back to the staged copy below)
            }
            apt::hip_check(hipStreamWaitEvent(S.down, s.decoded, 0), "hipStreamWaitEvent");
            bool consecutive = true;
            for (int b = 1; b < cnt; ++b) consecutive = consecutive && sl[static_cast<size_t>(b)] == sl[0] + b;
            if (consecutive) {
                apt::hip_check(hipMemcpyAsync(s.h_res, plan->d_results.ptr + sl[0], static_cast<size_t>(cnt) * sizeof(apt::gpu::Result),
                                              hipMemcpyDeviceToHost, S.down), "hipMemcpyAsync results");
            } else {
                for (int b = 0; b < cnt; ++b)
                    apt::hip_check(hipMemcpyAsync(s.h_res + b, plan->d_results.ptr + sl[static_cast<size_t>(b)], sizeof(apt::gpu::Result),
                                                  hipMemcpyDeviceToHost, S.down), "hipMemcpyAsync results");
            }
            for (int b = 0; b < cnt; ++b) {
                const uint64_t fl = std::min<uint64_t>(rows_bound(items[from + static_cast<size_t>(b)].n), s.out_cap);
                Dest &d = dests[c][static_cast<size_t>(b)];
                float *dst = d.locked ? d.rows : staging_rows(s, static_cast<size_t>(S.key.per_call)) + static_cast<size_t>(b) * s.out_cap;
                apt::hip_check(hipMemcpyAsync(dst, s.out[static_cast<size_t>(b)].ptr, fl * sizeof(float), hipMemcpyDeviceToHost, S.down),
                               "hipMemcpyAsync D2H rows");
            }
            apt::hip_check(hipEventRecord(s.downloaded, S.down), "hipEventRecord");
            used[c % Session::kSets] = 1;
            t_d2h += seconds(a, clock::now());
        };
        // host: wait for chunk c's download, hand the rows over
        auto collect = [&](size_t c) {
            IoSet &s = S.sets[c % Session::kSets];
            size_t from, to;
            chunk_of(c, &from, &to);
            const auto a = clock::now();
            apt::hip_check(hipEventSynchronize(s.downloaded), "hipEventSynchronize");
            for (size_t k = from; k < to; ++k) {
                const Item &it = items[k];
                Dest &d = dests[c][k - from];
                aptgpu_result r{};
                static_assert(sizeof(aptgpu_result) == sizeof(apt::gpu::Result), "the device record IS the ABI record");
                std::memcpy(&r, s.h_res + (k - from), sizeof r);
                if (sh.results) sh.results[it.index] = r;
                const bool was_locked = d.locked;
                if (d.locked) (void)hipHostUnregister(d.rows);
                d.locked = false;
                if (r.status != APTGPU_OK) {
                    std::free(d.rows);
                    d.rows = nullptr;
                    fail_item(sh, it, r.status);
                    continue;
                }
                if (!was_locked && r.n_out) std::memcpy(d.rows, s.h_rows + (k - from) * s.out_cap, r.n_out * sizeof(float));
                b_d2h += r.n_out * sizeof(float);
                sh.rows_out[it.index] = d.rows;
                d.rows = nullptr;
                sh.n_out[it.index] = r.n_out;
                sh.status[it.index] = APTGPU_OK;
            }
            t_d2h += seconds(a, clock::now());
        };
        struct DestGuard {  // an exception half-way: nothing stays page-locked, nothing leaks
            std::vector<std::vector<Dest>> &d;
            hipStream_t down;
            ~DestGuard()
            {
                bool any = false;
                for (auto &v : d)
                    for (Dest &x : v) any = any || x.rows;
                if (!any) return;
                (void)hipStreamSynchronize(down);
                for (auto &v : d)
                    for (Dest &x : v) {
                        if (x.locked) (void)hipHostUnregister(x.rows);
                        if (x.rows) std::free(x.rows);
                        x.rows = nullptr;
                    }
            }
        } dest_guard{dests, S.down};
        (void)release_dests;

        for (size_t c = 0; c < std::min<size_t>(2, n_chunks); ++c) upload(c);
        for (size_t c = 0; c < n_chunks; ++c) {
            decode(c);
            download(c);
            if (c + 2 < n_chunks) upload(c + 2);
            if (c >= 1) collect(c - 1);
        }
        if (n_chunks >= 1) collect(n_chunks - 1);
        std::lock_guard<std::mutex> lock(sh.mu);
        sh.h2d_seconds += t_h2d;
        sh.d2h_seconds += t_d2h;
        sh.h2d_bytes += b_h2d;
        sh.d2h_bytes += b_d2h;
        sh.gate_wait_seconds += t_gate;
        sh.setup_seconds += t_setup;
        sh.sessions_created += created ? 1 : 0;
        sh.workers_pinned += pinned ? 1 : 0;
    } catch (const Error &e) {
        lease.poison();  // (its destructor synchronises the streams before anything is freed)
        std::lock_guard<std::mutex> lock(sh.mu);
        if (sh.first_error_code == APTGPU_OK) {
            sh.first_error_code = static_cast<int>(e.kind);
            sh.first_error = e.message;
        }
        for (const Item &it : items)
            if (sh.status[it.index] == -1) fail_item(sh, it, static_cast<int>(e.kind));
    } catch (const std::exception &e) {
        lease.poison();
        std::lock_guard<std::mutex> lock(sh.mu);
        if (sh.first_error_code == APTGPU_OK) {
            sh.first_error_code = APTGPU_ERR_INTERNAL;
            sh.first_error = e.what();
        }
        for (const Item &it : items)
            if (sh.status[it.index] == -1) fail_item(sh, it, APTGPU_ERR_INTERNAL);
    }
}

int decode_batch_impl(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                      int count, const void *const *inputs, const size_t *n, bool wav_images, const int32_t *devices,
                      int n_devices, int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                      aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    if (!settings || count < 0 || (count && (!inputs || !n || !rows_out || !n_out || !status)) || n_devices < 0 ||
        (n_devices && !devices)) {
        put_err(err, err_cap, "bad argument to aptgpu_decode_batch");
        return APTGPU_ERR_INVALID;
    }
    // (the caller's sizeof, set before the call; 0 — a zero-initialised struct, or the first half of a 0.1.0 caller's
    // `double seconds` — is not a size the library may guess: include/aptgpu.h)
    if (stats && stats->struct_size < 2 * sizeof(uint32_t)) {
        put_err(err, err_cap, "aptgpu_batch_stats.struct_size is not set (sizeof(aptgpu_batch_stats) of the caller's header)");
        return APTGPU_ERR_INVALID;
    }
    return guarded(err, err_cap, [&]() -> int {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<int32_t> devs(devices, devices + n_devices);
        if (devs.empty()) devs.push_back(ctx ? ctx->device : 0);
        int visible = 0;
        apt::hip_check(hipGetDeviceCount(&visible), "hipGetDeviceCount");
        for (int32_t d : devs)
            if (d < 0 || d >= visible) throw Error{ErrorKind::Invalid, "device ordinal out of range"};

        Shared sh{};
        sh.settings = settings;
        sh.rate = input_rate_hz;
        sh.sync = sync != 0;
        sh.mode = ctx ? ctx->mode : APTGPU_MODE_STRICT;
        sh.per_call = std::max(1, std::min(recordings_per_call <= 0 ? 16 : recordings_per_call, apt::gpu::kMaxCall));
        sh.rows_out = rows_out;
        sh.n_out = n_out;
        sh.status = status;
        sh.results = results;
        std::vector<Item> items;
        uint64_t total_samples = 0;
        for (int i = 0; i < count; ++i) {
            status[i] = -1;  // pending
            rows_out[i] = nullptr;
            n_out[i] = 0;
            if (results) results[i] = aptgpu_result{};
            Item it;
            it.index = i;
            if (!inputs[i] && n[i]) {
                status[i] = APTGPU_ERR_INVALID;
                continue;
            }
            if (wav_images) {
                try {
                    it.wav = apt::parse_wav(static_cast<const uint8_t *>(inputs[i]), n[i]);
                } catch (const Error &e) {
                    status[i] = static_cast<int>(e.kind);  // the reference's error for this file (err.rs:72-83)
                    continue;
                }
                if (it.wav.sample_rate != input_rate_hz) {
                    status[i] = APTGPU_ERR_INVALID;  // one (settings, rate) per batch: the plans are built for it
                    continue;
                }
                it.is_wav = true;
                it.data = static_cast<const uint8_t *>(inputs[i]) + it.wav.data_offset;
                it.bytes = it.wav.data_len;
                it.n = it.wav.n_frames;
            } else {
                it.data = static_cast<const uint8_t *>(inputs[i]);
                it.bytes = static_cast<uint64_t>(n[i]) * sizeof(float);
                it.n = n[i];
            }
            total_samples += it.n;
            items.push_back(it);
        }
        // longest first onto the least loaded entry (LPT; noaa_apt_amd/shard.py uses the same rule across ranks)
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.n > b.n; });
        std::vector<std::vector<Item>> share(devs.size());
        std::vector<uint64_t> load(devs.size(), 0);
        for (const Item &it : items) {
            const size_t k = static_cast<size_t>(std::min_element(load.begin(), load.end()) - load.begin());
            share[k].push_back(it);
            load[k] += it.n;
        }
        std::vector<std::thread> threads;
        for (size_t k = 0; k < devs.size(); ++k)
            if (!share[k].empty()) threads.emplace_back(worker, std::ref(sh), static_cast<int>(devs[k]), std::move(share[k]));
        for (auto &t : threads) t.join();
        if (stats) {
            // (filled in a local copy, then at most the caller's struct_size bytes go out: include/aptgpu.h)
            aptgpu_batch_stats full{};
            const uint32_t want = stats->struct_size;
            full.struct_size = want;
            full.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            full.samples = total_samples;
            full.h2d_bytes = sh.h2d_bytes;
            full.d2h_bytes = sh.d2h_bytes;
            full.h2d_seconds = sh.h2d_seconds;
            full.d2h_seconds = sh.d2h_seconds;
            full.workers = static_cast<int32_t>(threads.size());
            full.recordings_per_call = sh.per_call;
            full.gate_wait_seconds = sh.gate_wait_seconds;
            full.setup_seconds = sh.setup_seconds;
            full.sessions_created = sh.sessions_created;
            full.workers_pinned = sh.workers_pinned;
            std::memcpy(stats, &full, std::min<size_t>(want, sizeof full));
        }
        if (sh.first_error_code != APTGPU_OK) {
            put_err(err, err_cap, sh.first_error);
            return sh.first_error_code;
        }
        return APTGPU_OK;
    });
}

}  // namespace

extern "C" {

int aptgpu_decode_batch(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                        int count, const float *const *signals, const size_t *n, const int32_t *devices, int n_devices,
                        int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                        aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    return decode_batch_impl(ctx, settings, input_rate_hz, sync, count, reinterpret_cast<const void *const *>(signals), n,
                             false, devices, n_devices, recordings_per_call, rows_out, n_out, status, results, stats, err,
                             err_cap);
}

int aptgpu_decode_batch_wav(const aptgpu_context *ctx, const aptgpu_settings *settings, uint32_t input_rate_hz, int sync,
                            int count, const void *const *wav_images, const size_t *wav_bytes, const int32_t *devices,
                            int n_devices, int recordings_per_call, float **rows_out, size_t *n_out, int32_t *status,
                            aptgpu_result *results, aptgpu_batch_stats *stats, char *err, size_t err_cap)
{
    return decode_batch_impl(ctx, settings, input_rate_hz, sync, count, wav_images, wav_bytes, true, devices, n_devices,
                             recordings_per_call, rows_out, n_out, status, results, stats, err, err_cap);
}

// Pinned host memory for inputs that should go over PCIe by direct DMA (a WAV reader can read a file
// straight into it).  Plain malloc'd buffers work everywhere in this API, at the runtime's staged rate.
void *aptgpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void aptgpu_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int aptgpu_host_affinity(int device, char *pci_bdf, int32_t *numa_node, char *cpulist, size_t cpulist_cap)
{
    std::string bdf;
    const HostAffinity h = affinity_of_device(device, &bdf);
    if (pci_bdf) std::snprintf(pci_bdf, 16, "%s", bdf.c_str());
    return put_affinity(h, numa_node, cpulist, cpulist_cap);
}

int aptgpu_host_affinity_from_sysfs(const char *sysfs_root, const char *pci_bdf, int32_t *numa_node, char *cpulist,
                                    size_t cpulist_cap)
{
    if (!sysfs_root || !pci_bdf) return APTGPU_ERR_INVALID;
    return put_affinity(affinity_from_sysfs(sysfs_root, pci_bdf), numa_node, cpulist, cpulist_cap);
}

}  // extern "C"
