// tools/ubench/lds_bw.hip — LDS read throughput of 8-byte reads for a given per-lane start pattern (entries of 8 bytes),
// many independent reads in flight, several waves per SIMD: bytes per clock and CU.  Patterns: consecutive entries;
// a constant stride; the window starts of the PHASE stage 1 at 44 100 Hz (c(t) = ceil((rb + t*735) / 208)), with and
// without a lane permutation u = (t*q) mod 208.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k(const uint32_t *start, int rep, uint64_t *out, float *sink)
{
    __shared__ f2 z[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) z[i] = (f2){1.f * i, 2.f * i};
    __syncthreads();
    uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(z + start[threadIdx.x]));  // LDS byte address
    f2 acc = {0.f, 0.f};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
        f2 v0, v1, v2, v3, v4, v5, v6, v7;
        // sixteen 8-byte reads (eight two-address instructions: consecutive taps), all issued before the first use
        asm volatile("ds_read2_b64 %0, %8 offset0:0 offset1:1\n\t"
                     "ds_read2_b64 %1, %8 offset0:2 offset1:3\n\t"
                     "ds_read2_b64 %2, %8 offset0:4 offset1:5\n\t"
                     "ds_read2_b64 %3, %8 offset0:6 offset1:7\n\t"
                     "ds_read2_b64 %4, %8 offset0:8 offset1:9\n\t"
                     "ds_read2_b64 %5, %8 offset0:10 offset1:11\n\t"
                     "ds_read2_b64 %6, %8 offset0:12 offset1:13\n\t"
                     "ds_read2_b64 %7, %8 offset0:14 offset1:15\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v0), "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v1),
                       "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v2), "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v3),
                       "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v4), "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v5),
                       "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v6), "=&v"(*(float __attribute__((ext_vector_type(4))) *)&v7)
                     : "v"(addr)
                     : "memory");
        acc += v0 + v2 + v4 + v6;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc.x == 123.456f) sink[0] = acc.y;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    uint32_t *d_start;
    uint64_t *d_out;
    float *d_sink;
    hipMalloc(&d_start, 256 * 4);
    hipMalloc(&d_out, 8 * 4096);
    hipMalloc(&d_sink, 64);
    auto run = [&](const char *name, const std::vector<uint32_t> &st) {
        hipMemcpy(d_start, st.data(), 256 * 4, hipMemcpyHostToDevice);
        const int rep = 400, wgs_per_cu = 3, blocks = 256 * wgs_per_cu;
        uint64_t c[8];
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_start, rep, d_out, d_sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(c, d_out, 64, hipMemcpyDeviceToHost);
        // per CU: wgs_per_cu workgroups x 4 waves x rep x 16 reads x 512 bytes in c[0] cycles (the workgroups run concurrently)
        const double bytes = double(wgs_per_cu) * 4 * rep * 16 * 512;
        printf("%-46s %6.1f bytes per clock and CU (%.2f cycles per 8-byte wave-read)\n", name, bytes / double(c[0]),
               double(c[0]) / (double(wgs_per_cu) * 4 * rep * 16));
    };
    std::vector<uint32_t> st(256);
    // a thread's 16 outputs are separate measurements in the real kernel; here one region
    for (int t = 0; t < 256; ++t) st[t] = t; run("consecutive entries", st);
    for (int s : {2, 3, 4, 5, 7, 25}) { for (int t = 0; t < 256; ++t) st[t] = (t * s) % 3000; char b[64]; snprintf(b, 64, "stride %d entries", s); run(b, st); }
    for (int q : {1, 25, 33, 5, 3}) {
        for (int rb : {0, 100}) {
            for (int t = 0; t < 256; ++t) { const int u = t < 208 ? (t * q) % 208 : 0; st[t] = (rb + u * 735 + 207) / 208; }
            char b[64]; snprintf(b, 64, "PHASE 44.1 kHz starts, lane stride q = %d, rb = %d", q, rb); run(b, st);
        }
    }
    // ---- search: which assignment of a stride's 208 outputs to the lanes reads fastest?  Hill climbing by swaps, the cost
    // measured on the device over four tile phases rb.
    {
        auto measure = [&](const std::vector<int> &perm) -> double {
            double tot = 0;
            for (int rb : {0, 52, 104, 156}) {
                for (int t = 0; t < 256; ++t) st[t] = t < 208 ? (rb + perm[t] * 735 + 207) / 208 : 0;
                hipMemcpy(d_start, st.data(), 256 * 4, hipMemcpyHostToDevice);
                const int rep = 100, blocks = 256 * 3;
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_start, rep, d_out, d_sink);
                hipDeviceSynchronize();
                uint64_t c;
                hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                tot += double(c) / (3.0 * 4 * rep * 16);
            }
            return tot / 4;
        };
        std::vector<int> perm(208);
        for (int t = 0; t < 208; ++t) perm[t] = t;
        double best = measure(perm);
        printf("search: identity %.3f cycles per wave-read\n", best);
        uint64_t rng = 88172645463325252ull;
        auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        for (int it = 0; it < 1500; ++it) {
            const int a = rnd() % 208, b = rnd() % 208;
            if (a == b) continue;
            std::swap(perm[a], perm[b]);
            const double c = measure(perm);
            if (c < best - 0.002) best = c;
            else std::swap(perm[a], perm[b]);
            if (it % 500 == 499) printf("search: iteration %d best %.3f\n", it + 1, best);
        }
        printf("search: best %.3f; permutation:", best);
        for (int t = 0; t < 208; ++t) printf(" %d", perm[t]);
        printf("\n");
    }
    for (int t = 0; t < 256; ++t) st[t] = (t * 7) / 2; run("3.5 entries per lane (floor)", st);
    for (int t = 0; t < 256; ++t) st[t] = ((t * 7) / 2) | 1; run("3.5 entries per lane, odd", st);
    return 0;
}
