// tools/ubench/lds_pat.hip <patterns.txt> — LDS read throughput of 8-byte reads for arbitrary per-lane start entries: each line of the
// file is "name e0 e1 ... e255" (entries of 8 bytes, one per thread of a 256-thread workgroup; -1 = lane idle).  Eight
// ds_read_b64 in flight per wave, three workgroups per CU: cycles per wave-read (round 5: the model behind the PHASE
// stage 1's thread assignment lists, apt_kernels_fused.hip fused_phase_table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k(const int *start, int rep, uint64_t *out, float *sink)
{
    __shared__ f2 z[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) z[i] = (f2){1.f * i, 2.f * i};
    __syncthreads();
    const int st = start[threadIdx.x];
    f2 acc = {0.f, 0.f};
    const uint64_t t0 = __builtin_readcyclecounter();
    if (st >= 0) {
        uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(z + st));  // LDS byte address
        for (int r = 0; r < rep; ++r) {
            f2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b64 %0, %8 offset:0\n\t"
                         "ds_read_b64 %1, %8 offset:8\n\t"
                         "ds_read_b64 %2, %8 offset:16\n\t"
                         "ds_read_b64 %3, %8 offset:24\n\t"
                         "ds_read_b64 %4, %8 offset:32\n\t"
                         "ds_read_b64 %5, %8 offset:40\n\t"
                         "ds_read_b64 %6, %8 offset:48\n\t"
                         "ds_read_b64 %7, %8 offset:56\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(addr)
                         : "memory");
            acc += v0 + v2 + v4 + v6 + v1 + v3 + v5 + v7;
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc.x == 123.456f) sink[0] = acc.y;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    int *d_start;
    uint64_t *d_out;
    float *d_sink;
    hipMalloc(&d_start, 256 * 4);
    hipMalloc(&d_out, 8 * 4096);
    hipMalloc(&d_sink, 64);
    std::ifstream f(argv[1]);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string name;
        ss >> name;
        std::vector<int> st(256, -1);
        int waves = 0;
        for (int t = 0; t < 256; ++t) ss >> st[t];
        for (int w = 0; w < 4; ++w) {
            bool any = false;
            for (int e = 0; e < 64; ++e) any = any || st[64 * w + e] >= 0;
            waves += any;
        }
        hipMemcpy(d_start, st.data(), 256 * 4, hipMemcpyHostToDevice);
        const int rep = 400, wgs_per_cu = 3, blocks = 256 * wgs_per_cu;
        uint64_t c[8];
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_start, rep, d_out, d_sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(c, d_out, 64, hipMemcpyDeviceToHost);
        // wave 0's clock over its rep x 8 reads while wgs_per_cu x waves waves share the CU's LDS
        printf("%-40s %6.2f cycles per 8-byte wave-read (%d active waves per workgroup)\n", name.c_str(),
               double(c[0]) / (double(wgs_per_cu) * waves * rep * 8), waves);
    }
    return 0;
}
