// tools/ubench/lds_phase.hip — the window reads of the PHASE stage 1 (k_fused, 44 100 Hz: l = 208, m = 735) as a
// microbenchmark: 8-byte LDS reads at entry c(u) + i, c(u) = ceil((rb + u*m) / l), with u = lane (consecutive outputs)
// or u = (lane * q) mod S (the lane permutation of phase_lane_stride), as single reads and as the two-address form the
// compiler merges consecutive taps into.  Prints LDS cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int TWO>
__global__ void __launch_bounds__(256) k(const uint32_t *start, int rep, uint64_t *out, float *sink)
{
    __shared__ f2 z[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) z[i] = (f2){1.f * i, 2.f * i};
    __syncthreads();
    const f2 *w = z + start[threadIdx.x];
    f2 acc = {0.f, 0.f};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 64; i += 2) {
            if (TWO) {
                const f2 a = w[i], b = w[i + 1];
                acc += a + b;
            } else {
                const f2 a = w[i];
                acc += a;
                asm volatile("" : "+v"(acc));
                const f2 b = w[i + 1];
                acc += b;
                asm volatile("" : "+v"(acc));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (acc.x == 123.456f) sink[0] = acc.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main()
{
    const int l = 208, m = 735, S = 208;
    uint32_t *d_start;
    uint64_t *d_out;
    float *d_sink;
    hipMalloc(&d_start, 256 * 4);
    hipMalloc(&d_out, 64);
    hipMalloc(&d_sink, 64);
    for (int q : {1, 25, 33, 2, 3, 5, 7, 9}) {
        for (int rb : {0, 100}) {
            std::vector<uint32_t> st(256, 0);
            for (int t = 0; t < 256; ++t) {
                const int u = t < S ? (t * q) % S : 0;
                st[t] = (rb + u * m + l - 1) / l;
            }
            hipMemcpy(d_start, st.data(), 256 * 4, hipMemcpyHostToDevice);
            for (int two = 0; two < 2; ++two) {
                const int rep = 200;
                for (int pass = 0; pass < 2; ++pass) {
                    if (two) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d_start, rep, d_out, d_sink);
                    else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d_start, rep, d_out, d_sink);
                    hipDeviceSynchronize();
                }
                uint64_t c;
                hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                // one workgroup per CU (256 blocks on 256 CUs): four waves share the CU's LDS; 64 reads per rep and wave
                printf("q %2d rb %3d %s: %.2f cycles per 8-byte wave-read (4 waves per CU)\n", q, rb, two ? "paired reads" : "single reads",
                       double(c) / (double(rep) * 64.0 * 4.0));
            }
        }
    }
    return 0;
}
