// Probe: can a 1024-thread / 147 KB-LDS workgroup start promptly while a persistent kernel
// occupies every other CU's LDS, if that kernel vacates one CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ uint32_t cu_key()
{
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    return ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
}
__global__ void __launch_bounds__(256) big(float *out, int iters, int reserve, uint32_t *landed)
{
    extern __shared__ float lds[];
    const uint32_t key = cu_key();
    if ((reserve == 1 && key == 0) || (reserve == 2 && key == 0x700) || (reserve == 3 && (key & 0xff) == 0) || (reserve == 4 && (key >> 8) == 7 && (key & 0x1f) == 0)) { if (threadIdx.x == 0) atomicAdd(landed, 1u); return; }
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) { lds[threadIdx.x] = a; a = a * 1.0001f + lds[(threadIdx.x + 1) & 255]; }
    if (a == 1234.5f) out[0] = a;
}
__global__ void __launch_bounds__(1024) small(uint32_t *info, long long *t)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) { info[0] = cu_key(); t[0] = wall_clock64(); }
}
int main()
{
    float *out; uint32_t *landed, *info; long long *t;
    hipMalloc(&out, 4); hipMalloc(&landed, 4); hipMalloc(&info, 4); hipMalloc(&t, 8);
    hipFuncSetAttribute((const void*)big, hipFuncAttributeMaxDynamicSharedMemorySize, 51504);
    hipFuncSetAttribute((const void*)small, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipStream_t a, b; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
    hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    for (int reserve = 4; reserve <= 4; ++reserve) for (int rep = 0; rep < 10; ++rep) {
        hipMemset(landed, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, a);
        hipLaunchKernelGGL(big, dim3(768), dim3(256), 51504, a, out, 20000, reserve, landed);
        hipEventRecord(e1, a);
        // give the big kernel time to fill the chip, then launch the LDS-heavy single workgroup
        hipEventRecord(e2, b);
        hipLaunchKernelGGL(small, dim3(1), dim3(1024), 150000, b, info, t);
        hipEventRecord(e3, b);
        hipDeviceSynchronize();
        float tb, ts, gap; hipEventElapsedTime(&tb, e0, e1); hipEventElapsedTime(&ts, e2, e3); hipEventElapsedTime(&gap, e0, e3);
        uint32_t hl, hi2; hipMemcpy(&hl, landed, 4, hipMemcpyDeviceToHost); hipMemcpy(&hi2, info, 4, hipMemcpyDeviceToHost);
        printf("reserve=%d big=%.1f us  small(start..end)=%.1f us  small done at %.1f us after big start  landed_on_reserved=%u small_cu=%03x\n",
               reserve, tb * 1e3, ts * 1e3, gap * 1e3, hl, hi2);
    }
    return 0;
}
