// Probe: which hardware-id fields identify a CU on this GPU?  Launches many workgroups,
// records (XCC_ID, HW_ID) per workgroup and prints the distinct combinations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void k(uint32_t *out)
{
    // burn a little time so the grid spreads over the whole chip
    float a = threadIdx.x;
    for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc + (a == 12345.f ? 1 : 0);
    }
}
int main()
{
    const int n = 4096;
    uint32_t *d;
    hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    std::vector<uint32_t> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::set<std::pair<uint32_t, uint32_t>> keys;
    uint32_t or_hw = 0, and_hw = ~0u, or_x = 0;
    for (int i = 0; i < n; ++i) { or_hw |= h[2*i]; and_hw &= h[2*i]; or_x |= h[2*i+1]; }
    printf("HW_ID varying bits: %08x  XCC_ID varying bits: %08x\n", or_hw & ~and_hw, or_x);
    // candidate key: xcc[3:0], se[15:13], sh[12], cu[11:8]
    std::set<uint32_t> k1, k2;
    for (int i = 0; i < n; ++i) {
        uint32_t hw = h[2*i], x = h[2*i+1] & 0xf;
        k1.insert((x << 16) | (hw & 0xff00));
        k2.insert((x << 16) | (hw & 0xfff00));
    }
    printf("distinct (xcc, hw[15:8]) = %zu ; distinct (xcc, hw[19:8]) = %zu\n", k1.size(), k2.size());
    std::map<uint32_t,int> per;
    for (int i = 0; i < n; ++i) per[((h[2*i+1]&0xf) << 16) | (h[2*i] & 0xff00)]++;
    int shown = 0;
    for (auto &kv : per) { if (shown++ < 12) printf("  key %05x : %d wgs\n", kv.first, kv.second); }
    return 0;
}
