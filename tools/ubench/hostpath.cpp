// tools/ubench/hostpath.cpp — what is the fastest way to bring a PAGEABLE host buffer to the GPU and rows back?
//   hipcc -O2 -o hostpath tools/ubench/hostpath.cpp -lpthread && ./hostpath
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main()
{
    const size_t bytes = 43200000ull * 4;  // one 15-minute 48 kHz recording as f32
    const int reps = 6;
    char *d;
    CK(hipMalloc(&d, bytes));
    std::vector<char *> src(reps);
    for (auto &p : src) { p = static_cast<char *>(malloc(bytes)); memset(p, 1, bytes); }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // (a) plain hipMemcpy from pageable memory
    { auto a = clk::now(); for (int r = 0; r < reps; ++r) CK(hipMemcpy(d, src[r], bytes, hipMemcpyHostToDevice)); auto b = clk::now();
      printf("hipMemcpy H2D pageable:                 %.2f GB/s\n", reps * bytes / secs(a, b) / 1e9); }
    { auto a = clk::now(); for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(d, src[r], bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); auto b = clk::now();
      printf("hipMemcpyAsync H2D pageable:            %.2f GB/s\n", reps * bytes / secs(a, b) / 1e9); }
    // (b) register, copy, unregister
    { double tr = 0, tc = 0, tu = 0;
      for (int r = 0; r < reps; ++r) {
          auto a = clk::now(); CK(hipHostRegister(src[r], bytes, hipHostRegisterDefault)); auto b = clk::now();
          CK(hipMemcpyAsync(d, src[r], bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); auto c = clk::now();
          CK(hipHostUnregister(src[r])); auto e = clk::now();
          tr += secs(a, b); tc += secs(b, c); tu += secs(c, e);
      }
      printf("hipHostRegister %.2f ms, DMA %.2f GB/s, unregister %.2f ms per 173 MB -> %.2f GB/s all in\n", 1e3 * tr / reps,
             reps * bytes / tc / 1e9, 1e3 * tu / reps, reps * bytes / (tr + tc + tu) / 1e9); }
    // (c) pinned source
    { char *p; CK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); memset(p, 2, bytes);
      auto a = clk::now(); for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); auto b = clk::now();
      printf("hipMemcpyAsync H2D pinned:              %.2f GB/s\n", reps * bytes / secs(a, b) / 1e9);
      a = clk::now(); for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); b = clk::now();
      printf("hipMemcpyAsync D2H pinned:              %.2f GB/s\n", reps * bytes / secs(a, b) / 1e9);
      // both directions at once on two streams
      hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); char *d2; CK(hipMalloc(&d2, bytes)); char *p2; CK(hipHostMalloc(&p2, bytes, hipHostMallocDefault));
      a = clk::now(); for (int r = 0; r < reps; ++r) { CK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(p2, d2, bytes, hipMemcpyDeviceToHost, s2)); }
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); b = clk::now();
      printf("H2D + D2H pinned, concurrently:         %.2f GB/s each way\n", reps * bytes / secs(a, b) / 1e9);
      // (d) staged: T threads copy pageable -> pinned ring of chunks, DMA per chunk
      for (int T : {1, 2, 4, 8}) {
          const size_t chunk = 8u << 20;
          const int ring = 8;
          char *stage; CK(hipHostMalloc(&stage, chunk * ring, hipHostMallocDefault));
          std::vector<hipEvent_t> ev(ring); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          a = clk::now();
          size_t issued = 0;
          for (int r = 0; r < reps; ++r) {
              const size_t nch = (bytes + chunk - 1) / chunk;
              for (size_t c = 0; c < nch; ++c, ++issued) {
                  const int slot = issued % ring;
                  if (issued >= (size_t)ring) CK(hipEventSynchronize(ev[slot]));
                  const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                  // T threads split the chunk
                  std::vector<std::thread> th;
                  for (int t = 1; t < T; ++t) th.emplace_back([&, t] { size_t a0 = len * t / T, a1 = len * (t + 1) / T; memcpy(stage + slot * chunk + a0, src[r] + off + a0, a1 - a0); });
                  memcpy(stage + slot * chunk, src[r] + off, len / T);
                  for (auto &x : th) x.join();
                  CK(hipMemcpyAsync(d + off, stage + slot * chunk, len, hipMemcpyHostToDevice, s));
                  CK(hipEventRecord(ev[slot], s));
              }
          }
          CK(hipStreamSynchronize(s)); b = clk::now();
          printf("staged ring, %d copier thread(s):        %.2f GB/s\n", T, reps * bytes / secs(a, b) / 1e9);
          CK(hipHostFree(stage));
      }
      // D2H into pageable
      char *dst = static_cast<char *>(malloc(bytes)); memset(dst, 0, bytes);
      a = clk::now(); for (int r = 0; r < reps; ++r) CK(hipMemcpy(dst, d, bytes, hipMemcpyDeviceToHost)); b = clk::now();
      printf("hipMemcpy D2H pageable:                 %.2f GB/s\n", reps * bytes / secs(a, b) / 1e9);
    }
    printf("host threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
