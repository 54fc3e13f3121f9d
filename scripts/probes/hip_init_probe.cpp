// how long the HIP runtime takes to come up in a fresh process on this box, with nothing of ours linked in:
//   hipcc -O2 scripts/probes/hip_init_probe.cpp -o /tmp/hip_init_probe && /tmp/hip_init_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
   auto t0 = std::chrono::steady_clock::now();
   auto lap = [&](const char *w) {
      auto t = std::chrono::steady_clock::now();
      std::printf("%-28s %8.2f ms\n", w, std::chrono::duration<double>(t - t0).count() * 1e3);
      t0 = t;
   };
   int n = 0;
   (void)hipGetDeviceCount(&n);
   lap("hipGetDeviceCount");
   (void)hipSetDevice(0);
   (void)hipFree(nullptr);
   lap("hipSetDevice + hipFree(0)");
   void *p = nullptr;
   (void)hipMalloc(&p, 1 << 20);
   lap("first hipMalloc (1 MB)");
   void *q = nullptr;
   (void)hipMalloc(&q, (size_t)12 << 30);
   lap("hipMalloc 12 GB");
   void *h = nullptr;
   (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
   lap("hipHostMalloc 64 MB");
   hipStream_t s;
   (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
   lap("hipStreamCreate");
   (void)hipMemsetAsync(q, 0x55, (size_t)12 << 30, s);
   (void)hipStreamSynchronize(s);
   lap("memset 12 GB");
   return 0;
}
