// upload_probe.cpp -- round 5: how fast can a page-cache-resident .bed reach HBM?  (flashpca.cpp:589-604 reads the file
// block by block into host memory, here it is streamed to the device once.)  Four ways of moving FILE_GB of a file in /tmp to the GPU:
//   A  the shipped pipeline: 2 pinned slots of 64 MB, 16 reader threads spawned per chunk, read i+1 under copy i
//   B  persistent reader threads over 8 MB pieces, 4 slots, the copy of a chunk enqueued when its last piece has landed
//   C  mmap + hipHostRegister of 64 MB windows (the DMA engine reads the page cache itself, no CPU copy), 3 windows in flight
//   D  mmap + ONE hipHostRegister of the whole file + one copy
// build + run (GPU box): hipcc -O2 -std=c++17 -o /tmp/upload_probe scripts/upload_probe.cpp -lpthread && /tmp/upload_probe [GB] [threads]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                     \
   do {                                                                                           \
      hipError_t e_ = (x);                                                                        \
      if (e_ != hipSuccess) {                                                                     \
         std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
         std::exit(1);                                                                            \
      }                                                                                           \
   } while (0)

static double now()
{
   return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void pread_all(int fd, uint8_t *dst, uint64_t n, off_t off)
{
   uint64_t got = 0;
   while (got < n) {
      const ssize_t k = pread(fd, dst + got, n - got, off + (off_t)got);
      if (k <= 0) {
         std::perror("pread");
         std::exit(1);
      }
      got += (uint64_t)k;
   }
}

int main(int argc, char **argv)
{
   const double gb = argc > 1 ? atof(argv[1]) : 4.0;
   const int nthreads = argc > 2 ? atoi(argv[2]) : 16;
   const uint64_t CH = 64ull << 20;
   const uint64_t total = (uint64_t)(gb * (1ull << 30)) / CH * CH;
   const uint64_t nch = total / CH;
   const char *path = "/tmp/upload_probe.bin";
   {
      int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0600);
      std::vector<uint8_t> buf(CH);
      uint64_t x = 88172645463325252ull;
      for (uint64_t i = 0; i < CH / 8; i++) {
         x ^= x << 13, x ^= x >> 7, x ^= x << 17;
         reinterpret_cast<uint64_t *>(buf.data())[i] = x;
      }
      for (uint64_t c = 0; c < nch; c++) {
         buf[0] = (uint8_t)c;
         if (write(fd, buf.data(), CH) != (ssize_t)CH) {
            std::perror("write");
            return 1;
         }
      }
      close(fd);
   }
   CK(hipSetDevice(0));
   uint8_t *dev = nullptr;
   CK(hipMalloc(&dev, total));
   hipStream_t s;
   CK(hipStreamCreate(&s));
   const int fd = open(path, O_RDONLY);
   std::printf("file %.1f GB in %s, %d reader threads, chunk 64 MB\n", total / 1e9, path, nthreads);

   // ---- A ----------------------------------------------------------------------------------------------------------------
   for (int rep = 0; rep < 2; rep++) {
      uint8_t *bounce[2];
      hipEvent_t done[2];
      for (int i = 0; i < 2; i++) {
         CK(hipHostMalloc(&bounce[i], CH, hipHostMallocDefault));
         CK(hipEventCreate(&done[i]));
      }
      const double t0 = now();
      for (uint64_t c = 0; c < nch; c++) {
         const int slot = c & 1;
         CK(hipEventSynchronize(done[slot]));
         std::vector<std::thread> th;
         auto work = [&](int t) { pread_all(fd, bounce[slot] + CH * t / nthreads, CH * (t + 1) / nthreads - CH * t / nthreads, (off_t)(c * CH + CH * t / nthreads)); };
         for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
         work(0);
         for (auto &x : th) x.join();
         CK(hipMemcpyAsync(dev + c * CH, bounce[slot], CH, hipMemcpyHostToDevice, s));
         CK(hipEventRecord(done[slot], s));
      }
      CK(hipStreamSynchronize(s));
      const double t = now() - t0;
      std::printf("A  threads per chunk, 2 slots           : %7.1f ms  %5.1f GB/s\n", t * 1e3, total / t / 1e9);
      for (int i = 0; i < 2; i++) {
         CK(hipHostFree(bounce[i]));
         CK(hipEventDestroy(done[i]));
      }
   }

   // ---- B ----------------------------------------------------------------------------------------------------------------
   for (int K : {3, 4, 6})
      for (uint64_t PIECE : {4ull << 20, 8ull << 20}) {
         std::vector<uint8_t *> slot(K);
         std::vector<hipEvent_t> done(K);
         for (int i = 0; i < K; i++) {
            CK(hipHostMalloc(&slot[i], CH, hipHostMallocDefault));
            CK(hipEventCreate(&done[i]));
         }
         const uint64_t ppc = CH / PIECE, npieces = nch * ppc;
         std::vector<std::atomic<int>> landed(nch);
         for (auto &a : landed) a = 0;
         std::atomic<uint64_t> next(0);
         std::atomic<int64_t> free_upto(K - 1); // readers may fill chunks <= free_upto
         const double t0 = now();
         std::vector<std::thread> th;
         for (int t = 0; t < nthreads; t++)
            th.emplace_back([&] {
               for (;;) {
                  const uint64_t p = next.fetch_add(1);
                  if (p >= npieces) return;
                  const uint64_t c = p / ppc, q = p % ppc;
                  while ((int64_t)c > free_upto.load(std::memory_order_acquire)) std::this_thread::yield();
                  pread_all(fd, slot[c % K] + q * PIECE, PIECE, (off_t)(c * CH + q * PIECE));
                  landed[c].fetch_add(1, std::memory_order_release);
               }
            });
         for (uint64_t c = 0; c < nch; c++) {
            while (landed[c].load(std::memory_order_acquire) < (int)ppc) std::this_thread::yield();
            CK(hipMemcpyAsync(dev + c * CH, slot[c % K], CH, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(done[c % K], s));
            if (c >= 1) { // the copy before this one has (nearly) finished: its slot may be refilled
               CK(hipEventSynchronize(done[(c - 1) % K]));
               free_upto.store((int64_t)(c - 1 + K), std::memory_order_release);
            }
         }
         CK(hipStreamSynchronize(s));
         const double t = now() - t0;
         for (auto &x : th) x.join();
         std::printf("B  persistent readers, %d slots, %llu MB pieces: %7.1f ms  %5.1f GB/s\n", K, (unsigned long long)(PIECE >> 20), t * 1e3, total / t / 1e9);
         for (int i = 0; i < K; i++) {
            CK(hipHostFree(slot[i]));
            CK(hipEventDestroy(done[i]));
         }
      }

   // ---- reading alone (no GPU): the page-cache -> pinned memcpy ceiling -----------------------------------------------------
   {
      uint8_t *b = nullptr;
      CK(hipHostMalloc(&b, CH * 4, hipHostMallocDefault));
      std::atomic<uint64_t> next(0);
      const uint64_t PIECE = 8ull << 20, npieces = total / PIECE;
      const double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; t++)
         th.emplace_back([&] {
            for (;;) {
               const uint64_t p = next.fetch_add(1);
               if (p >= npieces) return;
               pread_all(fd, b + (p % 32) * PIECE, PIECE, (off_t)(p * PIECE));
            }
         });
      for (auto &x : th) x.join();
      const double t = now() - t0;
      std::printf("   pread alone into pinned memory       : %7.1f ms  %5.1f GB/s\n", t * 1e3, total / t / 1e9);
      // the copy alone
      const double t1 = now();
      for (uint64_t c = 0; c < nch; c++) CK(hipMemcpyAsync(dev + c * CH, b + (c % 4) * CH, CH, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      const double tc = now() - t1;
      std::printf("   H2D copies alone from pinned memory  : %7.1f ms  %5.1f GB/s\n", tc * 1e3, total / tc / 1e9);
      CK(hipHostFree(b));
   }

   // ---- C / D: the DMA engine reads the page cache --------------------------------------------------------------------------
   void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
   if (map == MAP_FAILED) {
      std::perror("mmap");
      return 0;
   }
   {
      const double t0 = now();
      hipError_t e = hipHostRegister(map, CH, hipHostRegisterDefault);
      const double t = now() - t0;
      if (e != hipSuccess) {
         std::printf("C  hipHostRegister of a read-only file mapping: %s -- not available\n", hipGetErrorString(e));
         (void)hipGetLastError();
         // a private writable mapping instead (copy-on-write pages are never written: still the page cache's pages)
         munmap(map, total);
         map = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);
         e = hipHostRegister(map, CH, hipHostRegisterDefault);
         if (e != hipSuccess) {
            std::printf("C  ... nor of a private writable mapping: %s\n", hipGetErrorString(e));
            return 0;
         }
         std::printf("C  a private writable mapping registers\n");
      } else
         std::printf("C  hipHostRegister of the first 64 MB window: %.2f ms\n", t * 1e3);
      CK(hipHostUnregister(map));
   }
   {
      const int W = 3;
      hipEvent_t done[W];
      for (int i = 0; i < W; i++) CK(hipEventCreate(&done[i]));
      const double t0 = now();
      double treg = 0;
      for (uint64_t c = 0; c < nch; c++) {
         if (c >= W) {
            CK(hipEventSynchronize(done[c % W]));
            CK(hipHostUnregister((uint8_t *)map + (c - W) * CH));
         }
         const double r0 = now();
         CK(hipHostRegister((uint8_t *)map + c * CH, CH, hipHostRegisterDefault));
         treg += now() - r0;
         CK(hipMemcpyAsync(dev + c * CH, (uint8_t *)map + c * CH, CH, hipMemcpyHostToDevice, s));
         CK(hipEventRecord(done[c % W], s));
      }
      CK(hipStreamSynchronize(s));
      const double t = now() - t0;
      for (uint64_t c = nch >= W ? nch - W : 0; c < nch; c++) CK(hipHostUnregister((uint8_t *)map + c * CH));
      std::printf("C  registered 64 MB windows, %d in flight : %7.1f ms  %5.1f GB/s   (registering: %.1f ms in all)\n", W, t * 1e3, total / t / 1e9, treg * 1e3);
   }
   // E: window size of the registered-mapping route, and of plain pinned-memory copies (is it the size of a copy that matters?)
   for (uint64_t WSZ : {64ull << 20, 256ull << 20, 1024ull << 20}) {
      const uint64_t nw = total / WSZ;
      if (nw < 2) continue;
      const int W = 2;
      hipEvent_t done[W];
      for (int i = 0; i < W; i++) CK(hipEventCreate(&done[i]));
      const double t0 = now();
      double treg = 0;
      for (uint64_t c = 0; c < nw; c++) {
         if (c >= W) {
            CK(hipEventSynchronize(done[c % W]));
            const double r0 = now();
            CK(hipHostUnregister((uint8_t *)map + (c - W) * WSZ));
            treg += now() - r0;
         }
         const double r0 = now();
         CK(hipHostRegister((uint8_t *)map + c * WSZ, WSZ, hipHostRegisterDefault));
         treg += now() - r0;
         CK(hipMemcpyAsync(dev + c * WSZ, (uint8_t *)map + c * WSZ, WSZ, hipMemcpyHostToDevice, s));
         CK(hipEventRecord(done[c % W], s));
      }
      CK(hipStreamSynchronize(s));
      const double t = now() - t0;
      for (uint64_t c = nw >= W ? nw - W : 0; c < nw; c++) CK(hipHostUnregister((uint8_t *)map + c * WSZ));
      std::printf("E  registered %4llu MB windows, 2 in flight: %7.1f ms  %5.1f GB/s   (register + unregister: %.1f ms in all)\n",
                  (unsigned long long)(WSZ >> 20), t * 1e3, nw * WSZ / t / 1e9, treg * 1e3);
      uint8_t *pin = nullptr;
      CK(hipHostMalloc(&pin, WSZ, hipHostMallocDefault));
      std::memset(pin, 1, WSZ);
      const double t1 = now();
      for (uint64_t c = 0; c < nw; c++) CK(hipMemcpyAsync(dev + c * WSZ, pin, WSZ, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      const double tp = now() - t1;
      std::printf("   copies of %4llu MB from hipHostMalloc memory : %7.1f ms  %5.1f GB/s\n", (unsigned long long)(WSZ >> 20), tp * 1e3, nw * WSZ / tp / 1e9);
      CK(hipHostFree(pin));
   }
   {
      std::vector<unsigned char> vec((total + 4095) / 4096);
      const double m0 = now();
      const int mr = mincore(map, total, vec.data());
      uint64_t res = 0;
      for (unsigned char v : vec) res += v & 1;
      std::printf("   mincore over the mapping: rc %d, %.1f %% resident, %.2f ms\n", mr, 100.0 * res / vec.size(), (now() - m0) * 1e3);
   }
   {
      const double t0 = now();
      CK(hipHostRegister(map, total, hipHostRegisterDefault));
      const double t1 = now();
      CK(hipMemcpyAsync(dev, map, total, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      const double t2 = now();
      CK(hipHostUnregister(map));
      std::printf("D  one registration %.1f ms + one copy %.1f ms = %7.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, total / (t2 - t0) / 1e9);
   }
   unlink(path);
   return 0;
}
