// download.hip -- every result leaves the device through pinned memory of the library's own, large ones pipelined.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ctx.hpp"

using namespace fpca;

namespace fpca {

constexpr size_t DL_CHUNK = (size_t)8 << 20, DL_SLOTS = 4, DL_PIN_BYTES = DL_CHUNK * DL_SLOTS;

// Every result leaves the device through pinned memory of our own -- a direct copy into the caller's pageable buffer makes
// the runtime register those pages for DMA, and when the caller later frees them (a Python loop dropping the previous
// result) the invalidation stalls the next submission by 20-30 ms.  Large results (U: 80 MB at 500,000 x 20) are
// pipelined: the column-major image is cut into DL_CHUNK-byte pieces that cycle through DL_SLOTS pinned slots; while piece
// i+1 is on the wire, worker threads scatter piece i into the caller's U (memcpy) and Px (scaled copy, randompca.cpp:207)
// -- first-touch page faults of fresh output arrays included, which is what a single-threaded copy spends its time on.
// d_img: device, column-major N x ncols with leading dimension N.
void staged_download(fpca_ctx *c_, const double *d_img, uint64_t N, int ncols, double *host, int64_t ld, double *host2, int64_t ld2,
                  const double *scale)
{
   if ((!host && !host2) || N == 0 || ncols <= 0) return;
   const size_t total = (size_t)N * ncols; // doubles
   if (!c_->dl_pin) HIP_CHECK(hipHostMalloc(&c_->dl_pin, DL_PIN_BYTES, hipHostMallocDefault));
   const double *pin = static_cast<const double *>(c_->dl_pin);
   // flat range [lo, hi) of the image, whose first element sits at src: column by column into the caller's matrices
   auto scatter = [&](size_t lo, size_t hi, const double *src) {
      for (size_t i = lo; i < hi;) {
         const size_t col = i / N, row = i - col * N, n = std::min(hi - i, (size_t)N - row);
         if (host) std::memcpy(host + col * (size_t)ld + row, src, n * sizeof(double));
         if (host2) {
            const double sc = scale[col];
            double *o = host2 + col * (size_t)ld2 + row;
            for (size_t j = 0; j < n; j++) o[j] = src[j] * sc;
         }
         src += n;
         i += n;
      }
   };
   constexpr size_t CH = DL_CHUNK / sizeof(double);
   if (total <= CH) {
      HIP_CHECK(hipMemcpyAsync(c_->dl_pin, d_img, total * sizeof(double), hipMemcpyDeviceToHost, c_->stream));
      HIP_CHECK(hipStreamSynchronize(c_->stream));
      scatter(0, total, pin);
      return;
   }
   for (hipEvent_t &e : c_->dl_ev)
      if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
   const size_t nch = (total + CH - 1) / CH;
   // helper threads: for results of two pieces and more (a single piece is scattered by the caller: the copy is over before a
   // thread has started), as many as this process may run at once (cgroup quota / affinity, not the host's hardware threads), at most 8;
   // they sleep on a condition variable until their piece has landed -- no spinning beside the thread that issues the copies
   static const unsigned cpus = usable_cpus();
   const int T = nch >= 2 ? (int)std::max(1u, std::min(8u, cpus > 1 ? cpus - 1 : 1u)) : 0;
   if (T == 0) {
      for (size_t c = 0; c < nch; c++) {
         const size_t c0 = c * CH, len = std::min(CH, total - c0);
         HIP_CHECK(hipMemcpyAsync(c_->dl_pin, d_img + c0, len * sizeof(double), hipMemcpyDeviceToHost, c_->stream));
         HIP_CHECK(hipStreamSynchronize(c_->stream));
         scatter(c0, c0 + len, pin);
      }
      return;
   }
   // (waiting = a short spin on an atomic -- a piece lands every ~150 us, and with T <= cpus - 1 helpers every thread has a CPU
   //  of its own -- then a sleep on the condition variable: nobody spins through a stall of the copy engine or a descheduled peer)
   std::mutex mu;
   std::condition_variable cv_ready, cv_done;
   std::atomic<size_t> ready(0);              // pieces [0, ready) have landed in their slots
   std::vector<std::atomic<int>> done(nch);   // workers finished with piece c
   for (auto &x : done) x.store(0);
   auto wait_for = [&](std::condition_variable &cv, auto &&pred) {
      for (int spin = 0; spin < 40000; spin++) { // (~0.5 ms: three pieces' worth; a longer wait is a stall, and stalls are slept through)
         if (pred()) return;
         __builtin_ia32_pause();
      }
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, pred);
   };
   std::vector<std::thread> workers;
   for (int t = 0; t < T; t++)
      workers.emplace_back([&, t] {
         for (size_t c = 0; c < nch; c++) {
            wait_for(cv_ready, [&] { return ready.load(std::memory_order_acquire) > c; });
            const size_t c0 = c * CH, len = std::min(CH, total - c0);
            const size_t lo = c0 + len * t / T, hi = c0 + len * (t + 1) / T;
            scatter(lo, hi, pin + (c % DL_SLOTS) * CH + (lo - c0));
            if (done[c].fetch_add(1, std::memory_order_acq_rel) + 1 == T) {
               { std::lock_guard<std::mutex> lk(mu); } // (pairs with the sleeper's predicate check under the mutex)
               cv_done.notify_one();
            }
         }
      });
   auto publish = [&](size_t upto) {
      ready.store(upto, std::memory_order_release);
      { std::lock_guard<std::mutex> lk(mu); }
      cv_ready.notify_all();
   };
   try {
      for (size_t c = 0; c < nch; c++) {
         const size_t slot = c % DL_SLOTS, c0 = c * CH, len = std::min(CH, total - c0);
         if (c >= DL_SLOTS) // the slot's previous piece has been scattered by every worker
            wait_for(cv_done, [&] { return done[c - DL_SLOTS].load(std::memory_order_acquire) == T; });
         HIP_CHECK(hipMemcpyAsync(static_cast<char *>(c_->dl_pin) + slot * DL_CHUNK, d_img + c0, len * sizeof(double), hipMemcpyDeviceToHost,
                                  c_->stream));
         HIP_CHECK(hipEventRecord(c_->dl_ev[slot], c_->stream));
         if (c >= 1) {
            HIP_CHECK(hipEventSynchronize(c_->dl_ev[(c - 1) % DL_SLOTS]));
            publish(c);
         }
      }
      HIP_CHECK(hipEventSynchronize(c_->dl_ev[(nch - 1) % DL_SLOTS]));
   } catch (...) {
      publish(nch); // let the workers run out (what they copy is discarded with the error)
      for (auto &w : workers) w.join();
      throw;
   }
   publish(nch);
   for (auto &w : workers) w.join();
}

} // namespace fpca
