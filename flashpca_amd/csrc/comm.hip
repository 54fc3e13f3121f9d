// comm.hip -- multi-GPU plumbing: RCCL loaded lazily, the three data-path collectives over whichever transport a
// context has (RCCL / a caller's sum / a caller's all-gather + reduce-scatter), the row-chunk plans of the overlapped
// exchange, and the self-test of the row-sharded solver's exchange.  The sum being sharded is svdwide.cpp:48-62.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "ctx.hpp"

using namespace fpca;

namespace fpca {

RcclApi &rccl()
{
   static RcclApi api;
   if (!api.handle) {
      const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char *n : names) {
         api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
         if (api.handle) break;
      }
      if (!api.handle) throw Error(FPCA_ECOMM, std::string("cannot load librccl: ") + dlerror());
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
      api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
      api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.handle, "ncclReduceScatter");
      api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
      api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
      api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.handle, "ncclCommGetAsyncError");
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.ReduceScatter || !api.AllGather || !api.CommDestroy)
         throw Error(FPCA_ECOMM, "librccl is missing expected symbols");
   }
   return api;
}

} // namespace fpca

static void refuse_dead(const fpca_ctx *c)
{
   if (c->comm_dead) throw Error(FPCA_ECOMM, "the transport of this context was abandoned after its ranks fell out of step; no further collective is issued");
}

void fpca_ctx::all_gather(const RowShard &sh, const double *slice, double *full, int b, hipStream_t s)
{
   refuse_dead(this);
   const size_t piece = (size_t)sh.plen * b, chunk = (size_t)sh.L * b;
   if (native_collectives()) {
      for (int c = 0; c < sh.nch; c++) {
         if (ag_fn) {
            if (ag_fn(coll_user, slice + c * piece, full + c * chunk, piece, (void *)s) != 0) throw Error(FPCA_ECOMM, "caller-supplied all-gather failed");
         } else
            RCCL_CHECK(rccl().AllGather(slice + c * piece, full + c * chunk, piece, ncclDouble, comm, s));
         coll_calls++;
         coll_bytes += piece * sizeof(double);
      }
      return;
   }
   // a caller-supplied transport only sums: every rank contributes its rows, zeros elsewhere
   HIP_CHECK(hipMemsetAsync(full, 0, (size_t)sh.full_rows() * b * sizeof(double), s));
   for (int c = 0; c < sh.nch; c++)
      HIP_CHECK(hipMemcpyAsync(full + c * chunk + (size_t)sh.rank * piece, slice + c * piece, piece * sizeof(double), hipMemcpyDeviceToDevice, s));
   allreduce(full, (uint64_t)sh.full_rows() * b, s);
}

void fpca_ctx::all_gather_bytes(const RowShard &sh, const int8_t *slice, int8_t *full, size_t bytes_per_row, hipStream_t s)
{
   refuse_dead(this);
   if (!native_collectives() || bytes_per_row % 8) throw Error(FPCA_ECOMM, "all_gather_bytes needs a transport with a real all-gather");
   const size_t piece = (size_t)sh.plen * bytes_per_row, chunk = (size_t)sh.L * bytes_per_row;
   for (int c = 0; c < sh.nch; c++) {
      if (ag_fn) { // (the caller's all-gather moves doubles: any 8-byte words travel unchanged)
         if (ag_fn(coll_user, reinterpret_cast<const double *>(slice + c * piece), reinterpret_cast<double *>(full + c * chunk), piece / 8, (void *)s) != 0)
            throw Error(FPCA_ECOMM, "caller-supplied all-gather failed");
      } else
         RCCL_CHECK(rccl().AllGather(slice + c * piece, full + c * chunk, piece, ncclInt8, comm, s));
      coll_calls++;
      coll_bytes += piece;
   }
}

void fpca_ctx::all_gather_small(const double *mine, double *all, size_t count, hipStream_t s)
{
   refuse_dead(this);
   if (!native_collectives()) throw Error(FPCA_ECOMM, "all_gather_small needs a transport with a real all-gather");
   if (ag_fn) {
      if (ag_fn(coll_user, mine, all, count, (void *)s) != 0) throw Error(FPCA_ECOMM, "caller-supplied all-gather failed");
   } else
      RCCL_CHECK(rccl().AllGather(mine, all, count, ncclDouble, comm, s));
   coll_calls++;
   coll_bytes += count * sizeof(double);
}

void fpca_ctx::reduce_scatter(const RowShard &sh, double *full, double *slice, int b, hipStream_t s, int only_chunk)
{
   refuse_dead(this);
   const size_t piece = (size_t)sh.plen * b, chunk = (size_t)sh.L * b;
   // (test builds: the n-th reduce-scatter of this context fails -- on every rank, the call sequence being the same everywhere)
   // (FPCA_DEBUG_RS_FAIL_RANK = r: on rank r ONLY -- the asymmetric failure the hook contract forbids: the job must end, not hang)
   if (const char *inj = FPCA_TEST_ENV("FPCA_DEBUG_RS_FAIL"))
      if (++dbg_rs_calls == atol(inj) && (!FPCA_TEST_ENV("FPCA_DEBUG_RS_FAIL_RANK") || atoi(FPCA_TEST_ENV("FPCA_DEBUG_RS_FAIL_RANK")) == rank)) throw Error(FPCA_ECOMM, "injected failure of reduce-scatter call " + std::to_string(dbg_rs_calls) + " (FPCA_DEBUG_RS_FAIL)");
   if (native_collectives()) {
      for (int c = 0; c < sh.nch; c++) {
         if (only_chunk >= 0 && c != only_chunk) continue;
         if (rs_fn) {
            if (rs_fn(coll_user, full + c * chunk, slice + c * piece, piece, (void *)s) != 0) throw Error(FPCA_ECOMM, "caller-supplied reduce-scatter failed");
         } else
            RCCL_CHECK(rccl().ReduceScatter(full + c * chunk, slice + c * piece, piece, ncclDouble, ncclSum, comm, s));
         coll_calls++;
         coll_bytes += piece * sizeof(double);
      }
      return;
   }
   allreduce(full, (uint64_t)sh.full_rows() * b, s);
   for (int c = 0; c < sh.nch; c++)
      HIP_CHECK(hipMemcpyAsync(slice + c * piece, full + c * chunk + (size_t)sh.rank * piece, piece * sizeof(double), hipMemcpyDeviceToDevice, s));
}

void fpca_ctx::allreduce(double *dbuf, uint64_t count, hipStream_t s)
{
   refuse_dead(this);
   coll_calls++;
   coll_bytes += count * sizeof(double);
   if (ar_fn) { // a caller-supplied hook wins over the built-in communicator (set after a failed / partial RCCL init)
      if (ar_fn(ar_user, dbuf, count, (void *)s) != 0) throw Error(FPCA_ECOMM, "caller-supplied all-reduce failed");
   } else if (comm)
      RCCL_CHECK(rccl().AllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, comm, s));
}

namespace fpca {

// Row chunks of Y for the overlapped all-reduce (built-in communicator only): the all-reduce of chunk i runs on the
// communication stream while K3 computes chunk i + 1, so only the last chunk's all-reduce is exposed.  Chunks are whole
// K3 row tiles.  Only for large N: measured on one GPU, two chunks cost +0.06 ms at N = 50k (each chunk is less than
// one round of workgroups) -- about what they would hide there -- and +0.07 ms of 23 ms at N = 500k.
// FPCA_AR_CHUNKS=n forces n (1 disables).
int ar_chunks(const fpca_ctx *c)
{
   if (!c->comm || c->ar_fn || !c->comm_stream) return 1;
   const char *env = FPCA_TEST_ENV("FPCA_AR_CHUNKS"); // read on every call: the tests switch it between contexts
   int n = c->N_pad >= 400000 ? 4 : c->N_pad >= 200000 ? 2 : 1;
   if (env && atoi(env) >= 1) n = std::min(atoi(env), 4);
   while (n > 1 && c->N_pad / n < 512) n--;
   return n;
}
// row chunks of the row-sharded solver's exchange (all-gather -> K2, K3 chunk by chunk -> reduce-scatter of chunk i under
// the computation of chunk i + 1): the same rule, for every transport that has real all-gather / reduce-scatter
int shard_chunks(const fpca_ctx *c)
{
   if (!c->native_collectives() || !c->comm_stream) return 1;
   const char *env = FPCA_TEST_ENV("FPCA_AR_CHUNKS");
   int n = c->N_pad >= 400000 ? 4 : c->N_pad >= 200000 ? 2 : 1;
   if (env && atoi(env) >= 1) n = std::min(atoi(env), 4);
   while (n > 1 && c->N_pad / n < 512) n--;
   return n;
}
uint64_t ar_chunk_begin(const fpca_ctx *c, int nchunks, int i) // multiples of 512 rows
{
   const uint64_t per = round_up((c->N_pad + nchunks - 1) / nchunks, 512);
   return std::min<uint64_t>(per * i, c->N_pad);
}

// The all-reduce of a finished Y, in the SAME sequence of collectives as the overlapped row chunks of the exact-integer
// path (ar_chunks depends on N and the communicator only): every rank issues identical calls whatever arithmetic it runs --
// a rank whose int8 buffers did not fit (FPCA_ACCUM_AUTO falls back to fp64 per rank) still matches the others.
void allreduce_rows(fpca_ctx *c, double *dY, int b, hipStream_t s)
{
   if (!c->multi()) return;
   const int nch = ar_chunks(c);
   if (nch <= 1) {
      c->allreduce(dY, (uint64_t)c->N_pad * b, s);
      return;
   }
   for (int i = 0; i < nch; i++) {
      const uint64_t r0 = ar_chunk_begin(c, nch, i), r1 = ar_chunk_begin(c, nch, i + 1);
      if (r1 > r0) RCCL_CHECK(rccl().AllReduce(dY + r0 * b, dY + r0 * b, (r1 - r0) * b, ncclDouble, ncclSum, c->comm, s));
   }
}

// The all-gather and the reduce-scatter of the row-sharded solver, as apply_sharded issues them, on a block whose every entry
// is known -- the rows a rank keeps of the random block with seed 4711 go out, the whole block must come back (all-gather),
// and nranks times a rank's own rows out of the whole block on every rank (reduce-scatter, chunk by chunk on the
// communication stream).  Returns THIS rank's verdict: empty = what came back is right; otherwise why not -- a wrong chunk /
// piece offset, a collective that pairs the wrong buffers, or a transport that reports a failure (FPCA_ECOMM is caught here:
// it is what is being tested).  The caller makes the verdict common to all ranks (agree_sum) before anybody acts on it.
std::string exchange_selftest(fpca_ctx *c_, const RowShard &sh_, int b_)
{
   hipStream_t s = c_->stream;
   std::string why;
   if (const char *inj = FPCA_TEST_ENV("FPCA_DEBUG_SELFTEST_FAIL")) // test builds: "all", or the rank that reports a failure
      if (std::string(inj) == "all" || atoi(inj) == sh_.rank) why = "injected failure of the exchange self-test (FPCA_DEBUG_SELFTEST_FAIL)";
   const size_t need_full = (size_t)sh_.full_rows() * b_;
   c_->ensure(c_->d_full_in, c_->full_in_cap, need_full);
   c_->ensure(c_->d_full_out, c_->full_out_cap, need_full);
   HIP_CHECK(hipMemsetAsync(c_->d_full_in, 0, c_->full_in_cap * sizeof(double), s));
   HIP_CHECK(hipMemsetAsync(c_->d_full_out, 0, c_->full_out_cap * sizeof(double), s));
   const uint64_t calls0 = c_->coll_calls, bytes0 = c_->coll_bytes; // (fpca_collective_stats counts the solver's data path only)
   double *slice = nullptr, *ref = nullptr;
   const size_t nslice = (size_t)sh_.slice_rows() * b_, nfull = (size_t)sh_.full_rows() * b_;
   HIP_CHECK(hipMalloc(&slice, nslice * sizeof(double)));
   HIP_CHECK(hipMalloc(&ref, nfull * sizeof(double)));
   auto finish = [&] {
      (void)hipFree(slice);
      (void)hipFree(ref);
   };
   try {
      unsigned long long *bits = reinterpret_cast<unsigned long long *>(c_->d_small + 64);
      HIP_CHECK(hipMemsetAsync(bits, 0, 2 * sizeof(unsigned long long), s));
      for (int c = 0; c < sh_.nch; c++)
         kern::fill_random(slice + (size_t)c * sh_.plen * b_, c_->N, sh_.plen, b_, 4711, s, (uint64_t)c * sh_.L + (uint64_t)sh_.rank * sh_.plen);
      kern::fill_random(ref, c_->N, sh_.full_rows(), b_, 4711, s, 0);
      c_->all_gather(sh_, slice, c_->d_full_in, b_, s);
      kern::max_abs_diff(c_->d_full_in, ref, 1.0, nfull, bits, s);
      // reduce-scatter of the whole block (identical on every rank): every rank must get G x its own rows
      HIP_CHECK(hipMemsetAsync(slice, 0xff, nslice * sizeof(double), s)); // NaNs: rows nobody writes would show
      if (sh_.nch > 1 && c_->native_collectives() && c_->comm_stream) {
         for (int i = 0; i < sh_.nch; i++) {
            HIP_CHECK(hipEventRecord(c_->ev_chunk[i], s));
            HIP_CHECK(hipStreamWaitEvent(c_->comm_stream, c_->ev_chunk[i], 0));
            c_->reduce_scatter(sh_, ref, slice, b_, c_->comm_stream, i);
         }
         HIP_CHECK(hipEventRecord(c_->ev_comm_done, c_->comm_stream));
         HIP_CHECK(hipStreamWaitEvent(s, c_->ev_comm_done, 0));
      } else {
         HIP_CHECK(hipMemcpyAsync(c_->d_full_out, ref, nfull * sizeof(double), hipMemcpyDeviceToDevice, s)); // (the sum-only route works in place)
         c_->reduce_scatter(sh_, c_->d_full_out, slice, b_, s);
         HIP_CHECK(hipMemsetAsync(c_->d_full_out, 0, nfull * sizeof(double), s));
      }
      for (int c = 0; c < sh_.nch; c++)
         kern::max_abs_diff(slice + (size_t)c * sh_.plen * b_, ref + ((size_t)c * sh_.L + (size_t)sh_.rank * sh_.plen) * b_, (double)sh_.G,
                            (size_t)sh_.plen * b_, bits + 1, s);
      HIP_CHECK(hipMemsetAsync(c_->d_full_in, 0, nfull * sizeof(double), s));
      double err[2] = {0, 0};
      HIP_CHECK(hipMemcpyAsync(err, bits, sizeof(err), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      // (entries are uniform in (-0.5, 0.5): the gathered block must be bit-equal, the sums of G equal terms within rounding)
      if (why.empty() && (err[0] != 0.0 || !(err[1] <= 1e-14 * sh_.G)))
         why = "rank " + std::to_string(sh_.rank) + " of " + std::to_string(sh_.G) + " (" + std::to_string(sh_.nch) + " row chunks): all-gather error " +
               std::to_string(err[0]) + ", reduce-scatter error " + std::to_string(err[1]);
   } catch (const Error &e) {
      if (e.code != FPCA_ECOMM) {
         finish();
         throw;
      }
      // (what this rank had enqueued before the failure may be waiting for peers that failed elsewhere: bounded)
      const bool drained = bounded_sync(c_, s) && (!c_->comm_stream || bounded_sync(c_, c_->comm_stream));
      if (!drained) {
         finish();
         abandon_comm(c_, std::string("rank ") + std::to_string(sh_.rank) + ": " + e.what() + "; the collectives enqueued before it never completed");
      }
      if (why.empty()) why = std::string("rank ") + std::to_string(sh_.rank) + ": " + e.what();
   } catch (...) {
      finish();
      throw;
   }
   finish();
   c_->coll_calls = calls0;
   c_->coll_bytes = bytes0;
   return why;
}

// One number summed over all ranks through the context's plainest collective -- an all-reduce of a single double, RCCL's
// most-trodden call -- so that a decision taken from a rank-local observation (the self-test's verdict, a collective that failed)
// is the SAME decision everywhere.  Not counted in fpca_collective_stats (it is not on the data path).
// time limit of a step on which the ranks must agree (seconds).  Generous: a peer may legitimately arrive late by a whole
// block apply plus its own error handling -- but never by minutes.
static double agree_limit_s()
{
   if (const char *e = FPCA_TEST_ENV("FPCA_AGREE_TIMEOUT_S"))
      if (atof(e) > 0) return atof(e);
   return 120.0;
}

void reset_exchange_state(fpca_ctx *c)
{
   c->exchange_tested = c->exchange_failed = -1;
   c->exchange_failed_path = 0;
   c->last_solver_path = 0;
}

void abandon_comm(fpca_ctx *c, const std::string &why)
{
   c->comm_dead = true;
   if (c->comm) {
      // ncclCommAbort tears down the kernels of this rank that spin on peers which will never arrive (CommDestroy would wait for them)
      try {
         if (rccl().CommAbort) (void)rccl().CommAbort(c->comm);
      } catch (...) {
      }
      c->comm = nullptr;
   }
   throw Error(FPCA_ECOMM, why + " -- the transport of this context is abandoned (this rank returns FPCA_ECOMM; the launcher ends the job)");
}

bool bounded_sync(fpca_ctx *c, hipStream_t s)
{
   const auto t0 = std::chrono::steady_clock::now();
   const double limit = agree_limit_s();
   for (long it = 0;; it++) {
      hipError_t q = hipStreamQuery(s);
      // (test builds: FPCA_DEBUG_AGREE_STALL makes the stream look pending for ever -- the peers that never arrive)
      if (q == hipSuccess && FPCA_TEST_ENV("FPCA_DEBUG_AGREE_STALL")) q = hipErrorNotReady;
      if (q == hipSuccess) return true;
      if (q != hipErrorNotReady) {
         (void)hipGetLastError();
         throw Error(FPCA_EHIP, std::string("hipStreamQuery failed: ") + hipGetErrorString(q));
      }
      if (c->comm && rccl().CommGetAsyncError && it % 256 == 255) { // an error RCCL noticed on its own: do not wait the limit out
         ncclResult_t ar = ncclSuccess;
         if (rccl().CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) return false;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
      std::this_thread::sleep_for(std::chrono::microseconds(it < 2000 ? 50 : 1000));
   }
}

// One number summed over all ranks through the context's plainest collective -- an all-reduce of a single double, RCCL's
// most-trodden call -- so that a decision taken from a rank-local observation (the self-test's verdict, a collective that failed)
// is the SAME decision everywhere.  Not counted in fpca_collective_stats (it is not on the data path).
// TIME-BOUNDED: the step is only safe when every rank reaches it -- a collective that failed on ALL ranks, or a
// verdict every rank computes.  A failure on ONE rank only (forbidden by the hook contract of fpca.h, possible with RCCL) leaves
// this rank here and its peers inside the collective that failed for it: the all-reduce below then never completes.  Instead of
// blocking for ever the rank gives up after agree_limit_s(), aborts the communicator and returns FPCA_ECOMM, which the
// launcher turns into the end of the job (flashpca --gpus: SIGCHLD -> the other ranks are killed).
double agree_sum(fpca_ctx *c, double mine)
{
   if (!c->multi()) return mine;
   const uint64_t calls0 = c->coll_calls, bytes0 = c->coll_bytes;
   double v = mine;
   HIP_CHECK(hipMemcpyAsync(c->d_small + 80, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
   try {
      c->allreduce(c->d_small + 80, 1, c->stream);
   } catch (const Error &e) {
      if (e.code != FPCA_ECOMM || c->comm_dead) throw;
      abandon_comm(c, std::string("the all-reduce on which the ranks agree failed (") + e.what() + ")");
   }
   HIP_CHECK(hipMemcpyAsync(&v, c->d_small + 80, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   if (!bounded_sync(c, c->stream))
      abandon_comm(c, "the ranks did not agree within " + std::to_string((int)agree_limit_s()) +
                         " s on how to continue: a collective failed on this rank while its peers are elsewhere");
   c->coll_calls = calls0;
   c->coll_bytes = bytes0;
   return v;
}

void comm_streams(fpca_ctx *c)
{
   if (c->comm_stream) return;
   HIP_CHECK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
   for (hipEvent_t &e : c->ev_chunk) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
   HIP_CHECK(hipEventCreateWithFlags(&c->ev_comm_done, hipEventDisableTiming));
}

} // namespace fpca
