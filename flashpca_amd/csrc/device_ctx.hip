// device_ctx.hip -- the C ABI of include/fpca.h: device context, HBM-resident data, operator launches,
// the HIP BlockBackend that the host eigensolver drives, RCCL plumbing and the measurement hooks.
// MI355X / gfx950 only; there is no CPU fallback anywhere in this file.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h> // types only: the library is dlopen()ed on first use (fpca_comm_*)

#include "../../include/fpca.h"
#include "../../include/fpca_debug.h"
#include "backend.hpp"
#include "common.hpp"
#include "kernels.hpp"
#include "pca_driver.hpp"
#include "synth.hpp"

namespace fpca {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

#define HIP_CHECK(expr)                                                                                      \
   do {                                                                                                      \
      hipError_t e__ = (expr);                                                                               \
      if (e__ != hipSuccess)                                                                                 \
         throw Error(FPCA_EHIP, std::string(#expr) + " failed: " + hipGetErrorString(e__));                  \
   } while (0)

// ---- RCCL, loaded lazily ----------------------------------------------------------------------------
struct RcclApi {
   void *handle = nullptr;
   ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
   ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
   ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
   const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi &rccl()
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
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.ReduceScatter || !api.AllGather || !api.CommDestroy)
         throw Error(FPCA_ECOMM, "librccl is missing expected symbols");
   }
   return api;
}

// allocations whose failure may legitimately be "does not fit": out-of-memory becomes FPCA_ENOMEM, anything else FPCA_EHIP
#define HIP_ALLOC(expr)                                                                                      \
   do {                                                                                                      \
      hipError_t e__ = (expr);                                                                               \
      if (e__ == hipErrorOutOfMemory) {                                                                      \
         (void)hipGetLastError();                                                                            \
         throw Error(FPCA_ENOMEM, std::string(#expr) + ": out of device memory");                            \
      }                                                                                                      \
      if (e__ != hipSuccess)                                                                                 \
         throw Error(FPCA_EHIP, std::string(#expr) + " failed: " + hipGetErrorString(e__));                  \
   } while (0)

#define RCCL_CHECK(expr)                                                                                     \
   do {                                                                                                      \
      ncclResult_t r__ = (expr);                                                                             \
      if (r__ != ncclSuccess)                                                                                \
         throw Error(FPCA_ECOMM, std::string(#expr) + " failed: " +                                          \
                                     (rccl().GetErrorString ? rccl().GetErrorString(r__) : "rccl error"));   \
   } while (0)

} // namespace fpca

using namespace fpca;

// ---- the context ---------------------------------------------------------------------------------------
struct fpca_ctx {
   int device = 0;
   hipStream_t stream = nullptr;
   uint64_t N = 0, P_g = 0, np = 0, N_pad = 0, P_pad = 0, P_total = 0;
   size_t pitch = 0;
   int stand = FPCA_STANDARDISE_BINOM2, accum = FPCA_ACCUM_FP64;
   uint8_t *d_packed = nullptr;
   double *d_Xd = nullptr; // dense (in-memory matrix) mode: standardised fp64 matrix [P_pad][N_pad]; d_packed unused
   bool dense = false;
   double *d_lut = nullptr, *d_mean = nullptr, *d_sd = nullptr, *d_sumsq = nullptr;
   bool stats_done = false;
   bool missing_known = false; // n_missing counted by K1 (not when mean/sd were preloaded)
   uint64_t n_missing = 0;     // missing calls in this shard
   double trace_local = 0;
   // workspaces (grown on demand)
   double *d_T = nullptr;
   size_t T_cap = 0;
   double *d_part = nullptr;
   size_t part_cap = 0;
   double *d_stage = nullptr; // column-major staging for the host-pointer API
   size_t stage_cap = 0;
   double *d_io_a = nullptr, *d_io_b = nullptr; // [N_pad][64] blocks for the host-pointer API
   double *d_small = nullptr;                   // small device scratch (scalars, column scales)
   // exact-integer mode (FPCA_ACCUM_I8(S)): sample-major packed copy, K3 row scales, sliced operands, int32 partials
   int i8_S = 0;
   int i8_S_req = 0; // the S the context was created with (i8_S drops to 0 if the mode's buffers do not fit; this does not)
   int i8_Sc = 0;    // slices of the passes being made NOW when that is fewer than i8_S (the eigensolver's cheap passes), else 0
   int cur_S() const { return (i8_Sc > 0 && i8_Sc < i8_S) ? i8_Sc : i8_S; }
   bool i8_auto = false; // mode chosen by FPCA_ACCUM_AUTO: falls back to fp64 if the extra buffers do not fit
   uint8_t *d_packedT = nullptr;
   size_t pitchT = 0;
   double *d_inv_sd = nullptr, *d_mu_inv_sd = nullptr, *d_i8w = nullptr;
   int8_t *d_Qb = nullptr, *d_Qg = nullptr, *d_Qm = nullptr;
   int i8_nsc = 0; // rows currently allocated (and zero-padded) in the Q buffers
   int i8_pad_zeroed_for = -1; // S*b for which rows [S*b, i8_nsc) of the Q buffers are known to be zero
   int i8_ws_for_S = 0, i8_ws_for_b = 0; // (S, b) the workspace was last sized for (the plan search is not free: 20 us)
   double *d_i8ws = nullptr;
   size_t i8ws_cap = 0;
   bool i8_scales_done = false, i8_transposed = false;
   // sparse missing indicator: index lists of the missing calls per SNP (sample indices) and per sample (SNP indices)
   std::vector<uint32_t> h_nmiss; // per-SNP counts from K1
   uint32_t *d_snp_ptr = nullptr, *d_snp_idx = nullptr, *d_smp_ptr = nullptr, *d_smp_idx = nullptr;
   double *d_eplane = nullptr; // E'Q of the current stage, [max(N_pad, P_pad)][b]; behind it, as much again: the row-major copy
   size_t eplane_cap = 0;      // of the scaled operand the gathers read (gather_src(): fp64, or fp32 under <= 4 slices)
   void *gather_src() const { return d_eplane + eplane_cap; }
   bool gather_f32() const { return cur_S() <= 4; } // the slices carry 30 bits: 24-bit rows of the (small) E term lose nothing
   bool sparse_ready = false;
   bool sparse_failed = false; // the index lists did not fit in device memory: the dense missing-indicator route is used
   // hybrid missing-indicator route: the SNPs whose missing calls are too many for the gathers (hyb_idx, hyb_n of them, padded
   // to hyb_pad) keep their indicator matrix E on the matrix cores as a compacted sub-matrix (SNP-major d_packedE, sample-major
   // d_packedET); the sample-major copy d_packedT then holds the VIEW of the matrix in which their missing calls read "dosage
   // 0" (same G.M), and the sparse lists hold the other SNPs' missing calls only
   int hyb_class = -1;        // -1 not classified yet, 0 no, 1 the shard qualifies
   bool hyb_view = false;     // d_packedT is that view (any route but the hybrid one needs the plain copy back: plain_view())
   bool hyb_failed = false;
   uint32_t hyb_n = 0, hyb_pad = 0;
   uint64_t hyb_sparse_nnz = 0;
   std::vector<uint32_t> h_hyb_idx;
   uint32_t *d_hyb_idx = nullptr;
   uint8_t *d_packedE = nullptr, *d_packedET = nullptr;
   size_t pitchET = 0;
   double *d_hyb_T = nullptr, *d_hyb_plane = nullptr;
   size_t hyb_T_cap = 0, hyb_plane_cap = 0;
   int8_t *d_Qd = nullptr;
   int hyb_qd_rows = 0, hyb_qd_zeroed_for = -1;
   hipStream_t aux_stream = nullptr; // the gather-sums run here, under the (MFMA-bound) GEMM of the same stage
   hipEvent_t ev_aux_go = nullptr, ev_aux_done = nullptr;
   // Krylov basis blocks of finished solves, kept for the next one (bytes, pointer): allocating and freeing a dozen
   // 128 MB blocks costs ~15 ms per fpca_pca at 500,000 samples; released by fpca_destroy
   std::vector<std::pair<size_t, double *>> block_pool;
   // the backend's small-matrix scratch lives here for the same reason (freeing a 100 MB partial stack and a pinned
   // buffer at the end of every solve costs ~10 ms)
   const double **be_ptrs = nullptr;
   double *be_C = nullptr, *be_gpart = nullptr;
   size_t be_C_cap = 0, be_gpart_cap = 0, be_pin_cap = 0;
   void *be_pin = nullptr;
   void *dl_pin = nullptr; // pinned landing zone of every download (HipBackend::download2): 4 slots of 8 MB
   hipEvent_t dl_ev[4] = {nullptr, nullptr, nullptr, nullptr};
   // communication
   ncclComm_t comm = nullptr;
   int nranks = 1, rank = 0;
   hipStream_t comm_stream = nullptr; // the all-reduce of a row chunk of Y runs here while the next chunk is computed
   hipEvent_t ev_chunk[4] = {nullptr, nullptr, nullptr, nullptr}, ev_comm_done = nullptr;
   fpca_allreduce_fn ar_fn = nullptr;
   void *ar_user = nullptr;
   fpca_allgather_fn ag_fn = nullptr; // caller-supplied all-gather / reduce-scatter (fpca_set_collectives): the row-sharded
   fpca_reducescatter_fn rs_fn = nullptr; // solver then runs exactly the call sequence it runs over RCCL
   void *coll_user = nullptr;
   // the transport has real all-gather / reduce-scatter (RCCL, or the caller's): the chunked, overlapped exchange of the
   // row-sharded solver; otherwise both are built from the caller's sum
   bool native_collectives() const { return (comm && !ar_fn) || (ar_fn && ag_fn && rs_fn); }
   bool rank_known = false; // nranks / rank are meaningful (fpca_comm_init_rank, or fpca_set_rank beside a caller's all-reduce)
   // row-sharded solver (backend.hpp RowShard): whole [full_rows][b] blocks either side of the operator
   double *d_full_in = nullptr, *d_full_out = nullptr;
   size_t full_in_cap = 0, full_out_cap = 0;
   uint64_t coll_calls = 0, coll_bytes = 0; // data-path collectives issued by this context (calls, payload bytes)
   long exchange_tested = -1; // layout (ranks, chunks, width) whose all-gather / reduce-scatter have passed the self-test
   // live profiling (fpca_profile_begin/end)
   std::vector<hipEvent_t> prof_ev;
   int prof_used = 0, prof_calls = 0, prof_stride = 1; // every prof_stride-th apply carries the events
   bool prof_on = false;

   void ensure(double *&p, size_t &cap, size_t need)
   {
      if (need <= cap) return;
      if (p) HIP_CHECK(hipFree(p));
      p = nullptr;
      cap = 0;
      HIP_CHECK(hipMalloc(&p, need * sizeof(double)));
      cap = need;
   }
   bool multi() const { return comm != nullptr || ar_fn != nullptr; }
   // slice [sh.slice_rows()][b] -> full [sh.full_rows()][b] on every rank
   void all_gather(const RowShard &sh, const double *slice, double *full, int b, hipStream_t s)
   {
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
   // sum over ranks of full [sh.full_rows()][b]; rank r keeps its rows in slice.  chunk >= 0: that chunk only.
   void reduce_scatter(const RowShard &sh, double *full, double *slice, int b, hipStream_t s, int only_chunk = -1)
   {
      const size_t piece = (size_t)sh.plen * b, chunk = (size_t)sh.L * b;
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
   void allreduce(double *dbuf, uint64_t count, hipStream_t s)
   {
      coll_calls++;
      coll_bytes += count * sizeof(double);
      if (ar_fn) { // a caller-supplied hook wins over the built-in communicator (set after a failed / partial RCCL init)
         if (ar_fn(ar_user, dbuf, count, (void *)s) != 0) throw Error(FPCA_ECOMM, "caller-supplied all-reduce failed");
      } else if (comm)
         RCCL_CHECK(rccl().AllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, comm, s));
   }
};

namespace {

void ctx_alloc_common(fpca_ctx *c, uint64_t N, uint64_t P_g, int stand, int device, int accum, bool dense = false)
{
   if (N == 0) throw Error(FPCA_EINVAL, "N must be > 0");
   if (!dense && stand != FPCA_STANDARDISE_BINOM && stand != FPCA_STANDARDISE_BINOM2)
      throw Error(FPCA_EINVAL, "unknown standardisation method: " + std::to_string(stand)); // data.cpp:283-288
   if (dense && (stand < FPCA_STANDARDISE_NONE || stand > FPCA_STANDARDISE_CENTER))
      throw Error(FPCA_EINVAL, "unknown standardization method"); // util.cpp:183
   if (accum == FPCA_ACCUM_AUTO) {
      // the exact-integer path when it applies (2-bit input, int32-safe sizes), else the fp64 MFMA path
      const bool fits = !dense && std::max<uint64_t>(N, P_g) <= (uint64_t)8000000;
      accum = fits ? FPCA_ACCUM_I8(7) : FPCA_ACCUM_FP64;
      c->i8_auto = fits;
   }
   const bool i8 = accum >= FPCA_ACCUM_I8(2) && accum <= FPCA_ACCUM_I8(8);
   if (accum != FPCA_ACCUM_FP64 && accum != FPCA_ACCUM_FP32 && !i8)
      throw Error(FPCA_EINVAL, "accum must be FPCA_ACCUM_AUTO, FPCA_ACCUM_FP64, FPCA_ACCUM_FP32 or FPCA_ACCUM_I8(2..8)");
   if (i8 && dense) throw Error(FPCA_EINVAL, "the int8-sliced mode needs 2-bit genotype input");
   const bool timing = std::getenv("FPCA_TIMING") != nullptr;
   auto tl = std::chrono::steady_clock::now();
   auto lap = [&](const char *what) {
      const auto now = std::chrono::steady_clock::now();
      if (timing) std::fprintf(stderr, "[fpca]   %-26s %8.3f ms\n", what, std::chrono::duration<double>(now - tl).count() * 1e3);
      tl = now;
   };
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw Error(FPCA_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
   lap("hipGetDeviceCount");
   if (device < 0 || device >= ndev) throw Error(FPCA_ENODEVICE, "device index out of range");
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, device) != hipSuccess) throw Error(FPCA_ENODEVICE, "hipGetDeviceProperties failed");
   if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw Error(FPCA_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
   if (hipSetDevice(device) != hipSuccess) throw Error(FPCA_ENODEVICE, "hipSetDevice failed");
   c->device = device;
   c->N = N;
   c->P_g = P_g;
   c->P_total = P_g;
   c->np = (N + 3) / 4;
   c->pitch = (size_t)round_up(c->np, ROW_ALIGN);
   c->N_pad = (uint64_t)c->pitch * 4;
   c->P_pad = round_up(std::max<uint64_t>(P_g, 1), SNP_ALIGN);
   c->stand = stand;
   c->accum = accum;
   c->i8_S = i8 ? accum - FPCA_ACCUM_I8(0) : 0;
   c->i8_S_req = c->i8_S;
   c->dense = dense;
   lap("device properties, hipSetDevice");
   HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
   lap("stream");
   if (dense) {
      HIP_CHECK(hipMalloc(&c->d_Xd, (size_t)c->P_pad * c->N_pad * sizeof(double)));
      HIP_CHECK(hipMemsetAsync(c->d_Xd, 0, (size_t)c->P_pad * c->N_pad * sizeof(double), c->stream));
   } else {
      // The matrix is RESIDENT (the reference streams any size from disk, svdwide.h:57-68): if it does not fit, say what would
      const size_t need = c->pitch * c->P_pad;
      const hipError_t e = hipMalloc(&c->d_packed, need);
      if (e == hipErrorOutOfMemory) {
         (void)hipGetLastError();
         size_t fr = 0, tot = 0;
         (void)hipMemGetInfo(&fr, &tot);
         const double gb = 1.0 / (1024.0 * 1024.0 * 1024.0);
         const int g1 = (int)std::ceil((double)need * 1.05 / std::max<double>((double)fr, 1.0)), g2 = (int)std::ceil((double)need * 2.1 / std::max<double>((double)fr, 1.0));
         char msg[640];
         std::snprintf(msg, sizeof(msg),
                       "the packed genotypes of this shard need %.1f GiB of device memory (%llu samples x %llu SNPs at 2 bits), %.1f of %.1f GiB are free "
                       "on device %d.  What fits: the SNPs sharded over at least %d GPUs (--gpus %d; %d for the default exact-integer arithmetic, "
                       "which keeps a second, sample-major copy -- --accum fp64 does not)",
                       (double)need * gb, (unsigned long long)N, (unsigned long long)P_g, (double)fr * gb, (double)tot * gb, device, std::max(g1, 2),
                       std::max(g1, 2), std::max(g2, 2));
         throw Error(FPCA_ENOMEM, msg);
      }
      if (e != hipSuccess) throw Error(FPCA_EHIP, std::string("hipMalloc of the packed genotypes failed: ") + hipGetErrorString(e));
      HIP_CHECK(hipMemsetAsync(c->d_packed, PAD_BYTE, c->pitch * c->P_pad, c->stream));
   }
   lap("hipMalloc packed + memset");
   HIP_CHECK(hipMalloc(&c->d_lut, c->P_pad * 4 * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_mean, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_sd, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_sumsq, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMemsetAsync(c->d_lut, 0, c->P_pad * 4 * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_mean, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_sd, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_sumsq, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMalloc(&c->d_small, 4096 * sizeof(double)));
}

void ctx_finish_upload(fpca_ctx *c)
{
   kern::fix_last_byte(c->d_packed, c->pitch, c->np, (int)(c->N % 4), c->P_g, c->stream);
   HIP_CHECK(hipStreamSynchronize(c->stream));
}

void ctx_free(fpca_ctx *c)
{
   if (!c) return;
   (void)hipSetDevice(c->device);
   if (c->comm) {
      try {
         rccl().CommDestroy(c->comm);
      } catch (...) {
      }
   }
   void *ptrs[] = {c->d_Xd, c->d_packed, c->d_lut, c->d_mean, c->d_sd,   c->d_sumsq, c->d_T,
                   c->d_part,   c->d_stage, c->d_io_a, c->d_io_b, c->d_small, c->d_packedT, c->d_inv_sd, c->d_mu_inv_sd,
                   c->d_i8w, c->d_Qb, c->d_Qg, c->d_Qm, c->d_i8ws, c->d_snp_ptr, c->d_snp_idx, c->d_smp_ptr, c->d_smp_idx, c->d_eplane,
                   c->d_full_in, c->d_full_out, c->d_hyb_idx, c->d_packedE, c->d_packedET, c->d_hyb_T, c->d_hyb_plane, c->d_Qd};
   for (void *p : ptrs)
      if (p) (void)hipFree(p);
   for (auto &pb : c->block_pool) (void)hipFree(pb.second);
   if (c->be_ptrs) (void)hipFree(c->be_ptrs);
   if (c->be_C) (void)hipFree(c->be_C);
   if (c->be_gpart) (void)hipFree(c->be_gpart);
   if (c->be_pin) (void)hipHostFree(c->be_pin);
   if (c->dl_pin) (void)hipHostFree(c->dl_pin);
   for (hipEvent_t e : c->dl_ev)
      if (e) (void)hipEventDestroy(e);
   for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
   for (hipEvent_t e : c->ev_chunk)
      if (e) (void)hipEventDestroy(e);
   if (c->ev_comm_done) (void)hipEventDestroy(c->ev_comm_done);
   if (c->ev_aux_go) (void)hipEventDestroy(c->ev_aux_go);
   if (c->ev_aux_done) (void)hipEventDestroy(c->ev_aux_done);
   if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
   if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
   if (c->stream) (void)hipStreamDestroy(c->stream);
   delete c;
}

void ensure_stats(fpca_ctx *c)
{
   if (c->stats_done) return;
   HIP_CHECK(hipSetDevice(c->device));
   uint32_t *d_nmiss = nullptr;
   std::vector<uint32_t> nm(c->P_g);
   if (c->P_g) HIP_CHECK(hipMalloc(&d_nmiss, c->P_g * sizeof(uint32_t)));
   kern::bed_stats(c->d_packed, c->pitch, c->N, c->P_g, c->stand, c->d_lut, c->d_mean, c->d_sd, c->d_sumsq, d_nmiss, c->stream);
   std::vector<double> ss(c->P_g);
   HIP_CHECK(hipMemcpyAsync(ss.data(), c->d_sumsq, c->P_g * sizeof(double), hipMemcpyDeviceToHost, c->stream));
   if (c->P_g) HIP_CHECK(hipMemcpyAsync(nm.data(), d_nmiss, c->P_g * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
   HIP_CHECK(hipStreamSynchronize(c->stream));
   if (d_nmiss) (void)hipFree(d_nmiss);
   c->n_missing = 0;
   for (uint32_t v : nm) c->n_missing += v;
   c->missing_known = true;
   c->h_nmiss.swap(nm);
   // pairwise-ish (blocked) summation for a reproducible, accurate trace
   double tot = 0;
   for (size_t i0 = 0; i0 < ss.size(); i0 += 1024) {
      double s = 0;
      const size_t i1 = std::min(ss.size(), i0 + 1024);
      for (size_t i = i0; i < i1; i++) s += ss[i];
      tot += s;
   }
   c->trace_local = tot;
   c->stats_done = true;
}

// ---- exact-integer mode --------------------------------------------------------------------------------
// layout of d_i8w in 8-byte words: three weight vectors (S*b <= 9*64 = 576 entries each, padded), then the region that is
// zeroed once per apply: column maxima (bit patterns) of the three operands and the column sums of the two M operands
// (I8W_D / I8W_MAXD: the K3 operand of the hybrid route's dense SNPs)
constexpr int I8W_B = 0, I8W_G = 640, I8W_M = 1280, I8W_D = 1920, I8W_ZERO = 2560, I8W_MAXB = I8W_ZERO, I8W_MAXG = I8W_MAXB + 64 * kern::I8_SHARDS,
              I8W_MAXM = I8W_MAXG + 64 * kern::I8_SHARDS, I8W_MAXD = I8W_MAXM + 64 * kern::I8_SHARDS, I8W_CSB = I8W_MAXD + 64 * kern::I8_SHARDS,
              I8W_CSM = I8W_CSB + kern::I8_CS_STRIDE * kern::I8_SHARDS, I8W_TOTAL = I8W_CSM + kern::I8_CS_STRIDE * kern::I8_SHARDS;

constexpr int I8M_FULL = 0, I8M_SKIP = 1, I8M_NONE = 2, I8M_SPARSE = 3, I8M_HYBRID = 4; // (see i8_mode)
constexpr double SPARSE_BREAK_EVEN = 0.005; // missing-call rate at which the gathers cost what the E half of the GEMMs costs
void hybrid_classify(fpca_ctx *c);
void ensure_i8_alloc(fpca_ctx *c, int b);
uint64_t ar_chunk_begin(const fpca_ctx *c, int nchunks, int i);
int shard_chunks(const fpca_ctx *c);

// true: the int8 path is ready for blocks of width b.  false (FPCA_ACCUM_AUTO only): its extra buffers did not fit, the
// context has been switched to the fp64 kernels for good.
bool ensure_i8(fpca_ctx *c, int b)
{
   try {
      ensure_i8_alloc(c, b);
      return true;
   } catch (const Error &e) {
      if (!c->i8_auto && e.code == FPCA_ENOMEM) { // asked for explicitly: no silent change of arithmetic -- say what would fit
         size_t fr = 0, tot = 0;
         (void)hipMemGetInfo(&fr, &tot);
         const double gb = 1.0 / (1024.0 * 1024.0 * 1024.0), copy = (double)c->N_pad * (double)c->P_pad / 4.0;
         char msg[640];
         std::snprintf(msg, sizeof(msg),
                       "the exact-integer arithmetic needs a second, sample-major copy of the packed genotypes (%.1f GiB) and its int8 operands; %.1f "
                       "of %.1f GiB are free (%s).  What fits: --accum auto (falls back to the fp64 kernels, which need no second copy: same "
                       "results, ~4x slower), --accum fp64, or the SNPs sharded over more GPUs (--gpus)",
                       copy * gb, (double)fr * gb, (double)tot * gb, e.what());
         throw Error(FPCA_ENOMEM, msg);
      }
      if (!c->i8_auto || e.code != FPCA_ENOMEM) throw; // only "does not fit"; a kernel or launch failure is not masked
      (void)hipGetLastError();
      std::fprintf(stderr, "[fpca] exact-integer mode needs more device memory than is free (%s); using the fp64 kernels\n", e.what());
      void **ptrs[] = {(void **)&c->d_packedT, (void **)&c->d_Qb, (void **)&c->d_Qg, (void **)&c->d_Qm, (void **)&c->d_i8ws};
      for (void **p : ptrs)
         if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
         }
      c->i8_transposed = false;
      c->i8_nsc = 0;
      c->i8_pad_zeroed_for = -1;
      c->i8ws_cap = 0;
      c->i8_ws_for_S = c->i8_ws_for_b = 0;
      c->i8_S = 0;
      c->accum = FPCA_ACCUM_FP64;
      return false;
   }
}

void ensure_i8_alloc(fpca_ctx *c, int b)
{
   hipStream_t s = c->stream;
   if (FPCA_TEST_ENV("FPCA_DEBUG_I8_NOMEM")) throw Error(FPCA_ENOMEM, "FPCA_DEBUG_I8_NOMEM is set"); // exercises the fallback in the tests
   // exact int32 accumulation: |sum| <= 2 * 128 * K must stay below 2^31
   if (std::max(c->N_pad, c->P_pad) > (uint64_t)8380000)
      throw Error(FPCA_EINVAL, "the int8-sliced mode supports up to 8,380,000 samples and SNPs per GPU (int32 accumulation)");
   if (!c->i8_transposed) {
      c->pitchT = (size_t)c->P_pad / 4;
      if (!c->d_inv_sd) HIP_ALLOC(hipMalloc(&c->d_inv_sd, c->P_pad * sizeof(double)));
      if (!c->d_mu_inv_sd) HIP_ALLOC(hipMalloc(&c->d_mu_inv_sd, c->P_pad * sizeof(double)));
      if (!c->d_i8w) HIP_ALLOC(hipMalloc(&c->d_i8w, I8W_TOTAL * sizeof(double)));
      HIP_ALLOC(hipMalloc(&c->d_packedT, c->pitchT * c->N_pad));
      // hybrid missing-indicator route (decided from K1's per-SNP counts, whatever b will be): the records of the dense SNPs are
      // copied out, their missing calls are rewritten to "dosage 0" for the duration of the transposition -- the sample-major
      // copy then IS the view the sparse lists and K3's G.M kernel want -- and the records are put back
      bool hyb = false;
      {
         const char *env = FPCA_TEST_ENV("FPCA_I8_MODE");
         hybrid_classify(c);
         hyb = c->hyb_class == 1 && !c->hyb_failed && !c->sparse_failed && (!env || atoi(env) == I8M_HYBRID);
      }
      if (hyb) {
         try {
            HIP_ALLOC(hipMalloc(&c->d_hyb_idx, c->hyb_pad * sizeof(uint32_t)));
            HIP_ALLOC(hipMalloc(&c->d_packedE, (size_t)c->hyb_pad * c->pitch));
            c->pitchET = (size_t)c->hyb_pad / 4;
            HIP_ALLOC(hipMalloc(&c->d_packedET, c->pitchET * c->N_pad));
         } catch (const Error &e) {
            if (e.code != FPCA_ENOMEM) throw;
            (void)hipGetLastError();
            for (void **q : {(void **)&c->d_hyb_idx, (void **)&c->d_packedE, (void **)&c->d_packedET})
               if (*q) {
                  (void)hipFree(*q);
                  *q = nullptr;
               }
            c->hyb_failed = true;
            hyb = false;
         }
      }
      if (hyb) {
         HIP_CHECK(hipMemcpyAsync(c->d_hyb_idx, c->h_hyb_idx.data(), c->hyb_n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
         kern::gather_packed_rows(c->d_packed, c->pitch, c->d_hyb_idx, c->hyb_n, c->hyb_pad, c->d_packedE, s);
         kern::patch_missing_rows(c->d_packed, c->pitch, c->d_hyb_idx, c->hyb_n, s);
      }
      kern::transpose_packed(c->d_packed, c->pitch, c->N_pad, c->P_pad, c->d_packedT, c->pitchT, s);
      if (hyb) {
         kern::scatter_packed_rows(c->d_packedE, c->pitch, c->d_hyb_idx, c->hyb_n, c->d_packed, s);
         kern::transpose_packed(c->d_packedE, c->pitch, c->N_pad, c->hyb_pad, c->d_packedET, c->pitchET, s);
         c->hyb_view = true;
      }
      c->i8_transposed = true;
   }
   if (!c->i8_scales_done) {
      kern::i8_rowscales(c->d_mean, c->d_sd, c->P_g, c->P_pad, c->d_inv_sd, c->d_mu_inv_sd, s);
      c->i8_scales_done = true;
   }
   const int Sc = c->cur_S();
   const int nsc_cur = kern::gemm_i8_nsc_pad(Sc, b);               // rows of Q the kernels of this pass read
   const int nsc = std::max(nsc_cur, kern::gemm_i8_nsc_pad(c->i8_S, b)); // rows to hold: the exact passes need the most
   if (nsc > c->i8_nsc) {
      for (int8_t **q : {&c->d_Qb, &c->d_Qg, &c->d_Qm})
         if (*q) {
            HIP_CHECK(hipFree(*q));
            *q = nullptr;
         }
      HIP_ALLOC(hipMalloc(&c->d_Qb, (size_t)nsc * c->N_pad));
      HIP_ALLOC(hipMalloc(&c->d_Qg, (size_t)nsc * c->P_pad));
      HIP_ALLOC(hipMalloc(&c->d_Qm, (size_t)nsc * c->P_pad));
      c->i8_nsc = nsc;
      c->i8_pad_zeroed_for = -1;
   }
   // rows >= S*b of the Q operands must be zero (they are multiplied like any other column); the slicing kernels never
   // write them, so once per (allocation, S*b) is enough -- this runs at the top of every apply
   // (... and once per change of the slice count: the cheap passes of the eigensolver leave their own padding rows behind)
   if (nsc_cur != Sc * b && c->i8_pad_zeroed_for != Sc * b) {
      c->i8_pad_zeroed_for = Sc * b;
      const size_t used = (size_t)Sc * b;
      HIP_CHECK(hipMemsetAsync(c->d_Qb + used * c->N_pad, 0, (nsc_cur - used) * c->N_pad, s));
      HIP_CHECK(hipMemsetAsync(c->d_Qg + used * c->P_pad, 0, (nsc_cur - used) * c->P_pad, s));
      HIP_CHECK(hipMemsetAsync(c->d_Qm + used * c->P_pad, 0, (nsc_cur - used) * c->P_pad, s));
   } else if (nsc_cur == Sc * b)
      c->i8_pad_zeroed_for = -1; // (whole tiles: nothing to zero now, but the next ragged count must not trust stale rows)
   if (Sc == c->i8_ws_for_S && b == c->i8_ws_for_b) return; // workspace already sized for this (S, b)
   size_t need = std::max(kern::gemm_i8_workspace_doubles(c->P_pad, c->N_pad, Sc, b, false),
                          std::max(kern::gemm_i8_workspace_doubles(c->N_pad, c->P_pad, Sc, b, true),
                                   kern::gemm_i8_workspace_doubles(c->N_pad, c->P_pad, Sc, b, false)));
   for (int nch = 2; nch <= 4; nch++) // K3 in row chunks (overlapped all-reduce): the plan of a chunk may use more planes
      for (int i = 0; i < nch; i++) {
         const uint64_t rows = ar_chunk_begin(c, nch, i + 1) - ar_chunk_begin(c, nch, i);
         if (rows)
            need = std::max(need, std::max(kern::gemm_i8_workspace_doubles(rows, c->P_pad, Sc, b, true),
                                           kern::gemm_i8_workspace_doubles(rows, c->P_pad, Sc, b, false)));
      }
   if (c->rank_known && c->nranks > 1) // K3 in the row chunks of the row-sharded solver (apply_sharded)
      for (int nch = 2; nch <= 4; nch++) {
         const RowShard sh = RowShard::make(c->N_pad, c->nranks, c->rank, nch, 512);
         for (int i = 0; i < nch; i++) {
            const uint64_t r0 = std::min<uint64_t>((uint64_t)i * sh.L, c->N_pad), r1 = std::min<uint64_t>((uint64_t)(i + 1) * sh.L, c->N_pad);
            if (r1 > r0)
               need = std::max(need, std::max(kern::gemm_i8_workspace_doubles(r1 - r0, c->P_pad, Sc, b, true),
                                              kern::gemm_i8_workspace_doubles(r1 - r0, c->P_pad, Sc, b, false)));
         }
      }
   if (need > c->i8ws_cap) {
      if (c->d_i8ws) HIP_CHECK(hipFree(c->d_i8ws));
      c->d_i8ws = nullptr;
      c->i8ws_cap = 0;
      HIP_ALLOC(hipMalloc(&c->d_i8ws, need * sizeof(double)));
      c->i8ws_cap = need;
   }
   c->i8_ws_for_S = Sc;
   c->i8_ws_for_b = b;
}

// how the int8 GEMMs treat the missing-indicator matrix (kernels_i8.hip: I8_FULL / I8_SKIP_EMPTY / I8_NO_MISSING)
// 0 both matrices on the matrix cores; 1 the same, skipping blocks of E without a missing call; 2 no missing call in the
// shard: G.M alone; 3 G.M alone on the matrix cores + the missing-indicator products as sparse fp64 gathers
// 4 = hybrid: G.M on the matrix cores; the missing-indicator products as sparse gathers for most SNPs and as a small
// integer GEMM over a compacted sub-matrix for the few SNPs that hold most of the missing calls (real arrays: failed assays)

// Per-SNP choice of the route (from K1's per-SNP counts): a SNP above the break-even rate goes dense.  The shard qualifies
// for the hybrid route when that leaves the rest at or below the break-even and the dense set is a minority of the SNPs.
void hybrid_classify(fpca_ctx *c)
{
   if (c->hyb_class >= 0) return;
   c->hyb_class = 0;
   if (!c->missing_known || c->h_nmiss.size() != c->P_g || c->P_g == 0) return;
   const double thr = SPARSE_BREAK_EVEN * (double)c->N;
   uint64_t dense_nnz = 0;
   std::vector<uint32_t> idx;
   for (uint64_t j = 0; j < c->P_g; j++)
      if ((double)c->h_nmiss[j] > thr) {
         idx.push_back((uint32_t)j);
         dense_nnz += c->h_nmiss[j];
      }
   const uint64_t rest = c->n_missing - dense_nnz;
   if (idx.empty() || idx.size() * 4 > c->P_g) return;
   if ((double)rest > SPARSE_BREAK_EVEN * (double)c->N * (double)(c->P_g - idx.size()) || rest >= (1ull << 31)) return;
   c->hyb_n = (uint32_t)idx.size();
   c->hyb_pad = (uint32_t)round_up(c->hyb_n, SNP_ALIGN);
   c->hyb_sparse_nnz = rest;
   c->h_hyb_idx.swap(idx);
   c->hyb_class = 1;
}

int i8_mode(fpca_ctx *c, int b) // (classifies the SNPs the first time a rate above the break-even makes the hybrid route a candidate)
{
   const char *env = FPCA_TEST_ENV("FPCA_I8_MODE"); // force (tests; 2 is wrong unless nothing is missing); read on every call
   const bool sparse_ok = c->missing_known && !c->sparse_failed && c->n_missing < (1ull << 31) && (b == 16 || b == 32 || b == 64);
   const bool lists_ok = c->missing_known && !c->sparse_failed && (b == 16 || b == 32 || b == 64);
   if (env && atoi(env) == I8M_HYBRID) {
      hybrid_classify(c);
      return (lists_ok && c->hyb_class == 1 && !c->hyb_failed) ? I8M_HYBRID : I8M_FULL;
   }
   if (env) return (atoi(env) == I8M_SPARSE && !sparse_ok) ? I8M_FULL : atoi(env);
   if (!c->missing_known) return I8M_FULL;
   if (c->n_missing == 0) return I8M_NONE;
   const double rate = (double)c->n_missing / ((double)c->N * (double)std::max<uint64_t>(c->P_g, 1));
   if (lists_ok && rate > SPARSE_BREAK_EVEN && !c->hyb_failed) {
      hybrid_classify(c);
      if (c->hyb_class == 1) return I8M_HYBRID;
   }
   // a gathered fp64 row costs 8 b bytes per missing call; the E half of the int8 GEMMs costs the same whatever the rate.
   // Measured at 500k x 100k (scripts/sparse_breakeven.py, profiles/r03_sparse_breakeven.txt; K2 / K3 stage in ms, sparse |
   // dense): b = 16: 0.3 % 6.8 / 7.4 | 9.0 / 9.8, 0.5 % 8.5 / 9.1 | 9.0 / 9.9, 1 % 12.6 / 13.0 | 9.0 / 9.8; b = 32: 0.3 % 12.7 / 13.7 |
   // 16.2 / 18.8, 0.5 % 15.9 / 16.8 | 16.1 / 18.8, 1 % 23.9 / 24.7 | 16.2 / 18.8 -- the lines cross at 0.51-0.63 % for both widths
   if (sparse_ok && rate <= SPARSE_BREAK_EVEN) return I8M_SPARSE;
   return rate < 3e-4 ? I8M_SKIP : I8M_FULL; // (block skipping: only where the sparse path does not apply)
}

// The gather-sum runs on the low-priority side stream, released together with the GEMM of its stage, when it is big enough
// to be worth two event hand-offs (~30 us): measured 11.0 / 11.9 ms vs 11.9 / 12.2 ms at cfg3, but 0.338 / 0.360 vs
// 0.306 / 0.334 ms at cfg2, where it stays inline.
// FPCA_SPARSE_SIDE_BYTES overrides the threshold (bytes gathered per stage).
bool sparse_on_side_stream(const fpca_ctx *c, int b)
{
   static const double thr = FPCA_TEST_ENV("FPCA_SPARSE_SIDE_BYTES") ? atof(FPCA_TEST_ENV("FPCA_SPARSE_SIDE_BYTES")) : 2e9;
   return (double)(c->hyb_view ? c->hyb_sparse_nnz : c->n_missing) * b * 8.0 > thr;
}

// index lists of the missing calls, built once (by SNP from the SNP-major stream, by sample from the sample-major copy)
void ensure_sparse(fpca_ctx *c, int b)
{
   hipStream_t s = c->stream;
   const size_t need = (size_t)std::max(c->N_pad, c->P_pad) * b;
   if (FPCA_TEST_ENV("FPCA_DEBUG_SPARSE_NOMEM")) throw Error(FPCA_ENOMEM, "FPCA_DEBUG_SPARSE_NOMEM is set"); // exercises the fallback
   if (need > c->eplane_cap) {
      if (c->d_eplane) HIP_CHECK(hipFree(c->d_eplane));
      c->d_eplane = nullptr;
      c->eplane_cap = 0;
      HIP_ALLOC(hipMalloc(&c->d_eplane, 2 * need * sizeof(double)));
      c->eplane_cap = need;
   }
   if (c->sparse_ready) return;
   if (!c->aux_stream) {
      int lo = 0, hi = 0; // lowest priority: the gather-sums should only fill what the GEMM's workgroups leave free
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      HIP_CHECK(hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, lo));
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_aux_go, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming));
   }
   // (hybrid view: the dense SNPs' missing calls are not listed -- their counts are zero here, and fill_missing leaves a record
   //  with an empty list untouched; the sample-major copy does not show them in the first place)
   const uint64_t nnz = c->hyb_view ? c->hyb_sparse_nnz : c->n_missing;
   std::vector<uint32_t> ptr(c->P_g + 1, 0);
   {
      size_t d = 0;
      for (uint64_t j = 0; j < c->P_g; j++) {
         const bool dense = c->hyb_view && d < c->h_hyb_idx.size() && c->h_hyb_idx[d] == j;
         if (dense) d++;
         ptr[j + 1] = ptr[j] + (dense ? 0u : c->h_nmiss[j]);
      }
   }
   HIP_ALLOC(hipMalloc(&c->d_snp_ptr, (c->P_g + 1) * sizeof(uint32_t)));
   HIP_ALLOC(hipMalloc(&c->d_snp_idx, std::max<uint64_t>(nnz, 1) * sizeof(uint32_t)));
   HIP_CHECK(hipMemcpyAsync(c->d_snp_ptr, ptr.data(), (c->P_g + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
   kern::fill_missing(c->d_packed, c->pitch, c->N, c->P_g, c->d_snp_ptr, c->d_snp_idx, s);
   HIP_CHECK(hipStreamSynchronize(s)); // ptr is reused below
   uint32_t *d_cnt = nullptr;
   HIP_ALLOC(hipMalloc(&d_cnt, c->N * sizeof(uint32_t)));
   kern::count_missing(c->d_packedT, c->pitchT, c->P_g, c->N, d_cnt, s);
   std::vector<uint32_t> cnt(c->N);
   HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, c->N * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
   HIP_CHECK(hipStreamSynchronize(s));
   (void)hipFree(d_cnt);
   ptr.assign(c->N + 1, 0);
   for (uint64_t i = 0; i < c->N; i++) ptr[i + 1] = ptr[i] + cnt[i];
   if (ptr[c->N] != nnz) throw Error(FPCA_EHIP, "missing-call counts by sample and by SNP disagree");
   HIP_ALLOC(hipMalloc(&c->d_smp_ptr, (c->N + 1) * sizeof(uint32_t)));
   HIP_ALLOC(hipMalloc(&c->d_smp_idx, std::max<uint64_t>(nnz, 1) * sizeof(uint32_t)));
   HIP_CHECK(hipMemcpyAsync(c->d_smp_ptr, ptr.data(), (c->N + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
   kern::fill_missing(c->d_packedT, c->pitchT, c->P_g, c->N, c->d_smp_ptr, c->d_smp_idx, s);
   HIP_CHECK(hipStreamSynchronize(s));
   c->sparse_ready = true;
}

// Any route but the hybrid one reads the sample-major copy as the plain transpose of the matrix: if it currently holds the
// hybrid view, it is transposed again (6 ms at 500,000 x 100,000) and the lists made for the view are dropped.  Rare: a
// block width without a gather kernel (48), a forced mode (tests), or the view's buffers not fitting after all.
void plain_view(fpca_ctx *c)
{
   if (!c->hyb_view) return;
   kern::transpose_packed(c->d_packed, c->pitch, c->N_pad, c->P_pad, c->d_packedT, c->pitchT, c->stream);
   HIP_CHECK(hipStreamSynchronize(c->stream));
   for (void **q : {(void **)&c->d_snp_ptr, (void **)&c->d_snp_idx, (void **)&c->d_smp_ptr, (void **)&c->d_smp_idx, (void **)&c->d_packedE,
                    (void **)&c->d_packedET, (void **)&c->d_hyb_idx})
      if (*q) {
         (void)hipFree(*q);
         *q = nullptr;
      }
   c->sparse_ready = false;
   c->hyb_view = false;
   c->hyb_failed = true;
}

// the hybrid route's per-width buffers: the dense SNPs' operand Td [hyb_pad][b] and its slices, one plane of E products
void ensure_hybrid(fpca_ctx *c, int b)
{
   const size_t need_t = (size_t)c->hyb_pad * b, need_p = (size_t)std::max<uint64_t>(c->hyb_pad, c->N_pad) * b;
   if (need_t > c->hyb_T_cap) {
      if (c->d_hyb_T) HIP_CHECK(hipFree(c->d_hyb_T));
      c->d_hyb_T = nullptr;
      c->hyb_T_cap = 0;
      HIP_ALLOC(hipMalloc(&c->d_hyb_T, need_t * sizeof(double)));
      c->hyb_T_cap = need_t;
   }
   if (need_p > c->hyb_plane_cap) {
      if (c->d_hyb_plane) HIP_CHECK(hipFree(c->d_hyb_plane));
      c->d_hyb_plane = nullptr;
      c->hyb_plane_cap = 0;
      HIP_ALLOC(hipMalloc(&c->d_hyb_plane, need_p * sizeof(double)));
      c->hyb_plane_cap = need_p;
   }
   const int Sc = c->cur_S(), nsc = std::max(kern::gemm_i8_nsc_pad(Sc, b), kern::gemm_i8_nsc_pad(c->i8_S, b));
   if (nsc > c->hyb_qd_rows) {
      if (c->d_Qd) HIP_CHECK(hipFree(c->d_Qd));
      c->d_Qd = nullptr;
      c->hyb_qd_rows = 0;
      HIP_ALLOC(hipMalloc(&c->d_Qd, (size_t)nsc * c->hyb_pad));
      c->hyb_qd_rows = nsc;
      c->hyb_qd_zeroed_for = -1;
   }
   if (c->hyb_qd_zeroed_for != Sc * b) { // rows behind S b are multiplied like any other: zero (small buffer: all of it)
      HIP_CHECK(hipMemsetAsync(c->d_Qd, 0, (size_t)c->hyb_qd_rows * c->hyb_pad, c->stream));
      c->hyb_qd_zeroed_for = Sc * b;
   }
   const size_t ws = std::max(kern::gemm_i8_workspace_doubles(c->hyb_pad, c->N_pad, Sc, b, false), kern::gemm_i8_workspace_doubles(c->N_pad, c->hyb_pad, Sc, b, false));
   if (ws > c->i8ws_cap) {
      HIP_CHECK(hipStreamSynchronize(c->stream));
      if (c->d_i8ws) HIP_CHECK(hipFree(c->d_i8ws));
      c->d_i8ws = nullptr;
      c->i8ws_cap = 0;
      HIP_ALLOC(hipMalloc(&c->d_i8ws, ws * sizeof(double)));
      c->i8ws_cap = ws;
   }
}

// The sparse route needs 8 bytes per missing call (2 GB at 500k x 100k and 0.5 %) plus one N x b plane.  If that does not
// fit, the context takes the dense missing-indicator route (both integer matrices on the matrix cores) from here on --
// out-of-memory only; any other failure is reported.  Returns the mode to use.
int sparse_or_dense(fpca_ctx *c, int b, int want = I8M_SPARSE)
{
   try {
      if (want == I8M_HYBRID && !c->hyb_view) { // (the shard qualified after the sample-major copy was made -- e.g. forced late)
         c->hyb_failed = true;
         return i8_mode(c, b);
      }
      if (want != I8M_HYBRID) plain_view(c);
      ensure_sparse(c, b);
      if (want == I8M_HYBRID) ensure_hybrid(c, b);
      return want;
   } catch (const Error &e) {
      if (e.code != FPCA_ENOMEM) throw;
      (void)hipGetLastError();
      std::fprintf(stderr, "[fpca] the missing-call index lists do not fit in device memory (%s); using the dense missing-indicator route\n", e.what());
      void **ptrs[] = {(void **)&c->d_snp_ptr, (void **)&c->d_snp_idx, (void **)&c->d_smp_ptr, (void **)&c->d_smp_idx, (void **)&c->d_eplane};
      for (void **p : ptrs)
         if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
         }
      c->eplane_cap = 0;
      c->sparse_ready = false;
      c->sparse_failed = true;
      plain_view(c);
      return i8_mode(c, b);
   }
}

kern::SliceOp i8_op_b(fpca_ctx *c)
{
   return kern::SliceOp{nullptr, reinterpret_cast<unsigned long long *>(c->d_i8w + I8W_MAXB), c->d_Qb, c->d_i8w + I8W_B,
                        reinterpret_cast<long long *>(c->d_i8w + I8W_CSB)};
}
void i8_ops_t(fpca_ctx *c, kern::SliceOp *ops) // the two K3 operands: T/sd and mean T/sd
{
   ops[0] = kern::SliceOp{c->d_inv_sd, reinterpret_cast<unsigned long long *>(c->d_i8w + I8W_MAXG), c->d_Qg, c->d_i8w + I8W_G, nullptr};
   ops[1] = kern::SliceOp{c->d_mu_inv_sd, reinterpret_cast<unsigned long long *>(c->d_i8w + I8W_MAXM), c->d_Qm, c->d_i8w + I8W_M,
                          reinterpret_cast<long long *>(c->d_i8w + I8W_CSM)};
}
void i8_zero_meta(fpca_ctx *c, hipStream_t s)
{
   HIP_CHECK(hipMemsetAsync(c->d_i8w + I8W_ZERO, 0, (I8W_TOTAL - I8W_ZERO) * sizeof(double), s));
}

// T = X' B : slices of B against the SNP-major stream, per-SNP mean / sd applied in the combine; with `chain` the
// combine also leaves the column maxima of the two K3 operands (the meta region must have been zeroed by the caller)
void xt_i8(fpca_ctx *c, const double *dB, int b, hipStream_t s, bool chain, hipEvent_t *gev = nullptr)
{
   kern::SliceOp ob = i8_op_b(c), ot[2];
   i8_ops_t(c, ot);
   int mode = i8_mode(c, b);
   const double *eplane = nullptr;
   hipEvent_t wait = nullptr;
   if (mode == I8M_SPARSE || mode == I8M_HYBRID)
      mode = sparse_or_dense(c, b, mode);
   else
      plain_view(c);
   const bool hyb = mode == I8M_HYBRID;
   const bool g32 = (mode == I8M_SPARSE || hyb) && c->gather_f32();
   if (g32) ob.copy32 = static_cast<float *>(c->gather_src()); // the slicing pass leaves the fp32 rows the gather reads
   kern::i8_colmax(dB, c->N, b, 1, &ob, s);
   kern::i8_slice(dB, c->N_pad, c->N, b, c->cur_S(), 1, &ob, s);
   if (hyb) // E_d' B of the dense SNPs on the matrix cores: their compacted records x the same slices of B -> [hyb_pad][b]
      kern::gemm_i8(c->d_packedE, c->pitch, c->d_Qb, c->d_Qb, ob.colw, ob.colw, nullptr, nullptr, nullptr, c->d_hyb_plane, c->d_i8ws, c->hyb_pad, c->N_pad,
                    c->hyb_n, I8M_NONE, nullptr, b, c->cur_S(), nullptr, s, nullptr, nullptr, true);
   if (mode == I8M_SPARSE || hyb) { // E'B: for every SNP the sum of the B rows of its missing samples, on the (low-priority) side
      hipStream_t gs = s;            // stream, released together with the GEMM
      if (sparse_on_side_stream(c, b)) {
         HIP_CHECK(hipEventRecord(c->ev_aux_go, s));
         HIP_CHECK(hipStreamWaitEvent(c->aux_stream, c->ev_aux_go, 0));
         gs = c->aux_stream;
      }
      if (g32)
         kern::sparse_rows_sum_f32(c->d_snp_ptr, c->d_snp_idx, ob.copy32, ob.colw, b, c->P_g, c->P_pad, c->d_eplane, gs);
      else
         kern::sparse_rows_sum(c->d_snp_ptr, c->d_snp_idx, dB, nullptr, b, c->P_g, c->P_pad, c->d_eplane, gs);
      if (hyb) kern::scatter_rows(c->d_hyb_plane, c->d_hyb_idx, c->hyb_n, b, c->d_eplane, gs); // (the gather wrote zeros there: empty lists)
      if (gs != s) {
         HIP_CHECK(hipEventRecord(c->ev_aux_done, c->aux_stream));
         wait = c->ev_aux_done;
      }
      eplane = c->d_eplane;
      mode = I8M_NONE;
   }
   kern::gemm_i8(c->d_packed, c->pitch, c->d_Qb, c->d_Qb, ob.colw, ob.colw, ob.colsum, c->d_mean, c->d_sd, c->d_T, c->d_i8ws, c->P_pad,
                 c->N_pad, c->P_g, mode, eplane, b, c->cur_S(), chain ? ot : nullptr, s, gev, wait);
}

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

// Y = X T : slices of T/sd and mean T/sd (one pass over T) against the sample-major copy, rows [r0, r1) of Y
void x_i8(fpca_ctx *c, int b, double *dY, hipStream_t s, bool have_max, bool do_slice = true, uint64_t r0 = 0, uint64_t r1 = 0,
          hipEvent_t *gev = nullptr)
{
   kern::SliceOp ot[2];
   i8_ops_t(c, ot);
   int mode = i8_mode(c, b);
   if (mode == I8M_SPARSE || mode == I8M_HYBRID)
      mode = sparse_or_dense(c, b, mode);
   else
      plain_view(c);
   const bool hyb = mode == I8M_HYBRID;
   if (hyb) mode = I8M_SPARSE; // (from here on the two routes differ only in what the gathered plane starts from)
   if (do_slice) {
      if (!have_max) kern::i8_colmax(c->d_T, c->P_g, b, 2, ot, s);
      const bool g32 = c->gather_f32();
      if (mode == I8M_SPARSE) { // the slicing pass leaves mean T / sd itself, row-major, for the gather (no per-entry row factor)
         if (g32)
            ot[1].copy32 = static_cast<float *>(c->gather_src());
         else
            ot[1].copy64 = static_cast<double *>(c->gather_src());
      }
      kern::i8_slice(c->d_T, c->P_pad, c->P_g, b, c->cur_S(), 2, ot, s);
      if (hyb) {
         // E_d (mean T / sd)_d: the dense SNPs' rows of the operand, gathered and scaled, sliced on their own (own column
         // scale), against the sample-major copy of their records -> one plane [N_pad][b] the gather below starts from
         kern::SliceOp od{nullptr, reinterpret_cast<unsigned long long *>(c->d_i8w + I8W_MAXD), c->d_Qd, c->d_i8w + I8W_D, nullptr};
         kern::gather_scaled_rows(c->d_T, c->d_mu_inv_sd, c->d_hyb_idx, c->hyb_n, c->hyb_pad, b, c->d_hyb_T, s);
         kern::i8_colmax(c->d_hyb_T, c->hyb_n, b, 1, &od, s);
         kern::i8_slice(c->d_hyb_T, c->hyb_pad, c->hyb_n, b, c->cur_S(), 1, &od, s);
         kern::gemm_i8(c->d_packedET, c->pitchET, c->d_Qd, c->d_Qd, od.colw, od.colw, nullptr, nullptr, nullptr, c->d_hyb_plane, c->d_i8ws, c->N_pad,
                       c->hyb_pad, c->N, I8M_NONE, nullptr, b, c->cur_S(), nullptr, s, nullptr, nullptr, true);
      }
      const double *init = hyb ? c->d_hyb_plane : nullptr;
      if (mode == I8M_SPARSE) { // E (mean T / sd): for every sample the sum of the scaled T rows of its missing SNPs
         hipStream_t gs = s;
         const double per_sample = (double)(c->hyb_view ? c->hyb_sparse_nnz : c->n_missing) / (double)std::max<uint64_t>(c->N, 1); // listed calls per sample
         if (sparse_on_side_stream(c, b)) {
            HIP_CHECK(hipEventRecord(c->ev_aux_go, s)); // T is complete on s here (and the K2 combine has consumed the plane)
            HIP_CHECK(hipStreamWaitEvent(c->aux_stream, c->ev_aux_go, 0));
            gs = c->aux_stream;
         }
         if (g32)
            kern::sparse_rows_sum_f32(c->d_smp_ptr, c->d_smp_idx, ot[1].copy32, ot[1].colw, b, c->N, c->N_pad, c->d_eplane, gs, init, true, per_sample);
         else
            kern::sparse_rows_sum(c->d_smp_ptr, c->d_smp_idx, ot[1].copy64, nullptr, b, c->N, c->N_pad, c->d_eplane, gs, init, true, per_sample);
         if (gs != s) HIP_CHECK(hipEventRecord(c->ev_aux_done, c->aux_stream));
      }
   }
   if (r1 == 0) r1 = c->N_pad;
   if (r1 <= r0) return;
   const double *eplane = nullptr;
   hipEvent_t wait = nullptr;
   if (mode == I8M_SPARSE) {
      eplane = c->d_eplane + r0 * b;
      if (sparse_on_side_stream(c, b)) wait = c->ev_aux_done;
      mode = I8M_NONE;
   }
   // G.M alone: one operand (Qm is still sliced: its column sums are 1'Qm, and M'Qm = 1'Qm - E'Qm)
   kern::gemm_i8(c->d_packedT + r0 * c->pitchT, c->pitchT, c->d_Qg, mode == I8M_NONE ? c->d_Qg : c->d_Qm, ot[0].colw, ot[1].colw,
                 ot[1].colsum, nullptr, nullptr, dY + r0 * b, c->d_i8ws, r1 - r0, c->P_pad, c->N > r0 ? std::min(c->N - r0, r1 - r0) : 0, mode,
                 eplane, b, c->cur_S(), nullptr, s, gev, wait);
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

// the operator on device-resident blocks: dY = X_g X_g' dB (+ all-reduce).  ev (optional): 4 events recorded
// at [start, after K2(+reduce), after K3(+reduce), after all-reduce].
void apply_xxt_dev(fpca_ctx *c, const double *dB, int b, double *dY, hipStream_t s, hipEvent_t *ev, bool reduce = true)
{
   ensure_stats(c);
   if (c->i8_S && ensure_i8(c, b)) {
      c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
      if (ev) HIP_CHECK(hipEventRecord(ev[0], s));
      i8_zero_meta(c, s);
      xt_i8(c, dB, b, s, true, ev ? ev + 4 : nullptr);
      if (ev) HIP_CHECK(hipEventRecord(ev[1], s));
      const int nch = reduce ? ar_chunks(c) : 1;
      if (nch > 1) {
         if (ev) HIP_CHECK(hipEventRecord(ev[6], s)); // (chunked: the "GEMM kernel" interval spans all chunks, slicing included)
         for (int i = 0; i < nch; i++) {
            const uint64_t r0 = ar_chunk_begin(c, nch, i), r1 = ar_chunk_begin(c, nch, i + 1);
            x_i8(c, b, dY, s, true, i == 0, r0, r1);
            if (r1 <= r0) continue;
            HIP_CHECK(hipEventRecord(c->ev_chunk[i], s));
            HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_chunk[i], 0));
            RCCL_CHECK(rccl().AllReduce(dY + r0 * b, dY + r0 * b, (r1 - r0) * b, ncclDouble, ncclSum, c->comm, c->comm_stream));
         }
         if (ev) HIP_CHECK(hipEventRecord(ev[7], s));
         if (ev) HIP_CHECK(hipEventRecord(ev[2], s));
         HIP_CHECK(hipEventRecord(c->ev_comm_done, c->comm_stream));
         HIP_CHECK(hipStreamWaitEvent(s, c->ev_comm_done, 0));
         if (ev) HIP_CHECK(hipEventRecord(ev[3], s));
         return;
      }
      x_i8(c, b, dY, s, true, true, 0, 0, ev ? ev + 6 : nullptr);
      if (ev) HIP_CHECK(hipEventRecord(ev[2], s));
      if (reduce) allreduce_rows(c, dY, b, s);
      if (ev) HIP_CHECK(hipEventRecord(ev[3], s));
      return;
   }
   const int s2 = c->dense ? kern::xt_b_dense_splits(c->N_pad, c->P_pad) : kern::xt_b_splits(c->N_pad, c->P_pad, b, c->accum == FPCA_ACCUM_FP32);
   const int s3 = c->dense ? kern::x_t_dense_splits(c->N_pad, c->P_pad) : kern::x_t_splits(c->N_pad, c->P_pad, b, c->accum == FPCA_ACCUM_FP32);
   c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
   size_t need = 0;
   if (s2 > 1) need = std::max(need, (size_t)s2 * c->P_pad * b);
   if (s3 > 1) need = std::max(need, (size_t)s3 * c->N_pad * b);
   if (need) c->ensure(c->d_part, c->part_cap, need);
   if (ev) HIP_CHECK(hipEventRecord(ev[0], s));
   if (ev) HIP_CHECK(hipEventRecord(ev[4], s));
   if (c->dense)
      kern::xt_b_dense(c->d_Xd, dB, s2 > 1 ? c->d_part : c->d_T, c->N_pad, c->P_pad, b, s2, s);
   else
      kern::xt_b(c->d_packed, c->pitch, c->d_lut, dB, s2 > 1 ? c->d_part : c->d_T, c->N_pad, c->P_pad, b, s2, c->accum == FPCA_ACCUM_FP32, s);
   if (ev) HIP_CHECK(hipEventRecord(ev[5], s));
   if (s2 > 1) kern::reduce_sum(c->d_part, c->d_T, (uint64_t)c->P_pad * b, s2, s);
   if (ev) HIP_CHECK(hipEventRecord(ev[1], s));
   if (ev) HIP_CHECK(hipEventRecord(ev[6], s));
   if (c->dense)
      kern::x_t_dense(c->d_Xd, c->d_T, s3 > 1 ? c->d_part : dY, c->N_pad, c->P_pad, b, s3, s);
   else
      kern::x_t(c->d_packed, c->pitch, c->d_lut, c->d_T, s3 > 1 ? c->d_part : dY, c->N_pad, c->P_pad, b, s3, c->accum == FPCA_ACCUM_FP32, s);
   if (ev) HIP_CHECK(hipEventRecord(ev[7], s));
   if (s3 > 1) kern::reduce_sum(c->d_part, dY, (uint64_t)c->N_pad * b, s3, s);
   if (ev) HIP_CHECK(hipEventRecord(ev[2], s));
   if (reduce) allreduce_rows(c, dY, b, s);
   if (ev) HIP_CHECK(hipEventRecord(ev[3], s));
}

// The operator on a ROW-SHARDED block (the eigensolver's view, backend.hpp RowShard): all-gather the rows of the input block
// (K2 sums over all samples), K2, K3 on the whole block, reduce-scatter the partial products -- the same bytes on the wire
// as the all-reduce of apply_xxt_dev, but every rank ends up with only ITS rows of the sum, which is all the
// orthogonalisation that follows needs.  With the built-in communicator and more than one chunk, K3 runs chunk by chunk
// and the reduce-scatter of chunk i rides on the communication stream under the computation of chunk i + 1.
void apply_sharded(fpca_ctx *c, const RowShard &sh, const double *in_slice, int b, double *out_slice, hipStream_t s)
{
   c->all_gather(sh, in_slice, c->d_full_in, b, s);
   if (sh.nch > 1 && c->native_collectives() && c->comm_stream) {
      ensure_stats(c);
      if (c->i8_S && ensure_i8(c, b)) {
         c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
         i8_zero_meta(c, s);
         xt_i8(c, c->d_full_in, b, s, true);
         for (int i = 0; i < sh.nch; i++) {
            const uint64_t r0 = std::min<uint64_t>((uint64_t)i * sh.L, c->N_pad), r1 = std::min<uint64_t>((uint64_t)(i + 1) * sh.L, c->N_pad);
            if (r1 <= r0 && i > 0) { // a chunk wholly behind the last row (the same on every rank): no K3, no collective, zeros out
               HIP_CHECK(hipMemsetAsync(out_slice + (size_t)i * sh.plen * b, 0, (size_t)sh.plen * b * sizeof(double), s));
               continue;
            }
            x_i8(c, b, c->d_full_out, s, true, i == 0, r0, r1); // (rows >= N_pad of d_full_out stay zero: nothing writes them)
            HIP_CHECK(hipEventRecord(c->ev_chunk[i], s));
            HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_chunk[i], 0));
            c->reduce_scatter(sh, c->d_full_out, out_slice, b, c->comm_stream, i);
         }
         HIP_CHECK(hipEventRecord(c->ev_comm_done, c->comm_stream));
         HIP_CHECK(hipStreamWaitEvent(s, c->ev_comm_done, 0));
         return;
      }
   }
   apply_xxt_dev(c, c->d_full_in, b, c->d_full_out, s, nullptr, false);
   c->reduce_scatter(sh, c->d_full_out, out_slice, b, s);
}

void xt_dev(fpca_ctx *c, const double *dB, int b, hipStream_t s)
{
   ensure_stats(c);
   if (c->i8_S && ensure_i8(c, b)) {
      c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
      i8_zero_meta(c, s);
      xt_i8(c, dB, b, s, false);
      return;
   }
   const int s2 = c->dense ? kern::xt_b_dense_splits(c->N_pad, c->P_pad) : kern::xt_b_splits(c->N_pad, c->P_pad, b, c->accum == FPCA_ACCUM_FP32);
   c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
   if (s2 > 1) c->ensure(c->d_part, c->part_cap, (size_t)s2 * c->P_pad * b);
   if (c->dense)
      kern::xt_b_dense(c->d_Xd, dB, s2 > 1 ? c->d_part : c->d_T, c->N_pad, c->P_pad, b, s2, s);
   else
      kern::xt_b(c->d_packed, c->pitch, c->d_lut, dB, s2 > 1 ? c->d_part : c->d_T, c->N_pad, c->P_pad, b, s2, c->accum == FPCA_ACCUM_FP32, s);
   if (s2 > 1) kern::reduce_sum(c->d_part, c->d_T, (uint64_t)c->P_pad * b, s2, s);
}

void x_dev(fpca_ctx *c, int b, double *dY, hipStream_t s)
{
   ensure_stats(c);
   if (c->i8_S && ensure_i8(c, b)) {
      i8_zero_meta(c, s);
      x_i8(c, b, dY, s, false);
      return;
   }
   const int s3 = c->dense ? kern::x_t_dense_splits(c->N_pad, c->P_pad) : kern::x_t_splits(c->N_pad, c->P_pad, b, c->accum == FPCA_ACCUM_FP32);
   if (s3 > 1) c->ensure(c->d_part, c->part_cap, (size_t)s3 * c->N_pad * b);
   if (c->dense)
      kern::x_t_dense(c->d_Xd, c->d_T, s3 > 1 ? c->d_part : dY, c->N_pad, c->P_pad, b, s3, s);
   else
      kern::x_t(c->d_packed, c->pitch, c->d_lut, c->d_T, s3 > 1 ? c->d_part : dY, c->N_pad, c->P_pad, b, s3, c->accum == FPCA_ACCUM_FP32, s);
   if (s3 > 1) kern::reduce_sum(c->d_part, dY, (uint64_t)c->N_pad * b, s3, s);
}

inline int pad16(int b) { return (int)round_up((uint64_t)b, 16); }

void ensure_io(fpca_ctx *c)
{
   if (!c->d_io_a) HIP_CHECK(hipMalloc(&c->d_io_a, (size_t)c->N_pad * MAX_BLOCKVEC * sizeof(double)));
   if (!c->d_io_b) HIP_CHECK(hipMalloc(&c->d_io_b, (size_t)c->N_pad * MAX_BLOCKVEC * sizeof(double)));
}

// ---- HIP backend for the eigensolver ---------------------------------------------------------------
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

class HipBackend : public BlockBackend {
 public:
   // cheap_S: byte slices of the eigensolver's cheap passes (0: none).  Whether they exist is decided from the arithmetic the
   // context was CREATED with, not from what it runs now: a rank whose exact-integer buffers did not fit runs the fp64 kernels
   // for every pass but must follow the same sequence of passes as the others.
   HipBackend(fpca_ctx *c, int b, bool replicated = false, int cheap_S = 0)
      : c_(c), b_(b), cheap_S_((c->i8_S_req > 0 && cheap_S >= 2 && cheap_S < c->i8_S_req) ? cheap_S : 0), d_ptrs_(c->be_ptrs), d_C_(c->be_C), d_gpart_(c->be_gpart), C_cap_(c->be_C_cap), gpart_cap_(c->be_gpart_cap),
        h_pin_(c->be_pin), pin_cap_(c->be_pin_cap)
   {
      HIP_CHECK(hipSetDevice(c->device));
      ensure_stats(c);
      rows_ = c->N_pad;
      // several ranks whose rank / size are known: the solver's N-sized objects are row slices (backend.hpp RowShard);
      // `replicated` keeps every rank's copy whole, as in round 2 (A/B, and contexts that only have an all-reduce hook
      // without fpca_set_rank).  FPCA_FORCE_ROWSHARD (test builds): the sharded code path with a single rank.
      const bool force = FPCA_TEST_ENV("FPCA_FORCE_ROWSHARD") != nullptr;
      if (!replicated && ((c->multi() && c->rank_known && c->nranks > 1) || force)) {
         const int G = (c->multi() && c->rank_known) ? c->nranks : 1;
         // (chunk count from N and the communicator only -- NOT from the arithmetic in effect: a rank whose int8 buffers did
         //  not fit runs the fp64 kernels but must issue the same sequence of collectives as the others)
         const int nch = shard_chunks(c);
         sh_ = RowShard::make(c->N_pad, G, (c->multi() && c->rank_known) ? c->rank : 0, nch, 512);
         rows_ = sh_.slice_rows();
         const size_t need = (size_t)sh_.full_rows() * b;
         if (need > c->full_in_cap || need > c->full_out_cap) {
            c->ensure(c->d_full_in, c->full_in_cap, need);
            c->ensure(c->d_full_out, c->full_out_cap, need);
         }
         // rows >= N_pad of the whole blocks are never written by the kernels and must read as zero in the collectives
         HIP_CHECK(hipMemsetAsync(c->d_full_in, 0, c->full_in_cap * sizeof(double), c->stream));
         HIP_CHECK(hipMemsetAsync(c->d_full_out, 0, c->full_out_cap * sizeof(double), c->stream));
      }
      if (sharded() && sh_.G > 1) exchange_selftest();
      if (!d_ptrs_) HIP_CHECK(hipMalloc(&d_ptrs_, 1024 * sizeof(double *)));
      HIP_CHECK(hipEventCreate(&e0_));
      HIP_CHECK(hipEventCreate(&e1_));
      HIP_CHECK(hipEventCreateWithFlags(&ev_pin_, hipEventDisableTiming));
      (void)pin_coeff((size_t)16 * b * b);
   }
   ~HipBackend() override
   {
      c_->i8_Sc = 0;
      (void)hipStreamSynchronize(c_->stream);
      for (double *p : blocks_)
         if (p) c_->block_pool.emplace_back(block_bytes(), p);
      (void)hipEventDestroy(ev_pin_);
      (void)hipEventDestroy(e0_);
      (void)hipEventDestroy(e1_);
   }
   // Once per context and layout (ranks, chunks, width): the all-gather and the reduce-scatter of the row-sharded solver, as
   // apply_sharded issues them, on a block whose every entry is known -- the rows a rank keeps of the random block with seed
   // 4711 go out, the whole block must come back (all-gather), and nranks times a rank's own rows out of the whole block on
   // every rank (reduce-scatter, chunk by chunk on the communication stream).  A wrong chunk / piece offset or a collective
   // that pairs the wrong buffers ends the solve here with FPCA_ECOMM instead of returning plausible-looking eigenvectors.
   void exchange_selftest()
   {
      const long key = ((long)sh_.G * 8 + sh_.nch) * 128 + b_;
      if (c_->exchange_tested == key) return;
      hipStream_t s = c_->stream;
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
         if (err[0] != 0.0 || !(err[1] <= 1e-14 * sh_.G))
            throw Error(FPCA_ECOMM, "self-test of the multi-rank exchange failed on rank " + std::to_string(sh_.rank) + " of " + std::to_string(sh_.G) + " (" +
                                        std::to_string(sh_.nch) + " row chunks): all-gather error " + std::to_string(err[0]) + ", reduce-scatter error " +
                                        std::to_string(err[1]));
      } catch (...) {
         finish();
         throw;
      }
      finish();
      c_->coll_calls = calls0;
      c_->coll_bytes = bytes0;
      c_->exchange_tested = key;
   }
   uint64_t nrows() const override { return c_->N; }
   int width() const override { return b_; }
   size_t block_bytes() const { return (size_t)rows_ * b_ * sizeof(double); }
   bool sharded() const { return sh_.on(); }
   const RowShard &shard() const { return sh_; }
   int alloc_block() override
   {
      for (size_t i = 0; i < used_.size(); i++)
         if (!used_[i]) {
            used_[i] = 1;
            return (int)i;
         }
      double *p = nullptr;
      for (size_t i = 0; i < c_->block_pool.size(); i++)
         if (c_->block_pool[i].first == block_bytes()) {
            p = c_->block_pool[i].second;
            c_->block_pool.erase(c_->block_pool.begin() + (long)i);
            break;
         }
      if (!p) HIP_CHECK(hipMalloc(&p, block_bytes()));
      blocks_.push_back(p);
      used_.push_back(1);
      return (int)blocks_.size() - 1;
   }
   void free_block(int h) override { used_[h] = 0; }
   // the WHOLE block h as the operator wants it ([N_pad][b] on this device): the block itself, or -- row-sharded -- its rows
   // gathered from all ranks into the context's scratch (a collective: every rank calls it)
   double *full_ptr(int h)
   {
      if (!sharded()) return blocks_[h];
      c_->all_gather(sh_, blocks_[h], c_->d_full_in, b_, c_->stream);
      return c_->d_full_in;
   }
   void fill_random(int h, uint64_t seed) override
   {
      if (!sharded()) {
         kern::fill_random(blocks_[h], c_->N, c_->N_pad, b_, seed, c_->stream);
         return;
      }
      for (int c = 0; c < sh_.nch; c++) // the rows this rank keeps of the block every rank would have generated
         kern::fill_random(blocks_[h] + (size_t)c * sh_.plen * b_, c_->N, sh_.plen, b_, seed, c_->stream, (uint64_t)c * sh_.L + (uint64_t)sh_.rank * sh_.plen);
   }
   void apply(int in, int out) override
   {
      apply_begin(in, out);
      apply_end();
   }
   void apply_begin(int in, int out) override
   {
      HIP_CHECK(hipEventRecord(e0_, c_->stream));
      if (sharded())
         apply_sharded(c_, sh_, blocks_[in], b_, blocks_[out], c_->stream);
      else
         apply_xxt_dev(c_, blocks_[in], b_, blocks_[out], c_->stream, nullptr);
      HIP_CHECK(hipEventRecord(e1_, c_->stream));
      inflight_ = true;
      inflight_exact_ = !c_->i8_Sc;
   }
   void apply_end() override
   {
      if (!inflight_) return;
      inflight_ = false;
      HIP_CHECK(hipEventSynchronize(e1_));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0_, e1_));
      sec_apply_ += ms * 1e-3;
      if (inflight_exact_) sec_exact_ += ms * 1e-3;
   }
   bool set_cheap(bool cheap) override
   {
      if (!cheap_S_) return false;
      c_->i8_Sc = cheap ? cheap_S_ : 0;
      return true;
   }
   int cheap_slices() const { return cheap_S_; }
   double seconds_exact() const { return sec_exact_; }
   // Host <-> device traffic of the small matrices goes through one pinned buffer ([1024 pointers][coefficients]); the
   // event marks the last asynchronous read of it, so a call never overwrites what an earlier copy has not picked up.
   void pin_wait()
   {
      if (pin_busy_) HIP_CHECK(hipEventSynchronize(ev_pin_));
      pin_busy_ = false;
   }
   double *pin_coeff(size_t cnt)
   {
      pin_wait();
      if (cnt > pin_cap_) {
         if (h_pin_) HIP_CHECK(hipHostFree(h_pin_));
         h_pin_ = nullptr;
         pin_cap_ = std::max(cnt, 2 * pin_cap_);
         HIP_CHECK(hipHostMalloc(&h_pin_, 1024 * sizeof(double *) + pin_cap_ * sizeof(double), hipHostMallocDefault));
      }
      return reinterpret_cast<double *>(static_cast<char *>(h_pin_) + 1024 * sizeof(double *));
   }
   void push_ptrs(const int *a, int nq)
   {
      if (nq > 1024) throw Error(FPCA_EINVAL, "too many basis blocks");
      const double **hp = static_cast<const double **>(h_pin_);
      for (int q = 0; q < nq; q++) hp[q] = blocks_[a[q]];
      HIP_CHECK(hipMemcpyAsync(d_ptrs_, hp, nq * sizeof(double *), hipMemcpyHostToDevice, c_->stream));
   }
   void gram(const int *a, int nq, int w, double *C) override
   {
      auto t0 = std::chrono::steady_clock::now();
      const size_t cnt = (size_t)nq * b_ * b_;
      const int rows = kern::gram_rows(rows_, nq);
      const int ns = kern::gram_splits(rows_, rows) * 4;
      grow(d_gpart_, gpart_cap_, cnt * ns);
      grow(d_C_, C_cap_, std::max(cnt, (size_t)1024 * b_ * 4));
      double *hc = pin_coeff(cnt);
      push_ptrs(a, nq);
      kern::gram(d_ptrs_, nq, blocks_[w], d_gpart_, rows_, b_, rows, c_->stream);
      kern::reduce_sum(d_gpart_, d_C_, cnt, ns, c_->stream);
      if (sharded() && sh_.G > 1) c_->allreduce(d_C_, cnt, c_->stream); // row slices: the only collective of the orthogonalisation
      HIP_CHECK(hipMemcpyAsync(hc, d_C_, cnt * sizeof(double), hipMemcpyDeviceToHost, c_->stream));
      HIP_CHECK(hipStreamSynchronize(c_->stream));
      std::memcpy(C, hc, cnt * sizeof(double));
      sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   }
   void gemm(const int *a, int nq, const double *C, int init, int out) override
   {
      auto t0 = std::chrono::steady_clock::now();
      const size_t cnt = (size_t)nq * b_ * b_;
      grow(d_C_, C_cap_, std::max(cnt, (size_t)1024 * b_ * 4));
      double *hc = pin_coeff(cnt);
      std::memcpy(hc, C, cnt * sizeof(double));
      push_ptrs(a, nq);
      HIP_CHECK(hipMemcpyAsync(d_C_, hc, cnt * sizeof(double), hipMemcpyHostToDevice, c_->stream));
      HIP_CHECK(hipEventRecord(ev_pin_, c_->stream));
      pin_busy_ = true;
      // not drained: d_C_ / d_ptrs_ are only rewritten by later copies on this same stream, i.e. after the kernel
      kern::block_gemm(d_ptrs_, nq, d_C_, init >= 0 ? blocks_[init] : nullptr, blocks_[out], rows_, b_, c_->stream);
      sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   }
   void download(int h, int ncols, double *host, int64_t ld) override { download2(h, ncols, host, ld, nullptr, 0, nullptr); }
   void download2(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale) override
   {
      if (!host && !host2 && !sharded()) return;
      const double *whole = full_ptr(h); // (row-sharded: a collective -- every rank comes here, whether it wants the result or not)
      if (!host && !host2) return;
      c_->ensure(c_->d_stage, c_->stage_cap, (size_t)c_->N * ncols);
      kern::block_to_colmajor(whole, c_->N, b_, ncols, c_->d_stage, c_->N, c_->stream);
      staged_download(c_, c_->d_stage, c_->N, ncols, host, ld, host2, ld2, scale);
   }
   void upload(int h, int ncols, const double *host, int64_t ld) override
   {
      c_->ensure(c_->d_stage, c_->stage_cap, (size_t)c_->N * ncols);
      HIP_CHECK(hipMemcpy2DAsync(c_->d_stage, c_->N * sizeof(double), host, (size_t)ld * sizeof(double),
                                 c_->N * sizeof(double), ncols, hipMemcpyHostToDevice, c_->stream));
      if (!sharded())
         kern::colmajor_to_block(c_->d_stage, c_->N, c_->N, c_->N_pad, b_, ncols, blocks_[h], c_->stream);
      else { // the whole block into the scratch, this rank's rows out of it
         kern::colmajor_to_block(c_->d_stage, c_->N, c_->N, c_->N_pad, b_, ncols, c_->d_full_out, c_->stream);
         for (int c = 0; c < sh_.nch; c++)
            HIP_CHECK(hipMemcpyAsync(blocks_[h] + (size_t)c * sh_.plen * b_, c_->d_full_out + ((size_t)c * sh_.L + (size_t)sh_.rank * sh_.plen) * b_,
                                     (size_t)sh_.plen * b_ * sizeof(double), hipMemcpyDeviceToDevice, c_->stream));
      }
      HIP_CHECK(hipStreamSynchronize(c_->stream));
   }
   double trace() override
   {
      double t = c_->trace_local;
      if (c_->multi()) {
         HIP_CHECK(hipMemcpyAsync(c_->d_small, &t, sizeof(double), hipMemcpyHostToDevice, c_->stream));
         c_->allreduce(c_->d_small, 1, c_->stream);
         HIP_CHECK(hipMemcpyAsync(&t, c_->d_small, sizeof(double), hipMemcpyDeviceToHost, c_->stream));
         HIP_CHECK(hipStreamSynchronize(c_->stream));
      }
      return t;
   }
   double seconds_apply() override { return sec_apply_; }
   double seconds_other() override { return sec_other_; }

 private:
   void grow(double *&p, size_t &cap, size_t need)
   {
      if (need <= cap) return;
      HIP_CHECK(hipStreamSynchronize(c_->stream));
      if (p) HIP_CHECK(hipFree(p));
      p = nullptr;
      need = std::max(need, 2 * cap); // geometric: the basis grows by one block per step
      HIP_CHECK(hipMalloc(&p, need * sizeof(double)));
      cap = need;
   }
   fpca_ctx *c_;
   int b_;
   int cheap_S_; // byte slices of the cheap passes (0: the backend has none)
   uint64_t rows_ = 0; // rows of a block as THIS rank stores it: N_pad, or its slice of the row-sharded solver
   RowShard sh_;
   std::vector<double *> blocks_;
   std::vector<unsigned char> used_;
   const double **&d_ptrs_; // scratch owned by the context (kept across solves)
   double *&d_C_, *&d_gpart_;
   size_t &C_cap_, &gpart_cap_;
   void *&h_pin_;
   size_t &pin_cap_;
   bool pin_busy_ = false, inflight_ = false, inflight_exact_ = false;
   hipEvent_t e0_, e1_, ev_pin_;
   double sec_apply_ = 0, sec_other_ = 0, sec_exact_ = 0;
};

template <typename F> int guarded(F &&f)
{
   try {
      f();
      return FPCA_OK;
   } catch (const Error &e) {
      set_last_error(e.what());
      return e.code;
   } catch (const std::bad_alloc &) {
      set_last_error("host allocation failed");
      return FPCA_ENOMEM;
   } catch (const std::exception &e) {
      set_last_error(e.what());
      return FPCA_EHIP;
   }
}

} // namespace

// =====================================================================================================
extern "C" {

const char *fpca_last_error(void) { return fpca::g_last_error.c_str(); }
const char *fpca_version(void) { return FPCA_VERSION; }
int fpca_abi_version(void) { return FPCA_ABI_VERSION; }

int fpca_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return -1;
   return n;
}

int fpca_warmup(int device)
{
   return guarded([&] {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(FPCA_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
      if (device < 0 || device >= ndev) throw Error(FPCA_ENODEVICE, "device index out of range");
      HIP_CHECK(hipSetDevice(device));
      HIP_CHECK(hipFree(nullptr)); // creates the primary context
      void *p = nullptr;          // first allocation + first pinned allocation: the runtime sets its pools up here
      HIP_CHECK(hipMalloc(&p, 1 << 20));
      HIP_CHECK(hipFree(p));
      HIP_CHECK(hipHostMalloc(&p, 1 << 20, hipHostMallocDefault));
      HIP_CHECK(hipHostFree(p));
   });
}

int fpca_device_name(int device, char *buf, int buflen)
{
   return guarded([&] {
      hipDeviceProp_t prop;
      HIP_CHECK(hipGetDeviceProperties(&prop, device));
      std::snprintf(buf, buflen, "%s (%s, %d CUs, %.1f GiB)", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                    (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
   });
}

int fpca_create(fpca_ctx **out, const uint8_t *packed, uint64_t N, uint64_t P_g, int stand_method, int device, int accum)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!packed && P_g > 0) throw Error(FPCA_EINVAL, "packed is NULL");
      ctx_alloc_common(c, N, P_g, stand_method, device, accum);
      if (P_g > 0)
         HIP_CHECK(hipMemcpy2DAsync(c->d_packed, c->pitch, packed, c->np, c->np, P_g, hipMemcpyHostToDevice, c->stream));
      ctx_finish_upload(c);
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

namespace {
// One pread stream tops out near 10 GB/s from the page cache (it is a single-threaded memcpy), a fifth of what the
// PCIe link takes; 16 threads reading disjoint slices of the chunk reach 17-23 GB/s (measured on the overlay file system of the test box).
bool parallel_pread(int fd, uint8_t *buf, uint64_t want, off_t off)
{
   static const unsigned hw = usable_cpus(); // (the CPUs this process may run on at once, not the host's hardware threads)
   static const int cap = getenv("FPCA_READ_THREADS") ? std::max(1, atoi(getenv("FPCA_READ_THREADS"))) : 16;
   const int nt = want < (8u << 20) ? 1 : (int)std::min<unsigned>((unsigned)cap, hw ? hw : 1);
   std::atomic<bool> ok(true);
   auto work = [&](int t) {
      const uint64_t b0 = want * t / nt, b1 = want * (t + 1) / nt;
      uint64_t got = b0;
      while (got < b1) {
         const ssize_t k = pread(fd, buf + got, b1 - got, off + (off_t)got);
         if (k <= 0) {
            ok = false;
            return;
         }
         got += (uint64_t)k;
      }
   };
   if (nt == 1) {
      work(0);
      return ok;
   }
   std::vector<std::thread> th;
   for (int t = 1; t < nt; t++) th.emplace_back(work, t);
   work(0);
   for (auto &x : th) x.join();
   return ok;
}
} // namespace

int fpca_create_from_bed(fpca_ctx **out, const char *bed_path, uint64_t N, uint64_t snp_begin, uint64_t P_g,
                         int stand_method, int device, int accum, uint64_t *P_total)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int fd = -1;
   int rc = guarded([&] {
      if (N == 0) throw Error(FPCA_EINVAL, "N must be > 0");
      fd = open(bed_path, O_RDONLY);
      if (fd < 0) // data.cpp:156-161
         throw Error(FPCA_EIO, std::string("[Data::read_bed] Error reading file ") + bed_path + ", error " + strerror(errno));
      struct stat st;
      if (fstat(fd, &st) != 0 || st.st_size < 3) throw Error(FPCA_EIO, std::string("cannot stat ") + bed_path);
      // The reference skips the three header bytes unseen (data.cpp:218: seekg(3 + ...)); a sample-major file (third byte 0)
      // or something that is not a .bed at all would be decoded as garbage without a word.  Checked here.
      unsigned char magic[3] = {0, 0, 0};
      if (pread(fd, magic, 3, 0) != 3) throw Error(FPCA_EIO, std::string("[Data::read_bed] Error reading file ") + bed_path);
      if (magic[0] != 0x6c || magic[1] != 0x1b)
         throw Error(FPCA_EIO, std::string(bed_path) + " is not a PLINK .bed file (it does not start with the magic bytes 6c 1b)");
      if (magic[2] != 0x01)
         throw Error(FPCA_EIO, std::string(bed_path) + (magic[2] == 0x00 ? " is a sample-major .bed (header 6c 1b 00)" : " has an unknown .bed mode byte") +
                                   "; only SNP-major files (header 6c 1b 01) are supported -- convert with plink --make-bed");
      const uint64_t len = (uint64_t)st.st_size - 3; // data.cpp:165
      const uint64_t np = (N + 3) / 4;               // data.cpp:168
      const uint64_t nsnps = len / np;               // data.cpp:170 (integer division; .bim is not consulted)
      if (P_total) *P_total = nsnps;
      if (snp_begin > nsnps) throw Error(FPCA_EINVAL, "snp_begin beyond the end of the file");
      uint64_t pg = P_g ? P_g : nsnps - snp_begin;
      if (snp_begin + pg > nsnps) throw Error(FPCA_EINVAL, "SNP range beyond the end of the file");
      const bool timing = std::getenv("FPCA_TIMING") != nullptr;
      auto tl = std::chrono::steady_clock::now();
      auto lap = [&](const char *what) {
         const auto now = std::chrono::steady_clock::now();
         if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", what, std::chrono::duration<double>(now - tl).count() * 1e3);
         tl = now;
      };
      ctx_alloc_common(c, N, pg, stand_method, device, accum);
      lap("device init + allocations");
      c->P_total = nsnps;
      // stream the shard: contiguous byte range [3 + np*begin, 3 + np*(begin+pg)) of the file -> parallel pread into one of
      // two pinned bounce buffers -> 1-D H2D copy into a device staging buffer -> repitch kernel into the resident matrix;
      // the read of chunk i+1 overlaps the copy of chunk i
      const uint64_t rows_per_chunk = std::max<uint64_t>(1, (64ull << 20) / np);
      uint8_t *bounce[2] = {nullptr, nullptr}, *dstage[2] = {nullptr, nullptr};
      hipEvent_t done[2] = {nullptr, nullptr};
      auto cleanup = [&] {
         for (int i = 0; i < 2; i++) {
            if (bounce[i]) (void)hipHostFree(bounce[i]);
            if (dstage[i]) (void)hipFree(dstage[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
         }
      };
      try {
         int slot = 0;
         for (uint64_t r0 = 0; r0 < pg; r0 += rows_per_chunk, slot ^= 1) {
            const uint64_t nr = std::min(rows_per_chunk, pg - r0);
            if (!bounce[slot]) { // the second slot's 64 MB of pinned memory (14 ms to allocate) come while the first chunk is on the wire
               HIP_CHECK(hipHostMalloc(&bounce[slot], std::min(rows_per_chunk, pg) * np, hipHostMallocDefault));
               HIP_CHECK(hipMalloc(&dstage[slot], std::min(rows_per_chunk, pg) * np));
               HIP_CHECK(hipEventCreate(&done[slot]));
            }
            HIP_CHECK(hipEventSynchronize(done[slot])); // the previous copy out of this slot has finished
            const uint64_t want = nr * np;
            const off_t off = (off_t)(3 + np * (snp_begin + r0)); // data.cpp:218
            if (!parallel_pread(fd, bounce[slot], want, off)) throw Error(FPCA_EIO, std::string("short read from ") + bed_path);
            HIP_CHECK(hipMemcpyAsync(dstage[slot], bounce[slot], want, hipMemcpyHostToDevice, c->stream));
            kern::repitch(dstage[slot], np, nr, c->d_packed + r0 * c->pitch, c->pitch, c->stream);
            HIP_CHECK(hipEventRecord(done[slot], c->stream));
         }
         HIP_CHECK(hipStreamSynchronize(c->stream));
      } catch (...) {
         cleanup();
         throw;
      }
      cleanup();
      ctx_finish_upload(c);
      lap(".bed -> HBM");
   });
   if (fd >= 0) close(fd);
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

int fpca_create_synthetic(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed, int n_pop,
                          double fst, double missing_rate, int stand_method, int device, int accum)
{
   fpca_synth_model m;
   std::memset(&m, 0, sizeof(m));
   m.n_pop = n_pop;
   m.fst = fst;
   m.missing_rate = missing_rate;
   return fpca_create_synthetic_model(out, N, snp_begin, P_g, seed, &m, stand_method, device, accum);
}

int fpca_create_synthetic_model(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed, const fpca_synth_model *model,
                                int stand_method, int device, int accum)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!model) throw Error(FPCA_EINVAL, "model is NULL");
      const int n_pop = model->n_pop;
      const double fst = model->fst, missing_rate = model->missing_rate;
      if (n_pop < 1 || n_pop > synth::MAX_POP) throw Error(FPCA_EINVAL, "n_pop must be in 1..64");
      if (!(fst >= 0 && fst < 1) || !(missing_rate >= 0 && missing_rate < 1)) throw Error(FPCA_EINVAL, "fst / missing_rate out of range");
      if ((model->maf_model | 1) != 1 || (model->missing_model | 1) != 1 || !(model->conc_frac >= 0 && model->conc_frac <= 1))
         throw Error(FPCA_EINVAL, "maf_model / missing_model must be 0 or 1, conc_frac in [0, 1]");
      ctx_alloc_common(c, N, P_g, stand_method, device, accum);
      const uint32_t fst_fp = (uint32_t)std::llround(fst * 65536.0);
      const uint32_t miss_thr = (uint32_t)std::llround(missing_rate * 65536.0);
      kern::synth_generate(c->d_packed, c->pitch, N, snp_begin, P_g, seed, n_pop, fst_fp, miss_thr, c->stream, model->maf_model, model->missing_model,
                           (uint32_t)std::llround(model->conc_frac * 65536.0));
      HIP_CHECK(hipStreamSynchronize(c->stream));
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

int fpca_create_dense(fpca_ctx **out, const double *X, int64_t ldx, uint64_t N, uint64_t P_g, int stand_method, int device)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!X || ldx < (int64_t)N || P_g == 0) throw Error(FPCA_EINVAL, "bad argument to fpca_create_dense");
      ctx_alloc_common(c, N, P_g, stand_method, device, FPCA_ACCUM_FP64, true);
      // column-major N x P on the host == row-major [P][N] : one strided copy into the padded [P_pad][N_pad] image
      HIP_CHECK(hipMemcpy2DAsync(c->d_Xd, c->N_pad * sizeof(double), X, (size_t)ldx * sizeof(double), N * sizeof(double), P_g,
                                 hipMemcpyHostToDevice, c->stream));
      kern::dense_standardise(c->d_Xd, c->N_pad, N, P_g, stand_method, c->d_mean, c->d_sd, c->d_sumsq, c->stream);
      std::vector<double> ss(P_g);
      HIP_CHECK(hipMemcpyAsync(ss.data(), c->d_sumsq, P_g * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_CHECK(hipStreamSynchronize(c->stream));
      double tot = 0;
      for (size_t i0 = 0; i0 < ss.size(); i0 += 1024) {
         double sblk = 0;
         for (size_t i = i0; i < std::min(ss.size(), i0 + 1024); i++) sblk += ss[i];
         tot += sblk;
      }
      c->trace_local = tot; // randompca.cpp:154: sum X^2 of the standardised matrix
      c->stats_done = true;
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

void fpca_destroy(fpca_ctx *ctx) { ctx_free(ctx); }

uint64_t fpca_nsamples(const fpca_ctx *ctx) { return ctx ? ctx->N : 0; }
uint64_t fpca_nsnps(const fpca_ctx *ctx) { return ctx ? ctx->P_g : 0; }
uint64_t fpca_block_rows(const fpca_ctx *ctx) { return ctx ? ctx->N_pad : 0; }
void *fpca_stream(fpca_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int fpca_synchronize(fpca_ctx *ctx)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      HIP_CHECK(hipDeviceSynchronize());
   });
}

int fpca_download_packed(fpca_ctx *ctx, uint8_t *out)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      if (ctx->dense) throw Error(FPCA_EINVAL, "this context holds a dense matrix, not a packed stream");
      if (ctx->P_g == 0) return;
      HIP_CHECK(hipMemcpy2D(out, ctx->np, ctx->d_packed, ctx->pitch, ctx->np, ctx->P_g, hipMemcpyDeviceToHost));
      // the pad bits of the last byte were rewritten to "missing" on upload; PLINK writes them as 0
      if (ctx->N % 4) {
         const uint8_t keep = (uint8_t)((1u << (2 * (ctx->N % 4))) - 1u);
         for (uint64_t j = 0; j < ctx->P_g; j++) out[j * ctx->np + ctx->np - 1] &= keep;
      }
   });
}

int fpca_stats(fpca_ctx *ctx, double *mean_sd, double *trace_out)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      if (mean_sd && ctx->P_g) {
         HIP_CHECK(hipMemcpy(mean_sd, ctx->d_mean, ctx->P_g * sizeof(double), hipMemcpyDeviceToHost));
         HIP_CHECK(hipMemcpy(mean_sd + ctx->P_g, ctx->d_sd, ctx->P_g * sizeof(double), hipMemcpyDeviceToHost));
      }
      if (trace_out) *trace_out = ctx->trace_local;
   });
}

int fpca_set_meansd(fpca_ctx *ctx, const double *mean_sd)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      if (!mean_sd) throw Error(FPCA_EINVAL, "mean_sd is NULL");
      if (ctx->dense) throw Error(FPCA_EINVAL, "preloaded mean/sd applies to packed genotypes only");
      HIP_CHECK(hipMemcpy(ctx->d_mean, mean_sd, ctx->P_g * sizeof(double), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(ctx->d_sd, mean_sd + ctx->P_g, ctx->P_g * sizeof(double), hipMemcpyHostToDevice));
      kern::lut_from_meansd(ctx->d_mean, ctx->d_sd, ctx->P_g, ctx->d_lut, ctx->stream);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      ctx->i8_scales_done = false;
      ctx->stats_done = true; // trace of the preloaded standardisation is not defined by the reference path
      ctx->trace_local = 0;
   });
}

int fpca_accum(const fpca_ctx *ctx) { return ctx ? ctx->accum : FPCA_EINVAL; }

int fpca_missing_mode(fpca_ctx *ctx, int b)
{
   int mode = -1;
   int rc = guarded([&] {
      if (!ctx || b < 1 || b > MAX_BLOCKVEC) throw Error(FPCA_EINVAL, "bad argument to fpca_missing_mode");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      mode = ctx->i8_S ? i8_mode(ctx, (int)round_up((uint64_t)b, 16)) : -1;
   });
   return rc == FPCA_OK ? mode : rc;
}

int fpca_allreduce_chunks(fpca_ctx *ctx) { return ctx ? ar_chunks(ctx) : FPCA_EINVAL; }

// ---- operator, host pointers -------------------------------------------------------------------------
int fpca_apply_xxt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *Y, int64_t ldy)
{
   return guarded([&] {
      if (!ctx || !B || !Y || b < 1 || ldb < (int64_t)ctx->N || ldy < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xxt");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)ctx->N * nc);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->N * sizeof(double), B + (size_t)c0 * ldb, (size_t)ldb * sizeof(double),
                                    ctx->N * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_block(ctx->d_stage, ctx->N, ctx->N, ctx->N_pad, bw, nc, ctx->d_io_a, ctx->stream);
         apply_xxt_dev(ctx, ctx->d_io_a, bw, ctx->d_io_b, ctx->stream, nullptr);
         kern::block_to_colmajor(ctx->d_io_b, ctx->N, bw, nc, ctx->d_stage, ctx->N, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->N, nc, Y + (size_t)c0 * ldy, ldy, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_xt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *T, int64_t ldt)
{
   return guarded([&] {
      if (!ctx || !B || !T || b < 1 || ldb < (int64_t)ctx->N || ldt < (int64_t)ctx->P_g) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xt");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * nc);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->N * sizeof(double), B + (size_t)c0 * ldb, (size_t)ldb * sizeof(double),
                                    ctx->N * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_block(ctx->d_stage, ctx->N, ctx->N, ctx->N_pad, bw, nc, ctx->d_io_a, ctx->stream);
         xt_dev(ctx, ctx->d_io_a, bw, ctx->stream);
         kern::t_to_colmajor(ctx->d_T, ctx->P_g, bw, nc, nullptr, ctx->d_stage, ctx->P_g, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->P_g, nc, T + (size_t)c0 * ldt, ldt, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_x(fpca_ctx *ctx, const double *T, int64_t ldt, int b, double *Y, int64_t ldy)
{
   return guarded([&] {
      if (!ctx || !T || !Y || b < 1 || ldt < (int64_t)ctx->P_g || ldy < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_x");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * nc);
         ctx->ensure(ctx->d_T, ctx->T_cap, (size_t)ctx->P_pad * MAX_BLOCKVEC);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->P_g * sizeof(double), T + (size_t)c0 * ldt, (size_t)ldt * sizeof(double),
                                    ctx->P_g * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_t(ctx->d_stage, ctx->P_g, ctx->P_g, ctx->P_pad, bw, nc, ctx->d_T, ctx->stream);
         x_dev(ctx, bw, ctx->d_io_b, ctx->stream);
         kern::block_to_colmajor(ctx->d_io_b, ctx->N, bw, nc, ctx->d_stage, ctx->N, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->N, nc, Y + (size_t)c0 * ldy, ldy, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_xxt_dev(fpca_ctx *ctx, const double *dB, int b, double *dY, void *stream)
{
   return guarded([&] {
      if (!ctx || !dB || !dY) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xxt_dev");
      if (b != 16 && b != 32 && b != 48 && b != 64) throw Error(FPCA_EINVAL, "device blocks must be 16, 32, 48 or 64 wide");
      HIP_CHECK(hipSetDevice(ctx->device));
      hipEvent_t *ev = nullptr;
      if (ctx->prof_on && ctx->prof_calls++ % ctx->prof_stride == 0 && (size_t)(ctx->prof_used + 1) * 8 <= ctx->prof_ev.size())
         ev = &ctx->prof_ev[(size_t)ctx->prof_used++ * 8];
      apply_xxt_dev(ctx, dB, b, dY, stream ? (hipStream_t)stream : ctx->stream, ev);
   });
}

// ---- multi-GPU ---------------------------------------------------------------------------------------
int fpca_comm_unique_id(uint8_t id[FPCA_UNIQUE_ID_BYTES])
{
   return guarded([&] {
      static_assert(sizeof(ncclUniqueId) == FPCA_UNIQUE_ID_BYTES, "ncclUniqueId size");
      ncclUniqueId u;
      RCCL_CHECK(rccl().GetUniqueId(&u));
      std::memcpy(id, &u, sizeof(u));
   });
}

int fpca_comm_init_rank(fpca_ctx *ctx, int nranks, int rank, const uint8_t id[FPCA_UNIQUE_ID_BYTES])
{
   return guarded([&] {
      if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) throw Error(FPCA_EINVAL, "bad argument to fpca_comm_init_rank");
      HIP_CHECK(hipSetDevice(ctx->device));
      ncclUniqueId u;
      std::memcpy(&u, id, sizeof(u));
      RCCL_CHECK(rccl().CommInitRank(&ctx->comm, nranks, u, rank));
      ctx->nranks = nranks;
      ctx->rank = rank;
      ctx->rank_known = true;
      ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0; // (the row chunks of the multi-rank K3 enter the workspace size)
      if (!ctx->comm_stream) {
         HIP_CHECK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
         for (hipEvent_t &e : ctx->ev_chunk) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
         HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
      }
      // self-test: sum of (rank+1) over ranks must be n(n+1)/2 on every rank
      double v = rank + 1.0;
      HIP_CHECK(hipMemcpyAsync(ctx->d_small, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ctx->allreduce(ctx->d_small, 1, ctx->stream);
      HIP_CHECK(hipMemcpyAsync(&v, ctx->d_small, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (v != nranks * (nranks + 1) / 2.0) throw Error(FPCA_ECOMM, "RCCL all-reduce self-test returned a wrong sum");
   });
}

int fpca_set_allreduce(fpca_ctx *ctx, fpca_allreduce_fn fn, void *user)
{
   if (!ctx) return FPCA_EINVAL;
   ctx->ar_fn = fn;
   ctx->ar_user = user;
   return FPCA_OK;
}

int fpca_set_collectives(fpca_ctx *ctx, fpca_allgather_fn ag, fpca_reducescatter_fn rs, void *user)
{
   return guarded([&] {
      if (!ctx || (ag == nullptr) != (rs == nullptr)) throw Error(FPCA_EINVAL, "bad argument to fpca_set_collectives");
      HIP_CHECK(hipSetDevice(ctx->device));
      ctx->ag_fn = ag;
      ctx->rs_fn = rs;
      ctx->coll_user = user;
      if (ag && !ctx->comm_stream) { // the stream on which a chunk's reduce-scatter runs while K3 computes the next chunk
         HIP_CHECK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
         for (hipEvent_t &e : ctx->ev_chunk) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
         HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
      }
      ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0;
   });
}

int fpca_set_rank(fpca_ctx *ctx, int nranks, int rank)
{
   if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return FPCA_EINVAL;
   ctx->nranks = nranks;
   ctx->rank = rank;
   ctx->rank_known = true;
   ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0;
   return FPCA_OK;
}

int fpca_collective_stats(const fpca_ctx *ctx, uint64_t *calls, uint64_t *bytes)
{
   if (!ctx) return FPCA_EINVAL;
   if (calls) *calls = ctx->coll_calls;
   if (bytes) *bytes = ctx->coll_bytes;
   return FPCA_OK;
}

int fpca_set_total_snps(fpca_ctx *ctx, uint64_t P_total)
{
   if (!ctx || P_total < ctx->P_g) return FPCA_EINVAL;
   ctx->P_total = P_total;
   return FPCA_OK;
}

// ---- driver -------------------------------------------------------------------------------------------
void fpca_pca_default_opts(fpca_pca_opts *o)
{
   std::memset(o, 0, sizeof(*o));
   o->struct_size = (uint32_t)sizeof(fpca_pca_opts);
   o->info_size = (uint32_t)sizeof(fpca_pca_info);
   o->ndim = 10;      // flashpca.cpp:325
   o->blockvec = 0;
   o->maxiter = 500;  // flashpca.cpp:426
   o->tol = 1e-6;     // flashpca.cpp:440
   o->divisor = FPCA_DIVISOR_P; // flashpca.cpp:484
   o->do_loadings = 0;
   o->max_blocks = 0;
   o->verbose = 0;
   o->seed = 1;       // flashpca.cpp:276
}

int fpca_pca(fpca_ctx *ctx, const fpca_pca_opts *opts, double *U, double *d, double *Px, double *pve, double *V,
             double *mean_sd, fpca_pca_info *info)
{
   int solver_rc = FPCA_OK;
   int rc = guarded([&] {
      if (!opts) throw Error(FPCA_EINVAL, "bad argument to fpca_pca");
      if (opts->struct_size != sizeof(fpca_pca_opts) || opts->info_size != sizeof(fpca_pca_info))
         throw Error(FPCA_EINVAL, "fpca_pca_opts was not filled by this library's fpca_pca_default_opts (struct sizes " + std::to_string(opts->struct_size) +
                                      " / " + std::to_string(opts->info_size) + ", expected " + std::to_string(sizeof(fpca_pca_opts)) + " / " +
                                      std::to_string(sizeof(fpca_pca_info)) + "): the caller was built against another include/fpca.h");
      if (!ctx) throw Error(FPCA_EINVAL, "bad argument to fpca_pca");
      HIP_CHECK(hipSetDevice(ctx->device));
      const int k = opts->ndim;
      // Spectra's requirement nev < ncv = 2 nev + 1 <= n, enforced by the reference CLI (flashpca.cpp:623-633)
      const uint64_t lim = std::min(ctx->N, ctx->P_total);
      const uint64_t max_dim = lim >= 1 ? (lim - 1) / 2 : 0;
      if (k < 1 || (uint64_t)k > max_dim)
         throw Error(FPCA_EINVAL, "You asked for " + std::to_string(k) + " dimensions, but only " + std::to_string(max_dim) + " allowed");
      const int b = choose_blockvec(k, opts->blockvec); // automatic: 16 (32 / 64 for ndim > 64 / > 128), never 48
      const bool timing = std::getenv("FPCA_TIMING") != nullptr;
      auto tp0 = std::chrono::steady_clock::now();
      auto lap = [&](const char *what) {
         const auto now = std::chrono::steady_clock::now();
         if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", what, std::chrono::duration<double>(now - tp0).count() * 1e3);
         tp0 = now;
      };
      if (opts->cheap_slices != 0 && (opts->cheap_slices < 3 || opts->cheap_slices > 7)) throw Error(FPCA_EINVAL, "cheap_slices must be 0 or 3..7");
      const int cheap_S = opts->mixed < 0 ? 0 : (opts->cheap_slices ? opts->cheap_slices : 4);
      HipBackend be(ctx, b, opts->replicated_solver != 0, cheap_S);
      lap("backend setup");
      PcaOutputs out;
      out.U = U;
      out.d = d;
      out.Px = Px;
      out.pve = pve;
      std::vector<int> ritz;
      double div = 1;
      std::vector<double> dloc(k);
      if (!out.d) out.d = dloc.data();
      solver_rc = run_pca(be, *opts, ctx->P_total, out, info, &ritz, &div);
      if (info) {
         info->cheap_slices = info->cheap_applies > 0 ? be.cheap_slices() : 0;
         info->seconds_exact = be.seconds_exact();
      }
      lap("run_pca");
      const auto tpost = std::chrono::steady_clock::now();
      if (opts->do_loadings && V) {
         // randompca.cpp:191-204: V[:, j] = X' u_j / sqrt(d_j) / sqrt(div); one K2 pass for all k columns
         // (b eigenvectors per Ritz block; ndim > b takes several)
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * std::min(k, b));
         std::vector<double> sc(b);
         for (int j0 = 0, q = 0; j0 < k; j0 += b, q++) {
            const int nc = std::min(b, k - j0);
            xt_dev(ctx, be.full_ptr(ritz[q]), b, ctx->stream); // (row-sharded solver: gathers the rows of the block from all ranks)
            std::fill(sc.begin(), sc.end(), 0.0);
            for (int j = 0; j < nc; j++) sc[j] = (1.0 / std::sqrt(out.d[j0 + j])) / std::sqrt(div);
            HIP_CHECK(hipMemcpyAsync(ctx->d_small, sc.data(), b * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            kern::t_to_colmajor(ctx->d_T, ctx->P_g, b, nc, ctx->d_small, ctx->d_stage, ctx->P_g, ctx->stream);
            staged_download(ctx, ctx->d_stage, ctx->P_g, nc, V + (size_t)j0 * ctx->P_g, (int64_t)ctx->P_g, nullptr, 0, nullptr); // (synchronises:
                                                                                               // sc / d_stage are reused by the next block)
         }
      }
      for (int h : ritz) be.free_block(h);
      if (mean_sd && ctx->P_g) {
         staged_download(ctx, ctx->d_mean, ctx->P_g, 1, mean_sd, (int64_t)ctx->P_g, nullptr, 0, nullptr);
         staged_download(ctx, ctx->d_sd, ctx->P_g, 1, mean_sd + ctx->P_g, (int64_t)ctx->P_g, nullptr, 0, nullptr);
      }
      if (info) {
         info->seconds_post = std::chrono::duration<double>(std::chrono::steady_clock::now() - tpost).count();
         info->seconds_total += info->seconds_post;
      }
      lap("loadings, mean/sd");
   });
   if (std::getenv("FPCA_TIMING")) std::fprintf(stderr, "[fpca] %-28s (backend teardown follows)\n", "fpca_pca body done");
   if (rc != FPCA_OK) return rc;
   if (solver_rc == FPCA_ENOTCONVERGED) set_last_error("eigen-decomposition was not successful (not converged within maxiter)");
   return solver_rc;
}

int fpca_check(fpca_ctx *ctx, const double *evec, int64_t ldu, const double *eval, int k, int divisor, double *err,
               double *mse, double *rmse)
{
   return guarded([&] {
      if (!ctx || !evec || !eval || k < 1 || ldu < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_check");
      const uint64_t N = ctx->N;
      double div = 1; // randompca.cpp:676-680
      if (divisor == FPCA_DIVISOR_N1)
         div = (double)N - 1;
      else if (divisor == FPCA_DIVISOR_P)
         div = (double)ctx->P_total;
      std::vector<double> Y((size_t)N * k);
      int rc = fpca_apply_xxt(ctx, evec, ldu, k, Y.data(), (int64_t)N);
      if (rc != FPCA_OK) throw Error(rc, fpca_last_error());
      double tot = 0;
      for (int j = 0; j < k; j++) {
         double s = 0;
         for (uint64_t i = 0; i < N; i++) {
            const double e = Y[i + (size_t)j * N] / div - evec[i + (size_t)j * ldu] * eval[j];
            s += e * e;
         }
         if (err) err[j] = s;
         tot += s;
      }
      const double m = tot / ((double)N * k); // randompca.cpp:694
      if (mse) *mse = m;
      if (rmse) *rmse = std::sqrt(m);
   });
}

// ---- measurement ---------------------------------------------------------------------------------------
int fpca_bench_apply(fpca_ctx *ctx, int b, int steps, int warmup, fpca_bench_result *res)
{
   return guarded([&] {
      if (!ctx || !res || steps < 1 || warmup < 0) throw Error(FPCA_EINVAL, "bad argument to fpca_bench_apply");
      if (b != 16 && b != 32 && b != 48 && b != 64) throw Error(FPCA_EINVAL, "b must be 16, 32, 48 or 64");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      double *dB = nullptr, *dY = nullptr;
      HIP_CHECK(hipMalloc(&dB, (size_t)ctx->N_pad * b * sizeof(double)));
      HIP_CHECK(hipMalloc(&dY, (size_t)ctx->N_pad * b * sizeof(double)));
      kern::fill_random(dB, ctx->N, ctx->N_pad, b, 12345, ctx->stream);
      std::vector<hipEvent_t> ev((size_t)steps * 8); // per step: stage boundaries [0..3], K2 / K3 GEMM kernel [4,5] / [6,7]
      for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
      for (int i = 0; i < warmup; i++) apply_xxt_dev(ctx, dB, b, dY, ctx->stream, nullptr);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      for (int i = 0; i < steps; i++) apply_xxt_dev(ctx, dB, b, dY, ctx->stream, &ev[(size_t)i * 8]);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      double t2 = 0, t3 = 0, ta = 0, g2 = 0, g3 = 0;
      float ms = 0;
      for (int i = 0; i < steps; i++) {
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 0], ev[i * 8 + 1]));
         t2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 1], ev[i * 8 + 2]));
         t3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 2], ev[i * 8 + 3]));
         ta += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 4], ev[i * 8 + 5]));
         g2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 6], ev[i * 8 + 7]));
         g3 += ms;
      }
      HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[(size_t)(steps - 1) * 8 + 3]));
      res->ms_total = ms;
      res->ms_xt = t2 / steps;
      res->ms_x = t3 / steps;
      res->ms_allreduce = ta / steps;
      res->ms_gemm_xt = g2 / steps;
      res->ms_gemm_x = g3 / steps;
      res->flops_per_step = 4.0 * (double)ctx->N * (double)ctx->P_g * b;
      res->packed_bytes_per_step = 2.0 * (double)ctx->np * (double)ctx->P_g;
      for (auto &e : ev) (void)hipEventDestroy(e);
      (void)hipFree(dB);
      (void)hipFree(dY);
   });
}

int fpca_profile_begin(fpca_ctx *ctx, int max_steps)
{
   return guarded([&] {
      if (!ctx || max_steps < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_begin");
      HIP_CHECK(hipSetDevice(ctx->device));
      while (ctx->prof_ev.size() < (size_t)max_steps * 8) {
         hipEvent_t e;
         HIP_CHECK(hipEventCreate(&e));
         ctx->prof_ev.push_back(e);
      }
      ctx->prof_used = 0;
      ctx->prof_calls = 0;
      ctx->prof_on = true;
   });
}

int fpca_profile_sample_every(fpca_ctx *ctx, int stride)
{
   return guarded([&] {
      if (!ctx || stride < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_sample_every");
      ctx->prof_stride = stride;
   });
}

int fpca_profile_end(fpca_ctx *ctx, int b, fpca_bench_result *res, int *nsteps)
{
   return guarded([&] {
      if (!ctx || !res) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_end");
      HIP_CHECK(hipSetDevice(ctx->device));
      ctx->prof_on = false;
      HIP_CHECK(hipDeviceSynchronize());
      const int n = ctx->prof_used;
      double t2 = 0, t3 = 0, ta = 0, tt = 0, g2 = 0, g3 = 0;
      float ms = 0;
      for (int i = 0; i < n; i++) {
         hipEvent_t *e = &ctx->prof_ev[(size_t)i * 8];
         HIP_CHECK(hipEventElapsedTime(&ms, e[4], e[5]));
         g2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[6], e[7]));
         g3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[1]));
         t2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[1], e[2]));
         t3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[2], e[3]));
         ta += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[3]));
         tt += ms;
      }
      std::memset(res, 0, sizeof(*res));
      if (n > 0) {
         res->ms_total = tt;
         res->ms_xt = t2 / n;
         res->ms_x = t3 / n;
         res->ms_allreduce = ta / n;
         res->ms_gemm_xt = g2 / n;
         res->ms_gemm_x = g3 / n;
      }
      res->flops_per_step = 4.0 * (double)ctx->N * (double)ctx->P_g * b;
      res->packed_bytes_per_step = 2.0 * (double)ctx->np * (double)ctx->P_g;
      if (nsteps) *nsteps = n;
   });
}

int fpca_bench_stats(fpca_ctx *ctx, int reps, double *ms_per_launch, double *bytes_per_launch)
{
   return guarded([&] {
      if (!ctx || reps < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_bench_stats");
      HIP_CHECK(hipSetDevice(ctx->device));
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      kern::bed_stats(ctx->d_packed, ctx->pitch, ctx->N, ctx->P_g, ctx->stand, ctx->d_lut, ctx->d_mean, ctx->d_sd, ctx->d_sumsq, nullptr, ctx->stream);
      HIP_CHECK(hipEventRecord(e0, ctx->stream));
      for (int i = 0; i < reps; i++)
         kern::bed_stats(ctx->d_packed, ctx->pitch, ctx->N, ctx->P_g, ctx->stand, ctx->d_lut, ctx->d_mean, ctx->d_sd, ctx->d_sumsq, nullptr, ctx->stream);
      HIP_CHECK(hipEventRecord(e1, ctx->stream));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms_per_launch) *ms_per_launch = ms / reps;
      if (bytes_per_launch) *bytes_per_launch = (double)ctx->np * (double)ctx->P_g;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
   });
}

// diagnostic used by tests/test_gpu_kernels.py: D = A(16x4) B(4x16) through the MFMA operand mapping of kernels.hip
int fpca_debug_mfma_probe(const double *A, const double *B, double *D)
{
   return guarded([&] {
      double *dA, *dB, *dD;
      HIP_CHECK(hipMalloc(&dA, 64 * sizeof(double)));
      HIP_CHECK(hipMalloc(&dB, 64 * sizeof(double)));
      HIP_CHECK(hipMalloc(&dD, 256 * sizeof(double)));
      HIP_CHECK(hipMemcpy(dA, A, 64 * sizeof(double), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dB, B, 64 * sizeof(double), hipMemcpyHostToDevice));
      kern::mfma_layout_probe(dA, dB, dD, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(D, dD, 256 * sizeof(double), hipMemcpyDeviceToHost));
      (void)hipFree(dA);
      (void)hipFree(dB);
      (void)hipFree(dD);
   });
}

int fpca_debug_mfma_i8_probe(const int8_t *A, const int8_t *Bt, int32_t *D)
{
   return guarded([&] {
      int8_t *dA, *dB;
      int *dD;
      HIP_CHECK(hipMalloc(&dA, 1024));
      HIP_CHECK(hipMalloc(&dB, 1024));
      HIP_CHECK(hipMalloc(&dD, 1024 * sizeof(int)));
      HIP_CHECK(hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dB, Bt, 1024, hipMemcpyHostToDevice));
      kern::mfma_i8_probe(dA, dB, dD, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(D, dD, 1024 * sizeof(int), hipMemcpyDeviceToHost));
      (void)hipFree(dA);
      (void)hipFree(dB);
      (void)hipFree(dD);
   });
}

int fpca_debug_k4(fpca_ctx *ctx, int b, int nq, const double *V, const double *W, double *C_gram, const double *C_in, int use_init,
                  double *Out)
{
   return guarded([&] {
      if (!ctx || !V || !W || nq < 1 || nq > 1000 || (b != 16 && b != 32 && b != 48 && b != 64)) throw Error(FPCA_EINVAL, "bad argument to fpca_debug_k4");
      if (Out && !C_in) throw Error(FPCA_EINVAL, "fpca_debug_k4: Out needs C_in");
      HIP_CHECK(hipSetDevice(ctx->device));
      HipBackend be(ctx, b);
      const int64_t N = (int64_t)ctx->N;
      std::vector<int> hv(nq);
      for (int q = 0; q < nq; q++) {
         hv[q] = be.alloc_block();
         be.upload(hv[q], b, V + (size_t)q * b * N, N);
      }
      const int hw = be.alloc_block();
      be.upload(hw, b, W, N);
      if (C_gram) be.gram(hv.data(), nq, hw, C_gram);
      if (Out) {
         const int ho = be.alloc_block();
         be.gemm(hv.data(), nq, C_in, use_init ? hw : -1, ho);
         be.download(ho, b, Out, N);
         be.free_block(ho);
      }
      be.free_block(hw);
      for (int h : hv) be.free_block(h);
   });
}

int fpca_debug_census(int nwg, uint64_t lds_bytes, uint32_t *out)
{
   return guarded([&] {
      uint32_t *d = nullptr;
      HIP_CHECK(hipMalloc(&d, (size_t)nwg * 2 * sizeof(uint32_t)));
      kern::census(d, nwg, (size_t)lds_bytes, 2000000, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(out, d, (size_t)nwg * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
      (void)hipFree(d);
   });
}

int fpca_debug_mfma_peak(int waves_per_simd, int iters, int pattern, double *tflops)
{
   return guarded([&] {
      if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 10 || pattern < 0 || (pattern > 3 && (pattern < 10 || pattern > 13)) || !tflops)
         throw Error(FPCA_EINVAL, "bad argument");
      if (pattern >= 12) // the b = 16 column remainder: 12 = four 32-wide tiles (half of the last one padding), 13 = three + 16x16x64
         *tflops = kern::mfma_i8_mix_tops(pattern - 12, iters, nullptr);
      else if (pattern >= 10) // v_mfma_i32_32x32x32_i8 (TOP/s): 10 = zero operands, 11 = random operands
         *tflops = kern::mfma_i8_peak_tops(waves_per_simd, iters, pattern == 11 ? 0x1234567u : 0u, nullptr);
      else
         *tflops = kern::mfma_peak_tflops(waves_per_simd, iters, pattern, nullptr);
   });
}

} // extern "C"
