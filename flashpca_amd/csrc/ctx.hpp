// ctx.hpp -- PRIVATE to libfpca.so: the device context behind the opaque `fpca_ctx` of include/fpca.h, the error / HIP / RCCL
// check macros, and the internal functions the translation units of the library share.  Nothing here is part of the ABI.
//
//   cabi.cpp            extern "C" entry points of include/fpca.h (argument checks, error plumbing, fpca_pca / fpca_check)
//   context.hip         context life cycle: allocation, .bed / synthetic / dense upload, K1 statistics, teardown
//   missing_routes.hip  exact-integer mode: buffers, the choice of the missing-indicator route, the two sliced GEMM stages
//   operator.hip        the block operator on device-resident blocks: whole (all-reduce) and row-sharded (all-gather / reduce-scatter)
//   comm.hip            RCCL loader, the three collectives over RCCL / caller-supplied transports, row-chunk plans, self-test
//   download.hip        pinned, pipelined result download
//   hip_backend.hip     BlockBackend over HBM-resident blocks (what solver.cpp drives)
//   bench_hooks.hip     include/fpca_debug.h: measurement hooks and hardware probes
// MI355X / gfx950 only; there is no CPU fallback anywhere in the library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include <rccl/rccl.h> // types only: the library is dlopen()ed on first use (fpca_comm_*)

#include "../../include/fpca.h"
#include "backend.hpp"
#include "common.hpp"
#include "kernels.hpp"

namespace fpca {

#define HIP_CHECK(expr)                                                                                      \
   do {                                                                                                      \
      hipError_t e__ = (expr);                                                                               \
      if (e__ != hipSuccess)                                                                                 \
         throw Error(FPCA_EHIP, std::string(#expr) + " failed: " + hipGetErrorString(e__));                  \
   } while (0)

// ---- RCCL, loaded lazily (comm.hip) ------------------------------------------------------------------
struct RcclApi {
   void *handle = nullptr;
   ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
   ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
   ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
   ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                         // optional
   ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
   const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl();

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

using fpca::RowShard;

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
   // FPCA_ACCUM_AUTO, the sample-major copy does not fit but everything else does: K2 (which reads the SNP-major matrix the
   // context holds anyway) stays on the int8 matrix cores, K3 runs the FP64-MFMA kernel on the same matrix -- the largest inputs
   // pay 5.5 + 23 ms per 16-column apply at 500,000 x 100,000 instead of 23 + 23
   bool i8_k2_only = false;
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
   // The transport was ABANDONED: a step on which the ranks must agree did not complete within its time limit (a collective had
   // failed on this rank only, the peers are somewhere else), or RCCL reported an asynchronous error.  Every later collective of
   // this context fails with FPCA_ECOMM at once -- nothing is ever sent into a communicator whose ranks are out of step.
   bool comm_dead = false;
   bool rank_known = false; // nranks / rank are meaningful (fpca_comm_init_rank, or fpca_set_rank beside a caller's all-reduce)
   // row-sharded solver (backend.hpp RowShard): whole [full_rows][b] blocks either side of the operator
   double *d_full_in = nullptr, *d_full_out = nullptr;
   size_t full_in_cap = 0, full_out_cap = 0;
   // row-sharded operand exchange in byte slices (operator.hip apply_sharded): this rank's rows / all rows, row-major [rows][S b] int8;
   // xmeta: [I8_SHARDS][64] column maxima, 640 weights, 64 folded maxima to send, [nranks][64] gathered (8-byte words)
   int8_t *d_qrm_loc = nullptr, *d_qrm_full = nullptr;
   size_t qrm_loc_cap = 0, qrm_full_cap = 0;
   double *d_xmeta = nullptr;
   size_t xmeta_cap = 0;
   uint64_t coll_calls = 0, coll_bytes = 0; // data-path collectives issued by this context (calls, payload bytes)
   long exchange_tested = -1; // layout (ranks, chunks, width) whose all-gather / reduce-scatter have passed the self-test on ALL ranks
   long exchange_failed = -1; // ... or failed it somewhere: that layout is not tried again (fpca_pca takes the replicated solver)
   int exchange_failed_path = 0; // FPCA_SOLVER_REPLICATED_SELFTEST / _FAILURE: how that layout failed
   int last_solver_path = 0;  // FPCA_SOLVER_* the last fpca_pca of this context ended on
   long dbg_rs_calls = 0;     // reduce-scatter calls so far (failure injection of the test build)
   // live profiling (fpca_profile_begin/end)
   std::vector<hipEvent_t> prof_ev;
   int prof_used = 0, prof_calls = 0, prof_stride = 1; // every prof_stride-th apply carries the events
   bool prof_on = false;

   void ensure(double *&p, size_t &cap, size_t need); // grow a device workspace of doubles (context.hip)
   bool multi() const { return comm != nullptr || ar_fn != nullptr || comm_dead; }
   // the three data-path collectives over whichever transport is installed (comm.hip)
   // slice [sh.slice_rows()][b] -> full [sh.full_rows()][b] on every rank
   void all_gather(const RowShard &sh, const double *slice, double *full, int b, hipStream_t s);
   // sum over ranks of full [sh.full_rows()][b]; rank r keeps its rows in slice.  only_chunk >= 0: that chunk only.
   void reduce_scatter(const RowShard &sh, double *full, double *slice, int b, hipStream_t s, int only_chunk = -1);
   void allreduce(double *dbuf, uint64_t count, hipStream_t s);
   // native collectives only: `bytes_per_row` bytes per row of the slice (a multiple of 8) -> the same rows of every rank, chunk by
   // chunk like all_gather; and a small all-gather of `count` doubles per rank
   void all_gather_bytes(const RowShard &sh, const int8_t *slice, int8_t *full, size_t bytes_per_row, hipStream_t s);
   void all_gather_small(const double *mine, double *all, size_t count, hipStream_t s);
};

namespace fpca {

// ---- context.hip -----------------------------------------------------------------------------------------
void ctx_alloc_common(fpca_ctx *c, uint64_t N, uint64_t P_g, int stand, int device, int accum, bool dense = false);
void ctx_finish_upload(fpca_ctx *c);
void ctx_free(fpca_ctx *c);
void ensure_stats(fpca_ctx *c); // K1 once per context: mean / sd / table / sum of squares / per-SNP missing counts
void ensure_io(fpca_ctx *c);    // the two [N_pad][64] blocks of the host-pointer operator API
inline int pad16(int b) { return (int)round_up((uint64_t)b, 16); }

// ---- missing_routes.hip (exact-integer mode) ----------------------------------------------------------
constexpr int I8M_FULL = 0, I8M_SKIP = 1, I8M_NONE = 2, I8M_SPARSE = 3, I8M_HYBRID = 4; // fpca_missing_mode
// true: the int8 path is ready for blocks of width b.  false (FPCA_ACCUM_AUTO only): its extra buffers did not fit, the
// context has been switched to the fp64 kernels for good.
bool ensure_i8(fpca_ctx *c, int b);
int i8_mode(fpca_ctx *c, int b);
void i8_zero_meta(fpca_ctx *c, hipStream_t s);
// the operand of K2 already sliced elsewhere (row-sharded exchange): all rows' slices row-major, the column maxima / weights that go
// with them, and an [N_pad][b] fp64 buffer the scaled operand may be written to for the sparse gathers of an exact pass
struct PreSliced {
   const int8_t *Qrm;
   unsigned long long *maxbits;
   double *colw;
   double *dq64;
};
// T = X' B on slices of B (K2 stage); chain: the combine also leaves the column maxima of the two K3 operands
void xt_i8(fpca_ctx *c, const double *dB, int b, hipStream_t s, bool chain, hipEvent_t *gev = nullptr, const PreSliced *pre = nullptr);
// Y = X T on slices of T/sd and mean T/sd (K3 stage), rows [r0, r1) of Y (r1 = 0: all)
void x_i8(fpca_ctx *c, int b, double *dY, hipStream_t s, bool have_max, bool do_slice = true, uint64_t r0 = 0, uint64_t r1 = 0,
          hipEvent_t *gev = nullptr);

// ---- comm.hip ----------------------------------------------------------------------------------------------
int ar_chunks(const fpca_ctx *c);    // row chunks of the overlapped all-reduce of Y (built-in communicator)
int shard_chunks(const fpca_ctx *c); // row chunks of the row-sharded solver's exchange
uint64_t ar_chunk_begin(const fpca_ctx *c, int nchunks, int i);
void allreduce_rows(fpca_ctx *c, double *dY, int b, hipStream_t s);
// all-gather + reduce-scatter of the row-sharded solver on a block of known content: this rank's verdict (empty = right)
std::string exchange_selftest(fpca_ctx *c, const RowShard &sh, int b);
double agree_sum(fpca_ctx *c, double mine); // sum over ranks of one number (decisions from rank-local observations); time-bounded
// wait for a stream on which collectives may be pending, at most the agreement time limit: false = still pending (a peer is not
// there); the caller abandons the transport
bool bounded_sync(fpca_ctx *c, hipStream_t s);
void abandon_comm(fpca_ctx *c, const std::string &why); // marks the transport dead, aborts the RCCL communicator, throws FPCA_ECOMM
void reset_exchange_state(fpca_ctx *c);                 // a new transport / rank: nothing known about its exchange any more
void comm_streams(fpca_ctx *c); // the communication stream and its events, once

// ---- operator.hip -----------------------------------------------------------------------------------------
// dY = X_g X_g' dB (+ all-reduce).  ev (optional): 8 events, [start, after K2, after K3, after all-reduce, K2 GEMM begin/end, K3 GEMM begin/end]
void apply_xxt_dev(fpca_ctx *c, const double *dB, int b, double *dY, hipStream_t s, hipEvent_t *ev, bool reduce = true);
void apply_sharded(fpca_ctx *c, const RowShard &sh, const double *in_slice, int b, double *out_slice, hipStream_t s);
void xt_dev(fpca_ctx *c, const double *dB, int b, hipStream_t s); // T (in the context) = X_g' dB
void x_dev(fpca_ctx *c, int b, double *dY, hipStream_t s);        // dY = X_g T

// ---- download.hip -----------------------------------------------------------------------------------------
// d_img: device, column-major N x ncols with leading dimension N -> host (ld) and, scaled per column, host2 (ld2); synchronises
void staged_download(fpca_ctx *c, const double *d_img, uint64_t N, int ncols, double *host, int64_t ld, double *host2, int64_t ld2,
                     const double *scale);

// ---- error plumbing of the extern "C" layer ---------------------------------------------------------------
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

} // namespace fpca
