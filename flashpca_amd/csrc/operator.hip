// operator.hip -- the block operator Y = X_g X_g' B on device-resident blocks (SVDWideOnline::perform_op / perform_op_mat,
// svdwide.cpp:21-118, b columns at a time) and its halves (crossprod / prod, svdwide.cpp:122-226): dispatch between the
// exact-integer stages (missing_routes.hip) and the fp64 / fp32 / dense MFMA kernels, split-K combines, and the two multi-GPU
// shapes -- whole blocks + all-reduce, row-sharded blocks + all-gather / reduce-scatter.
#include <algorithm>

#include "ctx.hpp"

using namespace fpca;

namespace fpca {

// the operator on device-resident blocks: dY = X_g X_g' dB (+ all-reduce).  ev (optional): 4 events recorded
// at [start, after K2(+reduce), after K3(+reduce), after all-reduce].
void apply_xxt_dev(fpca_ctx *c, const double *dB, int b, double *dY, hipStream_t s, hipEvent_t *ev, bool reduce)
{
   ensure_stats(c);
   if (c->i8_S && ensure_i8(c, b)) {
      c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
      if (ev) HIP_CHECK(hipEventRecord(ev[0], s));
      i8_zero_meta(c, s);
      if (c->i8_k2_only) { // no sample-major copy (it did not fit): K2 on the int8 matrix cores, K3 on the FP64-MFMA kernel
         xt_i8(c, dB, b, s, false, ev ? ev + 4 : nullptr);
         if (ev) HIP_CHECK(hipEventRecord(ev[1], s));
         const int s3 = kern::x_t_splits(c->N_pad, c->P_pad, b, false);
         if (s3 > 1) c->ensure(c->d_part, c->part_cap, (size_t)s3 * c->N_pad * b);
         if (ev) HIP_CHECK(hipEventRecord(ev[6], s));
         kern::x_t(c->d_packed, c->pitch, c->d_lut, c->d_T, s3 > 1 ? c->d_part : dY, c->N_pad, c->P_pad, b, s3, false, s);
         if (ev) HIP_CHECK(hipEventRecord(ev[7], s));
         if (s3 > 1) kern::reduce_sum(c->d_part, dY, (uint64_t)c->N_pad * b, s3, s);
         if (ev) HIP_CHECK(hipEventRecord(ev[2], s));
         if (reduce) allreduce_rows(c, dY, b, s);
         if (ev) HIP_CHECK(hipEventRecord(ev[3], s));
         return;
      }
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

// The operator on a ROW-SHARDED block (the eigensolver's view, backend.hpp RowShard): the rows of the input block go to every rank
// (K2 sums over all samples), K2, K3 on the whole block, reduce-scatter of the partial products -- every rank ends up with only ITS
// rows of the sum, which is all the orthogonalisation that follows needs.  With real all-gather / reduce-scatter and more than one
// chunk, K3 runs chunk by chunk and the reduce-scatter of chunk i rides on the communication stream under the computation of chunk i + 1.
//
// What travels in the all-gather of a CHEAP pass (round 6): in the exact-integer arithmetic the operand of K2 is S byte slices of the
// block, so each rank cuts ITS rows into slices -- the column scales agreed through one all-gather of b maxima per rank -- and the
// ranks all-gather the SLICES, row-major [rows][S b] int8: S N b bytes instead of 8 N b (half at 4 slices), and nobody slices rows it
// does not own; what every rank still does for all N rows is a byte transposition into the GEMM's operand layout, which also
// leaves the operand the slices spell for the sparse gathers (kernels_i8.hip k_unpack_slices).  The format depends on the REQUESTED arithmetic and the transport only, never on what
// fitted on a rank: a rank that fell back to the fp64 kernels receives the same slices and multiplies the block they spell.
// Which passes: those on at most 4 slices -- the eigensolver's cheap passes, nine in ten of a long solve.  Measured per pass at 500,000
// rows (profiles/r06_exchange_kernel_cost.txt): the transposition + fp32 operand of a 4-slice pass 58 us, + 6 us of own-row slicing at 8
// ranks, against the 44 us of re-slicing they replace, for HALF the all-gather's bytes; on 7 slices the same pass costs 124 us (it also
// writes the fp64 operand of the exact gathers) for an eighth fewer bytes -- a loss, so exact passes keep the fp64 all-gather, whose
// block IS that operand.  (FPCA_EXCHANGE_SLICES=all, test builds: every pass, as first built.)
static int exchange_slices_S(const fpca_ctx *c)
{
   if (c->i8_S_req <= 0 || !c->native_collectives() || FPCA_TEST_ENV("FPCA_EXCHANGE_FP64")) return 0;
   const int S = (c->i8_Sc > 0 && c->i8_Sc < c->i8_S_req) ? c->i8_Sc : c->i8_S_req;
   return (S <= 4 || FPCA_TEST_ENV("FPCA_EXCHANGE_SLICES")) ? S : 0;
}

constexpr size_t XM_MAX = 0, XM_COLW = 64 * kern::I8_SHARDS, XM_SEND = XM_COLW + 640, XM_ALL = XM_SEND + 64; // d_xmeta, 8-byte words

static void exchange_slices(fpca_ctx *c, const RowShard &sh, const double *in_slice, int b, int S, hipStream_t s)
{
   const size_t SB = (size_t)S * b, need_loc = (size_t)sh.slice_rows() * SB, need_full = (size_t)sh.full_rows() * SB, need_meta = XM_ALL + 64 * (size_t)sh.G;
   auto grow = [&](void **p, size_t &cap, size_t need) {
      if (need <= cap) return;
      HIP_CHECK(hipStreamSynchronize(s));
      if (*p) HIP_CHECK(hipFree(*p));
      *p = nullptr;
      cap = 0;
      HIP_ALLOC(hipMalloc(p, need));
      cap = need;
   };
   grow((void **)&c->d_qrm_loc, c->qrm_loc_cap, need_loc);
   grow((void **)&c->d_qrm_full, c->qrm_full_cap, need_full);
   grow((void **)&c->d_xmeta, c->xmeta_cap, need_meta * sizeof(double));
   unsigned long long *maxbits = reinterpret_cast<unsigned long long *>(c->d_xmeta + XM_MAX);
   HIP_CHECK(hipMemsetAsync(maxbits, 0, 64 * kern::I8_SHARDS * sizeof(double), s));
   kern::SliceOp ox{nullptr, maxbits, nullptr, c->d_xmeta + XM_COLW, nullptr, nullptr, nullptr};
   kern::i8_colmax(in_slice, sh.slice_rows(), b, 1, &ox, s); // (rows >= N of a slice are zero)
   kern::i8_maxbits_fold(maxbits, c->d_xmeta + XM_SEND, s);
   c->all_gather_small(c->d_xmeta + XM_SEND, c->d_xmeta + XM_ALL, 64, s);
   kern::i8_maxbits_set(c->d_xmeta + XM_ALL, sh.G, maxbits, s);
   kern::i8_slice_rows(in_slice, sh.slice_rows(), b, S, ox, c->d_qrm_loc, s);
   c->all_gather_bytes(sh, c->d_qrm_loc, c->d_qrm_full, SB, s);
}

void apply_sharded(fpca_ctx *c, const RowShard &sh, const double *in_slice, int b, double *out_slice, hipStream_t s)
{
   const int S = exchange_slices_S(c);
   const bool chunked = sh.nch > 1 && c->native_collectives() && c->comm_stream;
   bool i8 = false;
   PreSliced pre{nullptr, nullptr, nullptr, nullptr};
   if (S) {
      exchange_slices(c, sh, in_slice, b, S, s);
      ensure_stats(c);
      i8 = c->i8_S && ensure_i8(c, b);
      kern::SliceOp ox{nullptr, reinterpret_cast<unsigned long long *>(c->d_xmeta + XM_MAX), nullptr, c->d_xmeta + XM_COLW, nullptr, nullptr, nullptr};
      if (!i8) // this rank runs the fp64 kernels (its int8 buffers did not fit): the block the slices spell, in fp64
         kern::i8_dequant_rows(c->d_qrm_full, c->N_pad, b, S, ox, nullptr, c->d_full_in, s);
      else
         pre = PreSliced{c->d_qrm_full, ox.maxbits, ox.colw, c->d_full_in};
   } else {
      c->all_gather(sh, in_slice, c->d_full_in, b, s);
      if (chunked) {
         ensure_stats(c);
         i8 = c->i8_S && ensure_i8(c, b);
      }
   }
   if (i8 && (S || (chunked && !c->i8_k2_only))) {
      c->ensure(c->d_T, c->T_cap, (size_t)c->P_pad * b);
      i8_zero_meta(c, s);
      if (c->i8_k2_only) { // K3 on the fp64 kernel (no sample-major copy), whole block
         xt_i8(c, c->d_full_in, b, s, false, nullptr, S ? &pre : nullptr);
         x_dev(c, b, c->d_full_out, s);
         c->reduce_scatter(sh, c->d_full_out, out_slice, b, s);
         return;
      }
      xt_i8(c, c->d_full_in, b, s, true, nullptr, S ? &pre : nullptr);
      if (!chunked) {
         x_i8(c, b, c->d_full_out, s, true);
         c->reduce_scatter(sh, c->d_full_out, out_slice, b, s);
         return;
      }
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
   if (c->i8_S && ensure_i8(c, b) && !c->i8_k2_only) {
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

} // namespace fpca
