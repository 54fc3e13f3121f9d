// missing_routes.hip -- the exact-integer mode (FPCA_ACCUM_I8(S), the default) above the kernels: its buffers, the
// choice of how the missing-call indicator E = 1 - M is handled (fpca_missing_mode: dense / block-skipping / none / sparse
// gathers / hybrid), and the two sliced GEMM stages T = X'B (xt_i8) and Y = X T (x_i8).  Reference semantics: a missing call
// standardises to 0 (data.cpp:300-320, table entry [1]).
#include <algorithm>
#include <cmath>

#include "ctx.hpp"

using namespace fpca;

namespace fpca {

// ---- exact-integer mode --------------------------------------------------------------------------------
// layout of d_i8w in 8-byte words: three weight vectors (S*b <= 9*64 = 576 entries each, padded), then the region that is
// zeroed once per apply: column maxima (bit patterns) of the three operands and the column sums of the two M operands
// (I8W_D / I8W_MAXD: the K3 operand of the hybrid route's dense SNPs)
constexpr int I8W_B = 0, I8W_G = 640, I8W_M = 1280, I8W_D = 1920, I8W_ZERO = 2560, I8W_MAXB = I8W_ZERO, I8W_MAXG = I8W_MAXB + 64 * kern::I8_SHARDS,
              I8W_MAXM = I8W_MAXG + 64 * kern::I8_SHARDS, I8W_MAXD = I8W_MAXM + 64 * kern::I8_SHARDS, I8W_CSB = I8W_MAXD + 64 * kern::I8_SHARDS,
              I8W_CSM = I8W_CSB + kern::I8_CS_STRIDE * kern::I8_SHARDS, I8W_TOTAL = I8W_CSM + kern::I8_CS_STRIDE * kern::I8_SHARDS;


constexpr double CHEAP_SPARSE_BREAK_EVEN = 0.002; // ... and for passes on <= 4 slices (i8_mode)
constexpr double SPARSE_BREAK_EVEN = 0.005; // missing-call rate at which the gathers cost what the E half of the GEMMs costs (= M2_CELL S / G_CALL at S = 7)
static void hybrid_classify(fpca_ctx *c);
static void ensure_i8_alloc(fpca_ctx *c, int b);

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
                       "of %.1f GiB are free (%s).  What fits: --accum auto (keeps X'B on the int8 cores and runs X T on the fp64 kernel, which "
                       "needs no second copy: same results, ~2.6x slower per pass), --accum fp64, or the SNPs sharded over more GPUs (--gpus)",
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
      c->i8_k2_only = false;
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
   if (!c->i8_transposed && !c->i8_k2_only) {
      c->pitchT = (size_t)c->P_pad / 4;
      if (!c->d_inv_sd) HIP_ALLOC(hipMalloc(&c->d_inv_sd, c->P_pad * sizeof(double)));
      if (!c->d_mu_inv_sd) HIP_ALLOC(hipMalloc(&c->d_mu_inv_sd, c->P_pad * sizeof(double)));
      if (!c->d_i8w) HIP_ALLOC(hipMalloc(&c->d_i8w, I8W_TOTAL * sizeof(double)));
      try {
         if (FPCA_TEST_ENV("FPCA_DEBUG_I8_NOCOPY")) throw Error(FPCA_ENOMEM, "FPCA_DEBUG_I8_NOCOPY is set"); // exercises the K2-only state
         HIP_ALLOC(hipMalloc(&c->d_packedT, c->pitchT * c->N_pad));
      } catch (const Error &e) {
         if (!c->i8_auto || e.code != FPCA_ENOMEM) throw;
         (void)hipGetLastError();
         std::fprintf(stderr, "[fpca] the sample-major copy of the packed genotypes (%.1f GiB) does not fit in device memory (%s): X'B stays on the int8 "
                              "matrix cores, X T runs the fp64 kernels on the SNP-major matrix\n",
                      (double)c->pitchT * (double)c->N_pad / (1024.0 * 1024.0 * 1024.0), e.what());
         c->i8_k2_only = true;
         c->hyb_failed = true; // (the hybrid route lives on a view of that copy)
      }
   }
   if (!c->i8_transposed && !c->i8_k2_only) {
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
      bool patched = false;
      try {
         if (hyb) {
            HIP_CHECK(hipMemcpyAsync(c->d_hyb_idx, c->h_hyb_idx.data(), c->hyb_n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            kern::gather_packed_rows(c->d_packed, c->pitch, c->d_hyb_idx, c->hyb_n, c->hyb_pad, c->d_packedE, s);
            kern::patch_missing_rows(c->d_packed, c->pitch, c->d_hyb_idx, c->hyb_n, s);
            patched = true;
         }
         kern::transpose_packed(c->d_packed, c->pitch, c->N_pad, c->P_pad, c->d_packedT, c->pitchT, s);
         if (hyb) {
            kern::scatter_packed_rows(c->d_packedE, c->pitch, c->d_hyb_idx, c->hyb_n, c->d_packed, s);
            patched = false;
            kern::transpose_packed(c->d_packedE, c->pitch, c->N_pad, c->hyb_pad, c->d_packedET, c->pitchET, s);
            HIP_CHECK(hipStreamSynchronize(s)); // (a failed launch of the set-up surfaces HERE, not in some later apply)
            c->hyb_view = true;
         }
      } catch (...) {
         // the resident SNP-major matrix must never stay altered: every later K1 pass, fp64 apply and download reads it
         if (patched) {
            (void)hipGetLastError();
            try {
               kern::scatter_packed_rows(c->d_packedE, c->pitch, c->d_hyb_idx, c->hyb_n, c->d_packed, s);
               (void)hipStreamSynchronize(s);
            } catch (...) {
            }
         }
         c->hyb_failed = true;
         throw;
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

// What a shard's missing calls cost per block column, beyond the one-matrix GEMMs every route runs -- three linear models fitted to
// the stage times at 500,000 x 100,000 (profiles/r03_sparse_breakeven.txt, r04_gather_ab.txt, r05_missing_routes.txt):
//   gathers (sparse / hybrid lists)          G_CALL per listed missing call (both stages: a 128-byte row each)
//   compacted indicator GEMMs (hybrid)       D_CELL x N x S per dense SNP (two one-matrix launches over the SNP's records)
//   two-matrix kernels (dense route)         M2_CELL x N x S per SNP of the shard (the second matrix shares loads and decode)
// Per SNP the hybrid route is free to choose: a SNP whose calls cost more to gather than its indicator row costs on the matrix
// cores goes dense -- above D_CELL S / G_CALL = 0.69 % of the samples at S = 7.  The shard takes the hybrid route when the sum of
// those per-SNP minima beats both alternatives; uniform rates have no minority to single out and land where they landed before
// (gathers up to M2_CELL S / G_CALL = 0.51 %, the two-matrix kernels above).  Round 4 decided by two fixed rules instead (dense SNPs
// at most a quarter, the rest at most 0.5 %): right for the concentrated profile, blind to everything between it and uniform.
constexpr double G_CALL = 2.0e-12, D_CELL = 1.96e-15, M2_CELL = 1.45e-15;
static double cost_sparse(const fpca_ctx *c, uint64_t calls) { return G_CALL * (double)calls; }
static double cost_two_matrix(const fpca_ctx *c) { return M2_CELL * (double)c->N * c->i8_S_req * (double)c->P_g; }
static double cost_hybrid(const fpca_ctx *c, uint64_t listed, uint64_t ndense)
{
   return G_CALL * (double)listed + D_CELL * (double)c->N * c->i8_S_req * (double)ndense;
}

void hybrid_classify(fpca_ctx *c)
{
   if (c->hyb_class >= 0) return;
   c->hyb_class = 0;
   if (!c->missing_known || c->h_nmiss.size() != c->P_g || c->P_g == 0 || c->i8_S_req <= 0) return;
   const double thr = D_CELL * c->i8_S_req / G_CALL * (double)c->N; // calls of one SNP above which its indicator row is cheaper dense
   uint64_t dense_nnz = 0;
   std::vector<uint32_t> idx;
   for (uint64_t j = 0; j < c->P_g; j++)
      if ((double)c->h_nmiss[j] > thr) {
         idx.push_back((uint32_t)j);
         dense_nnz += c->h_nmiss[j];
      }
   const uint64_t rest = c->n_missing - dense_nnz;
   if (idx.empty() || rest >= (1ull << 31)) return;
   const double ch = cost_hybrid(c, rest, idx.size());
   // (against the gathers alone a clear margin is asked for: the route has a one-off set-up -- two row shuffles of the dense SNPs and
   //  a second, small transposition -- and keeps a compacted copy of their records)
   if (!(ch < cost_two_matrix(c)) || !(ch < 0.85 * cost_sparse(c, c->n_missing))) return;
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
   if (lists_ok && !c->hyb_failed) {
      hybrid_classify(c);
      if (c->hyb_class == 1) return I8M_HYBRID;
   }
   // a gathered fp64 row costs 8 b bytes per missing call; the E half of the int8 GEMMs costs the same whatever the rate.
   // Measured at 500k x 100k (scripts/sparse_breakeven.py, profiles/r03_sparse_breakeven.txt; K2 / K3 stage in ms, sparse |
   // dense): b = 16: 0.3 % 6.8 / 7.4 | 9.0 / 9.8, 0.5 % 8.5 / 9.1 | 9.0 / 9.9, 1 % 12.6 / 13.0 | 9.0 / 9.8; b = 32: 0.3 % 12.7 / 13.7 |
   // 16.2 / 18.8, 0.5 % 15.9 / 16.8 | 16.1 / 18.8, 1 % 23.9 / 24.7 | 16.2 / 18.8 -- the lines cross at 0.51-0.63 % for both widths
   if (sparse_ok && rate <= SPARSE_BREAK_EVEN) {
      // The eigensolver's cheap passes (fewer slices) cross over much earlier: the gathers cost the same whatever the slice count, the
      // indicator's half of a 4-slice two-matrix launch does not -- measured at 500,000 x 100,000, 16 columns, 4 slices
      // (profiles/r06_cheap_route_breakeven.txt; apply in ms, gathers | two-matrix): 0.10 % 8.6 | 10.2, 0.15 % 9.3 | 10.2, 0.20 % 10.2 | 10.2,
      // 0.30 % 11.8 | 10.2, 0.50 % 14.1 | 10.2.  Both routes read the same resident matrices (the lists stay for the exact passes), so
      // the choice is per pass.  (fpca_missing_mode reports the route of the exact passes.)
      if (c->i8_Sc > 0 && c->i8_Sc < c->i8_S && c->i8_Sc <= 4 && rate > CHEAP_SPARSE_BREAK_EVEN && !c->hyb_view) return I8M_FULL;
      return I8M_SPARSE;
   }
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
   if (c->i8_k2_only) { // (no sample-major copy: X T runs the fp64 kernel, only the per-SNP lists of K2 exist)
      c->sparse_ready = true;
      return;
   }
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
void xt_i8(fpca_ctx *c, const double *dB, int b, hipStream_t s, bool chain, hipEvent_t *gev, const PreSliced *pre)
{
   kern::SliceOp ob = i8_op_b(c), ot[2];
   if (pre) { // the slices were cut by the ranks that own the rows, with the column scales all ranks agreed on
      ob.maxbits = pre->maxbits;
      ob.colw = pre->colw;
   }
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
   if (pre) {
      // column-major operand + exact column sums from the gathered rows; the gathers of the missing-call route read the operand
      // the slices spell -- X' applied to exactly the rounded block, G and E terms alike
      const bool gathers = mode == I8M_SPARSE || hyb;
      kern::i8_unpack_slices(pre->Qrm, c->N_pad, b, c->cur_S(), ob, s, c->N, (gathers && g32) ? ob.copy32 : nullptr, (gathers && !g32) ? pre->dq64 : nullptr);
      if (gathers && !g32) dB = pre->dq64;
   } else {
      kern::i8_colmax(dB, c->N, b, 1, &ob, s);
      kern::i8_slice(dB, c->N_pad, c->N, b, c->cur_S(), 1, &ob, s);
   }
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

// Y = X T : slices of T/sd and mean T/sd (one pass over T) against the sample-major copy, rows [r0, r1) of Y
void x_i8(fpca_ctx *c, int b, double *dY, hipStream_t s, bool have_max, bool do_slice, uint64_t r0, uint64_t r1, hipEvent_t *gev)
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

} // namespace fpca
