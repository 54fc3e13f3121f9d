// solver.cpp -- block Krylov-Schur driver (see solver.hpp).
//
// Relation to the reference: RandomPCA::pca_fast hands the operator y = X X' x to Spectra's single-vector
// implicitly restarted Lanczos with ncv = 2k+1 (randompca.cpp:173-178), which applies the operator one
// vector at a time (a GEMV per SNP block).  Here the Krylov space is built b vectors at a time:
//
//   V_0 = orth(random N x b)
//   repeat:  W = A V_{m-1}                                   one pass over the packed matrix (K2 + K3)
//            H = [V_0..V_{m-1}]' W ; W -= [V_0..V_{m-1}] H   classical Gram-Schmidt, repeated (full re-orth.)
//            W = Q R                                         scaled eigen-orthonormalisation (SVQB), twice
//            T[:, m-1] = H (+ symmetric mirror)              projected matrix T = V' A V, explicitly computed
//            (theta, S) = eig(T)                             host, (m b) x (m b)
//            res_i = || R S[last block, i] ||                == || A u_i - theta_i u_i ||, u_i = V S[:, i]
//            stop when res_i < tol * max(eps^(2/3), |theta_i|) for the k largest (Spectra's rule)
//            V_m = Q, or thick restart keeping the b best Ritz vectors when the basis is full
// (the loop below is the slight generalisation in which several basis blocks may be waiting for their pass through the
//  operator -- needed when Krylov passes in cheap arithmetic are verified by exact ones, see "mixed precision")
//
// All N-sized work is delegated to the backend; this file only does (m b)-sized dense algebra.
#include "solver.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "symeig.hpp"

namespace fpca {

namespace {

using clk = std::chrono::steady_clock;
inline double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// small column-major helpers ------------------------------------------------------------------------
inline void matmul(int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
   for (int j = 0; j < n; j++) {
      double *cj = C + (size_t)j * ldc;
      for (int i = 0; i < m; i++) cj[i] = 0;
      for (int p = 0; p < k; p++) {
         const double bpj = B[(size_t)p + (size_t)j * ldb];
         if (bpj == 0.0) continue;
         const double *ap = A + (size_t)p * lda;
         for (int i = 0; i < m; i++) cj[i] += ap[i] * bpj;
      }
   }
}

// Host part of the column-scaled eigen-orthonormalisation (SVQB) of a block with Gram matrix G (b x b, row-major ==
// symmetric): W_in = W_out R, W_out = W_in M.  M row-major [p][c] (the layout BlockBackend::gemm takes), R column-major.
// Robust to residual columns of very different norms, which a plain Cholesky QR is not.  Directions whose scaled Gram
// eigenvalue is below the noise floor are zeroed (returned count, dead_col); their rows of R are zero.
int svqb_factor(int b, std::vector<double> &G, double zero_scale, std::vector<double> &M, std::vector<double> &R,
                std::vector<unsigned char> &dead_col)
{
   std::vector<double> Gs((size_t)b * b), lam(b), d(b);
   M.assign((size_t)b * b, 0.0);
   for (int i = 0; i < b; i++)
      for (int j = 0; j < i; j++) {
         const double a = 0.5 * (G[(size_t)i * b + j] + G[(size_t)j * b + i]);
         G[(size_t)i * b + j] = G[(size_t)j * b + i] = a;
      }
   const double tiny = zero_scale * zero_scale;
   std::vector<unsigned char> zero_in(b, 0);
   for (int i = 0; i < b; i++) {
      const double g = G[(size_t)i * b + i];
      if (!(g > tiny) || !std::isfinite(g)) {
         zero_in[i] = 1;
         d[i] = 1.0;
      } else
         d[i] = std::sqrt(g);
   }
   for (int i = 0; i < b; i++)
      for (int j = 0; j < b; j++)
         Gs[(size_t)i + (size_t)j * b] = (zero_in[i] || zero_in[j]) ? 0.0 : G[(size_t)i * b + j] / (d[i] * d[j]);
   R.assign((size_t)b * b, 0.0);
   dead_col.assign(b, 0);
   // Well-conditioned blocks -- every second normalisation, most first ones -- take the Cholesky factor of the scaled Gram
   // matrix (b^3/3 flops against ~6 b^3 for the eigen-decomposition, which at b = 32 was a tenth of a millisecond twice per
   // step): Gs = U'U, W_in = W_out (U D), M = D^-1 U^-1.  Accepted only when every pivot stays above 1e-4 (unit
   // diagonal: condition number below ~1e8, so W_out is orthonormal to ~1e-8 x eps-level and the next pass finishes the
   // job); anything less clear-cut -- nearly dependent or vanished columns -- goes the eigen route below.
   bool any_zero = false;
   for (int i = 0; i < b; i++) any_zero = any_zero || zero_in[i];
   if (!any_zero) {
      std::vector<double> U(Gs), Ui((size_t)b * b);
      if (cholesky_upper(b, U.data(), b, 1e-4) == 0) {
         upper_inverse(b, U.data(), b, Ui.data(), b);
         for (int p = 0; p < b; p++)
            for (int c = 0; c < b; c++) {
               M[(size_t)p * b + c] = Ui[(size_t)p + (size_t)c * b] / d[p];   // (D^-1 U^-1)[p][c]
               R[(size_t)p + (size_t)c * b] = U[(size_t)p + (size_t)c * b] * d[c]; // (U D)(p, c), column-major
            }
         return 0;
      }
   }
   if (symeig_desc(b, Gs.data(), b, lam.data()) != 0) throw Error(-3, "svqb: small eigensolver failed");
   const double floor_rel = 64.0 * b * DBL_EPSILON;
   const double lmax = std::max(lam[0], 0.0);
   int ndead = 0;
   // new column j of W = sum_p W[:, p] * M[p][j],  M = D^-1 Z Lambda^-1/2 ; R = Lambda^1/2 Z' D
   for (int j = 0; j < b; j++) {
      if (!(lam[j] > floor_rel * lmax) || lmax == 0.0) {
         ndead++;
         dead_col[j] = 1;
         continue;
      }
      const double sq = std::sqrt(lam[j]);
      for (int p = 0; p < b; p++) {
         const double z = Gs[(size_t)p + (size_t)j * b];
         M[(size_t)p * b + j] = zero_in[p] ? 0.0 : z / (d[p] * sq); // row-major [p][c] for gemm
         R[(size_t)j + (size_t)p * b] = zero_in[p] ? 0.0 : sq * z * d[p]; // column-major R(j, p)
      }
   }
   return ndead;
}

// Orthonormalise the columns of backend block w in place: W_in = W_out * R (one Gram pass + one update pass).
int svqb_pass(BlockBackend &be, int w, double zero_scale, std::vector<double> &R, std::vector<unsigned char> &dead_col)
{
   const int b = be.width();
   std::vector<double> G((size_t)b * b), M;
   be.gram(&w, 1, w, G.data()); // G[p][c] row-major == symmetric
   const int ndead = svqb_factor(b, G, zero_scale, M, R, dead_col);
   be.gemm(&w, 1, M.data(), -1, w);
   return ndead;
}

// Fewer than three blocks of samples (N < 3 b): a block Krylov basis does not fit in N dimensions, but the whole operator
// is small.  A = X X' is formed column block by column block from applies on the identity and decomposed on the host --
// the reference handles such inputs through ncv = 2k+1 <= N single vectors (flashpca.cpp:623-633 admits k up to
// (min(N,P)-1)/2).
SolverResult dense_small(BlockBackend &be, const SolverOpts &o)
{
   const int b = be.width(), k = o.k;
   const uint64_t N = be.nrows();
   SolverResult res;
   auto t0 = clk::now();
   std::vector<double> A((size_t)N * N), E((size_t)N * b), w(N);
   const int in = be.alloc_block(), out = be.alloc_block();
   for (uint64_t j0 = 0; j0 < N; j0 += (uint64_t)b) {
      const int nc = (int)std::min<uint64_t>((uint64_t)b, N - j0);
      std::fill(E.begin(), E.end(), 0.0);
      for (int c = 0; c < nc; c++) E[(j0 + c) + (size_t)c * N] = 1.0;
      be.upload(in, nc, E.data(), (int64_t)N);
      be.apply(in, out);
      be.download(out, nc, &A[(size_t)j0 * N], (int64_t)N);
      res.block_applies++;
   }
   for (uint64_t i = 0; i < N; i++)
      for (uint64_t j = 0; j < i; j++) {
         const double a = 0.5 * (A[i + j * N] + A[j + i * N]);
         A[i + j * N] = A[j + i * N] = a;
      }
   const std::vector<double> A0 = A;
   if (symeig_desc((int)N, A.data(), (int)N, w.data()) != 0) throw Error(-3, "solver: dense eigensolver failed");
   res.evals.assign(w.begin(), w.begin() + k);
   res.residuals.assign(k, 0.0);
   for (int i = 0; i < k; i++) { // || A u - theta u || of the computed pairs
      double r2 = 0;
      for (uint64_t r = 0; r < N; r++) {
         double acc = -w[i] * A[r + (size_t)i * N];
         for (uint64_t c = 0; c < N; c++) acc += A0[r + c * N] * A[c + (size_t)i * N];
         r2 += acc * acc;
      }
      res.residuals[i] = std::sqrt(r2);
      res.max_rel_residual = std::max(res.max_rel_residual, res.residuals[i] / std::max(std::pow(DBL_EPSILON, 2.0 / 3.0), std::fabs(w[i])));
   }
   be.free_block(out);
   be.free_block(in);
   for (int j0 = 0; j0 < k; j0 += b) { // Ritz block j holds eigenvectors j b .. (j+1) b - 1
      const int h = be.alloc_block();
      be.upload(h, (int)std::min<uint64_t>((uint64_t)b, N - (uint64_t)j0), &A[(size_t)j0 * N], (int64_t)N);
      res.ritz_blocks.push_back(h);
   }
   res.converged = true;
   res.seconds_host = since(t0);
   if (o.verbose) std::fprintf(stderr, "[fpca] %llu samples leave no room for a Krylov basis of %d-column blocks holding %d Ritz vectors: dense eigendecomposition of X X' (%d applies)\n",
                               (unsigned long long)N, b, k, res.block_applies);
   return res;
}

} // namespace

SolverResult block_krylov_schur(BlockBackend &be, const SolverOpts &o)
{
   const int b = be.width();
   const int k = o.k;
   const uint64_t N = be.nrows();
   if (k < 1 || (uint64_t)k > N) throw Error(-1, "solver: need 1 <= k <= N");
   // k > b (more wanted pairs than one block holds; the reference admits ndim up to (min(N,P)-1)/2, flashpca.cpp:623-633):
   // the wanted Ritz vectors occupy kb blocks, a thick restart keeps kb + 1 blocks of them, and the basis cap grows with k
   const int kb = (k + b - 1) / b;
   const int nk_wide = kb > 1 ? kb + 1 : 0; // Ritz blocks a restart keeps when k > b (0: the k <= b rule below)
   // (basis cap in columns: 512, or 384 for 16-column blocks -- where the sum of block applies and dense Rayleigh-Ritz cost
   // was smallest on the slowly converging test spectrum, scripts/hard_spectrum_cap.py)
   int mcap = o.max_blocks > 0 ? o.max_blocks : std::max(4, (b == 16 ? 384 : 512) / b);
   if (kb > 1) mcap = std::max(mcap, 2 * nk_wide + 2);
   if (kb > 1 && o.max_blocks <= 0) mcap = std::max(mcap, 3 * nk_wide); // room to grow between two restarts
   // the basis must fit in N dimensions
   const int fit = (int)std::min<uint64_t>(N / (uint64_t)b, 1u << 20) - 1;
   if (fit < 2 || (kb > 1 && fit < nk_wide + 2)) return dense_small(be, o);
   mcap = std::min(mcap, fit);
   if (mcap < 2) mcap = 2;
   // The basis V_0 .. V_{M-1} is a QUEUE: its first `na` blocks have been through the operator (column blocks 0 .. na-1 of
   // T = V'AV are known, and A V_c lies in span(V) for c < na), the others wait for their turn.  Plain block Lanczos has
   // exactly one block waiting (the newest); after the switch from cheap to exact passes (below) all the kept Ritz blocks
   // wait, and each of their passes appends one more block.  nqmax bounds the blocks waiting + the one being made.
   const int nqmax = std::max(2, nk_wide) + 2;
   const int nmax = (mcap + nqmax) * b;
   const double eps23 = std::pow(DBL_EPSILON, 2.0 / 3.0);

   SolverResult res;
   double host_s = 0;
   // FPCA_TIMING=1: wall-clock per phase of the iteration (host view: each phase ends at its own synchronisation point)
   const bool timing = std::getenv("FPCA_TIMING") != nullptr;
   enum { PH_APPLY, PH_PROJ, PH_SVQB, PH_RR, PH_RESTART, PH_N };
   double ph[PH_N] = {0, 0, 0, 0, 0};
   auto tph = clk::now();
   auto phase = [&](int which) {
      if (!timing) return;
      const auto now = clk::now();
      ph[which] += std::chrono::duration<double>(now - tph).count();
      tph = now;
   };
   std::vector<int> V;
   int na = 0; // blocks of V that have been through the operator
   std::vector<double> T((size_t)nmax * nmax, 0.0), Tw, theta(nmax);
   std::vector<double> H((size_t)nmax * b), C((size_t)(nmax + b) * b), negC((size_t)(nmax + b) * b);
   std::vector<double> R1, R2, R3, R((size_t)b * b), tmp((size_t)b * b), M1, M2, Gw;
   std::vector<int> VW;
   std::vector<unsigned char> dead, dead2;
   auto Tat = [&](int i, int j) -> double & { return T[(size_t)i + (size_t)j * nmax]; };

   // project W against all basis blocks once; accumulate coefficients into H (row-major [q][p][c])
   auto project_out = [&](int w, bool accumulate) {
      const int m = (int)V.size();
      be.gram(V.data(), m, w, C.data());
      const size_t cnt = (size_t)m * b * b;
      for (size_t i = 0; i < cnt; i++) negC[i] = -C[i];
      be.gemm(V.data(), m, negC.data(), w, w);
      if (accumulate)
         for (size_t i = 0; i < cnt; i++) H[i] += C[i];
   };

   // ---- mixed precision (exact-integer backends) -----------------------------------------------------------------------
   // A pass on fewer byte slices of the fp64 operand costs ~0.75 of an exact one and perturbs the operator by ~1e-9 of its
   // norm, which a Krylov iteration to tol = 1e-6 does not feel (measured: same number of passes down to 3 slices).  The
   // iteration starts EXACT -- a solve that converges within a handful of passes never leaves that mode and is what it always
   // was -- and switches to cheap passes once the measured decay of the residuals predicts enough passes to pay for the
   // verification: when the reference's rule holds on the cheap passes' estimates, the basis is compressed to its leading
   // Ritz blocks, those go through the EXACT operator one by one (rebuilding T = Y'AY from nothing), and the rule is judged
   // on the exact residuals || (I - YY') A Y s ||.  If it fails there, the iteration goes on from that state -- with cheap
   // passes towards a tighter threshold first, exact passes only after the third failure.
   // (the blocks waiting after a verification need room: N dimensions hold fit + 1 blocks, and a restart in that phase keeps
   //  up to nkv Ritz blocks + nkv waiting ones; tiny problems simply stay exact)
   const int Mmax = fit + 1, nkv_max = std::max(2, nk_wide);
   const bool can_cheap = o.mixed && Mmax >= 2 * nkv_max + 3 && be.set_cheap(true);
   if (can_cheap) be.set_cheap(false);
   bool cheap = false;        // mode of the passes being made
   bool tainted = false;      // T holds columns from cheap passes: convergence cannot be declared from it
   int verifications = 0;     // switches from cheap back to exact passes so far
   double tol_cheap = 0.8 * o.tol; // threshold for the estimates of the cheap passes (the exact residual sits within the noise of it)
   double best_cheap = 0;          // smallest worst-residual the current run of cheap passes has reached, and when
   int best_cheap_step = 0;
   // The verification is paid for out of the same budget: kb exact passes until the rule can be judged on the Ritz blocks.  They
   // are RESERVED while cheap passes run -- when only they are left, the cheap passes end whatever their estimates say, so that
   // a budget that runs out returns Rayleigh-Ritz pairs of the exact operator (and "converged" if the rule holds on them), never
   // a half-rebuilt T.  ritz_prefix: V[0 .. ritz_prefix) ARE the Ritz blocks of the last compression (until the next test).
   // (One exception, by construction a NOT-CONVERGED result: a budget that ends between the compression and the first test on the
   //  rebuilt T returns those Ritz blocks with res.evals of the last test of the cheap passes -- ESTIMATES within the cheap
   //  arithmetic's noise of the Rayleigh quotients, ~1e-9 relative at 4 slices, not exact ones; `n < k` branch below.)
   const int reserve = kb;
   int ritz_prefix = 0;

   // ---- start block ----------------------------------------------------------------------------
   int v0 = be.alloc_block();
   be.fill_random(v0, o.seed);
   {
      int nd = svqb_pass(be, v0, 0.0, R1, dead);
      nd += svqb_pass(be, v0, 0.0, R2, dead);
      if (nd) throw Error(-3, "solver: random start block is rank deficient");
   }
   V.push_back(v0);
   int W = be.alloc_block();
   double scale = 0; // running estimate of ||A|| (largest Ritz value)
   TridiagKeep keep;            // Householder reduction of the last Rayleigh-Ritz matrix
   std::vector<double> S, Srow; // eigenvectors of T (n x n, ld n; only when needed) / their last rows (nrows x n)
   std::vector<double> Cpl;     // coupling of the waiting blocks to the leading Ritz vectors (restart)
   int n = 0;                   // order of the last Rayleigh-Ritz problem (= na b at that time)
   uint64_t reseed = o.seed * 7919 + 13;
   int skip_rr = 0; // Rayleigh-Ritz tests to skip (set after a test that ended far from convergence)
   double prev_worst = 0; // worst relative residual of the previous test and the apply count it was made at
   int prev_step = 0;
   double rho_last = 0;   // decay of the worst residual per pass, measured between the last two tests (0: not known)

   // The pass of the NEXT step is launched before this step's projected eigenproblem is solved whenever that solve is unlikely
   // to end the iteration (no test at this step, or the last test was far from the threshold): the block it applies -- the
   // oldest one waiting -- and the block it fills are the same whether this step ends in a push or in a thick restart; only
   // convergence, the end of the budget or a verification discard it.  pend_*: that pass (in flight on the backend).
   int pend_in = -1, pend_out = -1;
   bool pend_cheap = false;
   auto drop_pending = [&] { // (a pass that ran for nothing is still a pass over the matrix: it is counted)
      if (pend_out < 0) return;
      be.apply_end();
      res.block_applies++;
      if (pend_cheap) res.cheap_applies++;
      res.discarded_applies++;
      be.free_block(pend_out);
      pend_in = pend_out = -1;
   };

   while (res.block_applies < o.max_applies) {
      const int M = (int)V.size(); // blocks in the basis; V[na] goes through the operator now, W becomes block M
      if (na >= M) throw Error(-3, "solver: internal error (no block waiting)");
      phase(PH_RESTART);
      bool this_cheap = cheap;
      if (pend_out >= 0) {
         if (pend_in != V[na] || pend_out != W) throw Error(-3, "solver: internal error (speculative pass on the wrong block)");
         be.apply_end();
         this_cheap = pend_cheap;
         pend_in = pend_out = -1;
      } else
         be.apply(V[na], W);
      phase(PH_APPLY);
      res.block_applies++;
      if (this_cheap) {
         res.cheap_applies++;
         tainted = true;
      }
      // Three passes over (basis, W), each ONE Gram launch -- C = V'W and G = W'W together, the block itself riding along
      // as the last "basis" block -- and ONE update launch:
      //   1  W <- W - V C1
      //   2  W <- (W - V C2) M1      M1 from the Gram matrix of W - V C2, which is G - C2'C2 (V is orthonormal; C2 is the
      //   3  W <- (W - V C3) M2      small second-pass correction, so nothing cancels) -- column-scaled eigen-
      //                              orthonormalisation, W_before = W_after R
      // i.e. classical Gram-Schmidt twice, normalisation, a third projection (the normalisation may have amplified
      // components along V) and a second normalisation: what used to be five Gram + five update launches, each Gram a
      // round trip to the host.  H = V'A V_na = C1 + C2 + C3 R1.
      VW.assign(V.begin(), V.end());
      VW.push_back(W);
      const size_t cnt = (size_t)M * b * b;
      auto gram_vw = [&]() { // C[0 .. M b b) = V'W, Gw = W'W
         be.gram(VW.data(), M + 1, W, C.data());
         Gw.assign(C.begin() + (long)cnt, C.begin() + (long)(cnt + (size_t)b * b));
      };
      auto minus_ctc = [&]() { // Gw -= C'C  (C: [q][p][c])
         for (size_t qp = 0; qp < (size_t)M * b; qp++) {
            const double *row = &C[qp * b];
            for (int c1 = 0; c1 < b; c1++) {
               const double x = row[c1];
               if (x == 0.0) continue;
               for (int c2 = 0; c2 < b; c2++) Gw[(size_t)c1 * b + c2] -= x * row[c2];
            }
         }
      };
      // W <- (W - V C) Mx : coefficients [-C_q Mx ; Mx]; gram != null: the launch also leaves W'W of the block it writes there
      // (BlockBackend::gemm_gram: no second pass over W, no second round trip)
      auto update_with = [&](const std::vector<double> &Mx, double *gram = nullptr) {
         for (size_t qp = 0; qp < (size_t)M * b; qp++) {
            const double *row = &C[qp * b];
            double *out = &negC[qp * b];
            for (int c = 0; c < b; c++) out[c] = 0.0;
            for (int j = 0; j < b; j++) {
               const double x = -row[j];
               if (x == 0.0) continue;
               const double *mj = &Mx[(size_t)j * b];
               for (int c = 0; c < b; c++) out[c] += x * mj[c];
            }
         }
         std::copy(Mx.begin(), Mx.end(), negC.begin() + (long)cnt);
         if (gram)
            be.gemm_gram(VW.data(), M + 1, negC.data(), -1, W, gram);
         else
            be.gemm(VW.data(), M + 1, negC.data(), -1, W);
      };
      std::fill(H.begin(), H.begin() + (long)cnt, 0.0);
      gram_vw();
      for (size_t i = 0; i < cnt; i++) {
         negC[i] = -C[i];
         H[i] += C[i];
      }
      // pass 1's update and pass 2's Gram matrices from one pass over the basis where the backend has it (BlockBackend::gemm_gramvw)
      be.gemm_gramvw(V.data(), M, negC.data(), W, W, C.data());
      Gw.assign(C.begin() + (long)cnt, C.begin() + (long)(cnt + (size_t)b * b));
      phase(PH_PROJ);

      auto t0 = clk::now();
      if (scale == 0) {
         // first step: ||A|| estimate from the diagonal block H_00 (Rayleigh quotients)
         for (int i = 0; i < b; i++) scale = std::max(scale, std::fabs(H[((size_t)na * b + i) * b + i]));
      }
      for (size_t i = 0; i < cnt; i++) H[i] += C[i];
      minus_ctc();
      const double zero_scale = scale * 16 * DBL_EPSILON * std::sqrt((double)N);
      int ndead = svqb_factor(b, Gw, zero_scale, M1, R1, dead);
      host_s += since(t0);
      // The third projection exists because the normalisation of pass 2 amplifies whatever component along V the first two
      // left behind (~eps of the column's norm BEFORE normalisation) by  max column norm / smallest pivot of the triangular
      // factor.  When that factor is small -- a well-conditioned block of comparable columns: every step of a slowly
      // converging solve except the last few -- the amplified component stays below 1e-12 and pass 3 only re-normalises W
      // (one Gram + one update of W alone instead of the whole basis: a third of the orthogonalisation's traffic).  Decided
      // from the factor only, i.e. from data every rank holds identically.
      bool light3 = false;
      if (ndead == 0) {
         bool upper = true; // (the Cholesky route of svqb_factor; the eigen route's factor is not triangular)
         double cmax = 0, pmin = 1e300;
         for (int c = 0; c < b && upper; c++) {
            double cn = 0;
            for (int r = 0; r < b; r++) {
               const double x = R1[(size_t)r + (size_t)c * b];
               if (r > c && x != 0.0) upper = false;
               cn += x * x;
            }
            cmax = std::max(cmax, std::sqrt(cn));
            pmin = std::min(pmin, std::fabs(R1[(size_t)c + (size_t)c * b]));
         }
         light3 = upper && pmin > 0 && cmax / pmin < 1e4;
      }
      // pass 2's update -- and, when pass 3 will only re-normalise, the Gram matrix of the block it writes from the same launch
      if (light3) {
         Gw.assign((size_t)b * b, 0.0);
         update_with(M1, Gw.data());
      } else
         update_with(M1);
      phase(PH_SVQB);
      int nd2;
      if (light3) {
         std::vector<double> &G3 = Gw; // (left there by the fused update + Gram launch of pass 2, just above)
         t0 = clk::now();
         nd2 = svqb_factor(b, G3, 0.0, M2, R2, dead2);
         host_s += since(t0);
         be.gemm(&W, 1, M2.data(), -1, W);
      } else {
         gram_vw();
         t0 = clk::now();
         // H += C3 * R1   (C: [q][p][c] row-major b x b per q; R1 column-major)
         for (int q = 0; q < M; q++)
            for (int p = 0; p < b; p++) {
               const double *cr = &C[((size_t)q * b + p) * b];
               for (int c = 0; c < b; c++) {
                  double sacc = 0;
                  for (int j = 0; j < b; j++) sacc += cr[j] * R1[(size_t)j + (size_t)c * b];
                  H[((size_t)q * b + p) * b + c] += sacc;
               }
            }
         minus_ctc();
         nd2 = svqb_factor(b, Gw, 0.0, M2, R2, dead2);
         host_s += since(t0);
         update_with(M2);
      }
      phase(PH_SVQB);
      // R = R2 * R1
      matmul(b, b, b, R2.data(), b, R1.data(), b, R.data(), b);
      // columns that died in either pass carry no information from A V: refill them with fresh random
      // directions orthogonal to everything, so that the basis keeps full width (invariant-subspace case)
      if (ndead + nd2 > 0) {
         std::vector<double> E((size_t)b * b, 0.0);
         int ndeadcols = 0;
         // after pass 2 the dead directions are the zero columns of W: detect through its Gram diagonal
         std::vector<double> G((size_t)b * b);
         be.gram(&W, 1, W, G.data());
         for (int c = 0; c < b; c++)
            if (!(G[(size_t)c * b + c] > 0.5)) {
               E[(size_t)c * b + c] = 1.0;
               ndeadcols++;
            }
         if (ndeadcols > 0) {
            if (o.verbose) std::fprintf(stderr, "[fpca] step %d: %d deflated direction(s) refilled\n", res.block_applies, ndeadcols);
            int rnd = be.alloc_block();
            for (int attempt = 0; attempt < 3; attempt++) {
               be.fill_random(rnd, reseed++);
               // zero the dead columns, add random ones there: W = W (I - E) + rnd E
               std::vector<double> IE((size_t)b * b, 0.0);
               for (int c = 0; c < b; c++) IE[(size_t)c * b + c] = 1.0 - E[(size_t)c * b + c];
               be.gemm(&W, 1, IE.data(), -1, W);
               be.gemm(&rnd, 1, E.data(), W, W);
               project_out(W, false);
               project_out(W, false);
               std::vector<double> Ra, Rb;
               std::vector<unsigned char> da;
               int bad = svqb_pass(be, W, 0.0, Ra, da);
               bad += svqb_pass(be, W, 0.0, Rb, da);
               // the good columns were already orthonormal: the rotation mixes them, so fold it into R
               // (rows of R belonging to refilled columns are zero, and stay zero in exact arithmetic)
               matmul(b, b, b, Rb.data(), b, Ra.data(), b, tmp.data(), b);
               // mask the contribution of the random columns: R_new = tmp * (I-E) * R
               std::vector<double> t2((size_t)b * b);
               for (int j = 0; j < b; j++)
                  for (int i = 0; i < b; i++) t2[(size_t)i + (size_t)j * b] = tmp[(size_t)i + (size_t)j * b] * (1.0 - E[(size_t)j * b + j]);
               std::vector<double> Rn((size_t)b * b);
               matmul(b, b, b, t2.data(), b, R.data(), b, Rn.data(), b);
               R = Rn;
               if (bad == 0) break;
               if (attempt == 2) throw Error(-3, "solver: could not complete the basis after deflation");
            }
            be.free_block(rnd);
         }
      }

      // ---- the next pass, ahead of this step's Rayleigh-Ritz (see pend_* above) ---------------------------------------------
      {
         const int na1 = na + 1, Mn1 = M + 1;
         const bool no_test = ((skip_rr > 0) || (na1 * b < k)) && res.block_applies < o.max_applies && na1 + 1 <= mcap && Mn1 + 1 <= Mmax;
         const double tol_est = cheap ? tol_cheap : o.tol;
         // (a test IS due: the worst residual expected at it, from the last test and the decay measured between the last two,
         //  must still be well above the threshold -- residuals of a fast solve fall by orders of magnitude per pass, and a
         //  pass launched before the test that ends it is a pass wasted)
         const double expected = (prev_worst > 0 && rho_last > 0 && rho_last < 1) ? prev_worst * std::pow(rho_last, (double)(res.block_applies - prev_step)) : 0.0;
         // "well above": by the decay of four more passes, at least a factor 2 -- a slow solve (decay 0.9-0.97 per pass) keeps
         // its Rayleigh-Ritz solves hidden until the last few tests, a fast one (0.1 and less) never launches ahead of a test
         const double margin = rho_last > 0 ? std::max(2.0, 1.0 / (rho_last * rho_last * rho_last * rho_last)) : 30.0;
         const bool reserved_next = cheap && res.block_applies + 1 >= o.max_applies - reserve; // (the next step ends the cheap passes)
         if (res.block_applies + 1 < o.max_applies && !reserved_next && (no_test || expected > margin * tol_est)) {
            pend_in = na1 < M ? V[na1] : W;
            pend_out = be.alloc_block();
            pend_cheap = cheap;
            be.apply_begin(pend_in, pend_out);
         }
      }

      // ---- projected matrix: column block na of T = V'AV over ALL blocks, and the coupling R of the new block M --------
      phase(PH_RESTART);
      t0 = clk::now();
      for (int q = 0; q < M; q++) {
         if (q == na) continue;
         for (int p = 0; p < b; p++)
            for (int c = 0; c < b; c++) {
               const double h = H[((size_t)q * b + p) * b + c];
               Tat(q * b + p, na * b + c) = h;
               Tat(na * b + c, q * b + p) = h;
            }
      }
      for (int p = 0; p < b; p++)
         for (int c = 0; c < b; c++) {
            const double h = 0.5 * (H[((size_t)na * b + p) * b + c] + H[((size_t)na * b + c) * b + p]);
            Tat(na * b + p, na * b + c) = h;
         }
      for (int j = 0; j < (M + 1) * b; j++) // block row / column M is new: it couples to block na (through R) and to nothing else
         for (int r = 0; r < b; r++) Tat(M * b + r, j) = Tat(j, M * b + r) = 0.0;
      for (int r = 0; r < b; r++)
         for (int c = 0; c < b; c++) {
            Tat(M * b + r, na * b + c) = R[(size_t)r + (size_t)c * b];
            Tat(na * b + c, M * b + r) = R[(size_t)r + (size_t)c * b];
         }
      na++;
      const int Mn = M + 1; // blocks now: V[0 .. M) and W
      n = na * b;
      // Far from convergence the Rayleigh-Ritz test cannot succeed at the very next step (residuals fall by one to two
      // orders of magnitude per block apply at best): it is skipped for a step (two when six orders away).  Convergence is
      // only ever declared by an actual test, so the worst case is one block apply more than strictly needed.
      if (n < k && skip_rr == 0) skip_rr = 1; // fewer basis columns than wanted pairs: nothing to test yet
      const bool last_cheap = cheap && tainted && res.block_applies >= o.max_applies - reserve; // only the reserved passes are left
      if (skip_rr > 0 && !last_cheap && res.block_applies < o.max_applies && na + 1 <= mcap && Mn + 1 <= Mmax) {
         skip_rr--;
         host_s += since(t0);
         phase(PH_RR);
         V.push_back(W);
         W = pend_out >= 0 ? pend_out : be.alloc_block();
         continue;
      }
      if (n < k) {
         // the budget ended between a compression and the first test on the rebuilt T: the Ritz blocks of that compression are
         // what there is (S = identity over them, eigenvalue estimates of the last test) -- returned as not converged
         if (ritz_prefix * b >= k && (int)V.size() >= ritz_prefix) {
            n = ritz_prefix * b;
            S.assign((size_t)n * n, 0.0);
            for (int i = 0; i < n; i++) S[(size_t)i + (size_t)i * n] = 1.0;
            host_s += since(t0);
            break;
         }
         throw Error(-1, "solver: max_applies allows fewer basis vectors than the wanted number of eigenpairs");
      }
      ritz_prefix = 0;
      // rows of T below the applied part: the waiting blocks' coupling.  Only its trailing columns are populated (a block
      // couples to what was applied after it joined the basis): row_lo = first populated column
      const int nu = (Mn - na) * b;
      int row_lo = n;
      for (int j = 0; j < n && row_lo == n; j++)
         for (int r = 0; r < nu; r++)
            if (Tat(n + r, j) != 0.0) {
               row_lo = j / b * b;
               break;
            }
      if (row_lo == n) row_lo = n - b;
      const int nrows = n - row_lo;
      Tw.resize((size_t)n * n);
      for (int j = 0; j < n; j++) std::memcpy(&Tw[(size_t)j * n], &T[(size_t)j * nmax], sizeof(double) * n);
      // eigenvalues + the trailing rows of the eigenvectors (all the residual test needs); the full eigenvector matrix is
      // formed only when it is used: convergence, thick restart, last step
      Srow.resize((size_t)nrows * n);
      if (symeig_desc_rows(n, Tw.data(), n, theta.data(), row_lo, nrows, Srow.data(), &keep) != 0)
         throw Error(-3, "solver: projected eigensolver failed");
      scale = std::max(scale, std::fabs(theta[0]));
      // residual estimates: || T[waiting rows, applied columns] s_i ||  (== || A u_i - theta_i u_i ||, u_i = V_applied s_i)
      res.residuals.assign(k, 0.0);
      res.evals.assign(theta.begin(), theta.begin() + k);
      const double tol_now = cheap ? tol_cheap : o.tol;
      bool all_conv = true;
      double worst = 0;
      for (int i = 0; i < k; i++) {
         const double *s = &Srow[(size_t)i * nrows];
         double r2 = 0;
         for (int r = 0; r < nu; r++) {
            double acc = 0;
            for (int c = 0; c < nrows; c++) acc += Tat(n + r, row_lo + c) * s[c];
            r2 += acc * acc;
         }
         const double rn = std::sqrt(r2);
         res.residuals[i] = rn;
         const double thr = tol_now * std::max(eps23, std::fabs(theta[i]));
         worst = std::max(worst, rn / std::max(eps23, std::fabs(theta[i])));
         if (!(rn < thr)) all_conv = false;
      }
      res.max_rel_residual = worst;
      const bool out_of_budget = res.block_applies >= o.max_applies;
      const bool full = na + 1 > mcap || Mn + 1 > Mmax; // the applied part has reached its cap, or the next block would not fit in N dimensions
      // Cheap passes that stop making progress have reached their noise floor (an operand rounded too coarsely for this
      // spectrum): no 5 % gain of the worst residual over two basis lengths of passes ends them for good -- the iteration
      // continues from the current Ritz vectors with exact passes, as after a verification.
      bool stalled = false;
      if (cheap) {
         if (best_cheap == 0 || worst < 0.95 * best_cheap) {
            best_cheap = worst;
            best_cheap_step = res.block_applies;
         } else if (res.block_applies - best_cheap_step > 2 * mcap + 8)
            stalled = true;
      }
      // cheap passes have met their threshold (or given up): the leading Ritz blocks go through the exact operator (below)
      const bool verify = ((all_conv && tainted) || stalled || last_cheap) && !out_of_budget && o.max_applies - res.block_applies >= reserve;
      // number of leading Ritz blocks a compression keeps
      // Round 4: more of them when the basis has room -- kb + 3 blocks, at most a quarter of the cap.  Measured at 500,000 x
      // 100,000 with the 24-block cap (passes / wall, kept blocks 3 -> 5 for k = 20, 2 -> 4 for k = 10, 5 -> 6 for k = 50):
      // slow spectrum 164 -> 150 / 1.82 -> 1.73 s, realistic profile 156 -> 141 / 1.73 -> 1.62 s, k = 10 120 -> 106 / 1.31 -> 1.20 s,
      // k = 50 228 -> 213 / 2.60 -> 2.48 s; the small problems' 12-block cap keeps what it kept (3).
      int nk = nk_wide ? std::min(nk_wide, na) : (mcap >= 6 && na >= 3) ? 2 : 1;
      nk = std::max(nk, std::min(std::min(kb + 3, mcap / 4), na - 1));
      if (all_conv || out_of_budget || full || verify) {
         // eigenvectors are needed now, but only the leading ones (Ritz vectors kept by a restart / returned):
         // selected columns by inverse iteration, verified inside; the full QL decomposition is the fallback
         const int need = std::min(n, std::max(std::max(2, nk_wide), nk) * b);
         S.assign((size_t)n * need, 0.0);
         // (from the reduction the residual test has just made: no second O(n^3) pass)
         if (n > need + b && keep.n == n && symeig_cols_from_keep(keep, theta.data(), need, S.data()) == 0) {
         } else {
            for (int j = 0; j < n; j++) std::memcpy(&Tw[(size_t)j * n], &T[(size_t)j * nmax], sizeof(double) * n);
            if (symeig_desc(n, Tw.data(), n, theta.data()) != 0) throw Error(-3, "solver: projected eigensolver failed");
            S.assign(Tw.begin(), Tw.begin() + (size_t)n * n);
         }
      }
      const double rr_s = since(t0);
      host_s += rr_s;
      phase(PH_RR);
      if (o.verbose)
         std::fprintf(stderr, "[fpca] apply %3d%s basis %4d (+%d waiting)  theta1 %.6g  theta_k %.6g  max rel resid %.3e  (projected eigenproblem %.2f ms%s)\n",
                      res.block_applies, cheap ? " (cheap)" : tainted ? " (exact*)" : "        ", n, Mn - na, theta[0], theta[k - 1], worst, rr_s * 1e3,
                      pend_out >= 0 ? ", next pass in flight" : "");
      if (all_conv && !tainted) {
         res.converged = true;
         break;
      }
      if (out_of_budget) break;
      if (verify) drop_pending();
      if (!verify) {
         skip_rr = worst > 1e6 * tol_now ? 2 : worst > 1e3 * tol_now ? 1 : 0;
         // Slowly converging spectra (k reaching into the bulk: 150+ applies) spend their host time in tests that cannot
         // succeed: the worst residual falls by a few per cent per apply.  From the decay between this test and the previous
         // one the number of applies still needed is estimated, and the next test is placed half-way there (at most 8 applies
         // ahead; a restart always tests).  Computed from the Ritz data only, so every rank skips alike.
         double n_est = -1;
         if (prev_worst > 0 && worst < prev_worst && worst > tol_now && res.block_applies > prev_step) {
            const double rho = std::pow(worst / prev_worst, 1.0 / (double)(res.block_applies - prev_step));
            rho_last = rho;
            if (rho < 1.0 && rho > 0.0) {
               n_est = std::log(tol_now / worst) / std::log(rho);
               const int rate_skip = (int)std::min(8.0, std::max(0.0, std::floor(n_est / 2.0) - 1.0));
               skip_rr = std::max(skip_rr, rate_skip);
            }
         }
         prev_worst = worst;
         prev_step = res.block_applies;
         // exact -> cheap passes: once the decay says that what is left pays for the kb exact passes of the verification
         // (a cheap pass saves about a quarter of an exact one).  From Ritz data only: every rank switches alike.
         if (can_cheap && !cheap && verifications < 3 && ((n_est > 4.5 * kb + 4) || (verifications > 0)) &&
             o.max_applies - res.block_applies > reserve + 1) { // (room for at least one cheap pass before the reserved ones)
            cheap = true;
            be.set_cheap(true);
            if (o.verbose) std::fprintf(stderr, "[fpca] apply %3d: switching to cheap passes (about %.0f passes to go)\n", res.block_applies, n_est);
         }
      }

      if (verify) {
         // ---- cheap passes done: compress to the leading Ritz blocks and put THEM through the exact operator ------------
         // Y_j = V_applied S_j, j < nkv (the wanted ones first).  Everything else -- the blocks waiting, T -- came from cheap
         // passes and is dropped; T is rebuilt from nothing by the exact passes that follow (na = 0: every Y_j waits).
         const int nkv = nk_wide ? std::min(nk_wide, na) : std::min(2, na);
         t0 = clk::now();
         std::vector<std::vector<double>> Sk(nkv, std::vector<double>((size_t)na * b * b));
         for (int j = 0; j < nkv; j++)
            for (int q = 0; q < na; q++)
               for (int p = 0; p < b; p++)
                  for (int c = 0; c < b; c++) Sk[j][((size_t)q * b + p) * b + c] = S[(size_t)(q * b + p) + (size_t)(j * b + c) * n];
         host_s += since(t0);
         std::vector<int> Y(nkv);
         for (int j = 0; j < nkv; j++) {
            Y[j] = be.alloc_block();
            be.gemm(V.data(), na, Sk[j].data(), -1, Y[j]);
         }
         for (int h : V) be.free_block(h);
         V.assign(Y.begin(), Y.end());
         std::fill(T.begin(), T.end(), 0.0);
         na = 0;
         cheap = false;
         tainted = false;
         be.set_cheap(false);
         verifications = stalled ? 3 : verifications + 1; // (stalled: no cheap passes any more)
         best_cheap = 0;
         tol_cheap *= 0.5; // (should the exact residuals fail the rule: the next round of cheap passes aims lower)
         // (until the next Rayleigh-Ritz the Ritz vectors ARE the blocks: what is returned if the budget ends right here)
         n = nkv * b;
         ritz_prefix = nkv;
         S.assign((size_t)n * n, 0.0);
         for (int i = 0; i < n; i++) S[(size_t)i + (size_t)i * n] = 1.0;
         skip_rr = 0;
         prev_worst = 0;
         rho_last = 0;
         keep.n = 0;
         if (o.verbose)
            std::fprintf(stderr, "[fpca] apply %3d: %s; %d Ritz block(s) go through the exact operator\n", res.block_applies,
                         stalled ? "the cheap passes have stopped converging (noise floor)" : all_conv ? "estimates of the cheap passes meet the rule" : "only the passes reserved for the verification are left", nkv);
         continue; // (W is free: it is overwritten by the next apply)
      }

      if (full) {
         // ---- thick restart: keep the best Ritz vectors + the blocks waiting (the new residual block) ----------------------
         // nk blocks of them: one when the cap is tight, two otherwise (the second block of Ritz vectors keeps the
         // neighbourhood of the wanted end of the spectrum in the basis, which is what slowly converging pairs need)
         t0 = clk::now();
         std::vector<std::vector<double>> Sk(nk, std::vector<double>((size_t)na * b * b));
         for (int j = 0; j < nk; j++)
            for (int q = 0; q < na; q++)
               for (int p = 0; p < b; p++)
                  for (int c = 0; c < b; c++) Sk[j][((size_t)q * b + p) * b + c] = S[(size_t)(q * b + p) + (size_t)(j * b + c) * n];
         // coupling of the waiting blocks to the kept Ritz vectors: T[waiting, applied] S[:, 0 : nk b]   (nu x nk b)
         Cpl.assign((size_t)nu * nk * b, 0.0);
         for (int c = 0; c < nk * b; c++)
            for (int j = row_lo; j < n; j++) {
               const double sjc = S[(size_t)j + (size_t)c * n];
               if (sjc == 0.0) continue;
               for (int r = 0; r < nu; r++) Cpl[(size_t)r + (size_t)c * nu] += Tat(n + r, j) * sjc;
            }
         host_s += since(t0);
         std::vector<int> Y(nk);
         for (int j = 0; j < nk; j++) {
            Y[j] = be.alloc_block();
            be.gemm(V.data(), na, Sk[j].data(), -1, Y[j]);
         }
         std::vector<int> waiting(V.begin() + na, V.end());
         waiting.push_back(W);
         for (int q = 0; q < na; q++) be.free_block(V[q]);
         V.clear();
         for (int j = 0; j < nk; j++) V.push_back(Y[j]);
         for (int h : waiting) V.push_back(h);
         W = pend_out >= 0 ? pend_out : be.alloc_block();
         std::fill(T.begin(), T.end(), 0.0);
         for (int i = 0; i < nk * b; i++) Tat(i, i) = theta[i];
         for (int r = 0; r < nu; r++)
            for (int c = 0; c < nk * b; c++) {
               Tat(nk * b + r, c) = Cpl[(size_t)r + (size_t)c * nu];
               Tat(c, nk * b + r) = Cpl[(size_t)r + (size_t)c * nu];
            }
         na = nk;
         res.restarts++;
      } else {
         V.push_back(W);
         W = pend_out >= 0 ? pend_out : be.alloc_block();
      }
   }
   if (pend_out >= 0) { // the iteration ended with a pass in flight (convergence, budget): wait for it; its block is not W yet
      if (pend_out == W) throw Error(-3, "solver: internal error (pending pass owns W)");
      drop_pending();
   }

   // ---- Ritz vectors of the last Rayleigh-Ritz: U_j = V_applied S[:, j b : (j+1) b], j < kb ------------------
   {
      const int m = n / b; // applied blocks at the last Rayleigh-Ritz (a prefix of V)
      const int have = (int)(S.size() / (size_t)std::max(n, 1)); // columns of S that were formed
      std::vector<double> Sk((size_t)m * b * b);
      for (int j = 0; j < kb; j++) {
         for (int q = 0; q < m; q++)
            for (int p = 0; p < b; p++)
               for (int c = 0; c < b; c++)
                  Sk[((size_t)q * b + p) * b + c] = (j * b + c < have) ? S[(size_t)(q * b + p) + (size_t)(j * b + c) * n] : 0.0;
         int U = be.alloc_block();
         be.gemm(V.data(), m, Sk.data(), -1, U);
         res.ritz_blocks.push_back(U);
      }
   }
   for (int h : V) be.free_block(h);
   be.free_block(W);
   if (can_cheap) be.set_cheap(false);
   res.seconds_host = host_s;
   res.verifications = verifications;
   phase(PH_RESTART);
   if (timing)
      std::fprintf(stderr, "[fpca] solver phases (ms): apply %.3f  projections %.3f  orthonormalisation %.3f  Rayleigh-Ritz %.3f  restart/Ritz vectors/other %.3f\n",
                   ph[PH_APPLY] * 1e3, ph[PH_PROJ] * 1e3, ph[PH_SVQB] * 1e3, ph[PH_RR] * 1e3, ph[PH_RESTART] * 1e3);
   return res;
}

} // namespace fpca
