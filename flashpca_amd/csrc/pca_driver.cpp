// pca_driver.cpp -- see pca_driver.hpp.  Post-processing follows randompca.cpp:180-208 line by line.
#include "pca_driver.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "common.hpp"
#include "solver.hpp"

namespace fpca {

int choose_blockvec(int ndim, int requested)
{
   if (ndim < 1) throw Error(FPCA_EINVAL, "ndim must be >= 1");
   int b = requested; // ndim > b is fine: the solver keeps ceil(ndim / b) + 1 Ritz blocks (solver.cpp)
   if (b <= 0) {
      // Automatic width: the NARROWEST block the kernels have.  A block apply on 16 columns costs 0.6 of one on 32 (12.6 vs
      // 20.5 ms at 500,000 x 100,000) and 0.3 of one on 64, while the Krylov iteration needs only 1.1-1.4x as many of them:
      // measured time to solution for k = 10 / 20 / 50 on an easy and on a slowly converging spectrum at 50,000 x 20,000
      // and 500,000 x 100,000 (scripts/width_vs_k.py) is shortest at b = 16 in all twelve cases -- by 20 % (k = 20, easy)
      // to 64 % (k = 50, slow) against the round-1 rule "smallest multiple of 16 >= k + 4".  Wider blocks only where k is so
      // large that the ceil(k/b) + 1 blocks of Ritz vectors a restart keeps would crowd the basis.
      b = ndim <= 64 ? 16 : ndim <= 128 ? 32 : MAX_BLOCKVEC;
   }
   if (b % 16 != 0 || b < 16 || b > MAX_BLOCKVEC) throw Error(FPCA_EINVAL, "blockvec must be 16, 32, 48 or 64");
   return b;
}

int run_pca(BlockBackend &be, const fpca_pca_opts &o, uint64_t P_div, const PcaOutputs &out, fpca_pca_info *info,
            std::vector<int> *ritz_blocks, double *div_out)
{
   auto t0 = std::chrono::steady_clock::now();
   const uint64_t N = be.nrows();
   const int k = o.ndim;
   SolverOpts so;
   so.k = k;
   // --maxiter counts what the reference counts: restarts of Spectra's ncv = 2k+1 factorisation (flashpca.cpp:423-433,
   // randompca.cpp:178).  Its budget of operator applications -- 2k+1 for the first factorisation, at most k+1 per restart
   // (op count = 1 + (ncv-1) + sum over restarts of (ncv - nev_adjusted)) -- is spent here b at a time.
   {
      const long long mi = o.maxiter > 0 ? o.maxiter : 500;
      // ... 16 at a time whatever the width: a wide block needs about as many PASSES as a narrow one (120 / 102 / 87+ passes for
      // b = 16 / 32 / 64 at k = 10 on the realistic profile), so dividing the budget by b would make the explicit wide widths run
      // out where the automatic one does not (profiles/r04_realistic_choices.txt: b = 64, k = 10 stopped at its 87 passes)
      const long long ops = 2LL * k + 1 + mi * (k + 1LL), b = be.width(), unit = std::min<long long>(b, 16);
      so.max_applies = (int)std::min<long long>((ops + unit - 1) / unit, 1LL << 30);
      if (o.max_applies > 0) so.max_applies = o.max_applies; // explicit cap in block applies (tests, bench warm-up)
      if ((long long)so.max_applies * b < k)
         throw Error(FPCA_EINVAL, "max_applies = " + std::to_string(so.max_applies) + " block applies of " + std::to_string((int)b) +
                                      " columns give fewer basis vectors than the " + std::to_string(k) + " eigenpairs asked for");
   }
   so.tol = o.tol > 0 ? o.tol : 1e-6;
   so.max_blocks = o.max_blocks;
   // basis cap when the caller leaves it open: the projected eigenproblem costs O((cap b)^3) on the host at every restart
   // whatever the data size, a block apply O(N P b) on the device -- small problems with slowly converging spectra are
   // better off with half the basis (256 columns; six blocks at least, so that a restart can keep two) and a few more applies.  Decided from the problem size, not from
   // timings, so that every rank of a multi-GPU run decides alike.
   if (so.max_blocks <= 0 && (double)N * (double)P_div < 1e10) {
      so.max_blocks = std::max(6, (be.width() == 16 ? 192 : 256) / be.width());
      const int kb = (k + be.width() - 1) / be.width();
      if (kb > 1) so.max_blocks = std::max(so.max_blocks, 3 * (kb + 1)); // k > b: a restart keeps kb + 1 blocks
   }
   so.seed = o.seed ? o.seed : 1;
   so.verbose = o.verbose;
   so.mixed = o.mixed >= 0;
   SolverResult r = block_krylov_schur(be, so);
   const bool timing = std::getenv("FPCA_TIMING") != nullptr;
   auto lap = [&, last = std::chrono::steady_clock::now()](const char *what) mutable {
      const auto now = std::chrono::steady_clock::now();
      if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", what, std::chrono::duration<double>(now - last).count() * 1e3);
      last = now;
   };
   if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", "solver", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
   lap("-");

   double div = 1; // randompca.cpp:180-184
   if (o.divisor == FPCA_DIVISOR_N1)
      div = (double)N - 1;
   else if (o.divisor == FPCA_DIVISOR_P)
      div = (double)P_div;
   if (div_out) *div_out = div;
   const double trace = be.trace() / div; // :205
   std::vector<double> d(k);
   for (int j = 0; j < k; j++) d[j] = r.evals[j] / div; // :190
   if (out.d)
      for (int j = 0; j < k; j++) out.d[j] = d[j];
   if (out.pve)
      for (int j = 0; j < k; j++) out.pve[j] = d[j] / trace; // :206
   double sec_download = 0;
   { // (always: with a row-sharded backend the download starts with an all-gather, which every rank must join -- a rank
     //  that passes no U / Px just takes part in it)
      lap("trace, eigenvalues");
      const auto td = std::chrono::steady_clock::now();
      // U and Px = U diag(sqrt(d)) (:207) leave the device in one pipelined pass: pinned chunks, the host side of each chunk
      // (copy into U, scaled copy into Px) running while the next chunk is on the wire (HipBackend::download2)
      std::vector<double> sq(k);
      for (int j = 0; j < k; j++) sq[j] = std::sqrt(d[j]);
      for (int j0 = 0, q = 0; j0 < k; j0 += be.width(), q++) {
         double *u = out.U ? out.U + (size_t)j0 * N : nullptr, *px = out.Px ? out.Px + (size_t)j0 * N : nullptr;
         if (out.partial_rows) // every rank its own rows: no gather of the Ritz blocks, no funnel through one PCIe link
            be.download_rows_mine(r.ritz_blocks[q], std::min(be.width(), k - j0), u, (int64_t)N, px, (int64_t)N, sq.data() + j0);
         else
            be.download2(r.ritz_blocks[q], std::min(be.width(), k - j0), u, (int64_t)N, px, (int64_t)N, sq.data() + j0);
      }
      sec_download = std::chrono::duration<double>(std::chrono::steady_clock::now() - td).count();
      lap("download U, Px = U sqrt(d)");
   }
   if (ritz_blocks)
      *ritz_blocks = r.ritz_blocks;
   else
      for (int h : r.ritz_blocks) be.free_block(h);
   if (info) {
      info->converged = r.converged ? 1 : 0;
      info->block_applies = r.block_applies;
      info->vector_ops = r.block_applies * be.width();
      info->restarts = r.restarts;
      info->cheap_applies = r.cheap_applies;
      info->cheap_slices = 0;  // (filled in by fpca_pca, which knows the backend's arithmetic)
      info->seconds_exact = 0;
      info->blockvec = be.width();
      info->trace = trace;
      info->max_residual = r.max_rel_residual;
      info->seconds_apply = be.seconds_apply();
      info->seconds_ortho = be.seconds_other();
      info->seconds_host = r.seconds_host;
      info->seconds_download = sec_download;
      info->seconds_post = 0; // (loadings / mean-sd: filled in by fpca_pca)
      info->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   }
   return r.converged ? FPCA_OK : FPCA_ENOTCONVERGED;
}

} // namespace fpca
