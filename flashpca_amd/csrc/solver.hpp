// solver.hpp -- block Krylov-Schur (thick-restart block Lanczos) eigensolver for the top-k eigenpairs of
// A = sum_g X_g X_g', driving a BlockBackend.  Takes the role Spectra::SymEigsSolver plays in the reference
// (randompca.cpp:173-178): same convergence rule per Ritz pair, but b vectors per pass over the matrix so
// that each pass is a tall-skinny MFMA GEMM instead of a GEMV.
#pragma once
#include <cstdint>
#include <vector>

#include "backend.hpp"

namespace fpca {

struct SolverOpts {
   int k = 10;            // wanted eigenpairs (any k <= N; k > b keeps ceil(k/b) + 1 Ritz blocks across restarts)
   int max_applies = 500; // operator applications (blocks) before giving up
   double tol = 1e-6;
   int max_blocks = 0;    // basis cap in blocks (>= 3); 0 = automatic
   uint64_t seed = 1;
   int verbose = 0;
   bool mixed = true;     // use the backend's cheap passes where it has them (BlockBackend::set_cheap), verified by exact ones
};

struct SolverResult {
   bool converged = false;
   int block_applies = 0;
   int restarts = 0;
   int cheap_applies = 0;        // of block_applies: passes in the backend's cheap arithmetic
   int verifications = 0;        // times the leading Ritz blocks were put through the exact operator after cheap passes
   int discarded_applies = 0;    // of block_applies: passes launched ahead of a Rayleigh-Ritz test that then ended the iteration
   double max_rel_residual = 0;  // max_i res_i / max(eps^(2/3), |theta_i|)
   double seconds_host = 0;      // projected eigenproblem + small dense algebra
   std::vector<double> evals;    // k eigenvalues of A, descending
   std::vector<double> residuals; // k residual norm estimates ||A u - theta u||
   std::vector<int> ritz_blocks; // ceil(k/b) backend blocks: block j holds eigenvectors j b .. (j+1) b - 1 (caller frees)
};

// Throws fpca::Error on backend failure.  Result.converged == false means max_applies was reached.
SolverResult block_krylov_schur(BlockBackend &be, const SolverOpts &opts);

} // namespace fpca
