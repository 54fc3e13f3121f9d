// symeig.hpp -- small dense host linear algebra for the projected (Rayleigh-Ritz) problem.
// The reference delegates this to Spectra/Eigen (TridiagEigen, UpperHessenbergQR); here it is own code,
// no third-party dependency.  All matrices column-major.
#pragma once
#include <vector>

namespace fpca {

// Eigen-decomposition of a dense symmetric n x n matrix (full storage, leading dimension lda).
// On return w[0..n) holds the eigenvalues in DESCENDING order and column j of A the unit eigenvector of w[j].
// Returns 0 on success, nonzero if the QL iteration failed to converge.
int symeig_desc(int n, double *A, int lda, double *w);

// Same eigenvalues, but only rows row0..row0+nrows-1 of the eigenvector matrix: Zr (nrows x n, ld nrows), column j
// belonging to w[j].  A is destroyed.  O(4/3 n^3) instead of O(6 n^3): the Krylov residual test needs only the last
// block of rows.
// With keep != nullptr the Householder reduction (original matrix, reflectors, tridiagonal form) is left in *keep, from
// which symeig_cols_from_keep forms leading eigenvectors later without a second reduction.
struct TridiagKeep {
   int n = 0;
   std::vector<double> A0, A, beta, d, e;
};
int symeig_desc_rows(int n, double *A, int lda, double *w, int row0, int nrows, double *Zr, TridiagKeep *keep = nullptr);
// First ncols eigenvectors (Z: n x ncols, ld n) of the matrix last reduced into keep; w = its eigenvalues (descending) as
// symeig_desc_rows returned them.  Verified like symeig_desc_cols; nonzero = "use symeig_desc".
int symeig_cols_from_keep(const TridiagKeep &keep, const double *w, int ncols, double *Z);

// Eigenvalues (all, descending) and only the first ncols eigenvectors (Z: n x ncols, ld n).  A is destroyed.  The vectors
// are verified (residual, orthonormality); a nonzero return means "use symeig_desc" -- w is then unspecified.
int symeig_desc_cols(int n, double *A, int lda, double *w, int ncols, double *Z);

// Upper Cholesky factor: G = R' R, R overwrites the upper triangle of G (strict lower part zeroed).
// Returns 0 on success, j+1 if the pivot of column j is not sufficiently positive (relative to rel_tol *
// the largest original diagonal entry).
int cholesky_upper(int n, double *G, int ld, double rel_tol);

// inverse of an upper-triangular matrix (Rinv may not alias R)
void upper_inverse(int n, const double *R, int ldr, double *Rinv, int ldi);

} // namespace fpca
