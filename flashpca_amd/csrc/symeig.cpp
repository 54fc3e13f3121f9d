// symeig.cpp -- dense symmetric eigensolver (Householder tridiagonalisation + implicit-shift QL) and
// small triangular helpers, used on the host for the (m*b x m*b) Rayleigh-Ritz problem of the block
// Krylov-Schur driver (the "k x k Rayleigh-Ritz / small SVD stays on the host" part of the design).
#include "symeig.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "common.hpp"

namespace fpca {

namespace {

// A = Q T Q', T tridiagonal (d, e).  On return A holds Q (column-major).
// With rows != nullptr Q is not formed: rows (nrows x n, ld nrows) receives rows row0..row0+nrows-1 of Q instead.
// With beta_out != nullptr Q is not formed: the reflectors stay below the sub-diagonal of A and their coefficients are
// returned, for the caller to apply (symeig_desc_cols); rows are still produced when asked for.
typedef double v4d __attribute__((vector_size(32)));
// dot product in a fixed evaluation order (4 lanes x 4 accumulators): the same bits whoever computes it, and vector code where a
// plain `s += a[i] * b[i]` loop is a 4-cycle dependency chain per element
inline double dot_fixed(const double *a, const double *b, int m)
{
   v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0;
   int i = 0;
   for (; i + 16 <= m; i += 16) {
      v4d a0, a1, a2, a3, b0, b1, b2, b3; // (memcpy = unaligned vector loads)
      std::memcpy(&a0, a + i, 32);
      std::memcpy(&a1, a + i + 4, 32);
      std::memcpy(&a2, a + i + 8, 32);
      std::memcpy(&a3, a + i + 12, 32);
      std::memcpy(&b0, b + i, 32);
      std::memcpy(&b1, b + i + 4, 32);
      std::memcpy(&b2, b + i + 8, 32);
      std::memcpy(&b3, b + i + 12, 32);
      s0 += a0 * b0;
      s1 += a1 * b1;
      s2 += a2 * b2;
      s3 += a3 * b3;
   }
   const v4d t = (s0 + s1) + (s2 + s3);
   double s = (t[0] + t[1]) + (t[2] + t[3]);
   for (; i < m; i++) s += a[i] * b[i];
   return s;
}
// One column of the symmetric matrix-vector product and of the rank-2 update on LOWER-triangle storage, fixed evaluation order
// (4 lanes x 2 accumulators) so that every rank of a multi-GPU run gets the same bits:
//   sym_col_mv:  returns col[1:].v[1:]  and adds col[1:] * vj to p[1:]      (column j of A22 from its diagonal down, length len)
//   sym_col_r2:  col[i] -= v[i] qj + q[i] vj                                 (the same column after the step's q is known)
inline double sym_col_mv(const double *col, const double *v, double *p, double vj, int len)
{
   v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
   const v4d vjv = {vj, vj, vj, vj};
   int i = 1;
   for (; i + 8 <= len; i += 8) {
      v4d c0, c1, v0, v1, p0, p1;
      std::memcpy(&c0, col + i, 32);
      std::memcpy(&c1, col + i + 4, 32);
      std::memcpy(&v0, v + i, 32);
      std::memcpy(&v1, v + i + 4, 32);
      std::memcpy(&p0, p + i, 32);
      std::memcpy(&p1, p + i + 4, 32);
      s0 += c0 * v0;
      s1 += c1 * v1;
      p0 += c0 * vjv;
      p1 += c1 * vjv;
      std::memcpy(p + i, &p0, 32);
      std::memcpy(p + i + 4, &p1, 32);
   }
   const v4d t = s0 + s1;
   double s = (t[0] + t[1]) + (t[2] + t[3]);
   for (; i < len; i++) {
      s += col[i] * v[i];
      p[i] += col[i] * vj;
   }
   return s;
}
inline void sym_col_r2(double *col, const double *v, const double *q, double vj, double qj, int len)
{
   const v4d vjv = {vj, vj, vj, vj}, qjv = {qj, qj, qj, qj};
   int i = 0;
   for (; i + 4 <= len; i += 4) {
      v4d c, vv, qq;
      std::memcpy(&c, col + i, 32);
      std::memcpy(&vv, v + i, 32);
      std::memcpy(&qq, q + i, 32);
      c -= vv * qjv + qq * vjv;
      std::memcpy(col + i, &c, 32);
   }
   for (; i < len; i++) col[i] -= v[i] * qj + q[i] * vj;
}

// Only the LOWER triangle of A is read and updated (round 5: rounds 1-4 kept both triangles current -- twice the flops and twice
// the traffic of a reduction that is bound by the L2 bandwidth of the one core it runs on: 11.2 -> 6 ms at n = 384).  On return the
// strict upper triangle of the trailing blocks is stale; nothing downstream reads it (the reflectors sit below the sub-diagonal).
void tridiagonalise(int n, double *A, int lda, double *d, double *e, int row0 = 0, int nrows = 0, double *rows = nullptr,
                    double *beta_out = nullptr)
{
#define AA(i, j) A[(size_t)(i) + (size_t)(j) * lda]
   std::vector<double> v(n), p(n), beta(n, 0.0);
   // Householder vectors are stored below the sub-diagonal of A (column k, rows k+2..n-1; v[k+1] implicit 1).
   // (Splitting the two O(m^2) passes of a step over 4-8 threads with spin barriers was measured on the 256-core host of
   // the GPU box: 6.8 vs 9.5 ms at n = 512, no gain at n <= 384, and the solves that matter got slower -- not kept.)
   for (int k = 0; k < n - 2; k++) {
      const int m = n - k - 1; // length of x = A[k+1.., k]
      double *x = &AA(k + 1, k);
      double nrm = 0;
      for (int i = 1; i < m; i++) nrm += x[i] * x[i];
      if (nrm == 0.0) {
         e[k] = x[0];
         d[k] = AA(k, k);
         beta[k] = 0;
         continue;
      }
      const double x0 = x[0];
      const double alpha = -std::copysign(std::sqrt(x0 * x0 + nrm), x0);
      // v = x - alpha e1, scaled so that v[0] = 1
      const double v0 = x0 - alpha;
      v[0] = 1.0;
      for (int i = 1; i < m; i++) v[i] = x[i] / v0;
      const double bk = -v0 / alpha; // 2 / (v'v) with v[0] = 1
      beta[k] = bk;
      // p = A22 v from the lower triangle: column j (from its diagonal down) gives p[j] += a_jj v_j + col[1:].v[j+1:] and
      // p[j+1:] += col[1:] v_j -- every stored element is read once
      for (int i = 0; i < m; i++) p[i] = 0.0;
      for (int j = 0; j < m; j++) {
         const double *col = &AA(k + 1 + j, k + 1 + j);
         p[j] += col[0] * v[j] + sym_col_mv(col, v.data() + j, p.data() + j, v[j], m - j);
      }
      double vp = 0;
      for (int i = 0; i < m; i++) {
         p[i] *= bk;
         vp += v[i] * p[i];
      }
      const double kk = 0.5 * bk * vp;
      for (int i = 0; i < m; i++) p[i] -= kk * v[i]; // q
      // A22 -= v q' + q v'  (lower triangle)
      for (int j = 0; j < m; j++) sym_col_r2(&AA(k + 1 + j, k + 1 + j), v.data() + j, p.data() + j, v[j], p[j], m - j);
      d[k] = AA(k, k);
      e[k] = alpha;
      for (int i = 1; i < m; i++) x[i] = v[i]; // keep the reflector (x[0] slot is not needed)
   }
   if (n >= 2) {
      d[n - 2] = AA(n - 2, n - 2);
      e[n - 2] = AA(n - 1, n - 2);
   }
   d[n - 1] = AA(n - 1, n - 1);
   if (beta_out) {
      for (int k = 0; k < n; k++) beta_out[k] = beta[k];
      if (!rows) return;
   }
   if (rows) {
      // e_r' Q = e_r' H_0 H_1 ... H_{n-3}: the reflectors applied to unit rows, front to back
      for (int j = 0; j < n; j++)
         for (int r = 0; r < nrows; r++) rows[(size_t)r + (size_t)j * nrows] = (j == row0 + r) ? 1.0 : 0.0;
      std::vector<double> s(nrows);
      for (int k = 0; k < n - 2; k++) {
         if (beta[k] == 0.0) continue;
         const int m = n - k - 1;
         v[0] = 1.0;
         for (int i = 1; i < m; i++) v[i] = AA(k + 1 + i, k);
         // 16 rows at a time in four vector accumulators (the residual test asks for the last block's 16 rows: left to the
         // compiler this loop nest ran scalar and cost as much as half the reduction itself), the remainder row by row
         int r0 = 0;
         for (; r0 + 16 <= nrows; r0 += 16) {
            v4d s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0;
            for (int i = 0; i < m; i++) {
               const double *col = &rows[(size_t)(k + 1 + i) * nrows + r0];
               const v4d vi = {v[i], v[i], v[i], v[i]};
               v4d c0, c1, c2, c3;
               std::memcpy(&c0, col, 32);
               std::memcpy(&c1, col + 4, 32);
               std::memcpy(&c2, col + 8, 32);
               std::memcpy(&c3, col + 12, 32);
               s0 += c0 * vi;
               s1 += c1 * vi;
               s2 += c2 * vi;
               s3 += c3 * vi;
            }
            for (int i = 0; i < m; i++) {
               double *col = &rows[(size_t)(k + 1 + i) * nrows + r0];
               const double bvi = beta[k] * v[i];
               const v4d bv = {bvi, bvi, bvi, bvi};
               v4d c0, c1, c2, c3;
               std::memcpy(&c0, col, 32);
               std::memcpy(&c1, col + 4, 32);
               std::memcpy(&c2, col + 8, 32);
               std::memcpy(&c3, col + 12, 32);
               c0 -= s0 * bv;
               c1 -= s1 * bv;
               c2 -= s2 * bv;
               c3 -= s3 * bv;
               std::memcpy(col, &c0, 32);
               std::memcpy(col + 4, &c1, 32);
               std::memcpy(col + 8, &c2, 32);
               std::memcpy(col + 12, &c3, 32);
            }
         }
         if (r0 < nrows) {
            std::fill(s.begin(), s.end(), 0.0);
            for (int i = 0; i < m; i++) {
               const double *col = &rows[(size_t)(k + 1 + i) * nrows];
               for (int r = r0; r < nrows; r++) s[r] += col[r] * v[i];
            }
            for (int i = 0; i < m; i++) {
               double *col = &rows[(size_t)(k + 1 + i) * nrows];
               const double bv = beta[k] * v[i];
               for (int r = r0; r < nrows; r++) col[r] -= s[r] * bv;
            }
         }
      }
      return;
   }
   // accumulate Q = H_0 H_1 ... H_{n-3} into A (overwriting the reduced matrix), back to front
   std::vector<double> Q((size_t)n * n, 0.0);
   for (int i = 0; i < n; i++) Q[(size_t)i + (size_t)i * n] = 1.0;
   for (int k = n - 3; k >= 0; k--) {
      if (beta[k] == 0.0) continue;
      const int m = n - k - 1;
      v[0] = 1.0;
      for (int i = 1; i < m; i++) v[i] = AA(k + 1 + i, k);
      // Q[k+1.., k+1..] -= beta v (v' Q[k+1.., k+1..])
      for (int j = 0; j < m; j++) {
         double *col = &Q[(size_t)(k + 1) + (size_t)(k + 1 + j) * n];
         double s = 0;
         for (int i = 0; i < m; i++) s += v[i] * col[i];
         s *= beta[k];
         for (int i = 0; i < m; i++) col[i] -= s * v[i];
      }
   }
   for (int j = 0; j < n; j++) std::memcpy(&AA(0, j), &Q[(size_t)j * n], sizeof(double) * n);
#undef AA
}

// sqrt(a^2 + b^2) without the cost of std::hypot's correct rounding (called once per rotation); the operands are
// entries of a tridiagonal matrix, scaled through the larger one so that tiny couplings do not underflow
static inline double pythag(double a, double b)
{
   const double x = std::fabs(a), y = std::fabs(b);
   const double hi = x > y ? x : y, lo = x > y ? y : x;
   if (hi == 0.0) return 0.0;
   const double r = lo / hi;
   return hi * std::sqrt(1.0 + r * r);
}

// the same on the rotation chain of the QL sweep, where its latency is what the sweep costs: the plain formula whenever
// the squares are safely inside the double range
static inline double pythag_fast(double a, double b)
{
   const double h = a * a + b * b;
   return (h > 1e-280 && h < 1e280) ? std::sqrt(h) : pythag(a, b);
}

// implicit-shift QL on (d, e) accumulating the rotations into the columns of Z (n x n, ld ldz)
int tridiag_ql(int n, double *d, double *e_in, double *Z, int ldz, int zrows)
{
   std::vector<double> e(n + 1, 0.0);
   for (int i = 0; i < n - 1; i++) e[i] = e_in[i];
   for (int l = 0; l < n; l++) {
      int iter = 0, m;
      do {
         for (m = l; m < n - 1; m++) {
            const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
            if (std::fabs(e[m]) <= DBL_EPSILON * dd) break;
         }
         if (m != l) {
            if (iter++ == 300) return 1;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = pythag(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
            double s = 1.0, c = 1.0, p = 0.0;
            int i;
            for (i = m - 1; i >= l; i--) {
               double f = s * e[i];
               const double bb = c * e[i];
               r = pythag_fast(f, g);
               e[i + 1] = r;
               if (r == 0.0) {
                  d[i + 1] -= p;
                  e[m] = 0.0;
                  break;
               }
               const double rinv = 1.0 / r;
               s = f * rinv;
               c = g * rinv;
               g = d[i + 1] - p;
               r = (d[i] - g) * s + 2.0 * c * bb;
               p = s * r;
               d[i + 1] = g + p;
               g = c * r - bb;
               double *z0 = Z + (size_t)i * ldz, *z1 = Z + (size_t)(i + 1) * ldz;
               for (int k = 0; k < zrows; k++) {
                  f = z1[k];
                  z1[k] = s * z0[k] + c * f;
                  z0[k] = c * z0[k] - s * f;
               }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
         }
      } while (m != l);
   }
   return 0;
}

} // namespace

int symeig_desc(int n, double *A, int lda, double *w)
{
   if (n <= 0) return 0;
   if (n == 1) {
      w[0] = A[0];
      A[0] = 1.0;
      return 0;
   }
   std::vector<double> d(n), e(n, 0.0);
   tridiagonalise(n, A, lda, d.data(), e.data());
   int rc = tridiag_ql(n, d.data(), e.data(), A, lda, n);
   if (rc) return rc;
   std::vector<int> idx(n);
   std::iota(idx.begin(), idx.end(), 0);
   std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] > d[b]; });
   std::vector<double> Z((size_t)n * n);
   for (int j = 0; j < n; j++) {
      w[j] = d[idx[j]];
      std::memcpy(&Z[(size_t)j * n], A + (size_t)idx[j] * lda, sizeof(double) * n);
   }
   for (int j = 0; j < n; j++) std::memcpy(A + (size_t)j * lda, &Z[(size_t)j * n], sizeof(double) * n);
   return 0;
}

int symeig_desc_rows(int n, double *A, int lda, double *w, int row0, int nrows, double *Zr, TridiagKeep *keep)
{
   if (keep) keep->n = 0;
   if (n <= 0) return 0;
   if (n == 1) {
      w[0] = A[0];
      if (nrows > 0) Zr[0] = 1.0;
      return 0;
   }
   std::vector<double> d(n), e(n, 0.0), rows((size_t)nrows * n);
   // The reduction runs on a copy whose column stride is an ODD number of cache lines: the projected matrices of the solver have
   // orders that are multiples of 16, and at a stride of 2,048 bytes (n = 256) every column falls into the same L1 sets -- the
   // reduction took 8.5 ms there against 5 at n = 320.
   int ldw = (n + 7) / 8 * 8;
   if ((ldw / 8) % 2 == 0) ldw += 8;
   std::vector<double> Wk((size_t)ldw * n);
   for (int j = 0; j < n; j++) std::memcpy(&Wk[(size_t)j * ldw], A + (size_t)j * lda, sizeof(double) * n);
   if (keep && n > 2) { // the reduction is kept for symeig_cols_from_keep: original matrix, reflectors, tridiagonal form
      keep->A0.resize((size_t)n * n);
      for (int j = 0; j < n; j++) std::memcpy(&keep->A0[(size_t)j * n], A + (size_t)j * lda, sizeof(double) * n);
      keep->beta.assign(n, 0.0);
      tridiagonalise(n, Wk.data(), ldw, d.data(), e.data(), row0, nrows, rows.data(), keep->beta.data());
      keep->A.resize((size_t)n * n);
      for (int j = 0; j < n; j++) std::memcpy(&keep->A[(size_t)j * n], &Wk[(size_t)j * ldw], sizeof(double) * n);
      keep->d = d;
      keep->e = e;
      keep->n = n;
   } else
      tridiagonalise(n, Wk.data(), ldw, d.data(), e.data(), row0, nrows, rows.data());
   int rc = tridiag_ql(n, d.data(), e.data(), rows.data(), nrows, nrows);
   if (rc) return rc;
   std::vector<int> idx(n);
   std::iota(idx.begin(), idx.end(), 0);
   std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] > d[b]; });
   for (int j = 0; j < n; j++) {
      w[j] = d[idx[j]];
      std::memcpy(Zr + (size_t)j * nrows, &rows[(size_t)idx[j] * nrows], sizeof(double) * nrows);
   }
   return 0;
}

// First ncols eigenvectors only (largest eigenvalues): tridiagonalise, all eigenvalues by QL without vectors, the wanted
// eigenvectors of the tridiagonal by inverse iteration (LU with partial pivoting of T - lambda I, modified Gram-Schmidt
// against the already computed neighbours of a cluster, as LAPACK's dstein does), back-transformation through the
// Householder reflectors.  O(4/3 n^3 + n^2 ncols) instead of O(6 n^3).  The result is VERIFIED -- residual
// ||A z - lambda z|| and orthonormality -- and a nonzero return tells the caller to use symeig_desc instead.
namespace {
int cols_core(int n, const double *A, int lda, const std::vector<double> &d, const std::vector<double> &e, const std::vector<double> &beta,
              const std::vector<double> &A0, const double *w, int ncols, double *Z);
}

int symeig_desc_cols(int n, double *A, int lda, double *w, int ncols, double *Z)
{
   if (n <= 2 || ncols <= 0 || ncols > n) return 1;
   std::vector<double> A0((size_t)n * n);
   for (int j = 0; j < n; j++) std::memcpy(&A0[(size_t)j * n], A + (size_t)j * lda, sizeof(double) * n);
   std::vector<double> d(n), e(n, 0.0), beta(n, 0.0), dq(n), eq(n);
   tridiagonalise(n, A, lda, d.data(), e.data(), 0, 0, nullptr, beta.data());
   dq = d;
   eq = e;
   if (tridiag_ql(n, dq.data(), eq.data(), nullptr, 0, 0) != 0) return 2;
   std::sort(dq.begin(), dq.end(), [](double a, double b) { return a > b; });
   for (int j = 0; j < n; j++) w[j] = dq[j];
   return cols_core(n, A, lda, d, e, beta, A0, w, ncols, Z);
}

// the same from a reduction symeig_desc_rows has already made (its eigenvalues w, descending): no second O(n^3) pass
int symeig_cols_from_keep(const TridiagKeep &keep, const double *w, int ncols, double *Z)
{
   const int n = keep.n;
   if (n <= 2 || ncols <= 0 || ncols > n) return 1;
   return cols_core(n, keep.A.data(), n, keep.d, keep.e, keep.beta, keep.A0, w, ncols, Z);
}

namespace {
int cols_core(int n, const double *A, int lda, const std::vector<double> &d, const std::vector<double> &e, const std::vector<double> &beta,
              const std::vector<double> &A0, const double *w, int ncols, double *Z)
{
   double tnorm = 0;
   for (int i = 0; i < n; i++) tnorm = std::max(tnorm, std::fabs(d[i]) + (i ? std::fabs(e[i - 1]) : 0.0) + (i < n - 1 ? std::fabs(e[i]) : 0.0));
   if (!(tnorm > 0.0) || !std::isfinite(tnorm)) return 3;
   const double eps = DBL_EPSILON, sep = 1e-3 * tnorm; // dstein's cluster criterion
   std::vector<double> dl(n), du(n), du2(n), dd(n), x(n), y((size_t)n * ncols);
   std::vector<int> piv(n);
   uint64_t rng = 0x9E3779B97F4A7C15ull;
   int cluster0 = 0; // first column of the current cluster
   double prev_shift = 0;
   for (int c = 0; c < ncols; c++) {
      double shift = w[c];
      if (c > 0 && std::fabs(w[c] - w[c - 1]) > sep) cluster0 = c;
      if (c > cluster0 && !(prev_shift - shift > 10.0 * eps * tnorm)) shift = prev_shift - 10.0 * eps * tnorm; // keep the shifts apart
      prev_shift = shift;
      // LU of (T - shift I) with partial pivoting: rows are (dl, dd, du), fill-in du2
      for (int i = 0; i < n; i++) {
         dd[i] = d[i] - shift;
         du[i] = i < n - 1 ? e[i] : 0.0;
         dl[i] = i < n - 1 ? e[i] : 0.0;
         du2[i] = 0.0;
      }
      for (int i = 0; i < n - 1; i++) {
         if (std::fabs(dd[i]) >= std::fabs(dl[i])) {
            piv[i] = 0;
            if (dd[i] == 0.0) dd[i] = eps * tnorm;
            const double f = dl[i] / dd[i];
            dl[i] = f;
            dd[i + 1] -= f * du[i];
         } else {
            piv[i] = 1;
            const double f = dd[i] / dl[i];
            dd[i] = dl[i];
            dl[i] = f;
            const double t = du[i];
            du[i] = dd[i + 1];
            dd[i + 1] = t - f * du[i];
            if (i < n - 2) {
               du2[i] = du[i + 1];
               du[i + 1] = -f * du2[i];
            }
         }
      }
      if (dd[n - 1] == 0.0) dd[n - 1] = eps * tnorm;
      for (int i = 0; i < n; i++) {
         rng ^= rng << 13;
         rng ^= rng >> 7;
         rng ^= rng << 17;
         x[i] = (double)(rng >> 11) / 9007199254740992.0 - 0.5;
      }
      bool done = false;
      for (int it = 0; it < 6 && !done; it++) {
         double nrm = 0;
         for (int i = 0; i < n; i++) nrm = std::max(nrm, std::fabs(x[i]));
         if (!(nrm > 0.0)) return 4;
         const double sc = (double)n * tnorm * eps / nrm; // dstein's scaling: keeps the solve from overflowing
         for (int i = 0; i < n; i++) x[i] *= sc;
         // forward: apply the row operations
         for (int i = 0; i < n - 1; i++) {
            if (piv[i]) std::swap(x[i], x[i + 1]);
            x[i + 1] -= dl[i] * x[i];
         }
         // backward
         x[n - 1] /= dd[n - 1];
         if (n >= 2) x[n - 2] = (x[n - 2] - du[n - 2] * x[n - 1]) / dd[n - 2];
         for (int i = n - 3; i >= 0; i--) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / dd[i];
         for (int j = cluster0; j < c; j++) { // modified Gram-Schmidt within the cluster
            const double *yj = &y[(size_t)j * n];
            double dot = 0;
            for (int i = 0; i < n; i++) dot += yj[i] * x[i];
            for (int i = 0; i < n; i++) x[i] -= dot * yj[i];
         }
         double n2 = 0, ninf = 0;
         for (int i = 0; i < n; i++) {
            n2 += x[i] * x[i];
            ninf = std::max(ninf, std::fabs(x[i]));
         }
         if (!(n2 > 0.0) || !std::isfinite(n2)) return 5;
         const double inv = 1.0 / std::sqrt(n2);
         for (int i = 0; i < n; i++) x[i] *= inv;
         if (it >= 1 && ninf >= std::sqrt(0.1 / n)) done = true; // grown enough (dstein: two good iterations)
      }
      std::memcpy(&y[(size_t)c * n], x.data(), sizeof(double) * n);
   }
   // Back-transformation and verification are independent per column: at the sizes of a thick restart (n ~ 400, 80 columns:
   // ~9 ms, which the pass in flight beside it does not cover) they are spread over a few threads.  Every column sees the same
   // arithmetic in the same order whatever the thread count: the result does not depend on it.
   const int T = (n >= 192 && ncols >= 32) ? std::max(1, std::min(std::min((int)usable_cpus(), 8), ncols / 8)) : 1;
   // (per-thread scratch is allocated here, before anything is spawned; a thread body that throws reports 8, and whatever
   //  has been started is joined before this function is left -- by return or by an exception from thread creation: a
   //  non-zero return sends the caller to symeig_desc, nothing ends in std::terminate)
   std::vector<std::vector<double>> scratch((size_t)T, std::vector<double>((size_t)8 * n)); // (a reflector; eight product columns)
   auto on_columns = [&](auto &&fn) { // fn(c0, c1, scratch) -> int
      std::vector<int> rcs((size_t)T, 0);
      std::vector<std::thread> th;
      struct Joiner {
         std::vector<std::thread> &t;
         ~Joiner()
         {
            for (auto &x : t)
               if (x.joinable()) x.join();
         }
      } joiner{th};
      auto body = [&](int t) {
         try {
            rcs[(size_t)t] = fn(ncols * t / T, ncols * (t + 1) / T, scratch[(size_t)t]);
         } catch (...) {
            rcs[(size_t)t] = 8;
         }
      };
      try {
         th.reserve((size_t)T);
         for (int t = 1; t < T; t++) th.emplace_back(body, t);
      } catch (...) { // thread creation failed: the columns of the threads that never started are done here
         const int started = (int)th.size() + 1;
         for (int t = started; t < T; t++) body(t);
      }
      body(0);
      for (auto &x : th) x.join();
      for (int rc : rcs)
         if (rc) return rc;
      return 0;
   };
   // back-transformation: z = H_0 H_1 ... H_{n-3} y
   const int brc = on_columns([&](int c0, int c1, std::vector<double> &v) {
      for (int k = n - 3; k >= 0; k--) {
         if (beta[k] == 0.0) continue;
         const int m = n - k - 1;
         v[0] = 1.0;
         for (int i = 1; i < m; i++) v[i] = A[(size_t)(k + 1 + i) + (size_t)k * lda];
         for (int c = c0; c < c1; c++) {
            double *col = &y[(size_t)c * n + (k + 1)];
            const double sdot = beta[k] * dot_fixed(v.data(), col, m);
            for (int i = 0; i < m; i++) col[i] -= sdot * v[i];
         }
      }
      return 0;
   });
   if (brc) return brc;
   // verification against the original matrix
   double anorm = 0;
   for (size_t i = 0; i < A0.size(); i++) anorm = std::max(anorm, std::fabs(A0[i]));
   anorm = std::max(anorm * n, tnorm);
   // (A0 z for a thread's columns eight at a time: every column of A0 is read once per eight products, not once per product -- the
   //  matrix does not fit in a core's L2 at restart sizes, and eight threads re-streaming it ten times each was most of this pass)
   const int vrc = on_columns([&](int c0, int c1, std::vector<double> &R) {
      for (int cb = c0; cb < c1; cb += 8) {
         const int nc = std::min(8, c1 - cb);
         std::fill(R.begin(), R.begin() + (size_t)nc * n, 0.0);
         for (int j = 0; j < n; j++) {
            const double *col = &A0[(size_t)j * n];
            for (int q = 0; q < nc; q++) {
               const double zj = y[(size_t)(cb + q) * n + j];
               double *r = &R[(size_t)q * n];
               for (int i = 0; i < n; i++) r[i] += col[i] * zj;
            }
         }
         for (int q = 0; q < nc; q++) {
            const double *z = &y[(size_t)(cb + q) * n], *r = &R[(size_t)q * n];
            double res = 0;
            for (int i = 0; i < n; i++) res = std::max(res, std::fabs(r[i] - w[cb + q] * z[i]));
            if (!(res <= 1e-11 * anorm)) return 6;
         }
      }
      for (int c = c0; c < c1; c++) {
         const double *z = &y[(size_t)c * n];
         for (int j = std::max(0, c - 64); j <= c; j++) {
            const double dot = dot_fixed(z, &y[(size_t)j * n], n);
            if (!(std::fabs(dot - (j == c ? 1.0 : 0.0)) <= 1e-10)) return 7;
         }
      }
      return 0;
   });
   if (vrc) return vrc;
   for (int c = 0; c < ncols; c++) std::memcpy(Z + (size_t)c * n, &y[(size_t)c * n], sizeof(double) * n);
   return 0;
}
} // namespace

int cholesky_upper(int n, double *G, int ld, double rel_tol)
{
#define GG(i, j) G[(size_t)(i) + (size_t)(j) * ld]
   double dmax = 0;
   for (int i = 0; i < n; i++) dmax = std::max(dmax, std::fabs(GG(i, i)));
   const double thresh = rel_tol * dmax;
   for (int j = 0; j < n; j++) {
      double s = GG(j, j);
      for (int k = 0; k < j; k++) s -= GG(k, j) * GG(k, j);
      if (!(s > thresh) || !std::isfinite(s)) return j + 1;
      const double rjj = std::sqrt(s);
      GG(j, j) = rjj;
      for (int c = j + 1; c < n; c++) {
         double t = GG(j, c);
         for (int k = 0; k < j; k++) t -= GG(k, j) * GG(k, c);
         GG(j, c) = t / rjj;
      }
   }
   for (int j = 0; j < n; j++)
      for (int i = j + 1; i < n; i++) GG(i, j) = 0.0;
#undef GG
   return 0;
}

void upper_inverse(int n, const double *R, int ldr, double *Rinv, int ldi)
{
   for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) Rinv[(size_t)i + (size_t)j * ldi] = 0.0;
   for (int j = 0; j < n; j++) {
      Rinv[(size_t)j + (size_t)j * ldi] = 1.0 / R[(size_t)j + (size_t)j * ldr];
      for (int i = j - 1; i >= 0; i--) {
         double s = 0;
         for (int k = i + 1; k <= j; k++) s += R[(size_t)i + (size_t)k * ldr] * Rinv[(size_t)k + (size_t)j * ldi];
         Rinv[(size_t)i + (size_t)j * ldi] = -s / R[(size_t)i + (size_t)i * ldr];
      }
   }
}

} // namespace fpca
