/*
 * oracle/fpca_oracle.h -- CPU restatement of flashpca's PCA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (flashpca_amd/, libfpca.so, the flashpca CLI) never links or calls it.
 *
 * PARITY PINNING STATUS: the reference cannot be built in this environment (needs Eigen, Spectra
 * v0.8.1 and Boost, none present) and its tests hold no stored numeric vectors (they recompute the
 * expectation live with R's dense eigen()/svd(): flashpcaR/tests/testthat/test_pca.R:24-43,
 * HapMap3/test_pca.R:121-246).  So this oracle is "parity unpinned" in the strict sense (no
 * reference-run outputs, no reference-held goldens); it IS pinned the way the reference's own tests
 * pin the reference: against an independent dense eigendecomposition of X X'/P on the reference's
 * bundled filesets (tests/golden/make_golden.py -> golden_*.json, tests/test_oracle_golden.py), and against the known
 * answers printed in SURVEY.md 8(c) (a third, independent computation).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 * All matrices are fp64 column-major, like the reference's Eigen::MatrixXd.
 */
#ifndef FPCA_ORACLE_H
#define FPCA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_STANDARDISE_BINOM 2  /* util.h:36 */
#define ORC_STANDARDISE_BINOM2 3 /* util.h:37 */
#define ORC_DIVISOR_NONE 0       /* randompca.h:41-43 */
#define ORC_DIVISOR_N1 1
#define ORC_DIVISOR_P 2

typedef struct orc_data orc_data; /* the slice of class Data (data.h:60-101) this path uses */
typedef struct orc_op orc_op;     /* SVDWideOnline (svdwide.h:40-107) */

/* data.cpp:65-126 / 128-148 */
void orc_decode_plink(unsigned char *out, const unsigned char *in, unsigned int n);
void orc_decode_plink_simple(unsigned char *out, const unsigned char *in, unsigned int n);

/* Data::get_size + Data::prepare (data.cpp:150-206) on a .bed file. N comes from the .fam. */
orc_data *orc_open_file(const char *bed_path, uint64_t N, int stand_method, char *err, int errlen);
/* Same object over an in-memory packed stream of P records of ceil(N/4) bytes (no 3-byte header). */
orc_data *orc_open_mem(const unsigned char *packed, uint64_t N, uint64_t P, int stand_method);
void orc_close(orc_data *d);
uint64_t orc_N(const orc_data *d);
uint64_t orc_nsnps(const orc_data *d);
uint64_t orc_np(const orc_data *d);
/* preloaded mean/sd (projection path, data.cpp:293-297); meansd is P x 2 column-major */
void orc_set_preloaded_meansd(orc_data *d, const double *meansd);
/* Data::read_snp_block(start, stop, false, false) (data.cpp:215-335); X is N x (stop-start+1) */
int orc_read_snp_block(orc_data *d, uint32_t start, uint32_t stop, double *X);
/* P x 2 column-major (mean | sd); rows are filled when a SNP is first visited */
const double *orc_meansd(const orc_data *d);
/* 4 x P column-major lookup table indexed by raw PLINK code (data.cpp:300-320) */
const double *orc_lookup(const orc_data *d);

/* SVDWideOnline ctor (svdwide.h:51-73) */
orc_op *orc_op_new(orc_data *d, uint32_t block_size, int nthreads);
void orc_op_free(orc_op *op);
void orc_perform_op(orc_op *op, const double *x_in, double *y_out);              /* svdwide.cpp:21-68 */
void orc_perform_op_mat(orc_op *op, const double *X, int ncols, double *Y);      /* svdwide.cpp:71-118 */
void orc_crossprod(orc_op *op, const double *x_in, double *y_out);               /* svdwide.cpp:122-153 */
void orc_prod(orc_op *op, const double *v_in, double *y_out);                    /* svdwide.cpp:193-226 */
double orc_op_trace(const orc_op *op);
uint32_t orc_op_nops(const orc_op *op);
uint32_t orc_op_nblocks(const orc_op *op);

/* Spectra::SymEigsSolver<double, LARGEST_ALGE, Op>(op, nev, ncv); init(); compute(maxit, tol)
 * (third-party, v0.8.1 pinned by the reference's Dockerfile:19-20; call sites randompca.cpp:173-178).
 * evals[nev] descending, evecs N x nev column-major.  Returns number of converged pairs;
 * *info = 0 on success (>= nev converged), 1 = not converging. */
int orc_symeigs(orc_op *op, int nev, int ncv, int maxit, double tol, double *evals, double *evecs,
                int *info, int *nrestarts);

/* RandomPCA::pca_fast(Data&, ...) (randompca.cpp:168-218).  Outputs (caller-allocated):
 *  U N x k, d k, V P x k (only if do_loadings), Px N x k, pve k, *trace.  Returns 0 / nonzero. */
int orc_pca_fast(orc_data *d, uint32_t block_size, int ndim, int maxiter, double tol, int divisor,
                 int do_loadings, int nthreads, double *U, double *dvals, double *V, double *Px,
                 double *pve, double *trace, uint32_t *nops);

/* RandomPCA::check(Data&, block_size, evec, eval) (randompca.cpp:663-703): err[k], *mse, *rmse */
int orc_check(orc_data *d, uint32_t block_size, int divisor, const double *evec, const double *eval,
              int k, double *err, double *mse, double *rmse);

/* standardise(MatrixXd&, method) (util.cpp:24-192), in place; methods 0 none, 1 sd, 2 binom, 3 binom2, 4 center */
int orc_standardise(double *X, uint64_t n, uint64_t p, int method, double *meansd);

/* block size heuristic of the CLI (flashpca.cpp:636-686); returns 0 if memory insufficient */
uint32_t orc_default_block_size(uint64_t N, uint64_t nsnps, int ndim, int do_loadings, int memory_mb);

/* save_text number formatting (util.h:69-108): std::setprecision(p) default-float == "%.{p}g" */
int orc_format_number(char *buf, int buflen, double v, int precision);

#ifdef __cplusplus
}
#endif
#endif
