/*
 * include/fpca.h -- C ABI of the MI355X-native flashpca PCA hot path (libfpca.so).
 *
 * The reference (gabraham/flashpca 2.1) has no C ABI of its own on this path; its seams are
 *   (1) the Spectra operator concept implemented by SVDWideOnline:
 *          unsigned rows(); unsigned cols(); void perform_op(const double* x_in, double* y_out);
 *                                                                    (svdwide.h:77-81, svdwide.cpp:21-68)
 *   (2) the C++ driver  RandomPCA::pca_fast(Data&, block_size, ndim, maxiter, tol, seed, do_loadings)
 *          with inputs  stand_method_x / divisor and outputs U, d, V, Px, pve, trace, X_meansd
 *                                                                    (randompca.h:56-80, randompca.cpp:168-218)
 *   (3) the data object  Data::{get_size, prepare, read_snp_block}   (data.h:60-101, data.cpp:150-335)
 * Every entry point below names the reference interface it replaces.  Conventions:
 *   - plain pointers and sizes only; no C++/torch types;
 *   - host matrices are fp64 COLUMN-major with an explicit leading dimension (what Eigen::MatrixXd /
 *     Map<VectorXd> hand to the reference operator);
 *   - functions return 0 on success, a negative FPCA_E* code on failure; fpca_last_error() returns the
 *     message of the last failure on the calling thread (the reference throws std::runtime_error,
 *     flashpca.cpp:882-892 turns that into EXIT_FAILURE);
 *   - a context owns ONE SNP shard [snp_begin, snp_begin+P_g) of the genotype matrix on ONE GPU:
 *     packed 2-bit stream resident in HBM, per-SNP mean/sd/lookup table, workspaces and a HIP stream.
 *     Multi-GPU = one context per process per GPU; the N x b product is summed across ranks either by
 *     RCCL inside the library (fpca_comm_init_rank) or by a caller-supplied all-reduce (fpca_set_allreduce);
 *   - there is NO CPU fallback: fpca_create* fail with FPCA_ENODEVICE when no gfx950 device is usable.
 */
#ifndef FPCA_H
#define FPCA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPCA_VERSION "0.3.0"
/* Binary interface revision: bumped whenever a struct below changes size or a field changes meaning.  A binding checks
 * fpca_abi_version() == FPCA_ABI_VERSION once after loading the library (flashpca_amd/_lib.py does).
 *   2 (library 0.2.0): fpca_pca_opts / fpca_pca_info carry their own size and the mixed-precision fields; `maxiter` counts the
 *     reference's restarts (it was a cap on block applies in 0.1.0 -- use max_applies for that).
 *   3 (library 0.3.0): fpca_pca_opts.partial_rows, fpca_pca_info.solver_path; the struct sizes are the CALLER's
 *     (FPCA_PCA_OPTS_INIT / fpca_pca_init_opts); fpca_pca_row_ranges. */
#define FPCA_ABI_VERSION 4

/* standardisation methods: same numeric values as the reference (util.h:34-38); the packed-genotype constructors
 * accept BINOM / BINOM2 like the CLI (flashpca.cpp:336-349), fpca_create_dense accepts all five */
#define FPCA_STANDARDISE_NONE 0
#define FPCA_STANDARDISE_SD 1
#define FPCA_STANDARDISE_BINOM 2
#define FPCA_STANDARDISE_BINOM2 3
#define FPCA_STANDARDISE_CENTER 4
/* eigenvalue divisor: same numeric values as the reference (randompca.h:41-43) */
#define FPCA_DIVISOR_NONE 0
#define FPCA_DIVISOR_N1 1
#define FPCA_DIVISOR_P 2
/* arithmetic of the two genotype GEMMs.  AUTO = the exact-integer path FPCA_ACCUM_I8(7) for 2-bit input, FP64 for dense input.  If
 * the second, sample-major 2-bit copy the integer X T needs does not fit in device memory, AUTO keeps X'B on the int8 cores and runs
 * X T on the FP64 kernel (same results; 30 instead of 11 ms per 16-column pass at 500,000 x 100,000); if the int8 operands do not
 * fit either, both stages run FP64 (47 ms) */
#define FPCA_ACCUM_AUTO 0
#define FPCA_ACCUM_FP64 64
#define FPCA_ACCUM_FP32 32
/* exact-integer mode: the fp64 operand is rounded to (8S-2)-bit fixed point (shared power-of-two scale per column), cut
 * into its S bytes and multiplied with the integer genotype matrices on the int8 matrix cores with exact int32
 * accumulation; S = 7 keeps 54 bits below the column maximum (fp64-equivalent, the default), S = 4 (30 bits) roughly
 * single precision but still without accumulation error.  S = 2..8. */
#define FPCA_ACCUM_I8(S) (800 + (S))

#define FPCA_OK 0
#define FPCA_EINVAL -1      /* bad argument */
#define FPCA_ENODEVICE -2   /* no usable gfx950 device / HIP runtime error at init */
#define FPCA_EHIP -3        /* HIP runtime or kernel failure */
#define FPCA_ENOMEM -4      /* host or device allocation failed */
#define FPCA_ENOTCONVERGED -5 /* eigensolver hit maxiter (reference: "Spectra eigen-decomposition was not successful", randompca.cpp:212-217) */
#define FPCA_ECOMM -6       /* RCCL / all-reduce failure */
#define FPCA_EIO -7         /* file error (reference: "[Data::read_bed] Error reading file", data.cpp:156-161) */

typedef struct fpca_ctx fpca_ctx;

const char *fpca_last_error(void);
const char *fpca_version(void);
int fpca_abi_version(void);
/* number of visible HIP devices (<0 on error); name/arch of one device into buf */
int fpca_device_count(void);
int fpca_device_name(int device, char *buf, int buflen);
/* start the HIP runtime and the device's primary context (0.1-0.2 s the first time in a process).  Thread-safe and
 * idempotent: a caller may run it on a helper thread while it parses its text inputs (the CLI does); every fpca_create*
 * does the same work itself if nobody has. */
int fpca_warmup(int device);

/* ------------------------------------------------------------------------------------------------
 * Data: replaces Data::get_size + Data::prepare (data.cpp:150-206).  `packed` is the body of a SNP-major
 * PLINK .bed for this shard: P_g records of ceil(N/4) bytes (i.e. file offset 3 + ceil(N/4)*snp_begin),
 * sample 4i+s of a record in bits 2s..2s+1 of byte i (data.cpp:128-148).  The bytes are copied to HBM
 * (re-pitched to 128-byte rows, pad positions rewritten to the "missing" code so they contribute 0);
 * the caller may free `packed` on return. */
int fpca_create(fpca_ctx **out, const uint8_t *packed, uint64_t N, uint64_t P_g, int stand_method,
                int device, int accum);

/* Same, reading the shard straight from a .bed file: records [snp_begin, snp_begin + P_g) (P_g = 0 means
 * "to the end of the file").  N comes from the .fam as in the reference (flashpca.cpp:589). The total SNP
 * count of the file, (filesize-3)/ceil(N/4) by integer division (data.cpp:165-170), is returned in *P_total. */
int fpca_create_from_bed(fpca_ctx **out, const char *bed_path, uint64_t N, uint64_t snp_begin, uint64_t P_g,
                         int stand_method, int device, int accum, uint64_t *P_total);

/* Synthetic shard generated directly in HBM (bench / scale tests): structured
 * population model, counter-based RNG keyed by (seed, snp, sample) so that any SNP range of the same
 * (N, seed, n_pop, fst, missing_rate) matrix can be generated independently on any rank. */
int fpca_create_synthetic(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed,
                          int n_pop, double fst, double missing_rate, int stand_method, int device, int accum);

/* The same generator with the knobs that make the matrix look like array / sequencing data instead of the survey's uniform model
 * (flashpca_amd/csrc/synth.hpp): maf_model 1 = rare-variant spectrum (minor-allele frequency 0.001 + 0.499 u^3: per-SNP sd over a
 * 16x range, SNPs monomorphic in small samples); missing_model 1 = missing calls concentrated in `conc_frac` of the SNPs (10-30 %
 * of their calls; every other SNP <= 0.1 %; missing_rate is ignored); missing_model 2 = per-SNP rates log-normally distributed
 * with MEAN missing_rate and log-sd lognormal_sigma (capped at 90 %): most SNPs below the mean, a long tail of poor assays.
 * maf_model = missing_model = 0 is fpca_create_synthetic. */
typedef struct fpca_synth_model {
   int n_pop;            /* sub-populations (1..64): n_pop - 1 structured eigenvalues */
   double fst;
   double missing_rate;  /* missing_model 0: every call; 2: the mean over SNPs */
   int maf_model, missing_model;
   double conc_frac;     /* missing_model 1: fraction of SNPs with 10-30 % missing calls (e.g. 0.05) */
   double lognormal_sigma; /* missing_model 2: sd of ln(rate) over SNPs (e.g. 1.5) */
} fpca_synth_model;
int fpca_create_synthetic_model(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed, const fpca_synth_model *model,
                                int stand_method, int device, int accum);

/* In-memory matrix input: replaces RandomPCA::pca_fast(MatrixXd& X, ...) + standardise(X, method) (randompca.cpp:121-166,
 * util.cpp:24-192; the R entry flashpca(X) for a numeric matrix, flashpcaR/src/flashpca.cpp:17-93).  X is N x P_g fp64
 * column-major with leading dimension ldx, NaN = missing.  It is copied to HBM and standardised there column by column
 * exactly as standardise() does (missing -> 0 after standardisation, or -> mean for "none"; sd <= 1e-9 -> the column
 * becomes its mean); every operator / driver entry point then works on it like on a packed context (fp64 only). */
int fpca_create_dense(fpca_ctx **out, const double *X, int64_t ldx, uint64_t N, uint64_t P_g, int stand_method, int device);

void fpca_destroy(fpca_ctx *ctx);

uint64_t fpca_nsamples(const fpca_ctx *ctx);   /* Data::N */
uint64_t fpca_nsnps(const fpca_ctx *ctx);      /* shard's P_g (Data::nsnps for a 1-shard run) */
int fpca_accum(const fpca_ctx *ctx);           /* the FPCA_ACCUM_* mode in effect (AUTO resolved; may drop to FP64 at the first apply) */
/* how the exact-integer path treats the missing-call indicator for blocks of b columns (chosen from K1's counts of this
 * shard): 0 = both integer matrices on the matrix cores, 1 = the same, skipping blocks without a missing call, 2 = the
 * shard has no missing call (one matrix), 3 = one matrix on the matrix cores + the missing-call products as sparse fp64
 * gathers, 4 = hybrid: as 3 for most SNPs, while the SNPs whose missing calls would cost more to gather than their indicator row
 * costs on the matrix cores (above ~0.7 % each at 7 slices: failed assays, the tail of a log-normal rate profile) keep their
 * indicator there as a compacted sub-matrix -- taken when the per-SNP minima beat both uniform routes; -1 = not the exact-integer path.  Negative
 * FPCA_E* on error. */
int fpca_missing_mode(fpca_ctx *ctx, int b);
/* row chunks of Y whose all-reduce the built-in communicator overlaps with the computation of the next chunk (1 = one
 * all-reduce after K3; always 1 without a communicator or with a caller-supplied all-reduce hook) */
int fpca_allreduce_chunks(fpca_ctx *ctx);
/* copy the shard's packed stream back (P_g * ceil(N/4) bytes, .bed body layout) -- used by the tests to feed
 * the CPU oracle the exact matrix a synthetic context holds */
int fpca_download_packed(fpca_ctx *ctx, uint8_t *out);

/* K1.  Replaces the first-visit branch of Data::read_snp_block (data.cpp:257-322): per-SNP mean over
 * non-missing dosages, sd = sqrt(2P(1-P)) [binom2] / sqrt(P(1-P)) [binom], lookup table by raw code, and
 * trace = sum X^2 (svdwide.cpp:44-45,60-61) for this shard.  mean_sd: P_g x 2 column-major (mean | sd), may be
 * NULL; trace_out may be NULL.  Called implicitly by the first apply if the caller did not. */
int fpca_stats(fpca_ctx *ctx, double *mean_sd, double *trace_out);
/* preloaded mean/sd (projection path, data.cpp:293-297, randompca.cpp:753-788); P_g x 2 column-major */
int fpca_set_meansd(fpca_ctx *ctx, const double *mean_sd);

/* ------------------------------------------------------------------------------------------------
 * Operator.  b columns at a time; b = 1 is exactly the reference's perform_op.
 *   fpca_apply_xxt : Y = X_g X_g' B        replaces SVDWideOnline::perform_op / perform_op_mat
 *                                           (svdwide.cpp:21-68, 71-118); summed over ranks when a
 *                                           communicator / all-reduce hook is installed
 *   fpca_apply_xt  : T = X_g' B  (P_g x b)  replaces SVDWideOnline::crossprod / crossprod2 (svdwide.cpp:122-188)
 *   fpca_apply_x   : Y = X_g T   (N x b)    replaces SVDWideOnline::prod / prod3 (svdwide.cpp:193-226, 312-343)
 * B, Y: N x b; T: P_g x b; host pointers, column-major, leading dimensions ldb/ldy/ldt (>= rows). */
int fpca_apply_xxt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *Y, int64_t ldy);
int fpca_apply_xt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *T, int64_t ldt);
int fpca_apply_x(fpca_ctx *ctx, const double *T, int64_t ldt, int b, double *Y, int64_t ldy);

/* Device-resident variants: B/Y are DEVICE pointers to row-major [fpca_block_rows()][b] fp64 blocks (rows
 * >= N must be zero on input and are zero on output); T lives in the context.  The kernels are enqueued on
 * `stream` (a hipStream_t; NULL = the context's own stream) and the call returns without synchronising.
 * This is what the solver and bench.py use so that B and Y never leave HBM during the iteration. */
uint64_t fpca_block_rows(const fpca_ctx *ctx); /* N rounded up to the kernels' sample tile (512) */
int fpca_apply_xxt_dev(fpca_ctx *ctx, const double *dB, int b, double *dY, void *stream);
/* contexts' own stream handle (hipStream_t) and a device-wide synchronise */
void *fpca_stream(fpca_ctx *ctx);
int fpca_synchronize(fpca_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU (SNP-sharded; one rank per GPU).  Two transports for the sum of the N x b partial products: */
#define FPCA_UNIQUE_ID_BYTES 128
/* (a) RCCL inside the library: rank 0 makes an id, the launcher broadcasts it, every rank joins. */
int fpca_comm_unique_id(uint8_t id[FPCA_UNIQUE_ID_BYTES]);
int fpca_comm_init_rank(fpca_ctx *ctx, int nranks, int rank, const uint8_t id[FPCA_UNIQUE_ID_BYTES]);
/* (b) caller-supplied in-place sum all-reduce of `count` fp64 at device pointer `dbuf`, ordered on `stream`
 * (e.g. torch.distributed over RCCL).  Must return 0 on success.
 * CONTRACT OF EVERY HOOK BELOW (all-reduce, all-gather, reduce-scatter): a collective fails AS A WHOLE -- a non-zero return must
 * be reported on EVERY rank for the same call, and must leave no collective of that call outstanding on any rank.  The
 * library reacts to a failed collective of the row-sharded solve by making the decision common (one all-reduce of a flag through
 * this hook) and starting over on the replicated solver; a hook that fails on one rank only leaves that rank in the
 * agreement while its peers are still inside the collective that failed for it.  A transport that cannot guarantee this by
 * construction should check that all ranks have entered the SAME call before moving data (tag + count; the CLI's test transport
 * does) and fail on all of them otherwise.  What the library does on its side: every step on which the ranks must agree is
 * time-bounded (120 s) -- a rank that does not get its answer aborts the built-in RCCL communicator (ncclCommAbort; RCCL's
 * asynchronous errors are polled meanwhile), marks the transport of the context dead (every later collective returns
 * FPCA_ECOMM at once) and returns FPCA_ECOMM, for the launcher to end the job.  A hook that BLOCKS inside the caller's own
 * code cannot be bounded from here. */
typedef int (*fpca_allreduce_fn)(void *user, double *dbuf, uint64_t count, void *stream);
int fpca_set_allreduce(fpca_ctx *ctx, fpca_allreduce_fn fn, void *user);
/* ... and, optionally beside it, the caller's all-gather and reduce-scatter (sum) of fp64 on device buffers, ordered on `stream`:
 * `send` holds count_per_rank doubles (all-gather) / nranks * count_per_rank (reduce-scatter), `recv` the opposite; rank r's
 * piece sits at offset r * count_per_rank of the long buffer.  With all three callbacks installed (and fpca_set_rank) the
 * row-sharded solver issues exactly the sequence of collectives it issues over RCCL -- per row chunk, the reduce-scatter of
 * chunk i on a second stream under the K3 of chunk i + 1 -- instead of building both from the sum (twice the bytes). */
/* The all-gather must move its payload as OPAQUE 8-byte words: in the eigensolver's passes on <= 4 slices the words are byte slices
 * of the operand (4 bytes per entry of the block instead of 8), not numbers -- no conversion, no reduction, no NaN canonicalisation. */
typedef int (*fpca_allgather_fn)(void *user, const double *send, double *recv, uint64_t count_per_rank, void *stream);
typedef int (*fpca_reducescatter_fn)(void *user, const double *send, double *recv, uint64_t count_per_rank, void *stream);
int fpca_set_collectives(fpca_ctx *ctx, fpca_allgather_fn allgather, fpca_reducescatter_fn reducescatter, void *user);
/* rank / size of this context among the shards when the transport is a caller's all-reduce (fpca_comm_init_rank sets
 * them itself).  Once they are known and nranks > 1, fpca_pca ROW-SHARDS the eigensolver's N-sized work: the operator becomes
 * all-gather -> K2, K3 -> reduce-scatter (the bytes of the one all-reduce, built from the caller's all-reduce if that is all
 * there is), every rank keeps and orthogonalises N / nranks rows of the Krylov basis, and the only other collective is the
 * all-reduce of the (m+1) b^2 Gram coefficients.  Without this call a hook-only context keeps round 2's replicated solver. */
int fpca_set_rank(fpca_ctx *ctx, int nranks, int rank);
/* data-path collectives this context has issued so far: number of calls and payload bytes (tests, scale model) */
int fpca_collective_stats(const fpca_ctx *ctx, uint64_t *calls, uint64_t *bytes);
/* total SNP count over all shards (the divisor "p", randompca.cpp:183) -- defaults to the shard's P_g */
int fpca_set_total_snps(fpca_ctx *ctx, uint64_t P_total);

/* ------------------------------------------------------------------------------------------------
 * Driver.  Replaces RandomPCA::pca_fast(Data&, block_size, ndim, maxiter, tol, seed, do_loadings)
 * (randompca.cpp:168-218) with a block Krylov-Schur eigensolver whose basis stays in HBM; the projected
 * (m*b x m*b) Rayleigh-Ritz problem is solved on the host.  Convergence rule per Ritz pair is the one the
 * reference's Spectra solver applies: ||A u - theta u|| < tol * max(eps^(2/3), |theta|) for the ndim largest.
 * With fewer than three block widths of samples (N < 3 b; the reference admits ndim <= (min(N,P)-1)/2, flashpca.cpp:623-633)
 * X X' is formed from ceil(N/b) applies on the identity and decomposed directly. */
typedef struct fpca_pca_opts {
   uint32_t struct_size; /* sizeof(fpca_pca_opts) / sizeof(fpca_pca_info) of the header the CALLER was compiled against: written by */
   uint32_t info_size;   /* FPCA_PCA_OPTS_INIT from the caller's own sizeof, checked by fpca_pca (FPCA_EINVAL on a mismatch) */
   int ndim;        /* --ndim (flashpca.cpp:325 default 10) */
   int blockvec;    /* block width b (16, 32, 48 or 64); 0 = automatic: 16 for ndim <= 64, 32 for ndim <= 128, else 64 -- the
                     * narrowest block gives the shortest time to solution (pca_driver.cpp).  ndim may exceed b -- up to the
                     * reference's (min(N,P)-1)/2 --: the solver then carries ceil(ndim/b) + 1 blocks of Ritz vectors across
                     * restarts */
   int maxiter;     /* --maxiter, counted like the reference counts it: restarts of Spectra's ncv = 2 ndim + 1 Lanczos
                     * factorisation (flashpca.cpp:423-433 default 500, randompca.cpp:178 compute(maxiter, tol)).  A restart
                     * re-applies the operator to at most ndim + 1 vectors, so the reference's budget is
                     * 2 ndim + 1 + maxiter (ndim + 1) operator applications; the block solver stops after
                     * ceil(that / 16) block applies -- 16 at a time whatever the width, because a wide block needs about as
                     * many passes as a narrow one (max_applies is the explicit cap in passes) */
   double tol;      /* --tol (flashpca.cpp:440 default 1e-6) */
   int divisor;     /* FPCA_DIVISOR_* (flashpca.cpp:484 default p) */
   int do_loadings; /* --outload given */
   int max_blocks;  /* basis cap in blocks before a thick restart; 0 = automatic */
   int verbose;
   uint64_t seed;   /* start block seed; the reference ignores --seed for PCA (randompca.cpp:168-218) */
   int replicated_solver; /* multi-GPU only: 1 = every rank keeps whole copies of the Krylov basis and repeats the
                     * orthogonalisation (round 2's scheme: one all-reduce per apply); 0 (default) = row-sharded, see fpca_set_rank */
   int max_applies; /* hard cap on block applies, overriding the budget derived from maxiter; 0 = none.  Must allow at least
                     * ceil(ndim / b) of them (FPCA_EINVAL otherwise: fewer basis vectors than wanted pairs) */
   int mixed;       /* exact-integer arithmetic only.  0 = automatic (on), 1 = on, -1 = off.  On: the Krylov passes run on
                     * `cheap_slices` byte slices of the fp64 operand instead of the context's S (4 slices = a 30-bit operand:
                     * ~1e-9 of operator noise, nothing accumulating -- a pass costs about 0.6 of an exact one), and when the
                     * reference's convergence rule holds on those passes the Ritz vectors are put through the EXACT operator
                     * (all S slices) once: the eigenvalues returned are Rayleigh quotients of the exact operator and the rule
                     * ||A u - theta u|| < tol max(eps^(2/3), |theta|) (randompca.cpp:173-178) is judged on exact residuals.  If
                     * it does not hold there, the iteration continues from those vectors with exact passes only. */
   int cheap_slices; /* 0 = automatic (4); 3..S-1 */
   int partial_rows; /* multi-GPU only.  1 = this rank writes ONLY ITS OWN ROWS of U and Px (still N x ndim, leading dimension N):
                     * its row slice of the row-sharded solver, or an even share of the rows under the replicated one -- no gather
                     * of the eigenvector blocks, 1 / nranks of the PCIe traffic per rank.  For callers whose ranks share the
                     * output memory (flashpca --gpus: one mmap'ed region) or gather the slices themselves
                     * (fpca_pca_row_ranges).  0 (default) = every rank that passes U / Px gets all N rows. */
} fpca_pca_opts;

/* fpca_pca_info.solver_path: which layout of the eigensolver's N-sized work the solve ran on */
#define FPCA_SOLVER_SINGLE 0              /* one rank */
#define FPCA_SOLVER_ROWSHARD 1            /* row-sharded: all-gather -> K2, K3 -> reduce-scatter per apply (default for nranks > 1) */
#define FPCA_SOLVER_REPLICATED 2          /* replicated, as asked (replicated_solver = 1, or an all-reduce hook without fpca_set_rank) */
#define FPCA_SOLVER_REPLICATED_SELFTEST 3 /* replicated after the self-test of the row-sharded exchange failed on some rank */
#define FPCA_SOLVER_REPLICATED_FAILURE 4  /* replicated, started over after a collective of the row-sharded solve reported a failure */

typedef struct fpca_pca_info {
   int converged;         /* all ndim pairs met the rule (mixed precision: on residuals of the exact operator) */
   int block_applies;     /* passes over the packed matrix (each = b single-vector operator applications) */
   int vector_ops;        /* block_applies * b  == the reference's nops unit (svdwide.cpp:67) */
   int restarts;          /* thick restarts */
   int blockvec;          /* b actually used */
   double trace;          /* sum X^2 / div (randompca.cpp:205) */
   double max_residual;   /* max_i ||A u_i - theta_i u_i|| / max(eps^(2/3), theta_i) at exit */
   double seconds_apply;  /* device time in the operator (K2+K3+all-reduce) */
   double seconds_ortho;  /* device time in basis orthogonalisation / Ritz rotation */
   double seconds_host;   /* host time in the projected eigenproblem */
   double seconds_total;
   double seconds_download; /* U and Px to the caller's memory (pinned, pipelined; Px = U sqrt(d) fused into the host side) */
   double seconds_post;     /* loadings (one K2 pass per block of eigenvectors) + mean/sd download */
   int cheap_applies;       /* of block_applies: passes on cheap_slices byte slices (0 when mixed precision is off) */
   int cheap_slices;        /* slices of those passes (0: none ran) */
   double seconds_exact;    /* of seconds_apply: the exact passes (verification and whatever followed it) */
   int solver_path;         /* FPCA_SOLVER_*: identical on every rank (a demotion is agreed through one all-reduce of a flag) */
} fpca_pca_info;

/* Defaults of the reference CLI (flashpca.cpp:276-484) + the CALLER's struct sizes, so that fpca_pca can refuse a caller built
 * against another revision of this header: never more than opts_size bytes are written.  C callers use the macro. */
void fpca_pca_init_opts(fpca_pca_opts *opts, size_t opts_size, size_t info_size);
#define FPCA_PCA_OPTS_INIT(o) fpca_pca_init_opts((o), sizeof(*(o)), sizeof(fpca_pca_info))
/* (Revision 4 removed fpca_pca_default_opts, which wrote the LIBRARY's sizeof into the caller's struct: a binary built against an
 * older header now fails to resolve the symbol at load time instead of being overrun by 8 bytes.) */
/* partial_rows = 1: the row ranges [begin, end) of U / Px this rank writes -- ranges[2 i], ranges[2 i + 1], at most max_ranges of
 * them are stored, the count is returned (<= 4; negative FPCA_E* on error).  After an fpca_pca call the answer reflects the
 * layout that call ended on (a demotion to the replicated solver changes the ranges). */
int fpca_pca_row_ranges(fpca_ctx *ctx, const fpca_pca_opts *opts, uint64_t *ranges, int max_ranges);
/* Outputs (host, column-major, caller-allocated; any may be NULL):
 *   U  N x ndim eigenvectors (unit 2-norm, sign arbitrary like Spectra's) ..... RandomPCA::U
 *   d  ndim eigenvalues of X X'/div, descending .............................. RandomPCA::d
 *   Px N x ndim = U diag(sqrt(d)) ............................................. RandomPCA::Px
 *   pve ndim = d / trace ....................................................... RandomPCA::pve
 *   V  P_g x ndim loadings rows of this shard = X_g' U diag(1/sqrt(d))/sqrt(div)  RandomPCA::V (randompca.cpp:191-204)
 *   mean_sd P_g x 2 ............................................................. RandomPCA::X_meansd */
int fpca_pca(fpca_ctx *ctx, const fpca_pca_opts *opts, double *U, double *d, double *Px, double *pve,
             double *V, double *mean_sd, fpca_pca_info *info);

/* Replaces RandomPCA::check(Data&, block_size, evec, eval) (randompca.cpp:663-703):
 * err[j] = || X X' u_j / div - u_j lambda_j ||^2, mse = sum(err)/(N k), rmse = sqrt(mse). */
int fpca_check(fpca_ctx *ctx, const double *evec, int64_t ldu, const double *eval, int k, int divisor,
               double *err, double *mse, double *rmse);

#ifdef __cplusplus
}
#endif
#endif /* FPCA_H */
