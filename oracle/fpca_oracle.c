/*
 * oracle/fpca_oracle.c -- CPU restatement of flashpca's PCA hot path.  TEST INFRASTRUCTURE ONLY.
 * See fpca_oracle.h for the rules on who may call this and for the parity-pinning statement.
 *
 * What is restated (reference file:line, relative to the reference root):
 *   decode of the PLINK 2-bit stream ........ data.cpp:65-148, data.h:42-45
 *   sizing / prepare ......................... data.cpp:150-206
 *   per-SNP mean/sd/LUT + dense block fill ... data.cpp:215-335
 *   block table and matrix-free operator ..... svdwide.h:51-73, svdwide.cpp:21-68, 71-118, 122-153, 193-226
 *   implicitly restarted Lanczos ............. Spectra v0.8.1 SymEigsSolver (third party, not in the
 *                                              reference tree; pinned by Dockerfile:19-20; used at
 *                                              randompca.cpp:173-178) -- restated from its published algorithm
 *   post-processing .......................... randompca.cpp:180-208
 *   check mode ............................... randompca.cpp:663-703
 *   CLI block-size heuristic ................. flashpca.cpp:636-686
 *   number formatting ........................ util.h:69-108
 *
 * Plain C99, no third-party code.  The two GEMVs of svdwide.cpp:42-43 are plain loops over the
 * column-major N x bs block (what Eigen's GEMV does, minus its vectorisation details); build with
 * -O3 -march=native -ffast-math as the reference Makefile:29,40-41 does.  nthreads > 1 is NOT what
 * the shipped reference does (it is single-threaded); it exists so that
 * bench.py can also quote a generous all-cores CPU number.
 */
#include "fpca_oracle.h"

#include <errno.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PACK_DENSITY 4 /* data.h:24 */
#define PLINK_NA 3     /* data.h:25 */
#define VAR_TOL 1e-9   /* util.h:33 */

/* ------------------------------------------------------------------------------------------ */
/* decode (data.cpp:65-148)                                                                    */

/* data.cpp:65-126: each byte holds 4 genotypes, sample 4i+s in bits 2s..2s+1.  Field value 1
 * (binary 01) is missing -> 3; otherwise dosage = !bit0 + !bit1, i.e. 00->2, 10->1, 11->0. */
void orc_decode_plink(unsigned char *out, const unsigned char *in, unsigned int n)
{
   for (unsigned int i = 0; i < n; i++) {
      unsigned char byte = in[i];
      for (unsigned int s = 0; s < PACK_DENSITY; s++) {
         unsigned char g = (unsigned char)((byte >> (2 * s)) & 3);
         unsigned char v;
         if (g == 1)
            v = PLINK_NA;
         else
            v = (unsigned char)(!(g & 1) + !(g >> 1));
         out[PACK_DENSITY * i + s] = v;
      }
   }
}

/* data.cpp:128-148: raw 2-bit fields, no mapping */
void orc_decode_plink_simple(unsigned char *out, const unsigned char *in, unsigned int n)
{
   for (unsigned int i = 0; i < n; i++) {
      unsigned char byte = in[i];
      out[PACK_DENSITY * i + 0] = (unsigned char)(byte & 3);
      out[PACK_DENSITY * i + 1] = (unsigned char)((byte >> 2) & 3);
      out[PACK_DENSITY * i + 2] = (unsigned char)((byte >> 4) & 3);
      out[PACK_DENSITY * i + 3] = (unsigned char)((byte >> 6) & 3);
   }
}

/* ------------------------------------------------------------------------------------------ */
/* Data (data.h:60-101, data.cpp:150-335)                                                      */

struct orc_data {
   uint64_t N, np, nsnps, len;
   int stand_method;
   FILE *fp;                    /* file-backed ... */
   const unsigned char *mem;    /* ... or memory-backed */
   unsigned char *tmp, *tmp2;   /* data.cpp:190-193 */
   unsigned char *visited;      /* data.cpp:196 */
   double *meansd;              /* X_meansd, P x 2 column-major (data.cpp:197) */
   double *lut;                 /* scaled_geno_lookup, 4 x P column-major (data.cpp:199) */
   int use_preloaded_maf;       /* data.h:79 */
};

static orc_data *orc_alloc_common(uint64_t N, uint64_t nsnps, int stand_method)
{
   orc_data *d = (orc_data *)calloc(1, sizeof(orc_data));
   if (!d) return NULL;
   d->N = N;
   d->np = (N + PACK_DENSITY - 1) / PACK_DENSITY; /* ceil(N/4), data.cpp:169 */
   d->nsnps = nsnps;
   d->stand_method = stand_method;
   d->tmp = (unsigned char *)malloc(d->np ? d->np : 1);
   d->tmp2 = (unsigned char *)malloc(d->np ? d->np * PACK_DENSITY : 1);
   d->visited = (unsigned char *)calloc(nsnps ? nsnps : 1, 1);
   d->meansd = (double *)calloc(nsnps ? 2 * nsnps : 1, sizeof(double));
   d->lut = (double *)calloc(nsnps ? 4 * nsnps : 1, sizeof(double));
   return d;
}

/* data.cpp:150-176 (get_size) + 179-206 (prepare): len = filesize-3, np = ceil(N/4),
 * nsnps = len/np by integer division; the 3 header bytes are skipped, never validated. */
orc_data *orc_open_file(const char *bed_path, uint64_t N, int stand_method, char *err, int errlen)
{
   FILE *fp = fopen(bed_path, "rb");
   if (!fp) {
      if (err) snprintf(err, errlen, "[Data::read_bed] Error reading file %s, error %s", bed_path, strerror(errno));
      return NULL;
   }
   fseek(fp, 0, SEEK_END);
   long long sz = ftell(fp);
   if (N == 0 || sz < 3) {
      if (err) snprintf(err, errlen, "empty input (N=%llu, size=%lld)", (unsigned long long)N, sz);
      fclose(fp);
      return NULL;
   }
   uint64_t len = (uint64_t)sz - 3;
   uint64_t np = (N + PACK_DENSITY - 1) / PACK_DENSITY;
   uint64_t nsnps = len / np;
   orc_data *d = orc_alloc_common(N, nsnps, stand_method);
   d->len = len;
   d->fp = fp;
   return d;
}

orc_data *orc_open_mem(const unsigned char *packed, uint64_t N, uint64_t P, int stand_method)
{
   orc_data *d = orc_alloc_common(N, P, stand_method);
   d->len = d->np * P;
   d->mem = packed;
   return d;
}

void orc_close(orc_data *d)
{
   if (!d) return;
   if (d->fp) fclose(d->fp);
   free(d->tmp);
   free(d->tmp2);
   free(d->visited);
   free(d->meansd);
   free(d->lut);
   free(d);
}

uint64_t orc_N(const orc_data *d) { return d->N; }
uint64_t orc_nsnps(const orc_data *d) { return d->nsnps; }
uint64_t orc_np(const orc_data *d) { return d->np; }
const double *orc_meansd(const orc_data *d) { return d->meansd; }
const double *orc_lookup(const orc_data *d) { return d->lut; }

void orc_set_preloaded_meansd(orc_data *d, const double *meansd)
{
   memcpy(d->meansd, meansd, sizeof(double) * 2 * d->nsnps);
   d->use_preloaded_maf = 1;
}

/* data.cpp:215-335.  For every SNP k in [start, stop]: read np bytes; on the first visit compute
 * mean over non-missing dosages, P = mean/2, sd by the standardisation method, store mean/sd, and
 * (if sd > VAR_TOL) the 4-entry table indexed by RAW code: [3]->(0-mean)/sd, [2]->(1-mean)/sd,
 * [0]->(2-mean)/sd, [1]->0; otherwise the table stays zero.  Every visit: X(i, j) = table[code_i]. */
static int read_snp_block_scratch(orc_data *d, uint32_t start, uint32_t stop, double *X, unsigned char *tmp, unsigned char *tmp2);

int orc_read_snp_block(orc_data *d, uint32_t start, uint32_t stop, double *X)
{
   return read_snp_block_scratch(d, start, stop, X, d->tmp, d->tmp2);
}

/* the body of read_snp_block with the two scratch rows (data.cpp:190-193) passed in, and positioned reads instead of
 * seek + read, so that several threads can fill blocks of DISJOINT SNP ranges at once (nthreads > 1 only; disjoint SNPs
 * touch disjoint entries of visited / meansd / lut) */
static int read_snp_block_scratch(orc_data *d, uint32_t start, uint32_t stop, double *X, unsigned char *tmp, unsigned char *tmp2)
{
   const uint64_t N = d->N, np = d->np, P = d->nsnps;
   uint32_t bs = stop - start + 1;
   for (uint32_t j = 0; j < bs; j++) {
      uint64_t k = (uint64_t)start + j;
      if (d->fp) {
         /* data.cpp:218 (seek to 3 + np*start) + :250 (read np bytes per SNP) */
         if (pread(fileno(d->fp), tmp, np, (off_t)(3 + np * k)) != (ssize_t)np) return -2;
      } else {
         memcpy(tmp, d->mem + np * k, np);
      }
      if (!d->visited[k]) { /* data.cpp:257-322 */
         double snp_avg = 0, sd = 0;
         if (!d->use_preloaded_maf) {
            orc_decode_plink(tmp2, tmp, (unsigned int)np);
            uint64_t ngood = 0;
            for (uint64_t i = 0; i < N; i++) {
               if (tmp2[i] != PLINK_NA) {
                  snp_avg += (double)tmp2[i];
                  ngood++;
               }
            }
            snp_avg /= (double)ngood;
            double pp = snp_avg / 2.0;
            if (d->stand_method == ORC_STANDARDISE_BINOM)
               sd = sqrt(pp * (1 - pp));
            else if (d->stand_method == ORC_STANDARDISE_BINOM2)
               sd = sqrt(2.0 * pp * (1 - pp));
            else
               return -3;
            d->meansd[k] = snp_avg;
            d->meansd[P + k] = sd;
         } else {
            snp_avg = d->meansd[k];
            sd = d->meansd[P + k];
         }
         double *lut = d->lut + 4 * k;
         if (sd > VAR_TOL) {
            lut[3] = (0 - snp_avg) / sd;
            lut[2] = (1 - snp_avg) / sd;
            lut[0] = (2 - snp_avg) / sd;
            lut[1] = 0; /* missing -> imputed to the mean, i.e. 0 after standardisation */
         }
         d->visited[k] = 1;
      }
      orc_decode_plink_simple(tmp2, tmp, (unsigned int)np); /* data.cpp:328 */
      const double *lut = d->lut + 4 * k;
      double *col = X + (uint64_t)j * N;
      for (uint64_t i = 0; i < N; i++) col[i] = lut[tmp2[i]]; /* data.cpp:330-333 */
   }
   return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SVDWideOnline (svdwide.h:40-107, svdwide.cpp)                                               */

struct orc_op {
   orc_data *dat;
   uint64_t n, p;
   uint32_t nblocks, block_size;
   uint32_t *start, *stop;
   uint32_t nops;
   int trace_done;
   double trace;
   double *X; /* dat.X, N x block_size */
   double *t; /* block_size scratch */
   int nthreads;
   void *mt;       /* nthreads > 1: per-thread blocks (orc_mt[nthreads]) */
   uint32_t mt_bs; /* SNPs per per-thread sub-block */
};

static void mt_free(orc_op *op);

/* svdwide.h:51-73 */
orc_op *orc_op_new(orc_data *d, uint32_t block_size, int nthreads)
{
   orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
   op->dat = d;
   op->n = d->N;
   op->p = d->nsnps;
   if (block_size > op->p) block_size = (uint32_t)op->p; /* flashpca.cpp:686 */
   if (block_size < 1) block_size = 1;
   op->block_size = block_size;
   op->nblocks = (uint32_t)((op->p + block_size - 1) / block_size);
   op->start = (uint32_t *)malloc(sizeof(uint32_t) * (op->nblocks ? op->nblocks : 1));
   op->stop = (uint32_t *)malloc(sizeof(uint32_t) * (op->nblocks ? op->nblocks : 1));
   for (uint32_t i = 0; i < op->nblocks; i++) {
      op->start[i] = i * block_size;
      uint64_t s = (uint64_t)op->start[i] + block_size - 1;
      op->stop[i] = (uint32_t)(s >= op->p ? op->p - 1 : s);
   }
   op->nops = 1;
   op->trace = 0;
   op->trace_done = 0;
   op->nthreads = nthreads < 1 ? 1 : nthreads;
   op->X = op->nthreads > 1 ? NULL : (double *)malloc(sizeof(double) * op->n * block_size); /* threads own their blocks */
   op->t = (double *)malloc(sizeof(double) * block_size * 64);
   return op;
}

void orc_op_free(orc_op *op)
{
   if (!op) return;
   mt_free(op);
   free(op->start);
   free(op->stop);
   free(op->X);
   free(op->t);
   free(op);
}

double orc_op_trace(const orc_op *op) { return op->trace; }
uint32_t orc_op_nops(const orc_op *op) { return op->nops; }
uint32_t orc_op_nblocks(const orc_op *op) { return op->nblocks; }

/* t = Xb' x  (Eigen GEMV, svdwide.cpp:42-43 inner product) */
static void gemv_t(const double *X, uint64_t n, uint32_t bs, const double *x, double *t)
{
   for (uint32_t j = 0; j < bs; j++) {
      const double *col = X + (uint64_t)j * n;
      double s = 0;
      for (uint64_t i = 0; i < n; i++) s += col[i] * x[i];
      t[j] = s;
   }
}

/* y (+)= Xb t (Eigen GEMV, svdwide.cpp:42-43 / 58-59 outer product); rows in chunks that stay in L1 while the bs columns
 * stream past, as a blocked GEMV does */
static void gemv_n_acc(const double *X, uint64_t n, uint32_t bs, const double *t, double *y, int overwrite)
{
   const uint64_t RC = 2048;
   for (uint64_t lo = 0; lo < n; lo += RC) {
      const uint64_t hi = lo + RC < n ? lo + RC : n;
      if (overwrite)
         for (uint64_t i = lo; i < hi; i++) y[i] = 0;
      for (uint32_t j = 0; j < bs; j++) {
         const double *col = X + (uint64_t)j * n;
         const double tj = t[j];
         for (uint64_t i = lo; i < hi; i++) y[i] += col[i] * tj;
      }
   }
}

static double sumsq(const double *X, uint64_t cnt)
{
   double s = 0;
   for (uint64_t i = 0; i < cnt; i++) s += X[i] * X[i];
   return s;
}

/* ---- nthreads > 1 (NOT what the shipped reference does; the generous all-cores number of bench.py and the full-size
 * oracle legs of the GPU tests): the SNP blocks of the block table are cut into sub-blocks of mt_bs SNPs, dealt to the
 * threads; every thread owns a dense N x mt_bs block, the two scratch rows and a partial y; partial y's are summed over
 * threads by row ranges at the end.  Same arithmetic per SNP as the serial path, different summation order over SNPs. */
typedef struct {
   double *X, *t, *y;
   unsigned char *tmp, *tmp2;
} orc_mt;

static void mt_free(orc_op *op)
{
   orc_mt *m = (orc_mt *)op->mt;
   if (!m) return;
   for (int i = 0; i < op->nthreads; i++) {
      free(m[i].X);
      free(m[i].t);
      free(m[i].y);
      free(m[i].tmp);
      free(m[i].tmp2);
   }
   free(m);
   op->mt = NULL;
}

static orc_mt *mt_get(orc_op *op)
{
   if (op->mt) return (orc_mt *)op->mt;
   /* per-thread dense block <= 4 MB where one SNP column allows it (it is written once and read twice per visit) */
   uint64_t bs = (4u << 20) / (8 * (op->n ? op->n : 1));
   if (bs < 1) bs = 1;
   if (bs > op->block_size) bs = op->block_size;
   uint64_t per = (op->p + (uint64_t)op->nthreads - 1) / (uint64_t)op->nthreads; /* at least one sub-block per thread */
   if (per >= 1 && bs > per) bs = per;
   op->mt_bs = (uint32_t)bs;
   op->mt = calloc((size_t)op->nthreads, sizeof(orc_mt));
   return (orc_mt *)op->mt;
}

static void mt_thread_alloc(orc_op *op, orc_mt *m)
{
   if (m->X) return;
   m->X = (double *)malloc(sizeof(double) * op->n * op->mt_bs);
   m->t = (double *)malloc(sizeof(double) * op->mt_bs);
   m->y = (double *)malloc(sizeof(double) * op->n);
   m->tmp = (unsigned char *)malloc(op->dat->np ? op->dat->np : 1);
   m->tmp2 = (unsigned char *)malloc(op->dat->np ? op->dat->np * PACK_DENSITY : 1);
}

/* what: 0 = y = X X' x (perform_op), 1 = out[P] = X' x (crossprod), 2 = y = X v (prod) */
static void op_mt(orc_op *op, int what, const double *in, double *out)
{
   orc_mt *mt = mt_get(op);
   const uint64_t n = op->n, p = op->p;
   const uint32_t bs = op->mt_bs;
   const int64_t nsub = (int64_t)((p + bs - 1) / bs);
   const int want_trace = (what == 0 && !op->trace_done);
   double trace = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(op->nthreads) reduction(+ : trace)
#endif
   {
#ifdef _OPENMP
      const int id = omp_get_thread_num(), nt = omp_get_num_threads();
#else
      const int id = 0, nt = 1;
#endif
      orc_mt *m = &mt[id];
      mt_thread_alloc(op, m);
      if (what != 1)
         for (uint64_t i = 0; i < n; i++) m->y[i] = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
      for (int64_t sb = 0; sb < nsub; sb++) {
         const uint32_t s0 = (uint32_t)(sb * bs);
         const uint32_t s1 = (uint32_t)((uint64_t)s0 + bs - 1 >= p ? p - 1 : s0 + bs - 1);
         const uint32_t w = s1 - s0 + 1;
         read_snp_block_scratch(op->dat, s0, s1, m->X, m->tmp, m->tmp2);
         if (what == 0) {
            gemv_t(m->X, n, w, in, m->t);
            gemv_n_acc(m->X, n, w, m->t, m->y, 0);
            if (want_trace) trace += sumsq(m->X, n * w);
         } else if (what == 1) {
            gemv_t(m->X, n, w, in, out + s0);
         } else {
            gemv_n_acc(m->X, n, w, in + s0, m->y, 0);
         }
      }
      /* implicit barrier above; every thread sums its row range over the threads' partials */
      if (what != 1) {
         const uint64_t lo = n * (uint64_t)id / (uint64_t)nt, hi = n * (uint64_t)(id + 1) / (uint64_t)nt;
         for (uint64_t i = lo; i < hi; i++) out[i] = 0;
         for (int q = 0; q < nt; q++) {
            const double *yq = mt[q].y;
            if (!yq) continue;
            for (uint64_t i = lo; i < hi; i++) out[i] += yq[i];
         }
      }
   }
   if (want_trace) {
      op->trace = trace;
      op->trace_done = 1;
   }
   op->nops++;
}

/* svdwide.cpp:21-68: y = sum_b X_b (X_b' x); trace accumulated on the first call only; block 0 is
 * re-read only when nblocks > 1 or on the very first op. */
void orc_perform_op(orc_op *op, const double *x_in, double *y_out)
{
   if (op->nthreads > 1) {
      op_mt(op, 0, x_in, y_out);
      return;
   }
   uint32_t bs = op->stop[0] - op->start[0] + 1;
   if (op->nblocks > 1 || op->nops == 1) orc_read_snp_block(op->dat, op->start[0], op->stop[0], op->X);
   gemv_t(op->X, op->n, bs, x_in, op->t);
   gemv_n_acc(op->X, op->n, bs, op->t, y_out, 1);
   if (!op->trace_done) op->trace = sumsq(op->X, op->n * bs);
   for (uint32_t k = 1; k < op->nblocks; k++) {
      bs = op->stop[k] - op->start[k] + 1;
      orc_read_snp_block(op->dat, op->start[k], op->stop[k], op->X);
      gemv_t(op->X, op->n, bs, x_in, op->t);
      gemv_n_acc(op->X, op->n, bs, op->t, y_out, 0);
      if (!op->trace_done) op->trace += sumsq(op->X, op->n * bs);
   }
   if (!op->trace_done) op->trace_done = 1;
   op->nops++;
}

/* svdwide.cpp:71-118: same with a matrix right-hand side (ncols columns, column-major) */
void orc_perform_op_mat(orc_op *op, const double *Xin, int ncols, double *Y)
{
   if (op->nthreads > 1) { /* column by column through the threaded operator (one nops tick, like the serial form) */
      const uint32_t nops0 = op->nops;
      for (int c = 0; c < ncols; c++) op_mt(op, 0, Xin + (uint64_t)c * op->n, Y + (uint64_t)c * op->n);
      op->nops = nops0 + 1;
      return;
   }
   double *t = (double *)malloc(sizeof(double) * op->block_size);
   for (uint32_t k = 0; k < op->nblocks; k++) {
      uint32_t bs = op->stop[k] - op->start[k] + 1;
      if (k > 0 || op->nblocks > 1 || op->nops == 1) orc_read_snp_block(op->dat, op->start[k], op->stop[k], op->X);
      for (int c = 0; c < ncols; c++) {
         gemv_t(op->X, op->n, bs, Xin + (uint64_t)c * op->n, t);
         gemv_n_acc(op->X, op->n, bs, t, Y + (uint64_t)c * op->n, k == 0);
      }
      if (!op->trace_done) op->trace = (k == 0 ? 0 : op->trace) + sumsq(op->X, op->n * bs);
   }
   if (!op->trace_done) op->trace_done = 1;
   op->nops++;
   free(t);
}

/* svdwide.cpp:122-153: y[P] = X' x, block by block */
void orc_crossprod(orc_op *op, const double *x_in, double *y_out)
{
   if (op->nthreads > 1) {
      op_mt(op, 1, x_in, y_out);
      return;
   }
   for (uint32_t k = 0; k < op->nblocks; k++) {
      uint32_t bs = op->stop[k] - op->start[k] + 1;
      orc_read_snp_block(op->dat, op->start[k], op->stop[k], op->X);
      gemv_t(op->X, op->n, bs, x_in, y_out + op->start[k]);
   }
   op->nops++;
}

/* svdwide.cpp:193-226: y[N] = X v */
void orc_prod(orc_op *op, const double *v_in, double *y_out)
{
   if (op->nthreads > 1) {
      op_mt(op, 2, v_in, y_out);
      return;
   }
   for (uint32_t k = 0; k < op->nblocks; k++) {
      uint32_t bs = op->stop[k] - op->start[k] + 1;
      orc_read_snp_block(op->dat, op->start[k], op->stop[k], op->X);
      gemv_n_acc(op->X, op->n, bs, v_in + op->start[k], y_out, k == 0);
   }
   op->nops++;
}

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers for the Lanczos driver                                                  */

static double dot(const double *a, const double *b, uint64_t n)
{
   double s = 0;
   for (uint64_t i = 0; i < n; i++) s += a[i] * b[i];
   return s;
}
static double nrm2(const double *a, uint64_t n) { return sqrt(dot(a, a, n)); }

/* Eigen-decomposition of a symmetric tridiagonal matrix (diag d[m], off-diagonal e[m-1]) by the
 * implicit-shift QL iteration; Z (m x m, column-major) receives the eigenvectors.  This plays the
 * role of Spectra's TridiagEigen inside retrieve_ritzpair(). */
static int tridiag_ql(int m, double *d, double *e_in, double *Z)
{
   double *e = (double *)malloc(sizeof(double) * (m + 1));
   for (int i = 0; i < m - 1; i++) e[i] = e_in[i];
   e[m - 1] = 0;
   for (int i = 0; i < m * m; i++) Z[i] = 0;
   for (int i = 0; i < m; i++) Z[i + (size_t)i * m] = 1;
   for (int l = 0; l < m; l++) {
      int iter = 0, mm;
      do {
         for (mm = l; mm < m - 1; mm++) {
            double dd = fabs(d[mm]) + fabs(d[mm + 1]);
            if (fabs(e[mm]) <= DBL_EPSILON * dd) break;
         }
         if (mm != l) {
            if (iter++ == 200) {
               free(e);
               return -1;
            }
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = hypot(g, 1.0);
            g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
            double s = 1, c = 1, p = 0;
            int i;
            for (i = mm - 1; i >= l; i--) {
               double f = s * e[i], b = c * e[i];
               r = hypot(f, g);
               e[i + 1] = r;
               if (r == 0.0) {
                  d[i + 1] -= p;
                  e[mm] = 0;
                  break;
               }
               s = f / r;
               c = g / r;
               g = d[i + 1] - p;
               r = (d[i] - g) * s + 2.0 * c * b;
               p = s * r;
               d[i + 1] = g + p;
               g = c * r - b;
               for (int k = 0; k < m; k++) {
                  double *zk1 = &Z[k + (size_t)(i + 1) * m], *zk0 = &Z[k + (size_t)i * m];
                  f = *zk1;
                  *zk1 = s * (*zk0) + c * f;
                  *zk0 = c * (*zk0) - s * f;
               }
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[mm] = 0;
         }
      } while (mm != l);
   }
   free(e);
   return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Spectra v0.8.1 SymEigsSolver<double, LARGEST_ALGE, Op> restated (algorithm per its published  */
/* source).  State names follow the upstream members for traceability.*/

typedef struct {
   orc_op *op;
   uint64_t n;
   int nev, ncv;
   double *V;  /* n x ncv */
   double *H;  /* ncv x ncv, column-major, symmetric tridiagonal after factorisation */
   double *f;  /* n */
   double *ritz_val, *ritz_est, *ritz_vec; /* ncv, ncv, ncv x nev */
   unsigned char *ritz_conv;
   double near_0, eps, eps23;
   long rng_state;
} irlm;

/* Spectra's SimpleRandom: Park-Miller minimal-standard LCG (a=16807, m=2^31-1), values mapped to
 * [-0.5, 0.5).  Seed 0 is replaced by 1. */
static void lcg_seed(irlm *s, unsigned long seed)
{
   const unsigned long mx = 2147483647UL;
   s->rng_state = seed ? (long)(seed & mx) : 1;
}
static double lcg_next(irlm *s)
{
   const unsigned long long a = 16807ULL, mx = 2147483647ULL;
   s->rng_state = (long)((a * (unsigned long long)s->rng_state) % mx);
   return (double)s->rng_state / (double)mx - 0.5;
}

#define HH(i, j) s->H[(i) + (size_t)(j) * s->ncv]

/* Lanczos expansion from column from_k up to to_m (exclusive): one operator application per new
 * column, three-term recurrence, then re-orthogonalisation against ALL previous columns with up to
 * 5 correction passes while max|V'f| > eps * ||f||. */
static void irlm_factorize_from(irlm *s, int from_k, int to_m, const double *fk)
{
   if (to_m <= from_k) return;
   const uint64_t n = s->n;
   const int ncv = s->ncv;
   memcpy(s->f, fk, sizeof(double) * n);
   double *w = (double *)malloc(sizeof(double) * n);
   double *Vf = (double *)malloc(sizeof(double) * ncv);
   double beta = nrm2(s->f, n);
   /* keep the leading from_k x from_k block of H, zero the rest */
   for (int j = from_k; j < ncv; j++)
      for (int i = 0; i < ncv; i++) HH(i, j) = 0;
   for (int j = 0; j < from_k; j++)
      for (int i = from_k; i < ncv; i++) HH(i, j) = 0;

   for (int i = from_k; i <= to_m - 1; i++) {
      int restart = 0;
      if (beta < s->near_0) {
         /* invariant subspace: new random residual orthogonal to V[:, :i] */
         lcg_seed(s, (unsigned long)(2 * i));
         for (uint64_t r = 0; r < n; r++) s->f[r] = lcg_next(s);
         for (int c = 0; c < i; c++) Vf[c] = dot(s->V + (size_t)c * n, s->f, n);
         for (int c = 0; c < i; c++) {
            const double *vc = s->V + (size_t)c * n;
            for (uint64_t r = 0; r < n; r++) s->f[r] -= vc[r] * Vf[c];
         }
         beta = nrm2(s->f, n);
         restart = 1;
      }
      double *vi = s->V + (size_t)i * n;
      for (uint64_t r = 0; r < n; r++) vi[r] = s->f[r] / beta;
      HH(i, i - 1) = restart ? 0.0 : beta;
      orc_perform_op(s->op, vi, w); /* the only place the operator is applied */
      double Hii = dot(vi, w, n);
      HH(i - 1, i) = HH(i, i - 1);
      HH(i, i) = Hii;
      const double *vim1 = s->V + (size_t)(i - 1) * n;
      if (restart)
         for (uint64_t r = 0; r < n; r++) s->f[r] = w[r] - Hii * vi[r];
      else {
         const double hb = HH(i, i - 1);
         for (uint64_t r = 0; r < n; r++) s->f[r] = w[r] - hb * vim1[r] - Hii * vi[r];
      }
      beta = nrm2(s->f, n);
      const int i1 = i + 1;
      double ortho_err = 0;
      for (int c = 0; c < i1; c++) {
         Vf[c] = dot(s->V + (size_t)c * n, s->f, n);
         if (fabs(Vf[c]) > ortho_err) ortho_err = fabs(Vf[c]);
      }
      int count = 0;
      while (count < 5 && ortho_err > s->eps * beta) {
         if (beta < s->near_0) {
            for (uint64_t r = 0; r < n; r++) s->f[r] = 0;
            beta = 0;
            break;
         }
         for (int c = 0; c < i1; c++) {
            const double *vc = s->V + (size_t)c * n;
            const double a = Vf[c];
            for (uint64_t r = 0; r < n; r++) s->f[r] -= vc[r] * a;
         }
         HH(i - 1, i) += Vf[i - 1];
         HH(i, i - 1) = HH(i - 1, i);
         HH(i, i) += Vf[i];
         beta = nrm2(s->f, n);
         ortho_err = 0;
         for (int c = 0; c < i1; c++) {
            Vf[c] = dot(s->V + (size_t)c * n, s->f, n);
            if (fabs(Vf[c]) > ortho_err) ortho_err = fabs(Vf[c]);
         }
         count++;
      }
   }
   free(w);
   free(Vf);
}

/* Ritz pairs of the ncv x ncv tridiagonal H, sorted by LARGEST_ALGE (descending value);
 * ritz_est = last row of the eigenvector matrix (the residual estimate factors). */
static void irlm_retrieve_ritzpair(irlm *s)
{
   const int m = s->ncv;
   double *d = (double *)malloc(sizeof(double) * m);
   double *e = (double *)malloc(sizeof(double) * m);
   double *Z = (double *)malloc(sizeof(double) * m * m);
   int *idx = (int *)malloc(sizeof(int) * m);
   for (int i = 0; i < m; i++) d[i] = HH(i, i);
   for (int i = 0; i < m - 1; i++) e[i] = HH(i + 1, i);
   tridiag_ql(m, d, e, Z);
   for (int i = 0; i < m; i++) idx[i] = i;
   for (int i = 1; i < m; i++) { /* insertion sort, descending */
      int t = idx[i], j = i - 1;
      while (j >= 0 && d[idx[j]] < d[t]) {
         idx[j + 1] = idx[j];
         j--;
      }
      idx[j + 1] = t;
   }
   for (int i = 0; i < m; i++) {
      s->ritz_val[i] = d[idx[i]];
      s->ritz_est[i] = Z[(m - 1) + (size_t)idx[i] * m];
   }
   for (int i = 0; i < s->nev; i++)
      for (int r = 0; r < m; r++) s->ritz_vec[r + (size_t)i * m] = Z[r + (size_t)idx[i] * m];
   free(d);
   free(e);
   free(Z);
   free(idx);
}

static int irlm_num_converged(irlm *s, double tol)
{
   const double fnorm = nrm2(s->f, s->n);
   int c = 0;
   for (int i = 0; i < s->nev; i++) {
      double thresh = tol * fmax(fabs(s->ritz_val[i]), s->eps23);
      double resid = fabs(s->ritz_est[i]) * fnorm;
      s->ritz_conv[i] = (unsigned char)(resid < thresh);
      c += s->ritz_conv[i];
   }
   return c;
}

static int irlm_nev_adjusted(irlm *s, int nconv)
{
   int nev_new = s->nev;
   for (int i = s->nev; i < s->ncv; i++)
      if (fabs(s->ritz_est[i]) < s->near_0) nev_new++;
   int half = (s->ncv - nev_new) / 2;
   nev_new += nconv < half ? nconv : half;
   if (nev_new == 1 && s->ncv >= 6)
      nev_new = s->ncv / 2;
   else if (nev_new == 1 && s->ncv > 2)
      nev_new = 2;
   if (nev_new > s->ncv - 1) nev_new = s->ncv - 1;
   return nev_new;
}

/* Implicit restart: for every unwanted Ritz value mu (indices k..ncv-1) factor H - mu I = QR
 * (Givens rotations on the tridiagonal), set H <- RQ + mu I and accumulate Q; then V <- V Q on the
 * first k+1 columns, f <- f Q[ncv-1, k-1] + V[:,k] H[k, k-1], and extend the factorisation. */
static void irlm_restart(irlm *s, int k)
{
   const int m = s->ncv;
   const uint64_t n = s->n;
   if (k >= m) return;
   double *Q = (double *)calloc((size_t)m * m, sizeof(double));
   double *R = (double *)malloc(sizeof(double) * (size_t)m * m);
   double *cs = (double *)malloc(sizeof(double) * m), *sn = (double *)malloc(sizeof(double) * m);
   for (int i = 0; i < m; i++) Q[i + (size_t)i * m] = 1;
#define RR(i, j) R[(i) + (size_t)(j) * m]
#define QQ(i, j) Q[(i) + (size_t)(j) * m]
   for (int sh = k; sh < m; sh++) {
      const double mu = s->ritz_val[sh];
      for (int j = 0; j < m; j++)
         for (int i = 0; i < m; i++) RR(i, j) = HH(i, j);
      for (int i = 0; i < m; i++) RR(i, i) -= mu;
      /* QR by Givens on rows (i, i+1): G_i^T ... G_0^T (H - mu I) = R */
      for (int i = 0; i < m - 1; i++) {
         double a = RR(i, i), b = RR(i + 1, i), r = hypot(a, b);
         double c = 1, z = 0;
         if (r > 0) {
            c = a / r;
            z = b / r;
         }
         cs[i] = c;
         sn[i] = z;
         int jmax = i + 3 < m ? i + 3 : m;
         for (int j = i; j < jmax; j++) {
            double x = RR(i, j), y = RR(i + 1, j);
            RR(i, j) = c * x + z * y;
            RR(i + 1, j) = -z * x + c * y;
         }
      }
      /* H <- R Q + mu I, where Q = G_0 G_1 ... ; apply rotations on columns (i, i+1) */
      for (int i = 0; i < m - 1; i++) {
         const double c = cs[i], z = sn[i];
         int rmax = i + 2 < m ? i + 2 : m;
         for (int r = 0; r < rmax; r++) {
            double x = RR(r, i), y = RR(r, i + 1);
            RR(r, i) = c * x + z * y;
            RR(r, i + 1) = -z * x + c * y;
         }
         for (int r = 0; r < m; r++) { /* accumulate Q <- Q G_i */
            double x = QQ(r, i), y = QQ(r, i + 1);
            QQ(r, i) = c * x + z * y;
            QQ(r, i + 1) = -z * x + c * y;
         }
      }
      for (int j = 0; j < m; j++)
         for (int i = 0; i < m; i++) HH(i, j) = 0;
      for (int i = 0; i < m; i++) {
         HH(i, i) = RR(i, i) + mu;
         if (i + 1 < m) {
            /* symmetric tridiagonal: take the sub-diagonal, mirror it */
            HH(i + 1, i) = RR(i + 1, i);
            HH(i, i + 1) = RR(i + 1, i);
         }
      }
   }
   /* V[:, :k+1] <- V Q[:, :k+1] */
   double *Vs = (double *)malloc(sizeof(double) * n * (size_t)(k + 1));
   for (int j = 0; j <= k; j++) {
      double *dst = Vs + (size_t)j * n;
      for (uint64_t r = 0; r < n; r++) dst[r] = 0;
      for (int c = 0; c < m; c++) {
         const double q = QQ(c, j);
         if (q == 0.0) continue;
         const double *vc = s->V + (size_t)c * n;
         for (uint64_t r = 0; r < n; r++) dst[r] += vc[r] * q;
      }
   }
   memcpy(s->V, Vs, sizeof(double) * n * (size_t)(k + 1));
   free(Vs);
   double *fk = (double *)malloc(sizeof(double) * n);
   const double qlast = QQ(m - 1, k - 1), hk = HH(k, k - 1);
   const double *vk = s->V + (size_t)k * n;
   for (uint64_t r = 0; r < n; r++) fk[r] = s->f[r] * qlast + vk[r] * hk;
   free(Q);
   free(R);
   free(cs);
   free(sn);
#undef RR
#undef QQ
   irlm_factorize_from(s, k, m, fk);
   free(fk);
   irlm_retrieve_ritzpair(s);
}

int orc_symeigs(orc_op *op, int nev, int ncv, int maxit, double tol, double *evals, double *evecs,
                int *info, int *nrestarts)
{
   irlm S, *s = &S;
   memset(s, 0, sizeof(S));
   const uint64_t n = op->n;
   if (nev < 1 || (uint64_t)nev > n - 1 || ncv <= nev || (uint64_t)ncv > n) {
      if (info) *info = 2;
      return -1;
   }
   s->op = op;
   s->n = n;
   s->nev = nev;
   s->ncv = ncv;
   s->V = (double *)calloc((size_t)n * ncv, sizeof(double));
   s->H = (double *)calloc((size_t)ncv * ncv, sizeof(double));
   s->f = (double *)calloc(n, sizeof(double));
   s->ritz_val = (double *)calloc(ncv, sizeof(double));
   s->ritz_est = (double *)calloc(ncv, sizeof(double));
   s->ritz_vec = (double *)calloc((size_t)ncv * nev, sizeof(double));
   s->ritz_conv = (unsigned char *)calloc(nev, 1);
   s->near_0 = DBL_MIN * 10.0;
   s->eps = DBL_EPSILON;
   s->eps23 = pow(DBL_EPSILON, 2.0 / 3.0);

   /* init(): deterministic pseudo-random residual (seed 0), one operator application */
   double *w = (double *)malloc(sizeof(double) * n);
   lcg_seed(s, 0);
   for (uint64_t r = 0; r < n; r++) s->V[r] = lcg_next(s);
   double vn = nrm2(s->V, n);
   for (uint64_t r = 0; r < n; r++) s->V[r] /= vn;
   orc_perform_op(op, s->V, w);
   HH(0, 0) = dot(s->V, w, n);
   for (uint64_t r = 0; r < n; r++) s->f[r] = w[r] - s->V[r] * HH(0, 0);
   free(w);

   /* compute(maxit, tol) */
   double *f0 = (double *)malloc(sizeof(double) * n);
   memcpy(f0, s->f, sizeof(double) * n);
   irlm_factorize_from(s, 1, ncv, f0);
   free(f0);
   irlm_retrieve_ritzpair(s);
   int i, nconv = 0, nrest = 0;
   for (i = 0; i < maxit; i++) {
      nconv = irlm_num_converged(s, tol);
      if (nconv >= nev) break;
      int nev_adj = irlm_nev_adjusted(s, nconv);
      irlm_restart(s, nev_adj);
      nrest++;
   }
   if (nrestarts) *nrestarts = nrest;
   int ok = nconv >= nev;
   if (info) *info = ok ? 0 : 1;
   int nout = nconv < nev ? nconv : nev;
   /* eigenvalues(): converged Ritz values, already descending; eigenvectors(): V * ritz_vec */
   int o = 0;
   for (int c = 0; c < nev && o < nout; c++) {
      if (!s->ritz_conv[c]) continue;
      evals[o] = s->ritz_val[c];
      double *dst = evecs + (size_t)o * n;
      for (uint64_t r = 0; r < n; r++) dst[r] = 0;
      for (int j = 0; j < ncv; j++) {
         const double q = s->ritz_vec[j + (size_t)c * ncv];
         const double *vj = s->V + (size_t)j * n;
         for (uint64_t r = 0; r < n; r++) dst[r] += vj[r] * q;
      }
      o++;
   }
   free(s->V);
   free(s->H);
   free(s->f);
   free(s->ritz_val);
   free(s->ritz_est);
   free(s->ritz_vec);
   free(s->ritz_conv);
   return nout;
}
#undef HH

/* ------------------------------------------------------------------------------------------ */
/* RandomPCA::pca_fast(Data&, ...) (randompca.cpp:168-218)                                     */

int orc_pca_fast(orc_data *d, uint32_t block_size, int ndim, int maxiter, double tol, int divisor,
                 int do_loadings, int nthreads, double *U, double *dvals, double *V, double *Px,
                 double *pve, double *trace, uint32_t *nops)
{
   const uint64_t N = d->N, p = d->nsnps;
   orc_op *op = orc_op_new(d, block_size, nthreads);
   int info = 0, nrest = 0;
   int got = orc_symeigs(op, ndim, ndim * 2 + 1, maxiter, tol, dvals, U, &info, &nrest);
   if (info != 0 || got < ndim) { /* randompca.cpp:210-217: reference throws */
      orc_op_free(op);
      return 1;
   }
   double div = 1;
   if (divisor == ORC_DIVISOR_N1)
      div = (double)N - 1;
   else if (divisor == ORC_DIVISOR_P)
      div = (double)p;
   for (int j = 0; j < ndim; j++) dvals[j] /= div; /* :190 eigenvalues, not singular values */
   if (do_loadings && V) {                          /* :191-204 */
      for (int j = 0; j < ndim; j++) {
         double *v = V + (size_t)j * p;
         orc_crossprod(op, U + (size_t)j * N, v);
         const double sc = (1.0 / sqrt(dvals[j])) / sqrt(div);
         for (uint64_t r = 0; r < p; r++) v[r] *= sc;
      }
   }
   const double tr = op->trace / div; /* :205 */
   if (trace) *trace = tr;
   for (int j = 0; j < ndim; j++) {
      pve[j] = dvals[j] / tr; /* :206 */
      const double sq = sqrt(dvals[j]);
      for (uint64_t r = 0; r < N; r++) Px[r + (size_t)j * N] = U[r + (size_t)j * N] * sq; /* :207 */
   }
   if (nops) *nops = op->nops - 1;
   orc_op_free(op);
   return 0;
}

/* randompca.cpp:663-703: err_j = || X X' u_j / div - u_j lambda_j ||^2; mse = sum(err)/(N K) */
int orc_check(orc_data *d, uint32_t block_size, int divisor, const double *evec, const double *eval,
              int k, double *err, double *mse, double *rmse)
{
   const uint64_t N = d->N;
   orc_op *op = orc_op_new(d, block_size, 1);
   double div = 1;
   if (divisor == ORC_DIVISOR_N1)
      div = (double)N - 1;
   else if (divisor == ORC_DIVISOR_P)
      div = (double)d->nsnps;
   double *Y = (double *)malloc(sizeof(double) * N * (size_t)k);
   orc_perform_op_mat(op, evec, k, Y);
   double tot = 0;
   for (int j = 0; j < k; j++) {
      double sacc = 0;
      for (uint64_t r = 0; r < N; r++) {
         double e = Y[r + (size_t)j * N] / div - evec[r + (size_t)j * N] * eval[j];
         sacc += e * e;
      }
      err[j] = sacc;
      tot += sacc;
   }
   *mse = tot / ((double)N * k);
   *rmse = sqrt(*mse);
   free(Y);
   orc_op_free(op);
   return 0;
}

/* standardise(MatrixXd&, method) (util.cpp:24-192): column-wise, in place, NaN = missing.  X is n x p column-major;
 * meansd (p x 2 column-major, mean | sd) may be NULL.  Methods: 0 none, 1 sd, 2 binom, 3 binom2, 4 center.
 * Returns 0, or -1 for an unknown method (the reference throws "unknown standardization method"). */
int orc_standardise(double *X, uint64_t n, uint64_t p, int method, double *meansd)
{
   if (method < 0 || method > 4) return -1;
   for (uint64_t j = 0; j < p; j++) {
      double *col = X + j * n;
      double mean = 0, sd = 1;
      if (method == 0 || method == 4) { /* util.cpp:37-71 */
         uint64_t nj = 0;
         for (uint64_t i = 0; i < n; i++)
            if (!isnan(col[i])) {
               mean += col[i];
               nj++;
            }
         mean /= (double)nj;
         if (method == 0) {
            for (uint64_t i = 0; i < n; i++)
               if (isnan(col[i])) col[i] = mean;
         } else {
            for (uint64_t i = 0; i < n; i++) col[i] = isnan(col[i]) ? 0 : col[i] - mean;
         }
      } else if (method == 1) { /* util.cpp:72-116: shifted-data variance, K = 1 */
         double sum = 0, sum_sqr = 0;
         const double K = 1;
         uint64_t nj = 0;
         for (uint64_t i = 0; i < n; i++)
            if (!isnan(col[i])) {
               sum += col[i] - K;
               sum_sqr += (col[i] - K) * (col[i] - K);
               nj++;
            }
         double varj = (sum_sqr - (sum * sum) / (double)nj) / (double)(nj - 1);
         mean = (sum + K * (double)nj) / (double)nj;
         sd = sqrt(varj);
         for (uint64_t i = 0; i < n; i++) {
            if (isnan(col[i]))
               col[i] = 0;
            else if (sd > VAR_TOL)
               col[i] = (col[i] - mean) / sd;
            else
               col[i] = mean;
         }
      } else { /* util.cpp:117-150, Price 2006 eqn 3 */
         const double mult = method == ORC_STANDARDISE_BINOM ? 1 : 2;
         double sum = 0;
         uint64_t nj = 0;
         for (uint64_t i = 0; i < n; i++)
            if (!isnan(col[i])) {
               sum += col[i];
               nj++;
            }
         mean = sum / (double)nj;
         const double r = mean / 2.0;
         sd = sqrt(mult * r * (1.0 - r));
         for (uint64_t i = 0; i < n; i++) {
            if (isnan(col[i]))
               col[i] = 0;
            else if (sd > VAR_TOL)
               col[i] = (col[i] - mean) / sd;
            else
               col[i] = mean;
         }
      }
      if (meansd) {
         meansd[j] = mean;
         meansd[p + j] = sd;
      }
   }
   return 0;
}

/* flashpca.cpp:636-686 */
uint32_t orc_default_block_size(uint64_t N, uint64_t nsnps, int ndim, int do_loadings, int memory_mb)
{
   long long mem = (long long)memory_mb * 1048576;
   long long req = 2 * (long long)nsnps * 8 * 2 + 3 * (long long)nsnps * 8 + (long long)N * ndim * 8 +
                   (do_loadings ? (long long)nsnps * ndim * 8 : 0) + 2 * (long long)N +
                   2 * (long long)(N + nsnps) * ndim * 8 + 2 * 1024 * 1024 + (long long)N * 8;
   long long remain = mem - req;
   if (remain <= 0) return 0;
   double bs = floor((double)remain / ((double)N * 8.0));
   if (bs < 1) return 0;
   if (bs > (double)nsnps) bs = (double)nsnps;
   return (uint32_t)bs;
}

/* util.h:77,97: operator<< with std::setprecision(p), default float field == printf("%.{p}g") */
int orc_format_number(char *buf, int buflen, double v, int precision)
{
   return snprintf(buf, buflen, "%.*g", precision, v);
}
