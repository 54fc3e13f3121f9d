/*
 * oracle/hostsim_backend.cpp -- host-memory implementation of fpca::BlockBackend over the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle/fpca_oracle.h): it lets tests run the product's host eigensolver
 * (flashpca_amd/csrc/solver.cpp, pca_driver.cpp -- the same sources that are linked into libfpca.so) and
 * its multi-rank sharding logic on machines without a GPU: blocks live in host memory, the operator is the
 * oracle's perform_op_mat on this rank's SNP shard, and the cross-rank sum is a caller-supplied callback
 * (torch.distributed/gloo in tests/test_multirank_gloo.py).  Never linked into libfpca.so or the CLI.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../flashpca_amd/csrc/backend.hpp"
#include "../flashpca_amd/csrc/common.hpp"
#include "../flashpca_amd/csrc/pca_driver.hpp"
#include "../flashpca_amd/csrc/plink_io.hpp"
#include "../flashpca_amd/csrc/solver.hpp"
#include "../flashpca_amd/csrc/symeig.hpp"
#include "fpca_oracle.h"

namespace fpca {
// libfpca.so defines this in context.hip; the test library needs its own copy
void set_last_error(const std::string &) {}
} // namespace fpca

namespace {

typedef int (*hostsim_allreduce_fn)(void *user, double *buf, uint64_t count);

class HostSimBackend : public fpca::BlockBackend {
 public:
   // nranks > 1 (with an all-reduce callback): the row-sharded solver of backend.hpp RowShard, on host memory -- every block
   // is this rank's slice of rows, the operator is all-gather -> oracle operator on the SNP shard -> reduce-scatter (both
   // built from the callback's sum, like the HIP backend does over a caller-supplied all-reduce), the Gram coefficients are
   // all-reduced.  nranks <= 1: whole blocks, one all-reduce of the N x b product per apply (round 2's scheme).
   // cheap_bits > 0: the backend offers "cheap passes" like the exact-integer mode of the HIP backend does -- the input block
   // is rounded to cheap_bits-bit fixed point per column (relative to the column maximum, as k_slice rounds the operand of
   // K2) before the oracle's operator is applied to it; 0: no cheap arithmetic (set_cheap returns false).
   HostSimBackend(orc_data *d, int b, uint32_t block_size, hostsim_allreduce_fn ar, void *user, int nranks = 1, int rank = 0, int cheap_bits = 0)
       : d_(d), b_(b), N_(orc_N(d)), ar_(ar), user_(user), cheap_bits_(cheap_bits)
   {
      op_ = orc_op_new(d, block_size ? block_size : (uint32_t)orc_nsnps(d), 1);
      rows_ = N_;
      if (ar && nranks > 1) {
         sh_ = fpca::RowShard::make(N_, nranks, rank, 1, 1);
         rows_ = sh_.slice_rows();
         row0_ = (uint64_t)rank * sh_.plen;
         full_in_.assign((size_t)N_ * b_, 0.0);
         full_out_.assign((size_t)N_ * b_, 0.0);
      }
   }
   ~HostSimBackend() override { orc_op_free(op_); }
   uint64_t nrows() const override { return N_; }
   int width() const override { return b_; }
   int alloc_block() override
   {
      for (size_t i = 0; i < used_.size(); i++)
         if (!used_[i]) {
            used_[i] = 1;
            return (int)i;
         }
      blocks_.emplace_back((size_t)rows_ * b_, 0.0);
      used_.push_back(1);
      return (int)blocks_.size() - 1;
   }
   void free_block(int h) override { used_[h] = 0; }
   void fill_random(int h, uint64_t seed) override
   {
      std::vector<double> whole((size_t)N_ * b_);
      uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
      for (auto &v : whole) {
         s ^= s << 13;
         s ^= s >> 7;
         s ^= s << 17;
         v = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      }
      take_rows(whole, blocks_[h]); // (sharded: the rows this rank keeps of the block every rank would have generated)
   }
   bool set_cheap(bool cheap) override
   {
      if (cheap_bits_ <= 0) return false;
      cheap_on_ = cheap;
      return true;
   }
   int cheap_applies() const { return n_cheap_; }
   // the cheap passes' operand: every column rounded to cheap_bits_ bits below the power of two above its largest entry
   void round_columns(const double *in, std::vector<double> &out) const
   {
      out.assign(in, in + (size_t)N_ * b_);
      for (int c = 0; c < b_; c++) {
         double m = 0;
         for (uint64_t i = 0; i < N_; i++) m = std::max(m, std::fabs(out[i + (size_t)c * N_]));
         if (!(m > 0)) continue;
         int e = 0;
         (void)std::frexp(m, &e);
         const double up = std::ldexp(1.0, cheap_bits_ - e), dn = std::ldexp(1.0, e - cheap_bits_);
         for (uint64_t i = 0; i < N_; i++) out[i + (size_t)c * N_] = std::nearbyint(out[i + (size_t)c * N_] * up) * dn;
      }
   }
   void apply(int in, int out) override
   {
      if (cheap_on_) n_cheap_++;
      if (!sh_.on()) {
         const double *src = blocks_[in].data();
         if (cheap_on_) {
            round_columns(src, rounded_);
            src = rounded_.data();
         }
         orc_perform_op_mat(op_, src, b_, blocks_[out].data());
         sum(blocks_[out].data(), N_ * (uint64_t)b_);
         return;
      }
      all_gather(blocks_[in], full_in_);
      if (cheap_on_) {
         round_columns(full_in_.data(), rounded_);
         full_in_ = rounded_;
      }
      orc_perform_op_mat(op_, full_in_.data(), b_, full_out_.data());
      sum(full_out_.data(), N_ * (uint64_t)b_); // reduce-scatter = sum + keep my rows
      take_rows(full_out_, blocks_[out]);
   }
   void gram(const int *a, int nq, int w, double *C) override
   {
      const double *W = blocks_[w].data();
      for (int q = 0; q < nq; q++) {
         const double *A = blocks_[a[q]].data();
         for (int p = 0; p < b_; p++)
            for (int c = 0; c < b_; c++) {
               const double *ap = A + (size_t)p * rows_, *wc = W + (size_t)c * rows_;
               double s = 0;
               for (uint64_t i = 0; i < rows_; i++) s += ap[i] * wc[i];
               C[((size_t)q * b_ + p) * b_ + c] = s;
            }
      }
      if (sh_.on()) sum(C, (uint64_t)nq * b_ * b_); // row slices: the only collective of the orthogonalisation
   }
   void gemm(const int *a, int nq, const double *C, int init, int out) override
   {
      std::vector<double> res((size_t)rows_ * b_, 0.0);
      if (init >= 0) res = blocks_[init];
      for (int q = 0; q < nq; q++) {
         const double *A = blocks_[a[q]].data();
         for (int c = 0; c < b_; c++)
            for (int p = 0; p < b_; p++) {
               const double cv = C[((size_t)q * b_ + p) * b_ + c];
               if (cv == 0.0) continue;
               const double *ap = A + (size_t)p * rows_;
               double *rc = res.data() + (size_t)c * rows_;
               for (uint64_t i = 0; i < rows_; i++) rc[i] += ap[i] * cv;
            }
      }
      blocks_[out] = res;
   }
   void download(int h, int ncols, double *host, int64_t ld) override
   {
      const std::vector<double> *src = &blocks_[h];
      if (sh_.on()) { // a collective: every rank comes here (pca_driver.cpp calls download2 on every rank)
         all_gather(blocks_[h], full_in_);
         src = &full_in_;
      }
      if (!host) return;
      for (int c = 0; c < ncols; c++) std::memcpy(host + (size_t)c * ld, src->data() + (size_t)c * N_, sizeof(double) * N_);
   }
   void download2(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale) override
   {
      if (!host && !host2) {
         if (sh_.on()) download(h, ncols, nullptr, 0);
         return;
      }
      fpca::BlockBackend::download2(h, ncols, host, ld, host2, ld2, scale);
   }
   void upload(int h, int ncols, const double *host, int64_t ld) override
   {
      std::vector<double> whole((size_t)N_ * b_, 0.0);
      for (int c = 0; c < ncols; c++) std::memcpy(whole.data() + (size_t)c * N_, host + (size_t)c * ld, sizeof(double) * N_);
      take_rows(whole, blocks_[h]);
   }
   double trace() override
   {
      double t = orc_op_trace(op_);
      sum(&t, 1);
      return t;
   }

 private:
   void sum(double *buf, uint64_t count)
   {
      if (ar_ && ar_(user_, buf, count) != 0) throw fpca::Error(-6, "allreduce failed");
   }
   // whole: column-major N x b; blk: column-major rows_ x b (rows beyond N are zero)
   void take_rows(const std::vector<double> &whole, std::vector<double> &blk) const
   {
      if (!sh_.on()) {
         blk = whole;
         return;
      }
      std::fill(blk.begin(), blk.end(), 0.0);
      for (int c = 0; c < b_; c++)
         for (uint64_t i = 0; i < rows_ && row0_ + i < N_; i++) blk[i + (size_t)c * rows_] = whole[row0_ + i + (size_t)c * N_];
   }
   void all_gather(const std::vector<double> &blk, std::vector<double> &whole)
   {
      std::fill(whole.begin(), whole.end(), 0.0);
      for (int c = 0; c < b_; c++)
         for (uint64_t i = 0; i < rows_ && row0_ + i < N_; i++) whole[row0_ + i + (size_t)c * N_] = blk[i + (size_t)c * rows_];
      sum(whole.data(), N_ * (uint64_t)b_);
   }
   orc_data *d_;
   orc_op *op_;
   int b_;
   uint64_t N_, rows_ = 0, row0_ = 0;
   fpca::RowShard sh_;
   hostsim_allreduce_fn ar_;
   void *user_;
   std::vector<std::vector<double>> blocks_; // column-major rows_ x b
   std::vector<unsigned char> used_;
   std::vector<double> full_in_, full_out_;  // row-sharded: whole N x b blocks either side of the operator
   int cheap_bits_ = 0, n_cheap_ = 0;
   bool cheap_on_ = false;
   std::vector<double> rounded_;
};

} // namespace

extern "C" {

/* Runs the product's block Krylov-Schur driver on this rank's shard `d` (an oracle Data object).
 * nranks / rank: > 1 ranks -> the row-sharded solver (RowShard), 0 or 1 -> whole blocks on every rank.
 * maxiter > 0: --maxiter as the reference counts it (fpca_pca_opts.maxiter); maxiter < 0: fpca_pca_opts.max_applies = -maxiter.
 * info_out: [converged, block_applies, restarts, blockvec].  Returns FPCA_OK / FPCA_ENOTCONVERGED / <0. */
int hostsim_pca2(orc_data *d, int ndim, int blockvec, int maxiter, double tol, int divisor, int max_blocks,
                 uint64_t seed, int verbose, uint64_t P_total, hostsim_allreduce_fn ar, void *user, double *U,
                 double *dvals, double *Px, double *pve, double *trace, int *info_out, int nranks, int rank, int cheap_bits);
int hostsim_pca(orc_data *d, int ndim, int blockvec, int maxiter, double tol, int divisor, int max_blocks,
                uint64_t seed, int verbose, uint64_t P_total, hostsim_allreduce_fn ar, void *user, double *U,
                double *dvals, double *Px, double *pve, double *trace, int *info_out, int nranks, int rank)
{
   return hostsim_pca2(d, ndim, blockvec, maxiter, tol, divisor, max_blocks, seed, verbose, P_total, ar, user, U, dvals, Px, pve, trace,
                       info_out, nranks, rank, 0);
}

/* The same with the solver's mixed-precision logic in play: cheap_bits > 0 makes the backend offer cheap passes (its input
 * rounded to cheap_bits bits per column), which the solver verifies by exact ones; info_out then has 6 entries:
 * [converged, block_applies, restarts, blockvec, cheap_applies, cheap passes the backend actually ran]. */
int hostsim_pca2(orc_data *d, int ndim, int blockvec, int maxiter, double tol, int divisor, int max_blocks,
                 uint64_t seed, int verbose, uint64_t P_total, hostsim_allreduce_fn ar, void *user, double *U,
                 double *dvals, double *Px, double *pve, double *trace, int *info_out, int nranks, int rank, int cheap_bits)
{
   try {
      const int b = fpca::choose_blockvec(ndim, blockvec);
      HostSimBackend be(d, b, 0, ar, user, nranks, rank, cheap_bits); // nranks > 1: row-sharded solver; <= 1: replicated (round 2)
      fpca_pca_opts o;
      std::memset(&o, 0, sizeof(o));
      o.ndim = ndim;
      o.blockvec = b;
      o.maxiter = maxiter; // the reference's unit (restarts); a NEGATIVE value is a hard cap of -maxiter block applies
      if (maxiter < 0) {
         o.maxiter = 500;
         o.max_applies = -maxiter;
      }
      o.tol = tol;
      o.divisor = divisor;
      o.max_blocks = max_blocks;
      o.verbose = verbose;
      o.seed = seed;
      fpca::PcaOutputs out;
      out.U = U;
      out.d = dvals;
      out.Px = Px;
      out.pve = pve;
      fpca_pca_info info;
      std::memset(&info, 0, sizeof(info));
      int rc = fpca::run_pca(be, o, P_total ? P_total : orc_nsnps(d), out, &info, nullptr, nullptr);
      if (trace) *trace = info.trace;
      if (info_out) {
         info_out[0] = info.converged;
         info_out[1] = info.block_applies;
         info_out[2] = info.restarts;
         info_out[3] = info.blockvec;
         if (cheap_bits > 0) {
            info_out[4] = info.cheap_applies;
            info_out[5] = be.cheap_applies();
         }
      }
      return rc;
   } catch (const fpca::Error &e) {
      std::fprintf(stderr, "hostsim_pca: %s\n", e.what());
      return e.code;
   } catch (const std::exception &e) {
      std::fprintf(stderr, "hostsim_pca: %s\n", e.what());
      return -3;
   }
}

/* the CLI's text writers / readers (plink_io.cpp) exposed for CPU unit tests */
int hostsim_save_text(const double *M, uint64_t rows, uint64_t cols, const char *colnames_tab, const char *rownames_nl,
                      const char *filename, unsigned precision)
{
   try {
      std::vector<std::string> cn, rn;
      auto split = [](const char *s, char sep, std::vector<std::string> &out) {
         if (!s || !*s) return;
         std::string cur;
         for (const char *p = s; *p; p++) {
            if (*p == sep) {
               out.push_back(cur);
               cur.clear();
            } else
               cur += *p;
         }
         out.push_back(cur);
      };
      split(colnames_tab, '|', cn);
      split(rownames_nl, '|', rn);
      return fpca::save_text(M, rows, cols, cn, rn, filename, precision) ? 0 : 1;
   } catch (const std::exception &e) {
      std::fprintf(stderr, "hostsim_save_text: %s\n", e.what());
      return -1;
   }
}

/* returns rows, writes cols; values copied column-major into out (capacity cap doubles); -1 + message on error */
long hostsim_read_text(const char *filename, unsigned firstcol, long nrows, unsigned skip, double *out, uint64_t cap,
                       uint64_t *cols, char *err, int errlen)
{
   try {
      fpca::TextMatrix M = fpca::read_text(filename, firstcol, nrows, skip);
      if (cols) *cols = M.cols;
      if (M.v.size() > cap) return -2;
      std::memcpy(out, M.v.data(), M.v.size() * sizeof(double));
      return (long)M.rows;
   } catch (const std::exception &e) {
      if (err) std::snprintf(err, errlen, "%s", e.what());
      return -1;
   }
}

long hostsim_read_fam(const char *filename, char *err, int errlen)
{
   try {
      std::vector<std::string> a, b;
      fpca::read_plink_fam(filename, a, b);
      return (long)a.size();
   } catch (const std::exception &e) {
      if (err) std::snprintf(err, errlen, "%s", e.what());
      return -1;
   }
}

/* the CLI's one-pass .fam reader (plink_io.cpp read_fam): returns N, the ids joined as "fid\tiid\n" into ids (capacity cap) */
long hostsim_read_fam_onepass(const char *filename, char *ids, uint64_t cap, char *err, int errlen)
{
   try {
      std::vector<std::string> a, b;
      const uint64_t n = fpca::read_fam(filename, a, b);
      std::string all;
      for (size_t i = 0; i < a.size(); i++) all += a[i] + "\t" + b[i] + "\n";
      if (ids && all.size() + 1 <= cap) std::memcpy(ids, all.c_str(), all.size() + 1);
      return (long)n;
   } catch (const std::exception &e) {
      if (err) std::snprintf(err, errlen, "%s", e.what());
      return -1;
   }
}

long hostsim_read_bim(const char *filename, char *err, int errlen)
{
   try {
      std::vector<std::string> a, b, c;
      fpca::read_plink_bim(filename, a, b, c);
      return (long)a.size();
   } catch (const std::exception &e) {
      if (err) std::snprintf(err, errlen, "%s", e.what());
      return -1;
   }
}

/* dense symmetric eigensolver of the product (symeig.cpp) exposed for unit tests */
int hostsim_symeig(int n, double *A, double *w) { return fpca::symeig_desc(n, A, n, w); }
int hostsim_symeig_cols(int n, double *A, double *w, int ncols, double *Z) { return fpca::symeig_desc_cols(n, A, n, w, ncols, Z); }
int hostsim_symeig_rows(int n, double *A, double *w, int row0, int nrows, double *Zr)
{
   return fpca::symeig_desc_rows(n, A, n, w, row0, nrows, Zr);
}

} // extern "C"
