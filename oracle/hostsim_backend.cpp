/*
 * oracle/hostsim_backend.cpp -- host-memory implementation of fpca::BlockBackend over the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle/fpca_oracle.h): it lets tests run the product's host eigensolver
 * (flashpca_amd/csrc/solver.cpp, pca_driver.cpp -- the same sources that are linked into libfpca.so) and
 * its multi-rank sharding logic on machines without a GPU: blocks live in host memory, the operator is the
 * oracle's perform_op_mat on this rank's SNP shard, and the cross-rank sum is a caller-supplied callback
 * (torch.distributed/gloo in tests/test_multirank_gloo.py).  Never linked into libfpca.so or the CLI.
 */
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
// libfpca.so defines this in device_ctx.hip; the test library needs its own copy
void set_last_error(const std::string &) {}
} // namespace fpca

namespace {

typedef int (*hostsim_allreduce_fn)(void *user, double *buf, uint64_t count);

class HostSimBackend : public fpca::BlockBackend {
 public:
   HostSimBackend(orc_data *d, int b, uint32_t block_size, hostsim_allreduce_fn ar, void *user)
       : d_(d), b_(b), N_(orc_N(d)), ar_(ar), user_(user)
   {
      op_ = orc_op_new(d, block_size ? block_size : (uint32_t)orc_nsnps(d), 1);
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
      blocks_.emplace_back((size_t)N_ * b_, 0.0);
      used_.push_back(1);
      return (int)blocks_.size() - 1;
   }
   void free_block(int h) override { used_[h] = 0; }
   void fill_random(int h, uint64_t seed) override
   {
      uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
      for (auto &v : blocks_[h]) {
         s ^= s << 13;
         s ^= s >> 7;
         s ^= s << 17;
         v = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      }
   }
   void apply(int in, int out) override
   {
      orc_perform_op_mat(op_, blocks_[in].data(), b_, blocks_[out].data());
      if (ar_ && ar_(user_, blocks_[out].data(), N_ * (uint64_t)b_) != 0) throw fpca::Error(-6, "allreduce failed");
   }
   void gram(const int *a, int nq, int w, double *C) override
   {
      const double *W = blocks_[w].data();
      for (int q = 0; q < nq; q++) {
         const double *A = blocks_[a[q]].data();
         for (int p = 0; p < b_; p++)
            for (int c = 0; c < b_; c++) {
               const double *ap = A + (size_t)p * N_, *wc = W + (size_t)c * N_;
               double s = 0;
               for (uint64_t i = 0; i < N_; i++) s += ap[i] * wc[i];
               C[((size_t)q * b_ + p) * b_ + c] = s;
            }
      }
   }
   void gemm(const int *a, int nq, const double *C, int init, int out) override
   {
      std::vector<double> res((size_t)N_ * b_, 0.0);
      if (init >= 0) res = blocks_[init];
      for (int q = 0; q < nq; q++) {
         const double *A = blocks_[a[q]].data();
         for (int c = 0; c < b_; c++)
            for (int p = 0; p < b_; p++) {
               const double cv = C[((size_t)q * b_ + p) * b_ + c];
               if (cv == 0.0) continue;
               const double *ap = A + (size_t)p * N_;
               double *rc = res.data() + (size_t)c * N_;
               for (uint64_t i = 0; i < N_; i++) rc[i] += ap[i] * cv;
            }
      }
      blocks_[out] = res;
   }
   void download(int h, int ncols, double *host, int64_t ld) override
   {
      for (int c = 0; c < ncols; c++) std::memcpy(host + (size_t)c * ld, blocks_[h].data() + (size_t)c * N_, sizeof(double) * N_);
   }
   void upload(int h, int ncols, const double *host, int64_t ld) override
   {
      std::fill(blocks_[h].begin(), blocks_[h].end(), 0.0);
      for (int c = 0; c < ncols; c++) std::memcpy(blocks_[h].data() + (size_t)c * N_, host + (size_t)c * ld, sizeof(double) * N_);
   }
   double trace() override
   {
      double t = orc_op_trace(op_);
      if (ar_ && ar_(user_, &t, 1) != 0) throw fpca::Error(-6, "allreduce failed");
      return t;
   }

 private:
   orc_data *d_;
   orc_op *op_;
   int b_;
   uint64_t N_;
   hostsim_allreduce_fn ar_;
   void *user_;
   std::vector<std::vector<double>> blocks_; // column-major N x b
   std::vector<unsigned char> used_;
};

} // namespace

extern "C" {

/* Runs the product's block Krylov-Schur driver on this rank's shard `d` (an oracle Data object).
 * maxiter > 0: --maxiter as the reference counts it (fpca_pca_opts.maxiter); maxiter < 0: fpca_pca_opts.max_applies = -maxiter.
 * info_out: [converged, block_applies, restarts, blockvec].  Returns FPCA_OK / FPCA_ENOTCONVERGED / <0. */
int hostsim_pca(orc_data *d, int ndim, int blockvec, int maxiter, double tol, int divisor, int max_blocks,
                uint64_t seed, int verbose, uint64_t P_total, hostsim_allreduce_fn ar, void *user, double *U,
                double *dvals, double *Px, double *pve, double *trace, int *info_out)
{
   try {
      const int b = fpca::choose_blockvec(ndim, blockvec);
      HostSimBackend be(d, b, 0, ar, user);
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
