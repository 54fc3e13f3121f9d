// hip_backend.hpp -- the product's BlockBackend (backend.hpp): the eigensolver's N-sized objects as fp64 blocks [rows][b] in HBM,
// every method a launch of hand-written gfx950 kernels (K2 / K3 through operator.hip, K4 gram / block_gemm).  PRIVATE to libfpca.so.
#pragma once
#include <vector>

#include "ctx.hpp"

namespace fpca {

class HipBackend : public BlockBackend {
 public:
   // How the solver's blocks are laid out over the ranks of a multi-GPU run.
   //   SINGLE      one rank (or a context that knows no rank): whole blocks, no collective but the operator's own
   //   ROWSHARD    blocks are row slices (backend.hpp RowShard): all-gather -> K2, K3 -> reduce-scatter per apply, local Gram + a
   //               small all-reduce, local updates
   //   REPLICATED  every rank keeps whole blocks and repeats the orthogonalisation: ONE all-reduce of the N x b product per
   //               apply and nothing else (north_star's literal scheme; svdwide.cpp:48-62 summed over ranks)
   enum Layout { SINGLE = 0, ROWSHARD = 1, REPLICATED = 2 };
   // the layout fpca_pca would take for this context (replicated: the caller asked for it / the sharded exchange was demoted)
   static Layout plan_layout(const fpca_ctx *c, bool replicated);
   // the row-shard geometry of that layout (RowShard::on() false unless ROWSHARD); chunk count from N and the transport only --
   // NOT from the arithmetic in effect: a rank whose int8 buffers did not fit must issue the same collectives as the others
   static RowShard plan_shard(const fpca_ctx *c, Layout layout);

   // cheap_S: byte slices of the eigensolver's cheap passes (0: none).  Whether they exist is decided from the arithmetic the
   // context was CREATED with, not from what it runs now: a rank whose exact-integer buffers did not fit runs the fp64 kernels
   // for every pass but must follow the same sequence of passes as the others.
   HipBackend(fpca_ctx *c, int b, bool replicated = false, int cheap_S = 0);
   ~HipBackend() override;

   uint64_t nrows() const override { return c_->N; }
   int width() const override { return b_; }
   int alloc_block() override;
   void free_block(int h) override { used_[h] = 0; }
   void fill_random(int h, uint64_t seed) override;
   void apply(int in, int out) override;
   void apply_begin(int in, int out) override;
   void apply_end() override;
   bool set_cheap(bool cheap) override;
   void gemm_gramvw(const int *a, int nq, const double *C, int init, int out, double *Cg) override;
   void gram(const int *a, int nq, int w, double *C) override;
   void gemm(const int *a, int nq, const double *C, int init, int out) override;
   void gemm_gram(const int *a, int nq, const double *C, int init, int out, double *G) override;
   void download(int h, int ncols, double *host, int64_t ld) override { download2(h, ncols, host, ld, nullptr, 0, nullptr); }
   void download2(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale) override;
   void download_rows_mine(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale) override;
   void upload(int h, int ncols, const double *host, int64_t ld) override;
   double trace() override;
   double seconds_apply() override { return sec_apply_; }
   double seconds_other() override { return sec_other_; }

   Layout layout() const { return layout_; }
   bool sharded() const { return sh_.on(); }
   const RowShard &shard() const { return sh_; }
   int cheap_slices() const { return cheap_S_; }
   double seconds_exact() const { return sec_exact_; }
   // the WHOLE block h as the operator wants it ([N_pad][b] on this device): the block itself, or -- row-sharded -- its rows
   // gathered from all ranks into the context's scratch (a collective: every rank calls it)
   double *full_ptr(int h);
   // rows [begin, end) of the blocks this rank writes in download_rows_mine (global sample indices, clipped to N); as a function
   // of the context and the shard geometry alone, for fpca_pca_row_ranges
   static std::vector<std::pair<uint64_t, uint64_t>> rows_of(const fpca_ctx *c, const RowShard &sh);
   std::vector<std::pair<uint64_t, uint64_t>> rows_mine() const { return rows_of(c_, sh_); }

 private:
   size_t block_bytes() const { return (size_t)rows_ * b_ * sizeof(double); }
   void pin_wait();
   double *pin_coeff(size_t cnt);
   void push_ptrs(const int *a, int nq);
   void gemm_launch(const int *a, int nq, const double *C, int init, int out, double *gram_part);
   void grow(double *&p, size_t &cap, size_t need);

   fpca_ctx *c_;
   int b_;
   int cheap_S_; // byte slices of the cheap passes (0: the backend has none)
   Layout layout_ = SINGLE;
   uint64_t rows_ = 0; // rows of a block as THIS rank stores it: N_pad, or its slice of the row-sharded solver
   RowShard sh_;
   std::vector<double *> blocks_;
   std::vector<unsigned char> used_;
   const double **&d_ptrs_; // scratch owned by the context (kept across solves)
   double *&d_C_, *&d_gpart_;
   size_t &C_cap_, &gpart_cap_;
   void *&h_pin_;
   size_t &pin_cap_;
   bool pin_busy_ = false, inflight_ = false, inflight_exact_ = false;
   hipEvent_t e0_ = nullptr, e1_ = nullptr, ev_pin_ = nullptr;
   double sec_apply_ = 0, sec_other_ = 0, sec_exact_ = 0;
};

} // namespace fpca
