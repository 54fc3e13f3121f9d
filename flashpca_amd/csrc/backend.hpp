// backend.hpp -- the narrow interface the host eigensolver drives.
//
// The solver (solver.cpp) never touches matrix data itself: every N-sized object is an opaque "block"
// (N x b, fp64) owned by a backend.  The product backend is HipBackend (hip_backend.hip): blocks live in
// HBM and every method enqueues hand-written gfx950 kernels.  oracle/hostsim_backend.cpp implements the
// same interface on host memory over the CPU oracle -- TEST INFRASTRUCTURE ONLY, used to exercise the
// solver and the multi-rank sharding logic on machines without a GPU (gloo tests); it is never linked
// into libfpca.so.
#pragma once
#include <cstddef>
#include <cstdint>

namespace fpca {

// Row-sharded solver state (multi-GPU).  The operator needs the whole N x b block on every rank (K2 sums over
// all samples), but nothing else does: between two applies every N-sized object of the eigensolver -- the Krylov basis, W,
// the Ritz blocks -- lives as a ROW SLICE on each rank.  The rows are cut into `nch` chunks of L = G plen rows (the chunks
// of the overlapped K3 / reduce-scatter pipeline); rank r keeps rows [c L + r plen, c L + (r+1) plen) of every chunk c, stored
// back to back: slice-local row c plen + j.  Orthogonalisation = local Gram + an all-reduce of (m+1) b^2 doubles + local
// update; the apply = all-gather -> K2, K3 -> reduce-scatter, the same bytes on the wire as the one all-reduce it replaces.
struct RowShard {
   int G = 1, rank = 0, nch = 1;
   uint64_t L = 0, plen = 0;
   bool on() const { return G > 1 || L > 0; }
   uint64_t slice_rows() const { return (uint64_t)nch * plen; }
   uint64_t full_rows() const { return (uint64_t)nch * L; }
   uint64_t global_row(uint64_t rho) const { return (rho / plen) * L + (uint64_t)rank * plen + rho % plen; }
   // rows: the (padded) height of a block; align: granularity of plen (512 for the HIP kernels, 1 on the host)
   static RowShard make(uint64_t rows, int G, int rank, int nch, uint64_t align)
   {
      RowShard s;
      s.G = G < 1 ? 1 : G;
      s.rank = rank;
      s.nch = nch < 1 ? 1 : nch;
      const uint64_t parts = (uint64_t)s.G * s.nch;
      s.plen = ((rows + parts - 1) / parts + align - 1) / align * align;
      s.L = s.plen * s.G;
      return s;
   }
};

class BlockBackend {
 public:
   virtual ~BlockBackend() {}
   virtual uint64_t nrows() const = 0; // N (samples)
   virtual int width() const = 0;      // b (columns per block)

   virtual int alloc_block() = 0;    // handle >= 0
   virtual void free_block(int h) = 0;
   virtual void fill_random(int h, uint64_t seed) = 0;

   // out = sum over ranks g of X_g X_g' in    (K2 + K3 + all-reduce); in != out
   virtual void apply(int in, int out) = 0;
   // The same in two halves, for a backend that can run the operator while the caller does something else: apply_begin
   // enqueues it, apply_end waits for it (at most one apply in flight).  The solver launches the next pass before it solves
   // the projected eigenproblem of the current one whenever that test is unlikely to end the iteration.
   virtual void apply_begin(int in, int out) { apply(in, out); }
   virtual void apply_end() {}
   // Arithmetic of the following apply() calls: cheap = the backend's reduced-precision passes (exact-integer mode: fewer
   // byte slices of the fp64 operand), else its exact ones.  Returns false -- and changes nothing -- when the backend has
   // no cheaper arithmetic than the one it runs; the solver then never asks again.
   virtual bool set_cheap(bool cheap) { (void)cheap; return false; }

   // C[q][p][c] = sum_s A_q[s][p] W[s][c]   for q < nq  (host result => synchronises)
   virtual void gram(const int *a, int nq, int w, double *C) = 0;
   // out = (init >= 0 ? block init : 0) + sum_q A_q C_q, C[q][p][c]; out may alias init or any a[q]
   virtual void gemm(const int *a, int nq, const double *C, int init, int out) = 0;

   // gemm + the Gram matrix of what it wrote: G[p][c] = sum_s Out[s][p] Out[s][c] (host result => synchronises).  A backend that
   // can produce G from the tile it has in registers saves a pass over the block; the default is the two calls.
   virtual void gemm_gram(const int *a, int nq, const double *C, int init, int out, double *G)
   {
      gemm(a, nq, C, init, out);
      gram(&out, 1, out, G);
   }

   // gemm (out = init + sum_q A_q C_q), then Cg[q][p][c] = sum_s A_q[s][p] Out[s][c] for q < nq and Cg[nq] = Out'Out -- the update of the
   // first Gram-Schmidt projection and the Gram matrices of the second from ONE pass over the basis where the backend can (host
   // result => synchronises).  Default: the two calls.
   virtual void gemm_gramvw(const int *a, int nq, const double *C, int init, int out, double *Cg)
   {
      gemm(a, nq, C, init, out);
      int vw[1024];
      for (int q = 0; q < nq && q < 1023; q++) vw[q] = a[q];
      vw[nq] = out;
      gram(vw, nq + 1, out, Cg);
   }

   // first ncols columns of a block <-> host column-major N x ncols
   virtual void download(int h, int ncols, double *host, int64_t ld) = 0;
   virtual void upload(int h, int ncols, const double *host, int64_t ld) = 0;
   // download with the Px pass fused in: column c of the block goes to host[:, c] (if host) and, scaled by scale[c], to
   // host2[:, c] (if host2) -- RandomPCA::Px = U diag(sqrt(d)) (randompca.cpp:207) without a second pass over U
   virtual void download2(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale)
   {
      const uint64_t N = nrows();
      double *dst = host ? host : host2;
      const int64_t ldd = host ? ld : ld2;
      if (!dst) return;
      download(h, ncols, dst, ldd);
      if (host2)
         for (int c = ncols - 1; c >= 0; c--) // (in place when host == nullptr)
            for (uint64_t i = 0; i < N; i++) host2[i + (std::size_t)c * ld2] = dst[i + (std::size_t)c * ldd] * scale[c];
   }

   // Several ranks: only the rows of the result THIS rank is responsible for (its slice of a row-sharded basis, or an even share of
   // whole blocks) into the caller's N x ncols matrices -- no collective, 1 / G of the traffic per rank.  Default: everything.
   virtual void download_rows_mine(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale)
   {
      download2(h, ncols, host, ld, host2, ld2, scale);
   }

   // sum over ranks of the shard traces sum X^2 (svdwide.cpp:44-45, 60-61)
   virtual double trace() = 0;

   // seconds spent in apply() / in the other device methods since construction (for fpca_pca_info)
   virtual double seconds_apply() { return 0; }
   virtual double seconds_other() { return 0; }
};

} // namespace fpca
