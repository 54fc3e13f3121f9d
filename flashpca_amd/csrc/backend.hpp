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

   // C[q][p][c] = sum_s A_q[s][p] W[s][c]   for q < nq  (host result => synchronises)
   virtual void gram(const int *a, int nq, int w, double *C) = 0;
   // out = (init >= 0 ? block init : 0) + sum_q A_q C_q, C[q][p][c]; out may alias init or any a[q]
   virtual void gemm(const int *a, int nq, const double *C, int init, int out) = 0;

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

   // sum over ranks of the shard traces sum X^2 (svdwide.cpp:44-45, 60-61)
   virtual double trace() = 0;

   // seconds spent in apply() / in the other device methods since construction (for fpca_pca_info)
   virtual double seconds_apply() { return 0; }
   virtual double seconds_other() { return 0; }
};

} // namespace fpca
