// hip_backend.hip -- see hip_backend.hpp.
#include <algorithm>
#include <chrono>

#include "hip_backend.hpp"

namespace fpca {

HipBackend::Layout HipBackend::plan_layout(const fpca_ctx *c, bool replicated)
{
   // FPCA_FORCE_ROWSHARD (test builds): the sharded code path with a single rank
   const bool force = FPCA_TEST_ENV("FPCA_FORCE_ROWSHARD") != nullptr;
   const bool ranks = c->multi() && c->rank_known && c->nranks > 1;
   if (!replicated && (ranks || force)) return ROWSHARD;
   return c->multi() ? REPLICATED : SINGLE; // (a context that only has an all-reduce hook and no rank keeps whole blocks)
}

RowShard HipBackend::plan_shard(const fpca_ctx *c, Layout layout)
{
   if (layout != ROWSHARD) return RowShard();
   const bool ranks = c->multi() && c->rank_known;
   return RowShard::make(c->N_pad, ranks ? c->nranks : 1, ranks ? c->rank : 0, shard_chunks(c), 512);
}

HipBackend::HipBackend(fpca_ctx *c, int b, bool replicated, int cheap_S)
   : c_(c), b_(b), cheap_S_((c->i8_S_req > 0 && cheap_S >= 2 && cheap_S < c->i8_S_req) ? cheap_S : 0), d_ptrs_(c->be_ptrs), d_C_(c->be_C),
     d_gpart_(c->be_gpart), C_cap_(c->be_C_cap), gpart_cap_(c->be_gpart_cap), h_pin_(c->be_pin), pin_cap_(c->be_pin_cap)
{
   HIP_CHECK(hipSetDevice(c->device));
   ensure_stats(c);
   rows_ = c->N_pad;
   layout_ = plan_layout(c, replicated);
   sh_ = plan_shard(c, layout_);
   if (sh_.on()) {
      rows_ = sh_.slice_rows();
      const size_t need = (size_t)sh_.full_rows() * b;
      c->ensure(c->d_full_in, c->full_in_cap, need);
      c->ensure(c->d_full_out, c->full_out_cap, need);
      // rows >= N_pad of the whole blocks are never written by the kernels and must read as zero in the collectives
      HIP_CHECK(hipMemsetAsync(c->d_full_in, 0, c->full_in_cap * sizeof(double), c->stream));
      HIP_CHECK(hipMemsetAsync(c->d_full_out, 0, c->full_out_cap * sizeof(double), c->stream));
   }
   if (!d_ptrs_) HIP_CHECK(hipMalloc(&d_ptrs_, 1024 * sizeof(double *)));
   HIP_CHECK(hipEventCreate(&e0_));
   HIP_CHECK(hipEventCreate(&e1_));
   HIP_CHECK(hipEventCreateWithFlags(&ev_pin_, hipEventDisableTiming));
   (void)pin_coeff((size_t)16 * b * b);
}

HipBackend::~HipBackend()
{
   c_->i8_Sc = 0;
   (void)hipStreamSynchronize(c_->stream);
   for (double *p : blocks_)
      if (p) c_->block_pool.emplace_back(block_bytes(), p);
   if (ev_pin_) (void)hipEventDestroy(ev_pin_);
   if (e0_) (void)hipEventDestroy(e0_);
   if (e1_) (void)hipEventDestroy(e1_);
}

int HipBackend::alloc_block()
{
   for (size_t i = 0; i < used_.size(); i++)
      if (!used_[i]) {
         used_[i] = 1;
         return (int)i;
      }
   double *p = nullptr;
   for (size_t i = 0; i < c_->block_pool.size(); i++)
      if (c_->block_pool[i].first == block_bytes()) {
         p = c_->block_pool[i].second;
         c_->block_pool.erase(c_->block_pool.begin() + (long)i);
         break;
      }
   if (!p) HIP_CHECK(hipMalloc(&p, block_bytes()));
   blocks_.push_back(p);
   used_.push_back(1);
   return (int)blocks_.size() - 1;
}

double *HipBackend::full_ptr(int h)
{
   if (!sharded()) return blocks_[h];
   c_->all_gather(sh_, blocks_[h], c_->d_full_in, b_, c_->stream);
   return c_->d_full_in;
}

void HipBackend::fill_random(int h, uint64_t seed)
{
   if (!sharded()) {
      kern::fill_random(blocks_[h], c_->N, c_->N_pad, b_, seed, c_->stream);
      return;
   }
   for (int c = 0; c < sh_.nch; c++) // the rows this rank keeps of the block every rank would have generated
      kern::fill_random(blocks_[h] + (size_t)c * sh_.plen * b_, c_->N, sh_.plen, b_, seed, c_->stream, (uint64_t)c * sh_.L + (uint64_t)sh_.rank * sh_.plen);
}

void HipBackend::apply(int in, int out)
{
   apply_begin(in, out);
   apply_end();
}

void HipBackend::apply_begin(int in, int out)
{
   HIP_CHECK(hipEventRecord(e0_, c_->stream));
   if (sharded())
      apply_sharded(c_, sh_, blocks_[in], b_, blocks_[out], c_->stream);
   else
      apply_xxt_dev(c_, blocks_[in], b_, blocks_[out], c_->stream, nullptr);
   HIP_CHECK(hipEventRecord(e1_, c_->stream));
   inflight_ = true;
   inflight_exact_ = !c_->i8_Sc;
}

void HipBackend::apply_end()
{
   if (!inflight_) return;
   inflight_ = false;
   HIP_CHECK(hipEventSynchronize(e1_));
   float ms = 0;
   HIP_CHECK(hipEventElapsedTime(&ms, e0_, e1_));
   sec_apply_ += ms * 1e-3;
   if (inflight_exact_) sec_exact_ += ms * 1e-3;
}

bool HipBackend::set_cheap(bool cheap)
{
   if (!cheap_S_) return false;
   c_->i8_Sc = cheap ? cheap_S_ : 0;
   return true;
}

// Host <-> device traffic of the small matrices goes through one pinned buffer ([1024 pointers][coefficients]); the
// event marks the last asynchronous read of it, so a call never overwrites what an earlier copy has not picked up.
void HipBackend::pin_wait()
{
   if (pin_busy_) HIP_CHECK(hipEventSynchronize(ev_pin_));
   pin_busy_ = false;
}

double *HipBackend::pin_coeff(size_t cnt)
{
   pin_wait();
   if (cnt > pin_cap_) {
      if (h_pin_) HIP_CHECK(hipHostFree(h_pin_));
      h_pin_ = nullptr;
      pin_cap_ = std::max(cnt, 2 * pin_cap_);
      HIP_CHECK(hipHostMalloc(&h_pin_, 1024 * sizeof(double *) + pin_cap_ * sizeof(double), hipHostMallocDefault));
   }
   return reinterpret_cast<double *>(static_cast<char *>(h_pin_) + 1024 * sizeof(double *));
}

void HipBackend::push_ptrs(const int *a, int nq)
{
   if (nq > 1024) throw Error(FPCA_EINVAL, "too many basis blocks");
   const double **hp = static_cast<const double **>(h_pin_);
   for (int q = 0; q < nq; q++) hp[q] = blocks_[a[q]];
   HIP_CHECK(hipMemcpyAsync(d_ptrs_, hp, nq * sizeof(double *), hipMemcpyHostToDevice, c_->stream));
}

void HipBackend::gram(const int *a, int nq, int w, double *C)
{
   auto t0 = std::chrono::steady_clock::now();
   const size_t cnt = (size_t)nq * b_ * b_;
   const int rows = kern::gram_rows(rows_, nq, b_);
   const int ns = kern::gram_splits(rows_, rows) * 4;
   grow(d_gpart_, gpart_cap_, cnt * ns);
   grow(d_C_, C_cap_, std::max(cnt, (size_t)1024 * b_ * 4));
   double *hc = pin_coeff(cnt);
   push_ptrs(a, nq);
   kern::gram(d_ptrs_, nq, blocks_[w], d_gpart_, rows_, b_, rows, c_->stream);
   kern::reduce_sum(d_gpart_, d_C_, cnt, ns, c_->stream);
   if (sharded() && sh_.G > 1) c_->allreduce(d_C_, cnt, c_->stream); // row slices: the only collective of the orthogonalisation
   HIP_CHECK(hipMemcpyAsync(hc, d_C_, cnt * sizeof(double), hipMemcpyDeviceToHost, c_->stream));
   HIP_CHECK(hipStreamSynchronize(c_->stream));
   std::memcpy(C, hc, cnt * sizeof(double));
   sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HipBackend::gemm_launch(const int *a, int nq, const double *C, int init, int out, double *gram_part)
{
   const size_t cnt = (size_t)nq * b_ * b_;
   grow(d_C_, C_cap_, std::max(cnt, (size_t)1024 * b_ * 4));
   double *hc = pin_coeff(cnt);
   std::memcpy(hc, C, cnt * sizeof(double));
   push_ptrs(a, nq);
   HIP_CHECK(hipMemcpyAsync(d_C_, hc, cnt * sizeof(double), hipMemcpyHostToDevice, c_->stream));
   HIP_CHECK(hipEventRecord(ev_pin_, c_->stream));
   pin_busy_ = true;
   // not drained: d_C_ / d_ptrs_ are only rewritten by later copies on this same stream, i.e. after the kernel
   kern::block_gemm(d_ptrs_, nq, d_C_, init >= 0 ? blocks_[init] : nullptr, blocks_[out], rows_, b_, c_->stream, gram_part);
}

void HipBackend::gemm(const int *a, int nq, const double *C, int init, int out)
{
   auto t0 = std::chrono::steady_clock::now();
   gemm_launch(a, nq, C, init, out, nullptr);
   sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the update and the Gram matrix of the block it writes in ONE pass over that block (k_block_gemm_lds<..., GRAM>)
void HipBackend::gemm_gram(const int *a, int nq, const double *C, int init, int out, double *G)
{
   const int planes = kern::block_gemm_gram_planes(rows_, b_);
   if (!planes) {
      BlockBackend::gemm_gram(a, nq, C, init, out, G);
      return;
   }
   auto t0 = std::chrono::steady_clock::now();
   const size_t bb = (size_t)b_ * b_;
   grow(d_gpart_, gpart_cap_, bb * planes + bb);
   gemm_launch(a, nq, C, init, out, d_gpart_ + bb);
   kern::reduce_sum(d_gpart_ + bb, d_gpart_, bb, planes, c_->stream);
   if (sharded() && sh_.G > 1) c_->allreduce(d_gpart_, bb, c_->stream);
   pin_wait(); // (the coefficients just pushed have been picked up before the pinned buffer is reused for the result)
   double *hc = pin_coeff(bb);
   HIP_CHECK(hipMemcpyAsync(hc, d_gpart_, bb * sizeof(double), hipMemcpyDeviceToHost, c_->stream));
   HIP_CHECK(hipStreamSynchronize(c_->stream));
   std::memcpy(G, hc, bb * sizeof(double));
   sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the first projection's update and the second projection's Gram matrices in one pass over the basis (k_update_gram16)
void HipBackend::gemm_gramvw(const int *a, int nq, const double *C, int init, int out, double *Cg)
{
   const int planes = kern::update_gram_planes(rows_, nq, b_);
   if (!planes || init < 0) {
      BlockBackend::gemm_gramvw(a, nq, C, init, out, Cg);
      return;
   }
   auto t0 = std::chrono::steady_clock::now();
   const size_t bb = (size_t)b_ * b_, cnt = (size_t)nq * bb, cntg = (size_t)(nq + 1) * bb;
   grow(d_gpart_, gpart_cap_, cntg * planes + cntg);
   grow(d_C_, C_cap_, std::max(cnt, (size_t)1024 * b_ * 4));
   double *hc = pin_coeff(std::max(cnt, cntg));
   std::memcpy(hc, C, cnt * sizeof(double));
   push_ptrs(a, nq);
   HIP_CHECK(hipMemcpyAsync(d_C_, hc, cnt * sizeof(double), hipMemcpyHostToDevice, c_->stream));
   kern::update_gram(d_ptrs_, nq, d_C_, blocks_[init], blocks_[out], rows_, b_, d_gpart_ + cntg, c_->stream);
   kern::reduce_sum(d_gpart_ + cntg, d_gpart_, cntg, planes, c_->stream);
   if (sharded() && sh_.G > 1) c_->allreduce(d_gpart_, cntg, c_->stream);
   HIP_CHECK(hipMemcpyAsync(hc, d_gpart_, cntg * sizeof(double), hipMemcpyDeviceToHost, c_->stream)); // (stream order: after the upload above was consumed)
   HIP_CHECK(hipStreamSynchronize(c_->stream));
   std::memcpy(Cg, hc, cntg * sizeof(double));
   sec_other_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HipBackend::download2(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale)
{
   if (!host && !host2 && !sharded()) return;
   const double *whole = full_ptr(h); // (row-sharded: a collective -- every rank comes here, whether it wants the result or not)
   if (!host && !host2) return;
   c_->ensure(c_->d_stage, c_->stage_cap, (size_t)c_->N * ncols);
   kern::block_to_colmajor(whole, c_->N, b_, ncols, c_->d_stage, c_->N, c_->stream);
   staged_download(c_, c_->d_stage, c_->N, ncols, host, ld, host2, ld2, scale);
}

// The rows of the result THIS rank is responsible for (fpca_pca_opts.partial_rows): its slice of the row-sharded solver, or -- whole
// blocks on several ranks -- an even share of the rows.  No collective, 1 / G of the PCIe traffic per rank; the caller's U / Px
// are N x ncols all the same (memory the ranks share, or slices it gathers itself: fpca_pca_row_ranges).
std::vector<std::pair<uint64_t, uint64_t>> HipBackend::rows_of(const fpca_ctx *c_, const RowShard &sh_)
{
   std::vector<std::pair<uint64_t, uint64_t>> r;
   const uint64_t N = c_->N;
   if (sh_.on()) {
      for (int c = 0; c < sh_.nch; c++) {
         const uint64_t g0 = std::min<uint64_t>((uint64_t)c * sh_.L + (uint64_t)sh_.rank * sh_.plen, N), g1 = std::min<uint64_t>(g0 + sh_.plen, N);
         if (g1 > g0) r.emplace_back(g0, g1);
      }
   } else if (c_->multi() && c_->rank_known && c_->nranks > 1) {
      const uint64_t per = round_up((c_->N_pad + c_->nranks - 1) / c_->nranks, 512);
      const uint64_t g0 = std::min<uint64_t>(per * c_->rank, N), g1 = std::min<uint64_t>(g0 + per, N);
      if (g1 > g0) r.emplace_back(g0, g1);
   } else
      r.emplace_back(0, N);
   return r;
}

void HipBackend::download_rows_mine(int h, int ncols, double *host, int64_t ld, double *host2, int64_t ld2, const double *scale)
{
   if (!host && !host2) return;
   for (const auto &rg : rows_mine()) {
      const uint64_t g0 = rg.first, n = rg.second - rg.first;
      // where global row g0 sits in this rank's copy of the block
      const uint64_t local = sharded() ? (g0 / sh_.L) * sh_.plen + (g0 % sh_.L - (uint64_t)sh_.rank * sh_.plen) : g0;
      c_->ensure(c_->d_stage, c_->stage_cap, (size_t)n * ncols);
      kern::block_to_colmajor(blocks_[h] + (size_t)local * b_, n, b_, ncols, c_->d_stage, n, c_->stream);
      staged_download(c_, c_->d_stage, n, ncols, host ? host + g0 : nullptr, ld, host2 ? host2 + g0 : nullptr, ld2, scale); // (synchronises)
   }
}

void HipBackend::upload(int h, int ncols, const double *host, int64_t ld)
{
   c_->ensure(c_->d_stage, c_->stage_cap, (size_t)c_->N * ncols);
   HIP_CHECK(hipMemcpy2DAsync(c_->d_stage, c_->N * sizeof(double), host, (size_t)ld * sizeof(double), c_->N * sizeof(double), ncols,
                              hipMemcpyHostToDevice, c_->stream));
   if (!sharded())
      kern::colmajor_to_block(c_->d_stage, c_->N, c_->N, c_->N_pad, b_, ncols, blocks_[h], c_->stream);
   else { // the whole block into the scratch, this rank's rows out of it
      kern::colmajor_to_block(c_->d_stage, c_->N, c_->N, c_->N_pad, b_, ncols, c_->d_full_out, c_->stream);
      for (int c = 0; c < sh_.nch; c++)
         HIP_CHECK(hipMemcpyAsync(blocks_[h] + (size_t)c * sh_.plen * b_, c_->d_full_out + ((size_t)c * sh_.L + (size_t)sh_.rank * sh_.plen) * b_,
                                  (size_t)sh_.plen * b_ * sizeof(double), hipMemcpyDeviceToDevice, c_->stream));
   }
   HIP_CHECK(hipStreamSynchronize(c_->stream));
}

double HipBackend::trace()
{
   double t = c_->trace_local;
   if (c_->multi()) {
      HIP_CHECK(hipMemcpyAsync(c_->d_small, &t, sizeof(double), hipMemcpyHostToDevice, c_->stream));
      c_->allreduce(c_->d_small, 1, c_->stream);
      HIP_CHECK(hipMemcpyAsync(&t, c_->d_small, sizeof(double), hipMemcpyDeviceToHost, c_->stream));
      HIP_CHECK(hipStreamSynchronize(c_->stream));
   }
   return t;
}

void HipBackend::grow(double *&p, size_t &cap, size_t need)
{
   if (need <= cap) return;
   HIP_CHECK(hipStreamSynchronize(c_->stream));
   if (p) HIP_CHECK(hipFree(p));
   p = nullptr;
   need = std::max(need, 2 * cap); // geometric: the basis grows by one block per step
   HIP_CHECK(hipMalloc(&p, need * sizeof(double)));
   cap = need;
}

} // namespace fpca
