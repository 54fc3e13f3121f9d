// bench_hooks.hip -- include/fpca_debug.h: measurement hooks (HIP-event timing of the operator's stages) and hardware probes.
// Not part of the drop-in boundary.
#include <algorithm>
#include <vector>

#include "../../include/fpca_debug.h"
#include "ctx.hpp"
#include "hip_backend.hpp"

using namespace fpca;

extern "C" {

// ---- measurement ---------------------------------------------------------------------------------------
int fpca_bench_apply(fpca_ctx *ctx, int b, int steps, int warmup, fpca_bench_result *res)
{
   return guarded([&] {
      if (!ctx || !res || steps < 1 || warmup < 0) throw Error(FPCA_EINVAL, "bad argument to fpca_bench_apply");
      if (b != 16 && b != 32 && b != 48 && b != 64) throw Error(FPCA_EINVAL, "b must be 16, 32, 48 or 64");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      double *dB = nullptr, *dY = nullptr;
      HIP_CHECK(hipMalloc(&dB, (size_t)ctx->N_pad * b * sizeof(double)));
      HIP_CHECK(hipMalloc(&dY, (size_t)ctx->N_pad * b * sizeof(double)));
      kern::fill_random(dB, ctx->N, ctx->N_pad, b, 12345, ctx->stream);
      std::vector<hipEvent_t> ev((size_t)steps * 8); // per step: stage boundaries [0..3], K2 / K3 GEMM kernel [4,5] / [6,7]
      for (auto &e : ev) HIP_CHECK(hipEventCreate(&e));
      for (int i = 0; i < warmup; i++) apply_xxt_dev(ctx, dB, b, dY, ctx->stream, nullptr);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      for (int i = 0; i < steps; i++) apply_xxt_dev(ctx, dB, b, dY, ctx->stream, &ev[(size_t)i * 8]);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      double t2 = 0, t3 = 0, ta = 0, g2 = 0, g3 = 0;
      float ms = 0;
      for (int i = 0; i < steps; i++) {
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 0], ev[i * 8 + 1]));
         t2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 1], ev[i * 8 + 2]));
         t3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 2], ev[i * 8 + 3]));
         ta += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 4], ev[i * 8 + 5]));
         g2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, ev[i * 8 + 6], ev[i * 8 + 7]));
         g3 += ms;
      }
      HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[(size_t)(steps - 1) * 8 + 3]));
      res->ms_total = ms;
      res->ms_xt = t2 / steps;
      res->ms_x = t3 / steps;
      res->ms_allreduce = ta / steps;
      res->ms_gemm_xt = g2 / steps;
      res->ms_gemm_x = g3 / steps;
      res->flops_per_step = 4.0 * (double)ctx->N * (double)ctx->P_g * b;
      res->packed_bytes_per_step = 2.0 * (double)ctx->np * (double)ctx->P_g;
      for (auto &e : ev) (void)hipEventDestroy(e);
      (void)hipFree(dB);
      (void)hipFree(dY);
   });
}

int fpca_profile_begin(fpca_ctx *ctx, int max_steps)
{
   return guarded([&] {
      if (!ctx || max_steps < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_begin");
      HIP_CHECK(hipSetDevice(ctx->device));
      while (ctx->prof_ev.size() < (size_t)max_steps * 8) {
         hipEvent_t e;
         HIP_CHECK(hipEventCreate(&e));
         ctx->prof_ev.push_back(e);
      }
      ctx->prof_used = 0;
      ctx->prof_calls = 0;
      ctx->prof_on = true;
   });
}

int fpca_profile_sample_every(fpca_ctx *ctx, int stride)
{
   return guarded([&] {
      if (!ctx || stride < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_sample_every");
      ctx->prof_stride = stride;
   });
}

int fpca_profile_end(fpca_ctx *ctx, int b, fpca_bench_result *res, int *nsteps)
{
   return guarded([&] {
      if (!ctx || !res) throw Error(FPCA_EINVAL, "bad argument to fpca_profile_end");
      HIP_CHECK(hipSetDevice(ctx->device));
      ctx->prof_on = false;
      HIP_CHECK(hipDeviceSynchronize());
      const int n = ctx->prof_used;
      double t2 = 0, t3 = 0, ta = 0, tt = 0, g2 = 0, g3 = 0;
      float ms = 0;
      for (int i = 0; i < n; i++) {
         hipEvent_t *e = &ctx->prof_ev[(size_t)i * 8];
         HIP_CHECK(hipEventElapsedTime(&ms, e[4], e[5]));
         g2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[6], e[7]));
         g3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[1]));
         t2 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[1], e[2]));
         t3 += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[2], e[3]));
         ta += ms;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[3]));
         tt += ms;
      }
      std::memset(res, 0, sizeof(*res));
      if (n > 0) {
         res->ms_total = tt;
         res->ms_xt = t2 / n;
         res->ms_x = t3 / n;
         res->ms_allreduce = ta / n;
         res->ms_gemm_xt = g2 / n;
         res->ms_gemm_x = g3 / n;
      }
      res->flops_per_step = 4.0 * (double)ctx->N * (double)ctx->P_g * b;
      res->packed_bytes_per_step = 2.0 * (double)ctx->np * (double)ctx->P_g;
      if (nsteps) *nsteps = n;
   });
}

int fpca_bench_stats(fpca_ctx *ctx, int reps, double *ms_per_launch, double *bytes_per_launch)
{
   return guarded([&] {
      if (!ctx || reps < 1) throw Error(FPCA_EINVAL, "bad argument to fpca_bench_stats");
      HIP_CHECK(hipSetDevice(ctx->device));
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      kern::bed_stats(ctx->d_packed, ctx->pitch, ctx->N, ctx->P_g, ctx->stand, ctx->d_lut, ctx->d_mean, ctx->d_sd, ctx->d_sumsq, nullptr, ctx->stream);
      HIP_CHECK(hipEventRecord(e0, ctx->stream));
      for (int i = 0; i < reps; i++)
         kern::bed_stats(ctx->d_packed, ctx->pitch, ctx->N, ctx->P_g, ctx->stand, ctx->d_lut, ctx->d_mean, ctx->d_sd, ctx->d_sumsq, nullptr, ctx->stream);
      HIP_CHECK(hipEventRecord(e1, ctx->stream));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms_per_launch) *ms_per_launch = ms / reps;
      if (bytes_per_launch) *bytes_per_launch = (double)ctx->np * (double)ctx->P_g;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
   });
}

// diagnostic used by tests/test_gpu_kernels.py: D = A(16x4) B(4x16) through the MFMA operand mapping of kernels.hip
int fpca_debug_mfma_probe(const double *A, const double *B, double *D)
{
   return guarded([&] {
      double *dA, *dB, *dD;
      HIP_CHECK(hipMalloc(&dA, 64 * sizeof(double)));
      HIP_CHECK(hipMalloc(&dB, 64 * sizeof(double)));
      HIP_CHECK(hipMalloc(&dD, 256 * sizeof(double)));
      HIP_CHECK(hipMemcpy(dA, A, 64 * sizeof(double), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dB, B, 64 * sizeof(double), hipMemcpyHostToDevice));
      kern::mfma_layout_probe(dA, dB, dD, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(D, dD, 256 * sizeof(double), hipMemcpyDeviceToHost));
      (void)hipFree(dA);
      (void)hipFree(dB);
      (void)hipFree(dD);
   });
}

int fpca_debug_mfma_i8_probe(const int8_t *A, const int8_t *Bt, int32_t *D)
{
   return guarded([&] {
      int8_t *dA, *dB;
      int *dD;
      HIP_CHECK(hipMalloc(&dA, 1024));
      HIP_CHECK(hipMalloc(&dB, 1024));
      HIP_CHECK(hipMalloc(&dD, 1024 * sizeof(int)));
      HIP_CHECK(hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dB, Bt, 1024, hipMemcpyHostToDevice));
      kern::mfma_i8_probe(dA, dB, dD, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(D, dD, 1024 * sizeof(int), hipMemcpyDeviceToHost));
      (void)hipFree(dA);
      (void)hipFree(dB);
      (void)hipFree(dD);
   });
}

int fpca_debug_k4(fpca_ctx *ctx, int b, int nq, const double *V, const double *W, double *C_gram, const double *C_in, int use_init,
                  double *Out, double *G_out)
{
   return guarded([&] {
      if (!ctx || !V || !W || nq < 1 || nq > 1000 || (b != 16 && b != 32 && b != 48 && b != 64)) throw Error(FPCA_EINVAL, "bad argument to fpca_debug_k4");
      if ((Out || G_out) && !C_in) throw Error(FPCA_EINVAL, "fpca_debug_k4: Out needs C_in");
      HIP_CHECK(hipSetDevice(ctx->device));
      HipBackend be(ctx, b);
      const int64_t N = (int64_t)ctx->N;
      std::vector<int> hv(nq);
      for (int q = 0; q < nq; q++) {
         hv[q] = be.alloc_block();
         be.upload(hv[q], b, V + (size_t)q * b * N, N);
      }
      const int hw = be.alloc_block();
      be.upload(hw, b, W, N);
      if (C_gram) be.gram(hv.data(), nq, hw, C_gram);
      if (Out || G_out) {
         const int ho = be.alloc_block();
         if (G_out) // the update and the Gram matrix of its output from one launch (HipBackend::gemm_gram)
            be.gemm_gram(hv.data(), nq, C_in, use_init ? hw : -1, ho, G_out);
         else
            be.gemm(hv.data(), nq, C_in, use_init ? hw : -1, ho);
         if (Out) be.download(ho, b, Out, N);
         be.free_block(ho);
      }
      be.free_block(hw);
      for (int h : hv) be.free_block(h);
   });
}

// HipBackend::gemm_gramvw through the backend object the solver drives: Out = W + sum_q V_q C_in[q], Cg[q] = V_q' Out (q < nq), Cg[nq] = Out' Out
int fpca_debug_k4_fused(fpca_ctx *ctx, int b, int nq, const double *V, const double *W, const double *C_in, double *Out, double *Cg)
{
   return guarded([&] {
      if (!ctx || !V || !W || !C_in || !Cg || nq < 1 || nq > 1000 || (b != 16 && b != 32 && b != 48 && b != 64)) throw Error(FPCA_EINVAL, "bad argument to fpca_debug_k4_fused");
      HIP_CHECK(hipSetDevice(ctx->device));
      HipBackend be(ctx, b);
      const int64_t N = (int64_t)ctx->N;
      std::vector<int> hv(nq);
      for (int q = 0; q < nq; q++) {
         hv[q] = be.alloc_block();
         be.upload(hv[q], b, V + (size_t)q * b * N, N);
      }
      const int hw = be.alloc_block();
      be.upload(hw, b, W, N);
      be.gemm_gramvw(hv.data(), nq, C_in, hw, hw, Cg); // (in place, as the solver calls it)
      if (Out) be.download(hw, b, Out, N);
      be.free_block(hw);
      for (int h : hv) be.free_block(h);
   });
}

// launch time of the fused update + Gram (kernel + plane reduction) against the two launches it replaces, nq random blocks
int fpca_debug_k4_fused_bench(fpca_ctx *ctx, int b, int nq, int reps, double *ms_fused)
{
   return guarded([&] {
      if (!ctx || nq < 1 || nq > 64 || reps < 1 || !ms_fused) throw Error(FPCA_EINVAL, "bad argument to fpca_debug_k4_fused_bench");
      HIP_CHECK(hipSetDevice(ctx->device));
      const uint64_t rows = ctx->N_pad;
      const int planes = kern::update_gram_planes(rows, nq, b);
      if (!planes) throw Error(FPCA_EINVAL, "no fused update + Gram kernel for this shape");
      std::vector<double *> blk(nq + 1, nullptr);
      const double **d_ptrs = nullptr;
      double *d_C = nullptr, *d_part = nullptr;
      hipEvent_t e[2] = {nullptr, nullptr};
      auto cleanup = [&] {
         for (double *p : blk)
            if (p) (void)hipFree(p);
         if (d_ptrs) (void)hipFree(d_ptrs);
         if (d_C) (void)hipFree(d_C);
         if (d_part) (void)hipFree(d_part);
         for (hipEvent_t x : e)
            if (x) (void)hipEventDestroy(x);
      };
      try {
         hipStream_t s = ctx->stream;
         for (int q = 0; q < nq + 1; q++) {
            HIP_CHECK(hipMalloc(&blk[q], rows * b * sizeof(double)));
            kern::fill_random(blk[q], ctx->N, rows, b, 100 + q, s);
         }
         HIP_CHECK(hipMalloc(&d_ptrs, (nq + 1) * sizeof(double *)));
         HIP_CHECK(hipMemcpyAsync(d_ptrs, blk.data(), (nq + 1) * sizeof(double *), hipMemcpyHostToDevice, s));
         const size_t cnt = (size_t)nq * b * b, cntg = (size_t)(nq + 1) * b * b;
         HIP_CHECK(hipMalloc(&d_C, cnt * sizeof(double)));
         HIP_CHECK(hipMalloc(&d_part, cntg * (planes + 1) * sizeof(double)));
         HIP_CHECK(hipMemsetAsync(d_C, 0, cnt * sizeof(double), s)); // (zero coefficients: the block stays bounded over the repetitions)
         for (hipEvent_t &x : e) HIP_CHECK(hipEventCreate(&x));
         for (int r = -2; r < reps; r++) {
            if (r == 0) HIP_CHECK(hipEventRecord(e[0], s));
            kern::update_gram(d_ptrs, nq, d_C, blk[nq], blk[nq], rows, b, d_part + cntg, s);
            kern::reduce_sum(d_part + cntg, d_part, cntg, planes, s);
         }
         HIP_CHECK(hipEventRecord(e[1], s));
         HIP_CHECK(hipEventSynchronize(e[1]));
         float ms = 0;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[1]));
         *ms_fused = ms / reps;
      } catch (...) {
         cleanup();
         throw;
      }
      cleanup();
   });
}

int fpca_debug_variant(int which, int variant)
{
   if (which == 0)
      kern::k4_variant(variant);
   else
      return FPCA_EINVAL;
   return FPCA_OK;
}

int fpca_debug_k4_bench(fpca_ctx *ctx, int b, int nq, int reps, double *ms_gram, double *ms_gemm)
{
   return guarded([&] {
      if (!ctx || nq < 1 || nq > 64 || reps < 1 || (b != 16 && b != 32 && b != 48 && b != 64)) throw Error(FPCA_EINVAL, "bad argument to fpca_debug_k4_bench");
      HIP_CHECK(hipSetDevice(ctx->device));
      const uint64_t rows = ctx->N_pad;
      std::vector<double *> blk(nq + 2, nullptr);
      const double **d_ptrs = nullptr;
      double *d_C = nullptr, *d_part = nullptr, *d_G = nullptr;
      hipEvent_t e[3] = {nullptr, nullptr, nullptr};
      auto cleanup = [&] {
         for (double *p : blk)
            if (p) (void)hipFree(p);
         if (d_ptrs) (void)hipFree(d_ptrs);
         if (d_C) (void)hipFree(d_C);
         if (d_part) (void)hipFree(d_part);
         if (d_G) (void)hipFree(d_G);
         for (hipEvent_t x : e)
            if (x) (void)hipEventDestroy(x);
      };
      try {
         hipStream_t s = ctx->stream;
         for (int q = 0; q < nq + 2; q++) {
            HIP_CHECK(hipMalloc(&blk[q], rows * b * sizeof(double)));
            kern::fill_random(blk[q], ctx->N, rows, b, 100 + q, s);
         }
         HIP_CHECK(hipMalloc(&d_ptrs, (nq + 2) * sizeof(double *)));
         HIP_CHECK(hipMemcpyAsync(d_ptrs, blk.data(), (nq + 2) * sizeof(double *), hipMemcpyHostToDevice, s));
         const size_t cnt = (size_t)nq * b * b;
         const int grows = kern::gram_rows(rows, nq, b), ns = kern::gram_splits(rows, grows) * 4;
         HIP_CHECK(hipMalloc(&d_C, cnt * sizeof(double)));
         HIP_CHECK(hipMalloc(&d_G, cnt * sizeof(double)));
         HIP_CHECK(hipMalloc(&d_part, cnt * ns * sizeof(double)));
         kern::fill_random(d_C, cnt / b, cnt / b, b, 7, s);
         for (hipEvent_t &x : e) HIP_CHECK(hipEventCreate(&x));
         for (int w = 0; w < 2; w++) { // warm-up
            kern::gram(d_ptrs, nq, blk[nq], d_part, rows, b, grows, s);
            kern::reduce_sum(d_part, d_G, cnt, ns, s);
            kern::block_gemm(d_ptrs, nq, d_C, blk[nq], blk[nq + 1], rows, b, s);
         }
         HIP_CHECK(hipEventRecord(e[0], s));
         for (int r = 0; r < reps; r++) {
            kern::gram(d_ptrs, nq, blk[nq], d_part, rows, b, grows, s);
            kern::reduce_sum(d_part, d_G, cnt, ns, s);
         }
         HIP_CHECK(hipEventRecord(e[1], s));
         for (int r = 0; r < reps; r++) kern::block_gemm(d_ptrs, nq, d_C, blk[nq], blk[nq + 1], rows, b, s);
         HIP_CHECK(hipEventRecord(e[2], s));
         HIP_CHECK(hipEventSynchronize(e[2]));
         float ms = 0;
         HIP_CHECK(hipEventElapsedTime(&ms, e[0], e[1]));
         if (ms_gram) *ms_gram = ms / reps;
         HIP_CHECK(hipEventElapsedTime(&ms, e[1], e[2]));
         if (ms_gemm) *ms_gemm = ms / reps;
      } catch (...) {
         cleanup();
         throw;
      }
      cleanup();
   });
}

int fpca_debug_census(int nwg, uint64_t lds_bytes, uint32_t *out)
{
   return guarded([&] {
      uint32_t *d = nullptr;
      HIP_CHECK(hipMalloc(&d, (size_t)nwg * 2 * sizeof(uint32_t)));
      kern::census(d, nwg, (size_t)lds_bytes, 2000000, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
      HIP_CHECK(hipMemcpy(out, d, (size_t)nwg * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
      (void)hipFree(d);
   });
}

int fpca_debug_mfma_peak(int waves_per_simd, int iters, int pattern, double *tflops)
{
   return guarded([&] {
      if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 10 || pattern < 0 || (pattern > 3 && (pattern < 10 || pattern > 13) && (pattern < 20 || pattern > 25) && pattern < 1000) || !tflops)
         throw Error(FPCA_EINVAL, "bad argument");
      if (pattern >= 1000) // 1000 + 100 LDS reads per MFMA + 10 VALU per MFMA + (1: bursts): v_mfma_f32_16x16x4_f32 with filler instructions
         *tflops = kern::mfma_valu_mix_tflops((pattern - 1000) % 100 / 10, (pattern & 1) != 0, (pattern - 1000) / 100, waves_per_simd, iters, nullptr);
      else if (pattern >= 20) // 20 / 21 v_mfma_f32_16x16x4_f32, 22 / 23 v_mfma_f32_32x32x2_f32, 24 / 25 v_mfma_f64_16x16x4_f64: zero / random operands
         *tflops = kern::mfma_fp_peak_tflops((pattern - 20) / 2, waves_per_simd, iters, (pattern & 1) ? 0x1234567u : 0u, nullptr);
      else if (pattern >= 12) // the b = 16 column remainder: 12 = four 32-wide tiles (half of the last one padding), 13 = three + 16x16x64
         *tflops = kern::mfma_i8_mix_tops(pattern - 12, iters, nullptr);
      else if (pattern >= 10) // v_mfma_i32_32x32x32_i8 (TOP/s): 10 = zero operands, 11 = random operands
         *tflops = kern::mfma_i8_peak_tops(waves_per_simd, iters, pattern == 11 ? 0x1234567u : 0u, nullptr);
      else
         *tflops = kern::mfma_peak_tflops(waves_per_simd, iters, pattern, nullptr);
   });
}

} // extern "C"
