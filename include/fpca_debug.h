/*
 * include/fpca_debug.h -- measurement hooks and hardware diagnostics of libfpca.so.
 *
 * NOT part of the drop-in boundary: a maintainer binding the reference's operator seam (svdwide.h:77-81) or driver
 * (randompca.h:77-80) needs include/fpca.h only.  What is declared here serves bench.py (HIP-event timing of the kernels
 * inside the timed region), the kernel parity tests (operand-layout probes, the eigensolver's K4 helpers on caller data) and
 * the lab scripts under scripts/ (MFMA stream rates, workgroup placement).  Same conventions as fpca.h.
 */
#ifndef FPCA_DEBUG_H
#define FPCA_DEBUG_H

#include "fpca.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement hooks (bench.py): run `steps` block-applies of width b on device-resident random blocks after
 * `warmup` untimed ones; HIP events on the context's stream bracket every kernel.  Times in milliseconds. */
typedef struct fpca_bench_result {
   double ms_total;      /* wall (event) time of the timed region, all steps */
   double ms_xt;         /* average per step in K2 (xt_b), incl. its split-K reduce */
   double ms_x;          /* average per step in K3 (x_t), incl. its split-K reduce */
   double ms_allreduce;  /* average per step in the all-reduce (0 for one rank) */
   double flops_per_step;           /* 4 N P_g b */
   double packed_bytes_per_step;    /* 2 ceil(N/4) P_g */
   double ms_gemm_xt;    /* average per step of the K2 GEMM kernel launch alone */
   double ms_gemm_x;     /* average per step of the K3 GEMM kernel launch alone */
} fpca_bench_result;
int fpca_bench_apply(fpca_ctx *ctx, int b, int steps, int warmup, fpca_bench_result *res);
/* Live profiling of caller-driven applies: between fpca_profile_begin and fpca_profile_end every
 * fpca_apply_xxt_dev call (up to max_steps of them) records HIP events on its stream around K2, K3 and the
 * all-reduce; fpca_profile_end synchronises and returns the per-step averages over the recorded calls
 * (ms_total = sum of all recorded steps; *nsteps = number recorded). */
int fpca_profile_begin(fpca_ctx *ctx, int max_steps);
/* Eight in-stream events per apply are not free at small sizes (0.62 vs 0.53 ms per apply at 50,000 x 20,000): with
 * stride > 1 only every stride-th apply of the profiled span carries them, the others run exactly as the solver runs
 * them.  Default 1. */
int fpca_profile_sample_every(fpca_ctx *ctx, int stride);
int fpca_profile_end(fpca_ctx *ctx, int b, fpca_bench_result *res, int *nsteps);
/* time the one-off statistics pass (K1) the same way: milliseconds per launch, bytes read */
int fpca_bench_stats(fpca_ctx *ctx, int reps, double *ms_per_launch, double *bytes_per_launch);

/* diagnostic: D(16x16, row-major) = A(16x4) B(4x16) through v_mfma_f64_16x16x4_f64 with the lane->operand mapping the
 * kernels assume; host pointers.  Used by tests/test_gpu_kernels.py as a guard on the hardware layout. */
int fpca_debug_mfma_probe(const double *A, const double *B, double *D);
/* diagnostic: D(32x32 int32, row-major) = A(32x32 int8, row-major) * Bt(32x32 int8, row j = column j of B)' through
 * v_mfma_i32_32x32x32_i8 with the lane->operand mapping of kernels_i8.hip; host pointers */
int fpca_debug_mfma_i8_probe(const int8_t *A, const int8_t *Bt, int32_t *D);
/* diagnostic: sustained rate (TFLOP/s) of a pure v_mfma_f64_16x16x4_f64 stream with `waves_per_simd` (1..8) resident
 * waves per SIMD and no memory traffic; pattern 0..3 selects the operand-register sharing pattern (kernels.hip).  The
 * practical ceiling to read the GEMM kernels' roofline fraction against (72-74 TFLOP/s at 2 waves/SIMD vs 78.6 datasheet).
 * pattern 10 / 11: v_mfma_i32_32x32x32_i8 in TOP/s with zero / pseudo-random operands.
 * pattern 20 / 21: v_mfma_f32_16x16x4_f32, 22 / 23: v_mfma_f32_32x32x2_f32, 24 / 25: v_mfma_f64_16x16x4_f64 -- zero / pseudo-random
 * operands (compiler-scheduled intrinsics): with random operands the package power cap sets the ceiling. */
int fpca_debug_mfma_peak(int waves_per_simd, int iters, int pattern, double *tflops);
/* diagnostic (tests/test_gpu_kernels.py): the K4 helpers the eigensolver runs on its HBM-resident basis, on caller data and
 * through the very backend object the solver drives (HipBackend::gram incl. its split-K plane reduction, HipBackend::gemm).
 * V: N x (nq b) fp64 column-major with leading dimension N, basis block q = columns [q b, (q+1) b); W: N x b.
 *   C_gram (may be NULL): [q][p][c] = sum_s V_q[s][p] W[s][c]                       nq b b doubles
 *   Out (may be NULL), N x b: (use_init ? W : 0) + sum_q V_q C_in[q]                C_in: [q][p][c], nq b b doubles
 *   G_out (may be NULL): [p][c] = sum_s Out[s][p] Out[s][c], from the SAME launch that writes Out (HipBackend::gemm_gram)  b b doubles */
int fpca_debug_k4(fpca_ctx *ctx, int b, int nq, const double *V, const double *W, double *C_gram, const double *C_in, int use_init,
                  double *Out, double *G_out);
/* lab (process-wide A/B switches, 1 = default = the round-5 kernels, 0 = the kernels of rounds 1-4):
 *   which 0: the K4 kernels of the eigensolver's orthogonalisation -- Gram with one W tile per 8 basis blocks, block GEMM with its
 *            coefficients in LDS and four row tiles per wave.
 * fpca_debug_k4_bench: launch times of the K4 kernels on nq device-resident random basis blocks of this context's height: ms per
 * Gram (kernel + plane reduction) and per block GEMM (Out = Init + sum_q V_q C_q) */
int fpca_debug_variant(int which, int variant);
/* round 6: the first Gram-Schmidt projection's update and the second projection's Gram matrices from ONE pass over the basis
 * (HipBackend::gemm_gramvw, k_update_gram16): Out (may be NULL) = W + sum_q V_q C_in[q]; Cg: [q][p][c] = sum_s V_q[s][p] Out[s][c] for
 * q < nq, Cg[nq] = Out' Out -- (nq + 1) b b doubles; and its launch time (kernel + plane reduction) on nq random blocks */
int fpca_debug_k4_fused(fpca_ctx *ctx, int b, int nq, const double *V, const double *W, const double *C_in, double *Out, double *Cg);
int fpca_debug_k4_fused_bench(fpca_ctx *ctx, int b, int nq, int reps, double *ms_fused);
int fpca_debug_k4_bench(fpca_ctx *ctx, int b, int nq, int reps, double *ms_gram, double *ms_gemm);
/* diagnostic: placement census of an nwg-workgroup grid (256 threads, lds_bytes dynamic LDS each): out[2i] = HW_ID,
 * out[2i+1] = XCC_ID of workgroup i */
int fpca_debug_census(int nwg, uint64_t lds_bytes, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* FPCA_DEBUG_H */
