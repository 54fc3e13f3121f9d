// kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (kernels.hip).
// All pointers are DEVICE pointers; every launch is asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace fpca {
namespace kern {

// rewrite the pad bits of the last valid byte of every record to the "missing" code (N % 4 != 0)
// back-to-back records of np bytes -> rows of `pitch` bytes (upload path)
void repitch(const uint8_t *src, uint64_t np, uint64_t nrec, uint8_t *dst, size_t pitch, hipStream_t stream);
void fix_last_byte(uint8_t *packed, size_t pitch, uint64_t np, int valid_in_last, uint64_t P_g, hipStream_t stream);

// K1: per-SNP code counts -> mean, sd, lookup table (by raw PLINK code), sum of squares
//   lut [P_pad][4], mean/sd/sumsq [P_pad]; rows >= P_g untouched (must be pre-zeroed)
void bed_stats(const uint8_t *packed, size_t pitch, uint64_t N, uint64_t P_g, int stand_method, double *lut,
               double *mean, double *sd, double *sumsq, uint32_t *nmiss /* per-SNP missing calls, or null */, hipStream_t stream);
// lookup table from preloaded mean/sd (projection path, data.cpp:293-320)
void lut_from_meansd(const double *mean, const double *sd, uint64_t P_g, double *lut, hipStream_t stream);

// K2: Tpart[split][P_pad][b] = X^T B over the split's sample chunks.   b = 16*NT, NT in 1..4
//   B: [N_pad][b] row-major.  nsplit==1 writes the final T directly.  fp32: v_mfma_f32 products / fp32 sums over at most 512 samples (4 chunks),
//   accumulation folded into fp64 accumulators (FPCA_ACCUM_FP32), else everything fp64.
void xt_b(const uint8_t *packed, size_t pitch, const double *lut, const double *B, double *Tpart, uint64_t N_pad,
          uint64_t P_pad, int b, int nsplit, bool fp32, hipStream_t stream);
// K3: Ypart[split][N_pad][b] = X T over the split's SNP chunks.  T: [P_pad][b] row-major
void x_t(const uint8_t *packed, size_t pitch, const double *lut, const double *T, double *Ypart, uint64_t N_pad,
         uint64_t P_pad, int b, int nsplit, bool fp32, hipStream_t stream);
// out[i] = sum_s part[s*count + i]   (deterministic split-K combine)
void reduce_sum(const double *part, double *out, uint64_t count, int nsplit, hipStream_t stream);

// heuristics (host): number of splits for K2 / K3 given the problem and the chip (256 CUs, 2 WGs/CU)
int xt_b_splits(uint64_t N_pad, uint64_t P_pad, int b, bool fp32);
int x_t_splits(uint64_t N_pad, uint64_t P_pad, int b, bool fp32);

// Dense (in-memory fp64 matrix) path: Xd [P_pad][N_pad] fp64, one row per column of the caller's N x P matrix
// util.cpp:24-192 in place (method 0 none, 1 sd, 2 binom, 3 binom2, 4 center; NaN = missing); mean/sd/sumsq [P_pad]
void dense_standardise(double *Xd, uint64_t N_pad, uint64_t N, uint64_t P_g, int method, double *mean, double *sd,
                       double *sumsq, hipStream_t stream);
int xt_b_dense_splits(uint64_t N_pad, uint64_t P_pad);
int x_t_dense_splits(uint64_t N_pad, uint64_t P_pad);
void xt_b_dense(const double *Xd, const double *B, double *Tpart, uint64_t N_pad, uint64_t P_pad, int b, int nsplit,
                hipStream_t stream);
void x_t_dense(const double *Xd, const double *T, double *Ypart, uint64_t N_pad, uint64_t P_pad, int b, int nsplit,
               hipStream_t stream);

// K4 helpers on row-major [N_pad][b] blocks ------------------------------------------------------------
// part[4*split + wave][q][b][b] (row-major p,c) = A_q^T W over the rows of one wave of split `split` (gram_splits(N_pad,
// rows) workgroups of `rows` = gram_rows(N_pad, nq, b) rows each); `blocks` = device array of nq pointers
void gram(const double *const *blocks, int nq, const double *W, double *part, uint64_t N_pad, int b, int rows, hipStream_t stream);
int gram_rows(uint64_t N_pad, int nq, int b);
void k4_variant(int v); // lab switch (fpca_debug_k4_variant): 0 = round 4's K4 kernels, 1 = the tiled ones (default)
int gram_splits(uint64_t N_pad, int rows);
// Out = (Init ? Init : 0) + sum_q A_q C_q,   C: [nq][b][b] row-major (C_q[p][c]); Out may alias Init or any A_q
// gram_part non-null (only where block_gemm_gram_planes(N_pad, b) > 0): the launch also leaves that many partial planes [b][b] of
// Out' Out there (sum them with reduce_sum)
void block_gemm(const double *const *blocks, int nq, const double *C, const double *Init, double *Out, uint64_t N_pad,
                int b, hipStream_t stream, double *gram_part = nullptr);
int block_gemm_gram_planes(uint64_t N_pad, int b);
// Out = Init + sum_q A_q C_q, and from the same pass over the basis G_q = A_q' Out (q < nq), G_nq = Out' Out: update_gram_planes()
// partial planes [nq + 1][b][b] in gpart (0 planes: no fused kernel for this shape -- 16 columns, nq <= 28 only)
int update_gram_planes(uint64_t N_pad, int nq, int b);
void update_gram(const double *const *blocks, int nq, const double *C, const double *Init, double *Out, uint64_t N_pad, int b, double *gpart,
                 hipStream_t stream);
// uniform(-0.5, 0.5) entries for rows < N, zero for rows in [N, N_pad)
void fill_random(double *blk, uint64_t N, uint64_t rows, int b, uint64_t seed, hipStream_t stream, uint64_t row0 = 0);
// *out_bits = max(*out_bits, bits of max |a - scale b|) over n doubles (NaN counts as +inf); out_bits zeroed by the caller
void max_abs_diff(const double *a, const double *b, double scale, uint64_t n, unsigned long long *out_bits, hipStream_t stream);
// row-major [N_pad][b] block <-> column-major N x ncols (ld) device staging buffer
void block_to_colmajor(const double *blk, uint64_t N, int b, int ncols, double *out, uint64_t ld, hipStream_t stream);
void colmajor_to_block(const double *in, uint64_t ld, uint64_t N, uint64_t N_pad, int b, int ncols, double *blk,
                       hipStream_t stream);
// T [P_pad][b] row-major -> column-major P_g x ncols, scaled per column
void t_to_colmajor(const double *T, uint64_t P_g, int b, int ncols, const double *colscale, double *out, uint64_t ld,
                   hipStream_t stream);
void colmajor_to_t(const double *in, uint64_t ld, uint64_t P_g, uint64_t P_pad, int b, int ncols, double *T,
                   hipStream_t stream);

// synthetic genotype generator (synth.hpp model), one workgroup per SNP record
//   maf_model 0: ancestral frequency uniform in [0.05, 0.95); 1: rare-variant spectrum 0.001 + 0.499 u^3
//   missing_model 0: every call missing with probability miss_thr / 65536; 1: concentrated in a fraction conc_fp / 65536 of
//   the SNPs (10-30 % of their calls), the others at most 0.1 %; 2: per-SNP rates log-normal, median med_q32 (0.32 fixed point),
//   log2-sd sig2_fp (16.16)  (synth.hpp)
void synth_generate(uint8_t *packed, size_t pitch, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed,
                    int n_pop, uint32_t fst_fp, uint32_t miss_thr, hipStream_t stream, int maf_model = 0, int missing_model = 0,
                    uint32_t conc_fp = 0, uint64_t med_q32 = 0, uint32_t sig2_fp = 0);

// diagnostic: D(16x16) = A(16x4) B(4x16) through v_mfma_f64_16x16x4_f64 with this file's operand mapping
void mfma_layout_probe(const double *A, const double *B, double *D, hipStream_t stream);

// issue-rate ceiling of the FP64 MFMA (TFLOP/s) with `waves_per_simd` resident waves per SIMD
double mfma_peak_tflops(int waves_per_simd, int iters, int pattern, hipStream_t stream);
// diagnostic: hardware placement (HW_ID, XCC_ID per workgroup) of an nwg-workgroup grid with lds_bytes of LDS each
void census(uint32_t *d_out, int nwg, size_t lds_bytes, long long spin, hipStream_t stream);

// ---- exact-integer int8-sliced path (kernels_i8.hip) ----
// one int8 operand cut from an fp64 matrix V[rows_pad][b]: Q[S*b][rows_pad] digits of V * rowscale[row], weights colw,
// optional exact column sums; maxbits = per-column max |V * rowscale| as double bit patterns (input of i8_slice)
constexpr int I8_SHARDS = 8, I8_CS_STRIDE = 640; // accumulator sharding (see kernels_i8.hip)
struct SliceOp {
   const double *rowscale;      // null: none
   unsigned long long *maxbits; // [I8_SHARDS][64]
   int8_t *Q;
   double *colw;       // [gemm_i8_nsc_pad(S, b)]
   long long *colsum;  // [I8_SHARDS][I8_CS_STRIDE] or null; accumulated with atomics: zero it first
   double *copy64;     // optional: the scaled operand itself, row-major [rows][b] (what the sparse gathers of the missing-call
   float *copy32;      // route read: no per-entry row factor; fp32, column-scaled, when the slices carry no more than 30 bits anyway)
};
void i8_colmax(const double *V, uint64_t rows, int b, int nops, const SliceOp *ops, hipStream_t stream); // maxbits must be zeroed
void i8_slice(const double *V, uint64_t rows_pad, uint64_t rows, int b, int S, int nops, const SliceOp *ops, hipStream_t stream);
// row-sharded operand exchange: slices of a rank's own rows in a row-major layout, and what every rank makes of the gathered rows
void i8_maxbits_fold(const unsigned long long *bits, double *out64, hipStream_t stream);      // [I8_SHARDS][64] -> 64 doubles
void i8_maxbits_set(const double *all, int G, unsigned long long *bits, hipStream_t stream);   // max over [G][64] -> shard 0
void i8_slice_rows(const double *V, uint64_t rows, int b, int S, const SliceOp &op, int8_t *Qrm, hipStream_t stream); // -> Qrm[rows][S*b], op.colw
// -> op.Q, op.colsum; optionally the scaled operand itself (rows < rows_valid) for the sparse gathers: fp32 (copy32) or fp64 (copy64)
void i8_unpack_slices(const int8_t *Qrm, uint64_t rows_pad, int b, int S, const SliceOp &op, hipStream_t stream, uint64_t rows_valid = 0,
                      float *copy32 = nullptr, double *copy64 = nullptr);
void i8_dequant_rows(const int8_t *Qrm, uint64_t rows, int b, int S, const SliceOp &op, float *copy32, double *copy64, hipStream_t stream);
int gemm_i8_nsc_pad(int S, int b); // rows of a Q operand: S*b rounded up to the 256-column workgroup tile
size_t gemm_i8_workspace_doubles(uint64_t rows_pad, uint64_t k_pad, int S, int b, bool two);
// out[rows_pad][b] = recombined ( (G.M) Qg' , M Qm' ); mean/sd non-null: K2 flavour (per-row standardisation)
void gemm_i8(const uint8_t *packed, size_t pitch, const int8_t *Qg, const int8_t *Qm, const double *wg, const double *wm,
             const long long *colsum_m, const double *mean, const double *sd, double *out, double *ws, uint64_t rows_pad, uint64_t k_pad,
             uint64_t rows_valid, int mode /* 0 full, 1 skip E blocks without a missing genotype, 2 G.M alone */,
             const double *eplane /* with mode 2: E'Q [rows_pad][b] from sparse_rows_sum, or null if nothing is missing */, int b, int S,
             const SliceOp *next_ops, hipStream_t stream, hipEvent_t *gemm_events = nullptr, hipEvent_t before_combine = nullptr,
             bool e_only = false /* mode 2 only: multiply the missing-indicator matrix E instead of G.M; out = E Q */);
// the hybrid missing-indicator route's row shuffles (kernels_i8.hip)
void gather_packed_rows(const uint8_t *src, size_t pitch, const uint32_t *idx, uint32_t nidx, uint32_t rows_out, uint8_t *dst, hipStream_t stream);
void patch_missing_rows(uint8_t *packed, size_t pitch, const uint32_t *idx, uint32_t nidx, hipStream_t stream);
void scatter_packed_rows(const uint8_t *src, size_t pitch, const uint32_t *idx, uint32_t nidx, uint8_t *packed, hipStream_t stream);
void gather_scaled_rows(const double *V, const double *scale, const uint32_t *idx, uint32_t nidx, uint64_t rows_out, int b, double *dst, hipStream_t stream);
void scatter_rows(const double *src, const uint32_t *idx, uint32_t nidx, int b, double *dst, hipStream_t stream);
// index lists of the missing calls of 2-bit records (positions < ncols), and the gather-sum over them
void count_missing(const uint8_t *packed, size_t pitch, uint64_t ncols, uint64_t nrec, uint32_t *cnt, hipStream_t stream);
void fill_missing(const uint8_t *packed, size_t pitch, uint64_t ncols, uint64_t nrec, const uint32_t *ptr, uint32_t *idx, hipStream_t stream);
void sparse_rows_sum(const uint32_t *ptr, const uint32_t *idx, const double *V, const double *rowscale, int b, uint64_t nrec,
                     uint64_t rows_out, double *out, hipStream_t stream, const double *init = nullptr /* [rows_out][b] added to the sums */,
                     bool short_lists = false /* many short lists (per sample): the batched index reads */,
                     double avg_len = 0 /* entries per list, if known: a dozen or so takes the several-rows-per-wave kernel */);
// the same over the fp32 rows k_slice leaves (SliceOp::copy32: each column scaled into (-2, 2) by its slice exponent; colw = that
// operand's slice weights, which carry the exponent back)
void sparse_rows_sum_f32(const uint32_t *ptr, const uint32_t *idx, const float *V, const double *colw, int b, uint64_t nrec, uint64_t rows_out,
                         double *out, hipStream_t stream, const double *init = nullptr, bool short_lists = false, double avg_len = 0);
void i8_rowscales(const double *mean, const double *sd, uint64_t P_g, uint64_t P_pad, double *inv_sd, double *mu_inv_sd,
                  hipStream_t stream);
void transpose_packed(const uint8_t *in, size_t pitch_in, uint64_t N_pad, uint64_t P_pad, uint8_t *out, size_t pitch_out,
                      hipStream_t stream);
double mfma_i8_peak_tops(int waves_per_simd, int iters, uint32_t fill, hipStream_t stream);
double mfma_valu_mix_tflops(int vpm, bool burst, int ldsr, int waves_per_simd, int iters, hipStream_t stream);
double mfma_fp_peak_tflops(int kind, int waves_per_simd, int iters, uint32_t fill, hipStream_t stream); // 0 f32 16x16x4, 1 f32 32x32x2, 2 f64 16x16x4
double mfma_i8_mix_tops(int mix, int iters, hipStream_t stream);
void mfma_i8_probe(const int8_t *A, const int8_t *Bt, int *D, hipStream_t stream);

} // namespace kern
} // namespace fpca
