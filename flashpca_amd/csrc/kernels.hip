// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the flashpca PCA hot path.
//
// The reference computes  y = sum_blocks X_b (X_b' x)  by decoding every SNP block of the PLINK 2-bit
// stream into a dense fp64 N x bs matrix through a per-SNP 4-entry lookup table and running two Eigen
// GEMVs over it (svdwide.cpp:21-68, data.cpp:215-335).  Here the packed stream stays resident in HBM and
// the decode + standardisation is fused into two tall-skinny FP64 MFMA GEMMs on b columns at a time:
//
//   K1 bed_stats : per-SNP code counts -> mean, sd, lookup table, sum of squares   (data.cpp:257-322)
//   K2 xt_b      : T[P x b] = X' B      (first GEMV of svdwide.cpp:42-43, crossprod of :122-153)
//   K3 x_t       : Y[N x b] = X  T      (second GEMV of svdwide.cpp:42-43, prod of :193-226)
//   K4 helpers   : Gram / block GEMM / fill / layout changes on N x b blocks for the host eigensolver
//
// MFMA: v_mfma_f64_16x16x4_f64, D(16x16) += A(16x4) B(4x16), one f64 of A and of B per lane:
//   lane l holds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the 4 accumulator registers r hold
//   D[row = (l>>4) + 4r][col = l&15].   (cdna_hip_programming.md section 3; checked at run time by
//   mfma_layout_probe / tests/test_gpu_kernels.py)
// A wave is 64 lanes; a workgroup is 4 waves (one per SIMD); 2 workgroups per CU hide the staging.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include "synth.hpp"

namespace fpca {
namespace kern {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));   // 16-byte load/store unit (native vector: stays in VGPRs)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#define FPCA_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define HIP_CHECK_LAUNCH()                                                                         \
   do {                                                                                             \
      hipError_t e__ = hipGetLastError();                                                           \
      if (e__ != hipSuccess) throw Error(-3, std::string("kernel launch failed: ") + hipGetErrorString(e__)); \
   } while (0)

// ------------------------------------------------------------------------------------------------
// padding fix-up: pad bits of the last valid byte -> "01" (missing)
__global__ void k_fix_last_byte(uint8_t *packed, size_t pitch, uint64_t np, uint32_t keep_mask, uint64_t P_g)
{
   uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (r >= P_g) return;
   uint8_t *p = packed + r * pitch + (np - 1);
   *p = (uint8_t)((*p & keep_mask) | (PAD_BYTE & ~keep_mask));
}

void fix_last_byte(uint8_t *packed, size_t pitch, uint64_t np, int valid_in_last, uint64_t P_g, hipStream_t stream)
{
   if (valid_in_last <= 0 || valid_in_last >= 4 || P_g == 0) return;
   uint32_t keep = (1u << (2 * valid_in_last)) - 1u;
   unsigned grid = (unsigned)((P_g + 255) / 256);
   hipLaunchKernelGGL(k_fix_last_byte, dim3(grid), dim3(256), 0, stream, packed, pitch, np, keep, P_g);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// records of np bytes, back to back (the .bed layout) -> pitched rows; one workgroup per record, byte granularity on the
// source side (np is arbitrary), so this is only for the upload path (a 2-D host-to-device copy runs at ~18 GB/s, a 1-D
// copy into a staging buffer at 57 GB/s, and this kernel at HBM speed)
__global__ __launch_bounds__(256) void k_repitch(const uint8_t *__restrict__ src, uint64_t np, uint8_t *__restrict__ dst, size_t pitch)
{
   const uint8_t *s = src + (uint64_t)blockIdx.x * np;
   uint8_t *d = dst + (uint64_t)blockIdx.x * pitch;
   const uint64_t mis = (16 - ((uint64_t)s & 15)) & 15; // bytes before the first 16-byte boundary of the source
   for (uint64_t i = threadIdx.x; i < mis && i < np; i += 256) d[i] = s[i];
   if (np > mis) {
      const uint64_t n16 = (np - mis) / 16;
      // source aligned, destination not necessarily: 16-byte loads, byte-wise funnel is not worth it -- 4 x 4-byte stores
      // when the destination offset is 4-byte aligned, else bytes
      const uint4 *s16 = reinterpret_cast<const uint4 *>(s + mis);
      uint8_t *dd = d + mis;
      if ((((uint64_t)dd) & 3) == 0) {
         for (uint64_t i = threadIdx.x; i < n16; i += 256) {
            const uint4 v = s16[i];
            uint32_t *o = reinterpret_cast<uint32_t *>(dd + i * 16);
            o[0] = v.x;
            o[1] = v.y;
            o[2] = v.z;
            o[3] = v.w;
         }
      } else {
         for (uint64_t i = threadIdx.x; i < n16 * 16; i += 256) dd[i] = s[mis + i];
      }
      for (uint64_t i = mis + n16 * 16 + threadIdx.x; i < np; i += 256) d[i] = s[i];
   }
}

void repitch(const uint8_t *src, uint64_t np, uint64_t nrec, uint8_t *dst, size_t pitch, hipStream_t stream)
{
   if (!nrec) return;
   hipLaunchKernelGGL(k_repitch, dim3((unsigned)nrec), dim3(256), 0, stream, src, np, dst, pitch);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K1 bed_stats.  One workgroup per SNP record; 16-byte coalesced loads of the packed stream; the four
// 2-bit codes are counted with popcount on the even/odd bit planes.  Padding bytes are "01" so they only
// inflate the missing count, which is not used.  HBM-bound: reads pitch bytes per SNP once.
__device__ __forceinline__ void count_word(uint32_t w, uint32_t &c01, uint32_t &c10, uint32_t &c11)
{
   const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
   c11 += __popc(lo & hi);
   c10 += __popc(hi & ~lo);
   c01 += __popc(lo & ~hi);
}

__device__ __forceinline__ void make_lut(double mean, double sd, double *lut4)
{
   // data.cpp:299-320: table indexed by RAW code; all-zero when sd <= VAR_TOL (or NaN)
   double v0 = 0, v2 = 0, v3 = 0;
   if (sd > 1e-9) {
      v3 = (0.0 - mean) / sd;
      v2 = (1.0 - mean) / sd;
      v0 = (2.0 - mean) / sd;
   }
   lut4[0] = v0;
   lut4[1] = 0.0;
   lut4[2] = v2;
   lut4[3] = v3;
}

__global__ __launch_bounds__(256) void k_bed_stats(const uint8_t *__restrict__ packed, size_t pitch, uint64_t N, int stand_method,
                                                    double *__restrict__ lut, double *__restrict__ mean_out,
                                                    double *__restrict__ sd_out, double *__restrict__ sumsq_out,
                                                    uint32_t *__restrict__ nmiss_out)
{
   const uint64_t snp = blockIdx.x;
   const uint4 *row = reinterpret_cast<const uint4 *>(packed + snp * pitch);
   const uint32_t nvec = (uint32_t)(pitch / 16);
   uint32_t c01 = 0, c10 = 0, c11 = 0;
   for (uint32_t v = threadIdx.x; v < nvec; v += 256) {
      const uint4 q = row[v];
      count_word(q.x, c01, c10, c11);
      count_word(q.y, c01, c10, c11);
      count_word(q.z, c01, c10, c11);
      count_word(q.w, c01, c10, c11);
   }
   for (int off = 32; off > 0; off >>= 1) {
      c01 += __shfl_down(c01, off);
      c10 += __shfl_down(c10, off);
      c11 += __shfl_down(c11, off);
   }
   __shared__ uint32_t red[4][3];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   if (lane == 0) {
      red[wave][0] = c01;
      red[wave][1] = c10;
      red[wave][2] = c11;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      uint64_t n01 = 0, n10 = 0, n11 = 0;
      for (int w = 0; w < 4; w++) {
         n01 += red[w][0];
         n10 += red[w][1];
         n11 += red[w][2];
      }
      const uint64_t cells = (uint64_t)pitch * 4;
      const uint64_t n00 = cells - n01 - n10 - n11; // dosage 2
      const uint64_t ngood = n00 + n10 + n11;
      // data.cpp:266-275: mean of the non-missing dosages (exact integer sum, one rounding in the divide)
      const double mean = (double)(2 * n00 + n10) / (double)ngood;
      const double pp = mean / 2.0;
      double sd;
      if (stand_method == 2)
         sd = sqrt(pp * (1 - pp)); // STANDARDISE_BINOM  (data.cpp:279)
      else
         sd = sqrt(2.0 * pp * (1 - pp)); // STANDARDISE_BINOM2 (data.cpp:281)
      double l[4];
      make_lut(mean, sd, l);
      double *lp = lut + snp * 4;
      lp[0] = l[0];
      lp[1] = l[1];
      lp[2] = l[2];
      lp[3] = l[3];
      mean_out[snp] = mean;
      sd_out[snp] = sd;
      if (nmiss_out) nmiss_out[snp] = (uint32_t)(n01 - (cells - N)); // missing calls among the N samples (padding is "01" too)
      // sum_i X_ij^2 in closed form from the counts (svdwide.cpp:44-45 sums the dense block)
      sumsq_out[snp] = (double)n00 * l[0] * l[0] + (double)n10 * l[2] * l[2] + (double)n11 * l[3] * l[3];
   }
}

void bed_stats(const uint8_t *packed, size_t pitch, uint64_t N, uint64_t P_g, int stand_method, double *lut,
               double *mean, double *sd, double *sumsq, uint32_t *nmiss, hipStream_t stream)
{
   if (P_g == 0) return;
   hipLaunchKernelGGL(k_bed_stats, dim3((unsigned)P_g), dim3(256), 0, stream, packed, pitch, N, stand_method, lut, mean,
                      sd, sumsq, nmiss);
   HIP_CHECK_LAUNCH();
}

__global__ void k_lut_from_meansd(const double *mean, const double *sd, uint64_t P_g, double *lut)
{
   uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (j >= P_g) return;
   double l[4];
   make_lut(mean[j], sd[j], l);
   for (int c = 0; c < 4; c++) lut[j * 4 + c] = l[c];
}

void lut_from_meansd(const double *mean, const double *sd, uint64_t P_g, double *lut, hipStream_t stream)
{
   if (P_g == 0) return;
   hipLaunchKernelGGL(k_lut_from_meansd, dim3((unsigned)((P_g + 255) / 256)), dim3(256), 0, stream, mean, sd, P_g, lut);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// The two GEMM kernels exist in two arithmetic flavours, selected by the context's `accum` setting:
//   RT = double : v_mfma_f64_16x16x4_f64, everything in fp64 (default; FPCA_ACCUM_FP64)
//   RT = float  : v_mfma_f32_16x16x4_f32 (twice the MFMA rate): the table of standardised values and the B / T tile
//                 are rounded to fp32 in LDS, products and the accumulation within FOLD_EVERY = 4 LDS chunks (512 samples / 256 SNPs)
//                 run in fp32, and every chunk's partial sums are added into fp64 accumulators, so the rounding error
//                 does not grow with N or P (BASELINE config 5, "fp32 accumulate (tolerance study)"; FPCA_ACCUM_FP32)
// Both MFMAs take A[i = lane&15][k = lane>>4] and B[k = lane>>4][j = lane&15], one value per lane; they differ in the
// C/D map: fp64 register r holds row (lane>>4) + 4r, fp32 register r holds row 4 (lane>>4) + r.
constexpr int FOLD_EVERY = 4; // mixed mode: LDS chunks between two folds of the fp32 partial sums into fp64 (power of two)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename RT> struct Mma;
template <> struct Mma<double> {
   typedef d4 acc_t;
   static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
   static __device__ __forceinline__ int row(int kq, int r) { return kq + 4 * r; }
};
template <> struct Mma<float> {
   typedef f4 acc_t;
   static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
   static __device__ __forceinline__ int row(int kq, int r) { return 4 * kq + r; }
};

// two table values fetched by one LDS read (pair decode of two consecutive samples)
template <typename RT> struct Pair;
template <> struct Pair<double> {
   typedef d2 type;
   static __device__ __forceinline__ d2 make(double a, double b) { return (d2){a, b}; }
};
template <> struct Pair<float> {
   typedef f2 type;
   static __device__ __forceinline__ f2 make(double a, double b) { return (f2){(float)a, (float)b}; }
};

// store a 16-byte piece of fp64 data (2 doubles) at logical element index 2*idx of an RT-typed LDS array
__device__ __forceinline__ void lds_put2(double *base, int idx, d2 v) { reinterpret_cast<d2 *>(base)[idx] = v; }
__device__ __forceinline__ void lds_put2(float *base, int idx, d2 v)
{
   reinterpret_cast<f2 *>(base)[idx] = (f2){(float)v.x, (float)v.y};
}

// Table gather of the pair decode with a TWO-instruction address (v_bfe_u32 + v_lshl_add_u32; the compiler's own form is shift, and,
// add).  Round 6 measured what plain VALU / LDS instructions cost a matrix-instruction stream (profiles/r06_mfma_valu_mix.txt):
// ~0.6 issue slots of 4 cycles each -- of a 32-cycle fp32 MFMA -- even with four waves per SIMD taking turns, and ~1.2 more per switch
// from MFMAs to other instructions; the FP kernels are bound by exactly that, not by power or memory (1,300 W, 2.38 GHz).
// base = LDS byte address of the lane's row 0 of the table; entry (nibble at bit `off` of w) is SH bytes-log2 apart.
template <typename PT, int SH, int OFF, int IMM>
__device__ __forceinline__ PT lds_pair_gather(uint32_t base, uint32_t w)
{
   uint32_t a;
   asm("v_bfe_u32 %0, %1, %2, 4\n\tv_lshl_add_u32 %0, %0, %3, %4" : "=&v"(a) : "v"(w), "n"(OFF), "n"(SH), "v"(base));
   return *reinterpret_cast<const __attribute__((address_space(3))) PT *>((uintptr_t)(a + IMM));
}
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>)
{
   (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f)
{
   sfor_impl(f, std::make_integer_sequence<int, N>{});
}
#define FPCA_ENV_INT(name, dflt) ((FPCA_TEST_ENV(name) && *FPCA_TEST_ENV(name)) ? atoi(FPCA_TEST_ENV(name)) : (dflt))

// ------------------------------------------------------------------------------------------------
// K2 xt_b:  T[snp][c] = sum_s X[s][snp] B[s][c]
//   workgroup = 256 SNPs x all b columns, wave = 64 SNPs (4 m-tiles of 16), K = samples.
//   MFMA roles: A[i = SNP in m-tile][k] = decoded genotype, B[k][j = column] = B tile from LDS.
//   Per KC-sample chunk each lane (i, kq) loads KC/16 packed bytes of ITS SNP record (samples kq*KC/4 .. of the
//   chunk) straight into registers -- the K order inside a chunk is permuted per lane group so that no cross-lane
//   shuffle is needed -- and the B tile (KC x b, contiguous in HBM) is staged through LDS once per workgroup.  The
//   workgroup's 256 x 4 table of standardised values also sits in LDS: decode = one LDS read indexed by the raw 2-bit
//   code.  The loads of chunk c+1 are issued before the MFMAs of chunk c.  Split-K over samples (blockIdx.y) with a
//   deterministic combine when there are too few SNP tiles to fill the chip.
template <int NT> struct XtCfg {
   // samples per LDS chunk: keeps the prefetch registers <= 32.  (Round 5 measured 256 for 16 columns -- half the barriers, staging
   // passes and fp64 folds per MFMA -- at 500,000 x 100,000: fp64 24.1 -> 24.8 ms, fp32 14.3 -> 14.4; the 40 extra registers cost a
   // resident wave per SIMD, which is worth more.  Not kept; K3 with 128-SNP chunks likewise: 24.4 -> 25.7 ms.)
   static constexpr int KC = (NT <= 2) ? 128 : 64;
};

template <typename RT, int NT, int MT /* m-tiles (16 SNPs each) per wave; workgroup tile = 64*MT SNPs */, int KC /* samples per LDS chunk */>
__global__ __launch_bounds__(256, 2) void k_xt_b(const uint8_t *__restrict__ packed, size_t pitch,
                                                  const double *__restrict__ lut, const double *__restrict__ B,
                                                  double *__restrict__ Tpart, uint64_t P_pad, int chunks_total,
                                                  int chunks_per_split)
{
   typedef typename Mma<RT>::acc_t acc_t;
   constexpr bool MIXED = sizeof(RT) == 4;
   constexpr int b = 16 * NT;
   constexpr int TILE = 64 * MT;
   constexpr int NW = KC / 64;                    // packed dwords per lane per m-tile per chunk (16 samples each)
   constexpr int NLOAD = KC * b * 8 / (256 * 16); // 16-byte pieces of the (fp64) B tile per thread
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
   RT *sB = reinterpret_cast<RT *>(smem_raw); // [KC][b]
   RT *sLut = sB + KC * b;                    // [TILE][4]

   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const uint64_t snp0 = (uint64_t)blockIdx.x * TILE + (uint64_t)wave * (16 * MT);
   const int c_begin = blockIdx.y * chunks_per_split;
   int c_end = c_begin + chunks_per_split;
   if (c_end > chunks_total) c_end = chunks_total;
   // The workgroup's table in LDS is laid out [16-SNP group][code][SNP in group]: the 16 lanes of an MFMA row group -- 16 different
   // SNPs, whatever their codes -- then read 16 consecutive words = every bank once.  (Rounds 1-4 kept the HBM layout [SNP][code]:
   // lanes li and li + 4 shared their banks whenever their codes agreed, 31-38 % of the LDS cycles of this kernel were conflicts.)
   // 16 columns (PAIR): the table holds PAIRS of values indexed by the 4 bits of two consecutive samples' codes, so that one LDS
   // read decodes two k-steps of an m-tile (see K3); [16-SNP group][pair index][SNP in group], conflict-free the same way.  Wider
   // blocks keep the per-sample table: there every gather already feeds NT MFMAs, and 16 x the table would not leave room for
   // three workgroups per CU.
   typedef typename Pair<RT>::type pair_t;
   constexpr bool PAIR = NT == 1;
   pair_t *sLutP = reinterpret_cast<pair_t *>(sLut);
   if (tid < TILE) { // one SNP's 4-entry table per thread; visible after the first barrier of the chunk loop
      const d2 *lsrc = reinterpret_cast<const d2 *>(lut + ((uint64_t)blockIdx.x * TILE + tid) * 4);
      const d2 e01 = lsrc[0], e23 = lsrc[1];
      if (PAIR) {
         const double l4[4] = {e01.x, e01.y, e23.x, e23.y};
         pair_t *dst = sLutP + (tid >> 4) * 256 + (tid & 15);
#pragma unroll
         for (int idx = 0; idx < 16; idx++) dst[idx * 16] = Pair<RT>::make(l4[idx & 3], l4[idx >> 2]);
      } else {
         RT *dst = sLut + (tid >> 4) * 64 + (tid & 15);
         dst[0] = (RT)e01.x;
         dst[16] = (RT)e01.y;
         dst[32] = (RT)e23.x;
         dst[48] = (RT)e23.y;
      }
   }
   // SUPER-CHUNKS (round 6; KC = 128 only): a lane's packed words of TWO consecutive chunks are one 16-byte load -- lane group kq owns
   // bytes [16 kq, 16 kq + 16) of the row's 64-byte segment -- i.e. half the load instructions, each touching 16 rows x 64 B instead
   // of 16 x 32 B (the packed-word loads were 0.7 ms of the 13.3 ms fp32 launch, profiles/r06_fp_k2_ablation.txt).  Sub-chunk s of
   // super-chunk S then holds, for lane group kq, samples S 2KC + kq KC/2 + s KC/4 + [0, KC/4): the B tile of a sub-chunk is four
   // 32-row segments of B instead of one contiguous piece, still 4 KB each.
   constexpr int SUP = (NW == 2) ? 2 : 1;
   constexpr int LB = SUP * KC / 16; // bytes per lane per super-chunk
   const uint8_t *rowp[MT];
#pragma unroll
   for (int m = 0; m < MT; m++) rowp[m] = packed + (snp0 + m * 16 + li) * pitch + kq * LB;

   acc_t acc[MT][NT];
   d4 acc64[MIXED ? MT : 1][MIXED ? NT : 1];
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
         acc[m][nt] = (acc_t){0, 0, 0, 0};
         if (MIXED) acc64[m][nt] = (d4){0.0, 0.0, 0.0, 0.0};
      }

   uint32_t pk_next[MT][NW * SUP];
   d2 breg[NLOAD];
   auto issue_p = [&](int sc) { // packed words of super-chunk sc
#pragma unroll
      for (int m = 0; m < MT; m++) {
         const uint32_t *pp = reinterpret_cast<const uint32_t *>(rowp[m] + (size_t)sc * (SUP * KC / 4));
         if constexpr (NW * SUP == 4) {
            const u4 w = *reinterpret_cast<const u4 *>(pp);
            pk_next[m][0] = w.x, pk_next[m][1] = w.y, pk_next[m][2] = w.z, pk_next[m][3] = w.w;
         } else {
#pragma unroll
            for (int h = 0; h < NW * SUP; h++) pk_next[m][h] = pp[h];
         }
      }
   };
   auto issue_b = [&](int cc) { // B tile of chunk cc (sub-chunk cc % SUP of super-chunk cc / SUP)
      if constexpr (SUP == 1) {
         const d2 *src = reinterpret_cast<const d2 *>(B + (size_t)cc * KC * b);
#pragma unroll
         for (int r = 0; r < NLOAD; r++) breg[r] = src[tid + 256 * r];
      } else {
         constexpr int RPR = 512 / b; // tile rows per 256-thread pass; a pass stays inside one 32-row segment
         static_assert((KC / 4) % RPR == 0, "B tile passes and lane-group segments");
         const size_t row0 = (size_t)(cc / SUP) * (SUP * KC) + (size_t)(cc % SUP) * (KC / 4);
#pragma unroll
         for (int r = 0; r < NLOAD; r++) {
            const int k0 = r * RPR; // first tile row of this pass: lane group k0 / (KC/4), row k0 % (KC/4) of its segment
            const d2 *src = reinterpret_cast<const d2 *>(B + (row0 + (size_t)(k0 / (KC / 4)) * (SUP * KC / 4) + k0 % (KC / 4)) * b);
            breg[r] = src[tid];
         }
      }
   };
   if (c_begin < c_end) {
      issue_p(c_begin / SUP);
      issue_b(c_begin);
   }

   const RT *sB_lane = sB + (size_t)((KC / 4) * kq) * b + li;
   const RT *sLut_lane = sLut + (size_t)(wave * MT) * 64 + li;        // m-tile m, code c: sLut_lane[m * 64 + c * 16]
   const pair_t *sLutP_lane = sLutP + (size_t)(wave * MT) * 256 + li; // PAIR: m-tile m, pair index i: sLutP_lane[m * 256 + i * 16]

   uint32_t pk[MT][NW * SUP];
   for (int c0 = c_begin; c0 < c_end; c0 += SUP) { // (c_begin and c_end are multiples of SUP: launch_xt_b)
      sfor<SUP>([&](auto subc) {
         constexpr int sub = decltype(subc)::value;
         const int c = c0 + sub;
         __syncthreads(); // every wave has finished reading the previous B tile
#pragma unroll
         for (int r = 0; r < NLOAD; r++) lds_put2(sB, tid + 256 * r, breg[r]);
         if constexpr (sub == 0) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
               for (int h = 0; h < NW * SUP; h++) pk[m][h] = pk_next[m][h];
         }
         __syncthreads();
         if (c + 1 < c_end) issue_b(c + 1);
         if (sub == SUP - 1 && c0 + SUP < c_end) issue_p(c0 / SUP + 1);
         constexpr int H0 = sub * NW; // first packed dword of this sub-chunk
         // k-step t + 1's operands (B fragment, MT table gathers) are read while step t's MFMAs run (see K3); PAIR: two k-steps at a time
         if constexpr (PAIR) {
            // (round 6) gathers first, the B fragment LAST: the first MFMA of a burst needs the B value, so ONE s_waitcnt -- for the
            // youngest load of its batch -- precedes the burst and none interrupts it (LDS answers in order); two-instruction addresses
            RT av[2][2][MT], bv[2][2][NT];
            constexpr int PSH = sizeof(pair_t) == 8 ? 7 : 8; // log2 bytes between the rows of two pair indices: 16 pairs per row
            const uint32_t lut_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) pair_t *)sLutP_lane;
            // the fetch group of a pair of k-steps goes out in two halves, one before each half of its 2 MT MFMAs (shorter groups:
            // 13.35 -> 13.2 ms at fp32 / 16 columns, like K3's; profiles/r06_fp_kernels.txt)
            auto fetch2 = [&](auto tpc, auto slotc, auto halfc) {
               constexpr int tp_ = decltype(tpc)::value, slot_ = decltype(slotc)::value, half_ = decltype(halfc)::value; // half_: 0 / 1 = first / second half of the group, 2 = all
               constexpr int M0 = half_ == 1 ? MT / 2 : 0, M1 = half_ == 0 ? MT / 2 : MT;
               sfor<M1 - M0>([&](auto mc) {
                  constexpr int m = M0 + decltype(mc)::value;
                  const pair_t pr_ = lds_pair_gather<pair_t, PSH, (4 * tp_) % 32, m * 256 * (int)sizeof(pair_t)>(lut_base, pk[m][H0 + (2 * tp_) / 16]);
                  av[slot_][0][m] = pr_.x;
                  av[slot_][1][m] = pr_.y;
               });
               if constexpr (half_ != 0) {
#pragma unroll
                  for (int w_ = 0; w_ < 2; w_++)
#pragma unroll
                     for (int nt = 0; nt < NT; nt++) bv[slot_][w_][nt] = sB_lane[(size_t)(2 * tp_ + w_) * b + nt * 16];
               }
            };
            fetch2(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            sfor<8 * NW>([&](auto tpc) {
               constexpr int tp = decltype(tpc)::value;
               if constexpr (tp + 1 < 8 * NW)
                  fetch2(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, (tp + 1) & 1>{}, std::integral_constant<int, 0>{});
               __builtin_amdgcn_sched_barrier(0);
#pragma unroll
               for (int w = 0; w < 2; w++) {
#pragma unroll
                  for (int m = 0; m < MT; m++)
#pragma unroll
                     for (int nt = 0; nt < NT; nt++) acc[m][nt] = Mma<RT>::mma(av[tp & 1][w][m], bv[tp & 1][w][nt], acc[m][nt]);
                  if (w == 0) {
                     __builtin_amdgcn_sched_barrier(0);
                     if constexpr (tp + 1 < 8 * NW)
                        fetch2(std::integral_constant<int, tp + 1>{}, std::integral_constant<int, (tp + 1) & 1>{}, std::integral_constant<int, 1>{});
                     __builtin_amdgcn_sched_barrier(0);
                  }
               }
               __builtin_amdgcn_sched_barrier(0);
            });
         } else {
            RT av[2][MT], bv[2][NT];
#define FPCA_XTB_FETCH(tt_, slot_)                                                                                              \
      {                                                                                                                            \
         _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bv[slot_][nt] = sB_lane[(size_t)(tt_) * b + nt * 16];                   \
         _Pragma("unroll") for (int m = 0; m < MT; m++) av[slot_][m] = sLut_lane[m * 64 + (((pk[m][H0 + (tt_) / 16] >> (2 * ((tt_) % 16))) & 3u) << 4)]; \
      }
            FPCA_XTB_FETCH(0, 0);
#pragma unroll
            for (int tt = 0; tt < 16 * NW; tt++) {
               if (tt + 1 < 16 * NW) FPCA_XTB_FETCH(tt + 1, (tt + 1) & 1);
               __builtin_amdgcn_sched_barrier(0);
#pragma unroll
               for (int m = 0; m < MT; m++)
#pragma unroll
                  for (int nt = 0; nt < NT; nt++) acc[m][nt] = Mma<RT>::mma(av[tt & 1][m], bv[tt & 1][nt], acc[m][nt]);
               __builtin_amdgcn_sched_barrier(0);
            }
#undef FPCA_XTB_FETCH
         }
         // fold the fp32 partial sums into the fp64 accumulators every FOLD_EVERY chunks (and at the end): fp32 sums then run over at most
         // 512 samples whatever N is -- the rounding error does not grow with N -- while the fold (4 converts + 4 fp64 adds per
         // accumulator register, 5 % of the kernel when made every chunk; round 5) stays off the MFMA-bound path
         if (MIXED && ((c & (FOLD_EVERY - 1)) == FOLD_EVERY - 1 || c + 1 == c_end)) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
               for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                  for (int r = 0; r < 4; r++) acc64[m][nt][r] += (double)acc[m][nt][r];
                  acc[m][nt] = (acc_t){0, 0, 0, 0};
               }
         }
         });
   }

   double *Tout = Tpart + (size_t)blockIdx.y * P_pad * b;
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) {
            const uint64_t row = snp0 + m * 16 + Mma<RT>::row(kq, r);
            Tout[row * b + nt * 16 + li] = MIXED ? acc64[m][nt][r] : (double)acc[m][nt][r];
         }
}

// Split-K plan of an MFMA-bound GEMM.  The co-resident workgroups of a CU share its four matrix pipes, so what a launch takes is
// the work the BUSIEST CU serialises through them -- ceil(workgroups / 256) x (chunks per workgroup + fill / drain) --, not the
// number of "rounds" of resident workgroups: 391 SNP tiles unsplit put two workgroups on 135 CUs and one on 121 (0.76 of the
// chip; this was the 16-column fp64 K2 of rounds 1-4), cut 13 ways they are 5,083 workgroups = 20 per CU with the last CU
// 1 % short.  More splits cost one partial plane each (a write + a read of the output in the combine pass) and ~1.5 chunks of
// fill / drain per workgroup; a CU left with a single workgroup cannot hide its staging behind a neighbour (x 1.4).
// `slots` = workgroups the chip holds at once: plans that fit are placed round-robin (the count per CU is exact); larger grids
// are fed as slots free up and balance themselves.
static int pick_splits(uint64_t tiles, uint64_t chunks, int min_chunks, int max_splits, uint64_t slots)
{
   uint64_t best_t = ~0ull;
   int best = 1;
   for (int s = 1; s <= max_splits && (uint64_t)s <= chunks; s++) {
      uint64_t cps = (chunks + s - 1) / s;
      if (s > 1 && cps < (uint64_t)min_chunks) break;
      uint64_t seff = (chunks + cps - 1) / cps;
      const uint64_t wgs = tiles * seff, per_cu = (wgs + 255) / 256;
      // time in 1/64 chunk units
      uint64_t t = per_cu * (cps * 64 + 96);
      if (wgs > slots) t += (cps * 64 + 96) / 2; // (dynamic placement: the last workgroups start when a slot frees, half a one late on average)
      if (per_cu < 2) t = t * 7 / 5;
      t += seff > 1 ? seff * 24 * ((tiles + 255) / 256) : 0; // the combine pass streams `seff` planes of the output
      if (t < best_t) {
         best_t = t;
         best = (int)seff;
      }
   }
   return best;
}

int xt_b_splits(uint64_t N_pad, uint64_t P_pad, int b, bool fp32)
{
   const int kc = b <= 32 ? 128 : 64; // XtCfg<NT>::KC
   const int tile = ((fp32 && b >= 48) || (!fp32 && b == 16)) ? 128 : 256; // 64 XtMt<RT, NT>::MT
   static const int forced = FPCA_ENV_INT("FPCA_XT_SPLITS", 0);
   if (forced > 0) return (int)std::min<uint64_t>(forced, N_pad / kc);
   // fp64 with b <= 32 needs 145 VGPRs and 40 KB of LDS: three workgroups per CU are resident
   const uint64_t slots = (!fp32 && b <= 32) ? 768 : 512;
   return pick_splits(P_pad / tile, N_pad / kc, 4, 64, slots);
}

template <typename RT, int NT> struct XtMt {
   // m-tiles of 16 SNPs per wave.  The mixed mode carries fp32 + fp64 accumulators (2 at b >= 48).  16 columns in fp64: 2 -- measured
   // at 500,000 x 100,000 (profiles/r05_fp_kernels.txt): 8 m-tiles 25.4 ms, 4 (rounds 1-4) 24.2, 2 23.9; the narrow tile has 70
   // registers, so more waves take turns on the matrix pipe, which at one MFMA per table gather is what this shape is short of.
   // (fp32: 4 stays -- 14.4 ms against 14.7 with 2)
   static constexpr int MT = (sizeof(RT) == 4 && NT >= 3) ? 2 : (sizeof(RT) == 8 && NT == 1) ? 2 : 4;
};

template <typename RT, int NT>
static void launch_xt_b(const uint8_t *packed, size_t pitch, const double *lut, const double *B, double *Tpart,
                        uint64_t N_pad, uint64_t P_pad, int nsplit, hipStream_t stream)
{
   constexpr int KC = XtCfg<NT>::KC;
   const int chunks_total = (int)(N_pad / KC);
   int cps = (chunks_total + nsplit - 1) / nsplit;
   if (KC == 128) cps += cps & 1; // super-chunks of two (k_xt_b): every split starts on an even chunk (N_pad is a multiple of 512)
   constexpr int MT = XtMt<RT, NT>::MT;
   const size_t smem = ((size_t)KC * 16 * NT + 64 * MT * (NT == 1 ? 32 : 4)) * sizeof(RT); // B tile + table (16 columns: [SNP][16] pairs)
   static bool attr_set = false;
   if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xt_b<RT, NT, MT, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
   }
   dim3 grid((unsigned)(P_pad / (64 * MT)), (unsigned)nsplit);
   hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xt_b<RT, NT, MT, KC>), grid, dim3(256), smem, stream, packed, pitch, lut, B, Tpart, P_pad,
                      chunks_total, cps);
   HIP_CHECK_LAUNCH();
}

void xt_b(const uint8_t *packed, size_t pitch, const double *lut, const double *B, double *Tpart, uint64_t N_pad,
          uint64_t P_pad, int b, int nsplit, bool fp32, hipStream_t stream)
{
#define FPCA_CASE(NT_)                                                                                          \
   case 16 * NT_:                                                                                               \
      if (fp32)                                                                                                 \
         launch_xt_b<float, NT_>(packed, pitch, lut, B, Tpart, N_pad, P_pad, nsplit, stream);                   \
      else                                                                                                      \
         launch_xt_b<double, NT_>(packed, pitch, lut, B, Tpart, N_pad, P_pad, nsplit, stream);                  \
      break;
   switch (b) {
      FPCA_CASE(1)
      FPCA_CASE(2)
      FPCA_CASE(3)
      FPCA_CASE(4)
   default: throw Error(-1, "xt_b: block width must be 16, 32, 48 or 64");
   }
#undef FPCA_CASE
}

// (Round 6 measured K3's fetch groups with two-instruction gather addresses, the T fragment last and one wait per group, and with 2 / 4
// k-steps per group: 13.41 / 13.59 / 14.42 ms against 13.25 for the compiler-formed ones below at fp32 / 16 columns -- the staged waits
// let the first MFMA of a group start a gather earlier, and with four waves per SIMD taking turns short groups win; r06_fp_kernels.txt.)
// ------------------------------------------------------------------------------------------------
// K3 x_t:  Y[s][c] = sum_snp X[s][snp] T[snp][c]
//   workgroup = 64*MT samples x all b columns, wave = 16*MT samples (MT m-tiles), K = SNPs.
//   The reduction runs over SNPs, which are pitch bytes apart in the packed stream, so a
//   [64 SNP x 16*MT byte] tile of the stream is staged through LDS with 16-byte coalesced loads (128-byte
//   lines for MT=8), together with the 64 x b tile of T and the 64 x 4 lookup-table tile.  Lane (i, kq) owns
//   samples MT*i .. MT*i+MT-1 of its wave (one ushort / byte of the record), so m-tile m row i is sample
//   MT*i + m: again a permutation chosen so that decode needs no cross-lane traffic.  The standardised
//   value is fetched from the LDS table with the raw 2-bit code as index.
template <typename RT, int MT, int NT, int KCX /* SNPs per LDS chunk */>
__global__ __launch_bounds__(256, 2) void k_x_t(const uint8_t *__restrict__ packed, size_t pitch,
                                                 const double *__restrict__ lut, const double *__restrict__ T,
                                                 double *__restrict__ Ypart, uint64_t N_pad, int chunks_total,
                                                 int chunks_per_split)
{
   typedef typename Mma<RT>::acc_t acc_t;
   constexpr bool MIXED = sizeof(RT) == 4;
   constexpr int b = 16 * NT;
   constexpr int ROWB = 16 * MT;                       // bytes of one record inside the workgroup tile
   constexpr int NPIECES = KCX * ROWB / 16;            // 16-byte pieces of the packed tile
   constexpr int NP = (NPIECES + 255) / 256;           // ... per thread (2 for MT=8, 1 for MT=4)
   constexpr int NTL = KCX * b * 8 / (256 * 16);      // (fp64) T-tile pieces per thread (= 2 NT)
   constexpr int SEG = ROWB / 16;                      // 16-byte pieces per record row
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
   unsigned char *sP = smem_raw;                                        // [KCX][ROWB]
   RT *sT = reinterpret_cast<RT *>(smem_raw + KCX * ROWB);             // [KCX][b]
   // [KCX][16] PAIRS of table values, indexed by the 4 bits of two consecutive samples' codes (c0 | c1 << 2): one 16-byte (fp64) or
   // 8-byte (fp32) LDS read decodes two samples -- half the gather instructions and address arithmetic of a per-sample lookup, and
   // ds_read_b128 reaches its rate from one wave per SIMD where ds_read_b64 needs four (MI355X_MICROARCH.md, LDS)
   typedef typename Pair<RT>::type pair_t;
   pair_t *sL = reinterpret_cast<pair_t *>(sT + KCX * b);

   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const size_t wg_byte0 = (size_t)blockIdx.x * ROWB;
   const int c_begin = blockIdx.y * chunks_per_split;
   int c_end = c_begin + chunks_per_split;
   if (c_end > chunks_total) c_end = chunks_total;

   acc_t acc[MT][NT];
   d4 acc64[MIXED ? MT : 1][MIXED ? NT : 1];
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
         acc[m][nt] = (acc_t){0, 0, 0, 0};
         if (MIXED) acc64[m][nt] = (d4){0.0, 0.0, 0.0, 0.0};
      }

   u4 preg[NP];
   d2 treg[NTL];
   d2 lreg[2]; // this thread's SNP (tid / 4) of the chunk: its four table values
#define FPCA_XT_ISSUE(cc)                                                                                     \
   {                                                                                                           \
      const uint64_t snp_c0 = (uint64_t)(cc) * KCX;                                                           \
      _Pragma("unroll") for (int r = 0; r < NP; r++)                                                           \
      {                                                                                                        \
         const int p = tid + 256 * r;                                                                          \
         const int prow = p / SEG, pseg = p % SEG;                                                             \
         if (p < NPIECES)                                                                                      \
            preg[r] = *reinterpret_cast<const u4 *>(packed + (snp_c0 + prow) * pitch + wg_byte0 + pseg * 16);  \
      }                                                                                                        \
      const d2 *tsrc = reinterpret_cast<const d2 *>(T + snp_c0 * b);                                           \
      _Pragma("unroll") for (int r = 0; r < NTL; r++) treg[r] = tsrc[tid + 256 * r];                           \
      if (tid < KCX * 4) {                                                                                     \
         const d2 *lsrc_ = reinterpret_cast<const d2 *>(lut + (snp_c0 + (tid >> 2)) * 4);                       \
         lreg[0] = lsrc_[0];                                                                                   \
         lreg[1] = lsrc_[1];                                                                                   \
      }                                                                                                        \
   }

   if (c_begin < c_end) FPCA_XT_ISSUE(c_begin);

   const unsigned char *sP_lane = sP + (size_t)kq * ROWB + wave * (4 * MT) + li * (MT / 4);
   const RT *sT_lane = sT + (size_t)kq * b + li;
   const pair_t *sL_lane = sL + (size_t)kq * 16;

   for (int c = c_begin; c < c_end; c++) {
      __syncthreads();
      {
#pragma unroll
         for (int r = 0; r < NP; r++)
            if (tid + 256 * r < NPIECES) reinterpret_cast<u4 *>(sP)[tid + 256 * r] = preg[r];
#pragma unroll
         for (int r = 0; r < NTL; r++) lds_put2(sT, tid + 256 * r, treg[r]);
         if (tid < KCX * 4) { // thread (SNP tid / 4, c1 = tid % 4) writes the four pairs (c0, c1), c0 = 0..3
            const double l4[4] = {lreg[0].x, lreg[0].y, lreg[1].x, lreg[1].y};
            const int c1 = tid & 3;
#pragma unroll
            for (int c0 = 0; c0 < 4; c0++) sL[(tid >> 2) * 16 + c1 * 4 + c0] = Pair<RT>::make(l4[c0], l4[c1]);
         }
      }
      __syncthreads();
      if (c + 1 < c_end) FPCA_XT_ISSUE(c + 1);
      // this lane's packed codes of the whole chunk go to registers in one batch of LDS reads: inside the loop the table gather of a
      // k-step then hangs on ONE LDS round trip (code -> value), not two (record -> code -> value), like K2's register-resident words
      uint32_t hh[KCX / 4];
#pragma unroll
      for (int t = 0; t < KCX / 4; t++) {
         if (MT == 8)
            hh[t] = *reinterpret_cast<const unsigned short *>(sP_lane + (size_t)(4 * t) * ROWB);
         else
            hh[t] = *(sP_lane + (size_t)(4 * t) * ROWB);
      }
      // k-step t + 1's operands -- the T fragment and the MT table gathers -- are read while step t's MFMAs run: left to itself
      // the compiler funnels every gather through one register pair (ds_read, s_waitcnt 0, v_mfma, MT times over), and a 64-cycle
      // MFMA behind a ~100-cycle LDS round trip kept the pipe 74 % busy with four waves per SIMD taking turns (rounds 1-4)
      RT av[2][MT], tv[2][NT];
#define FPCA_XT_FETCH(t_, slot_)                                                                                    \
   {                                                                                                                \
      const uint32_t h_ = hh[t_];                                                                                   \
      const pair_t *lrow_ = sL_lane + (size_t)(4 * (t_)) * 16;                                                      \
      _Pragma("unroll") for (int nt = 0; nt < NT; nt++) tv[slot_][nt] = sT_lane[(size_t)(4 * (t_)) * b + nt * 16];  \
      _Pragma("unroll") for (int m = 0; m < MT; m += 2)                                                             \
      {                                                                                                             \
         const pair_t pr_ = lrow_[(h_ >> (2 * m)) & 15u];                                                           \
         av[slot_][m] = pr_.x;                                                                                      \
         av[slot_][m + 1] = pr_.y;                                                                                  \
      }                                                                                                             \
   }
      FPCA_XT_FETCH(0, 0);
#pragma unroll
      for (int t = 0; t < KCX / 4; t++) {
         if (t + 1 < KCX / 4) FPCA_XT_FETCH(t + 1, (t + 1) & 1);
         __builtin_amdgcn_sched_barrier(0); // (the fetch group stays ahead of the MFMA group it does not feed)
#pragma unroll
         for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[m][nt] = Mma<RT>::mma(av[t & 1][m], tv[t & 1][nt], acc[m][nt]);
         __builtin_amdgcn_sched_barrier(0);
      }
#undef FPCA_XT_FETCH
      if (MIXED && ((c & (FOLD_EVERY - 1)) == FOLD_EVERY - 1 || c + 1 == c_end)) { // (as in K2: fp32 sums over at most 256 SNPs)
#pragma unroll
         for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
               for (int r = 0; r < 4; r++) acc64[m][nt][r] += (double)acc[m][nt][r];
               acc[m][nt] = (acc_t){0, 0, 0, 0};
            }
      }
   }
#undef FPCA_XT_ISSUE

   double *Yout = Ypart + (size_t)blockIdx.y * N_pad * b;
   const uint64_t s_wave = (uint64_t)blockIdx.x * (64 * MT) + (uint64_t)wave * (16 * MT);
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) {
            const uint64_t s = s_wave + (uint64_t)MT * Mma<RT>::row(kq, r) + m;
            Yout[s * b + nt * 16 + li] = MIXED ? acc64[m][nt][r] : (double)acc[m][nt][r];
         }
}

// shape of K3 per (arithmetic, block width): fp64 -> 8 m-tiles for b <= 32 else 4, 64-SNP chunks; the mixed fp32 mode
// carries fp32 + fp64 accumulators, so it uses 4 m-tiles and, for b >= 48, 32-SNP chunks (fewer prefetch registers)
template <typename RT, int NT> struct XCfg {
   static constexpr bool MIXED = sizeof(RT) == 4;
   static constexpr int MT = MIXED ? 4 : (NT <= 2 ? 8 : 4);
   static constexpr int KCX = (MIXED && NT >= 3) ? 32 : 64;
};
static inline int x_t_mt(int b, bool fp32) { return fp32 ? 4 : (b <= 32 ? 8 : 4); }
static inline int x_t_kc(int b, bool fp32) { return (fp32 && b >= 48) ? 32 : 64; }

int x_t_splits(uint64_t N_pad, uint64_t P_pad, int b, bool fp32)
{
   static const int forced = FPCA_ENV_INT("FPCA_X_SPLITS", 0);
   if (forced > 0) return (int)std::min<uint64_t>(forced, P_pad / x_t_kc(b, fp32));
   return pick_splits(N_pad / (64 * x_t_mt(b, fp32)), P_pad / x_t_kc(b, fp32), 4, 64, 512);
}

template <typename RT, int NT>
static void launch_x_t(const uint8_t *packed, size_t pitch, const double *lut, const double *T, double *Ypart,
                       uint64_t N_pad, uint64_t P_pad, int nsplit, hipStream_t stream)
{
   constexpr int MT = XCfg<RT, NT>::MT, KCX = XCfg<RT, NT>::KCX;
   const int chunks_total = (int)(P_pad / KCX);
   const int cps = (chunks_total + nsplit - 1) / nsplit;
   const size_t smem = (size_t)KCX * 16 * MT + ((size_t)KCX * 16 * NT + KCX * 32) * sizeof(RT); // packed tile, T tile, [KCX][16] table pairs
   static bool attr_set = false;
   if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_x_t<RT, MT, NT, KCX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
   }
   dim3 grid((unsigned)(N_pad / (64 * MT)), (unsigned)nsplit);
   hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_t<RT, MT, NT, KCX>), grid, dim3(256), smem, stream, packed, pitch, lut, T, Ypart,
                      N_pad, chunks_total, cps);
   HIP_CHECK_LAUNCH();
}

void x_t(const uint8_t *packed, size_t pitch, const double *lut, const double *T, double *Ypart, uint64_t N_pad,
         uint64_t P_pad, int b, int nsplit, bool fp32, hipStream_t stream)
{
#define FPCA_CASE(NT_)                                                                                          \
   case 16 * NT_:                                                                                               \
      if (fp32)                                                                                                 \
         launch_x_t<float, NT_>(packed, pitch, lut, T, Ypart, N_pad, P_pad, nsplit, stream);                    \
      else                                                                                                      \
         launch_x_t<double, NT_>(packed, pitch, lut, T, Ypart, N_pad, P_pad, nsplit, stream);                   \
      break;
   switch (b) {
      FPCA_CASE(1)
      FPCA_CASE(2)
      FPCA_CASE(3)
      FPCA_CASE(4)
   default: throw Error(-1, "x_t: block width must be 16, 32, 48 or 64");
   }
#undef FPCA_CASE
}

// ------------------------------------------------------------------------------------------------
// deterministic split-K combine: out[i] = sum_s part[s][i]
// Each thread owns 2 consecutive 16-byte elements and keeps 4 partial loads of each in flight, so the pass streams at
// HBM rate instead of paying one dependent-load latency per partial.
__global__ __launch_bounds__(256) void k_reduce_sum(const d2 *__restrict__ part, d2 *__restrict__ out, uint64_t count2,
                                                     int nsplit)
{
   const uint64_t stride = (uint64_t)gridDim.x * 256;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count2; i += 2 * stride) {
      const uint64_t i2 = i + stride;
      const bool has2 = i2 < count2;
      d2 s0 = (d2){0.0, 0.0}, s1 = (d2){0.0, 0.0};
      int k = 0;
      for (; k + 4 <= nsplit; k += 4) {
         const d2 a0 = part[(uint64_t)(k + 0) * count2 + i], a1 = part[(uint64_t)(k + 1) * count2 + i];
         const d2 a2 = part[(uint64_t)(k + 2) * count2 + i], a3 = part[(uint64_t)(k + 3) * count2 + i];
         d2 b0 = (d2){0.0, 0.0}, b1 = b0, b2 = b0, b3 = b0;
         if (has2) {
            b0 = part[(uint64_t)(k + 0) * count2 + i2];
            b1 = part[(uint64_t)(k + 1) * count2 + i2];
            b2 = part[(uint64_t)(k + 2) * count2 + i2];
            b3 = part[(uint64_t)(k + 3) * count2 + i2];
         }
         s0 += ((a0 + a1) + (a2 + a3)); // fixed association: deterministic for a given nsplit
         s1 += ((b0 + b1) + (b2 + b3));
      }
      for (; k < nsplit; k++) {
         s0 += part[(uint64_t)k * count2 + i];
         if (has2) s1 += part[(uint64_t)k * count2 + i2];
      }
      out[i] = s0;
      if (has2) out[i2] = s1;
   }
}

// Tall stacks of small planes (the Gram partials: hundreds of planes of a few b x b blocks): a block owns 8 consecutive
// 16-byte elements, 32 thread groups stride over the planes with four loads in flight each, and the groups are folded
// in a fixed order through LDS (deterministic for a given nsplit).
__global__ __launch_bounds__(256) void k_reduce_tall(const d2 *__restrict__ part, d2 *__restrict__ out, uint64_t count2, int nsplit)
{
   __shared__ d2 sh[32][8];
   const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
   const uint64_t i = (uint64_t)blockIdx.x * 8 + e;
   d2 s0 = (d2){0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0;
   if (i < count2) {
      int k = g;
      for (; k + 96 < nsplit; k += 128) {
         const d2 a0 = part[(uint64_t)k * count2 + i], a1 = part[(uint64_t)(k + 32) * count2 + i];
         const d2 a2 = part[(uint64_t)(k + 64) * count2 + i], a3 = part[(uint64_t)(k + 96) * count2 + i];
         s0 += a0;
         s1 += a1;
         s2 += a2;
         s3 += a3;
      }
      for (; k < nsplit; k += 32) s0 += part[(uint64_t)k * count2 + i];
   }
   sh[g][e] = (s0 + s1) + (s2 + s3);
   __syncthreads();
   if (g == 0 && i < count2) {
      d2 t = sh[0][e];
#pragma unroll
      for (int j = 1; j < 32; j++) t += sh[j][e];
      out[i] = t;
   }
}

void reduce_sum(const double *part, double *out, uint64_t count, int nsplit, hipStream_t stream)
{
   if (count == 0) return;
   const uint64_t count2 = count / 2; // all our buffers have even element counts
   if (nsplit >= 64 && count2 <= (1u << 16)) {
      hipLaunchKernelGGL(k_reduce_tall, dim3((unsigned)((count2 + 7) / 8)), dim3(256), 0, stream, reinterpret_cast<const d2 *>(part),
                         reinterpret_cast<d2 *>(out), count2, nsplit);
      HIP_CHECK_LAUNCH();
      return;
   }
   uint64_t blocks = (count2 + 511) / 512;
   if (blocks > 2048) blocks = 2048;
   if (blocks == 0) blocks = 1;
   hipLaunchKernelGGL(k_reduce_sum, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const d2 *>(part),
                      reinterpret_cast<d2 *>(out), count2, nsplit);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Dense (in-memory matrix) path: RandomPCA::pca_fast(MatrixXd&) (randompca.cpp:121-166) + standardise()
// (util.cpp:24-192) + SVDWide::perform_op (svdwide.cpp:4-12).  The fp64 matrix is kept in HBM as Xd[P_pad][N_pad]
// (one row per column of the caller's N x P column-major matrix, zero padded), standardised in place once; the two
// products are then plain tall-skinny FP64 MFMA GEMMs that read Xd once each and are HBM-bound (2 b flops per 8
// bytes = 8 flop/B at b = 32, below the ~10 flop/B ridge).

// util.cpp:24-192, one workgroup per column; method: 0 none, 1 sd, 2 binom, 3 binom2, 4 center.  NaN = missing.
__global__ __launch_bounds__(256) void k_dense_standardise(double *__restrict__ Xd, uint64_t N_pad, uint64_t N, int method,
                                                            double *__restrict__ mean_out, double *__restrict__ sd_out,
                                                            double *__restrict__ sumsq_out)
{
   double *col = Xd + (uint64_t)blockIdx.x * N_pad;
   __shared__ double red[3][4];
   __shared__ double bc[2];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   // pass 1: NaN-aware sums (sd uses the shifted-data variance with K = 1, util.cpp:84-100)
   double s1 = 0, s2 = 0, cnt = 0;
   for (uint64_t i = threadIdx.x; i < N; i += 256) {
      const double x = col[i];
      if (!isnan(x)) {
         const double xs = (method == 1) ? x - 1.0 : x;
         s1 += xs;
         s2 += xs * xs;
         cnt += 1.0;
      }
   }
   for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_down(s1, off);
      s2 += __shfl_down(s2, off);
      cnt += __shfl_down(cnt, off);
   }
   if (lane == 0) {
      red[0][wave] = s1;
      red[1][wave] = s2;
      red[2][wave] = cnt;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      const double sum = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      const double sum_sqr = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      const double nj = red[2][0] + red[2][1] + red[2][2] + red[2][3];
      double mean = 0.0, sd = 1.0;
      if (method == 0 || method == 4)
         mean = sum / nj;
      else if (method == 1) {
         const double varj = (sum_sqr - (sum * sum) / nj) / (nj - 1.0);
         mean = (sum + nj) / nj;
         sd = sqrt(varj);
      } else {
         mean = sum / nj;
         const double r = mean / 2.0;
         sd = sqrt((method == 2 ? 1.0 : 2.0) * r * (1.0 - r));
      }
      bc[0] = mean;
      bc[1] = sd;
      mean_out[blockIdx.x] = mean;
      sd_out[blockIdx.x] = sd;
   }
   __syncthreads();
   const double mean = bc[0], sd = bc[1];
   // pass 2: impute + standardise in place; accumulate sum of squares (trace, randompca.cpp:154)
   double sq = 0;
   for (uint64_t i = threadIdx.x; i < N; i += 256) {
      const double x = col[i];
      double y;
      if (method == 0)
         y = isnan(x) ? mean : x;
      else if (method == 4)
         y = isnan(x) ? 0.0 : x - mean;
      else
         y = isnan(x) ? 0.0 : (sd > 1e-9 ? (x - mean) / sd : mean); // util.cpp:107-113 / 141-147
      col[i] = y;
      sq += y * y;
   }
   for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
   __syncthreads();
   if (lane == 0) red[0][wave] = sq;
   __syncthreads();
   if (threadIdx.x == 0) sumsq_out[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
}

void dense_standardise(double *Xd, uint64_t N_pad, uint64_t N, uint64_t P_g, int method, double *mean, double *sd,
                       double *sumsq, hipStream_t stream)
{
   if (P_g == 0) return;
   hipLaunchKernelGGL(k_dense_standardise, dim3((unsigned)P_g), dim3(256), 0, stream, Xd, N_pad, N, method, mean, sd, sumsq);
   HIP_CHECK_LAUNCH();
}

// K2d: T[j][c] = sum_s Xd[j][s] B[s][c].  Workgroup = 128 columns-of-X (2 m-tiles per wave); lane (i, kq) streams
// 4 consecutive doubles of ITS row per 16-sample step (16 lanes x 32 B... a wave reads 16 rows x 128 B = full lines),
// with the K order permuted per lane group as in the packed kernel; the B tile goes through LDS.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_xt_b_dense(const double *__restrict__ Xd, uint64_t N_pad,
                                                        const double *__restrict__ B, double *__restrict__ Tpart,
                                                        uint64_t P_pad, int chunks_total, int chunks_per_split)
{
   constexpr int b = 16 * NT;
   constexpr int MT = 2;
   constexpr int KC = 64; // samples per chunk: lane group kq owns samples 16 kq .. 16 kq + 15
   constexpr int NLOAD = KC * b * 8 / (256 * 16);
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
   double *sB = reinterpret_cast<double *>(smem_raw); // [KC][b]
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const uint64_t j0 = (uint64_t)blockIdx.x * (64 * MT) + (uint64_t)wave * (16 * MT);
   const int c_begin = blockIdx.y * chunks_per_split;
   int c_end = c_begin + chunks_per_split;
   if (c_end > chunks_total) c_end = chunks_total;
   const double *rowp[MT];
#pragma unroll
   for (int m = 0; m < MT; m++) rowp[m] = Xd + (j0 + m * 16 + li) * N_pad + kq * 16;
   d4 acc[MT][NT];
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[m][nt] = (d4){0.0, 0.0, 0.0, 0.0};
   d2 xn[MT][8], breg[NLOAD];
#define FPCA_XTBD_ISSUE(cc)                                                                                   \
   {                                                                                                           \
      _Pragma("unroll") for (int m = 0; m < MT; m++)                                                           \
      {                                                                                                        \
         const d2 *pp = reinterpret_cast<const d2 *>(rowp[m] + (size_t)(cc) * KC);                             \
         _Pragma("unroll") for (int h = 0; h < 8; h++) xn[m][h] = pp[h];                                       \
      }                                                                                                        \
      const d2 *src = reinterpret_cast<const d2 *>(B + (size_t)(cc) * KC * b);                                 \
      _Pragma("unroll") for (int r = 0; r < NLOAD; r++) breg[r] = src[tid + 256 * r];                          \
   }
   if (c_begin < c_end) FPCA_XTBD_ISSUE(c_begin);
   const double *sB_lane = sB + (size_t)(16 * kq) * b + li;
   for (int c = c_begin; c < c_end; c++) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NLOAD; r++) reinterpret_cast<d2 *>(sB)[tid + 256 * r] = breg[r];
      d2 xc[MT][8];
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
         for (int h = 0; h < 8; h++) xc[m][h] = xn[m][h];
      __syncthreads();
      if (c + 1 < c_end) FPCA_XTBD_ISSUE(c + 1);
#pragma unroll
      for (int t = 0; t < 16; t++) {
         double bv[NT];
#pragma unroll
         for (int nt = 0; nt < NT; nt++) bv[nt] = sB_lane[(size_t)t * b + nt * 16];
#pragma unroll
         for (int m = 0; m < MT; m++) {
            const double a = xc[m][t >> 1][t & 1];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[m][nt] = FPCA_MFMA(a, bv[nt], acc[m][nt]);
         }
      }
   }
#undef FPCA_XTBD_ISSUE
   double *Tout = Tpart + (size_t)blockIdx.y * P_pad * b;
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) Tout[(j0 + m * 16 + kq + 4 * r) * b + nt * 16 + li] = acc[m][nt][r];
}

// K3d: Y[s][c] = sum_j Xd[j][s] T[j][c].  Workgroup = 256 samples (wave = 64 = 4 m-tiles); lane (i, kq) owns samples
// 4i .. 4i+3 of its wave and reads them as one 32-byte piece of row k (16 lanes = 512 contiguous bytes); the T tile
// (64 rows x b) goes through LDS.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_x_t_dense(const double *__restrict__ Xd, uint64_t N_pad,
                                                       const double *__restrict__ T, double *__restrict__ Ypart,
                                                       int chunks_total, int chunks_per_split)
{
   constexpr int b = 16 * NT;
   constexpr int MT = 4;
   constexpr int KC = 64; // rows of Xd (SNPs) per chunk
   constexpr int NTL = KC * b * 8 / (256 * 16);
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
   double *sT = reinterpret_cast<double *>(smem_raw); // [KC][b]
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const uint64_t s0 = (uint64_t)blockIdx.x * 256 + (uint64_t)wave * 64 + 4 * li;
   const int c_begin = blockIdx.y * chunks_per_split;
   int c_end = c_begin + chunks_per_split;
   if (c_end > chunks_total) c_end = chunks_total;
   d4 acc[MT][NT];
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[m][nt] = (d4){0.0, 0.0, 0.0, 0.0};
   d2 treg[NTL];
#define FPCA_XTD_ISSUE(cc)                                                                                    \
   {                                                                                                           \
      const d2 *tsrc = reinterpret_cast<const d2 *>(T + (size_t)(cc) * KC * b);                                \
      _Pragma("unroll") for (int r = 0; r < NTL; r++) treg[r] = tsrc[tid + 256 * r];                           \
   }
   if (c_begin < c_end) FPCA_XTD_ISSUE(c_begin);
   const double *sT_lane = sT + (size_t)kq * b + li;
   for (int c = c_begin; c < c_end; c++) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NTL; r++) reinterpret_cast<d2 *>(sT)[tid + 256 * r] = treg[r];
      __syncthreads();
      if (c + 1 < c_end) FPCA_XTD_ISSUE(c + 1);
      const double *xrow = Xd + ((uint64_t)c * KC + kq) * N_pad + s0;
#pragma unroll 4
      for (int t = 0; t < KC / 4; t++) {
         const d2 x01 = reinterpret_cast<const d2 *>(xrow + (size_t)(4 * t) * N_pad)[0];
         const d2 x23 = reinterpret_cast<const d2 *>(xrow + (size_t)(4 * t) * N_pad)[1];
         const double xa[4] = {x01.x, x01.y, x23.x, x23.y};
         double tv[NT];
#pragma unroll
         for (int nt = 0; nt < NT; nt++) tv[nt] = sT_lane[(size_t)(4 * t) * b + nt * 16];
#pragma unroll
         for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[m][nt] = FPCA_MFMA(xa[m], tv[nt], acc[m][nt]);
      }
   }
#undef FPCA_XTD_ISSUE
   double *Yout = Ypart + (size_t)blockIdx.y * N_pad * b;
   const uint64_t s_wave = (uint64_t)blockIdx.x * 256 + (uint64_t)wave * 64;
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) Yout[(s_wave + 4 * (kq + 4 * r) + m) * b + nt * 16 + li] = acc[m][nt][r];
}

int xt_b_dense_splits(uint64_t N_pad, uint64_t P_pad) { return pick_splits(P_pad / 128, N_pad / 64, 8, 64, 768); }
int x_t_dense_splits(uint64_t N_pad, uint64_t P_pad) { return pick_splits(N_pad / 256, P_pad / 64, 8, 64, 768); }

void xt_b_dense(const double *Xd, const double *B, double *Tpart, uint64_t N_pad, uint64_t P_pad, int b, int nsplit,
                hipStream_t stream)
{
   const int chunks_total = (int)(N_pad / 64);
   const int cps = (chunks_total + nsplit - 1) / nsplit;
   dim3 grid((unsigned)(P_pad / 128), (unsigned)nsplit);
   const size_t smem = (size_t)64 * b * 8;
   switch (b) {
   case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xt_b_dense<1>), grid, dim3(256), smem, stream, Xd, N_pad, B, Tpart, P_pad, chunks_total, cps); break;
   case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xt_b_dense<2>), grid, dim3(256), smem, stream, Xd, N_pad, B, Tpart, P_pad, chunks_total, cps); break;
   case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xt_b_dense<3>), grid, dim3(256), smem, stream, Xd, N_pad, B, Tpart, P_pad, chunks_total, cps); break;
   case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xt_b_dense<4>), grid, dim3(256), smem, stream, Xd, N_pad, B, Tpart, P_pad, chunks_total, cps); break;
   default: throw Error(-1, "xt_b_dense: block width must be 16, 32, 48 or 64");
   }
   HIP_CHECK_LAUNCH();
}

void x_t_dense(const double *Xd, const double *T, double *Ypart, uint64_t N_pad, uint64_t P_pad, int b, int nsplit,
               hipStream_t stream)
{
   const int chunks_total = (int)(P_pad / 64);
   const int cps = (chunks_total + nsplit - 1) / nsplit;
   dim3 grid((unsigned)(N_pad / 256), (unsigned)nsplit);
   const size_t smem = (size_t)64 * b * 8;
   switch (b) {
   case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_t_dense<1>), grid, dim3(256), smem, stream, Xd, N_pad, T, Ypart, chunks_total, cps); break;
   case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_t_dense<2>), grid, dim3(256), smem, stream, Xd, N_pad, T, Ypart, chunks_total, cps); break;
   case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_t_dense<3>), grid, dim3(256), smem, stream, Xd, N_pad, T, Ypart, chunks_total, cps); break;
   case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_x_t_dense<4>), grid, dim3(256), smem, stream, Xd, N_pad, T, Ypart, chunks_total, cps); break;
   default: throw Error(-1, "x_t_dense: block width must be 16, 32, 48 or 64");
   }
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K4 gram: part[(split*4+wave)][q][p][c] = sum_{rows of the wave} A_q[s][p] W[s][c]
//   MFMA roles: A[i = p][k = sample] and B[k = sample][j = c] are both read straight from HBM, 16 lanes
//   covering 128 contiguous bytes of a row; HBM-bound (each A_q read once, W once per q, the latter out of L2 / the Infinity
//   Cache).  Measured at N = 500,000 (rocprofv3, slow-spectrum solve): 4.0-4.4 TB/s of basis with each wave streaming its own
//   contiguous 256 KB range; dealing the chunks round-robin over the waves took 10 % off.
template <int NT>
__global__ __launch_bounds__(256) void k_gram(const double *const *__restrict__ blocks, const double *__restrict__ W,
                                               double *__restrict__ part, uint64_t N_pad, int rows_per_split, int nq)
{
   constexpr int b = 16 * NT;
   constexpr int U = NT == 1 ? 4 : 2; // k-steps (of 4 rows) per chunk: their loads are issued together
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int li = lane & 15, kq = lane >> 4;
   const int q = blockIdx.y;
   const double *__restrict__ A = blocks[q];
   // chunks of 4 U rows are dealt round-robin over all waves of this basis block: at any moment the waves read one contiguous
   // window of A_q and of W (DRAM pages stay open) instead of gridDim.x * 4 separate 256 KB streams
   (void)rows_per_split;
   const uint64_t nchunks = N_pad / (4 * U), total = (uint64_t)gridDim.x * 4;

   d4 acc[NT][NT];
#pragma unroll
   for (int pt = 0; pt < NT; pt++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[pt][nt] = (d4){0.0, 0.0, 0.0, 0.0};

   for (uint64_t ch = (uint64_t)blockIdx.x * 4 + wave; ch < nchunks; ch += total) {
      const uint64_t s = ch * (4 * U);
      double a[U][NT], w[U][NT];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const uint64_t row = s + 4 * u + kq;
#pragma unroll
         for (int pt = 0; pt < NT; pt++) a[u][pt] = A[row * b + pt * 16 + li];
#pragma unroll
         for (int nt = 0; nt < NT; nt++) w[u][nt] = W[row * b + nt * 16 + li];
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
         for (int pt = 0; pt < NT; pt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[pt][nt] = FPCA_MFMA(a[u][pt], w[u][nt], acc[pt][nt]);
   }
   double *out = part + ((size_t)(blockIdx.x * 4 + wave) * nq + q) * (size_t)(b * b);
#pragma unroll
   for (int pt = 0; pt < NT; pt++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) out[(size_t)(pt * 16 + kq + 4 * r) * b + nt * 16 + li] = acc[pt][nt][r];
}

// The same with ONE W tile serving QG basis blocks (blockIdx.y = group of QG blocks): what the chip moves for a Gram launch is
// the basis once PLUS W once per workgroup row -- W does not fit in the 4 MB L2s, so every re-read comes over the fabric out of
// the Infinity Cache, and with one basis block per workgroup row (k_gram) that is as many bytes again as the basis itself.
// With QG = 8 the fabric carries 9/8 of the basis instead of 2x.
template <int NT, int QG>
__global__ __launch_bounds__(256) void k_gram_tiled(const double *const *__restrict__ blocks, const double *__restrict__ W,
                                                     double *__restrict__ part, uint64_t N_pad, int nq)
{
   constexpr int b = 16 * NT;
   constexpr int U = 2; // k-steps (of 4 rows) per chunk: (QG + 1) NT U loads in flight per wave
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
   const int li = lane & 15, kq = lane >> 4;
   const int q0 = blockIdx.y * QG;
   const int nqg = min(QG, nq - q0);
   const double *__restrict__ A[QG];
#pragma unroll
   for (int i = 0; i < QG; i++) A[i] = blocks[q0 + (i < nqg ? i : 0)]; // (slots beyond the last block re-read block q0: never stored)
   const uint64_t nchunks = N_pad / (4 * U), total = (uint64_t)gridDim.x * 4;

   d4 acc[QG][NT][NT];
#pragma unroll
   for (int i = 0; i < QG; i++)
#pragma unroll
      for (int pt = 0; pt < NT; pt++)
#pragma unroll
         for (int nt = 0; nt < NT; nt++) acc[i][pt][nt] = (d4){0.0, 0.0, 0.0, 0.0};

   for (uint64_t ch = (uint64_t)blockIdx.x * 4 + wave; ch < nchunks; ch += total) {
      const uint64_t s = ch * (4 * U);
      double a[QG][U][NT], w[U][NT];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const uint64_t off = (s + 4 * u + kq) * b + li;
#pragma unroll
         for (int nt = 0; nt < NT; nt++) w[u][nt] = W[off + nt * 16];
#pragma unroll
         for (int i = 0; i < QG; i++)
#pragma unroll
            for (int pt = 0; pt < NT; pt++) a[i][u][pt] = A[i][off + pt * 16];
      }
#pragma unroll
      for (int i = 0; i < QG; i++)
#pragma unroll
         for (int u = 0; u < U; u++)
#pragma unroll
            for (int pt = 0; pt < NT; pt++)
#pragma unroll
               for (int nt = 0; nt < NT; nt++) acc[i][pt][nt] = FPCA_MFMA(a[i][u][pt], w[u][nt], acc[i][pt][nt]);
   }
#pragma unroll
   for (int i = 0; i < QG; i++) {
      if (i >= nqg) break;
      double *out = part + ((size_t)(blockIdx.x * 4 + wave) * nq + q0 + i) * (size_t)(b * b);
#pragma unroll
      for (int pt = 0; pt < NT; pt++)
#pragma unroll
         for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(size_t)(pt * 16 + kq + 4 * r) * b + nt * 16 + li] = acc[i][pt][nt][r];
   }
}

// basis blocks that share one W tile, by block width (accumulators: QG NT^2 x 8 registers)
static inline int gram_group(int b) { return b == 16 ? 8 : b == 32 ? 2 : 1; }
static int g_k4_variant = 1; // fpca_debug_k4_variant: 0 = round 4's kernels (one basis block per workgroup row / C from L1), 1 = tiled
void k4_variant(int v) { g_k4_variant = v; }

// Rows per workgroup: enough workgroups to fill the chip (groups x splits >= ~512: two per CU, each wave with (QG + 1) U NT loads
// in flight; ~1024 with one block per workgroup row) without making the stack of partial planes (4 per workgroup) taller than it
// has to be.
int gram_rows(uint64_t N_pad, int nq, int b)
{
   const int QG = g_k4_variant ? gram_group(b) : 1;
   const uint64_t groups = (uint64_t)((nq > 0 ? nq : 1) + QG - 1) / QG, want = QG > 1 ? 512 : 1024;
   uint64_t rows = 8192;
   while (rows > 256 && (N_pad + rows - 1) / rows * groups < want) rows /= 2;
   return (int)rows;
}

int gram_splits(uint64_t N_pad, int rows) { return (int)((N_pad + rows - 1) / rows); }

void gram(const double *const *blocks, int nq, const double *W, double *part, uint64_t N_pad, int b, int rows, hipStream_t stream)
{
   if (nq <= 0) return;
   if (N_pad % 16) throw Error(-1, "gram: the block height must be a multiple of 16 rows");
   const int QG = g_k4_variant ? gram_group(b) : 1;
   if (QG > 1) {
      dim3 grid((unsigned)gram_splits(N_pad, rows), (unsigned)((nq + QG - 1) / QG));
      if (b == 16)
         hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram_tiled<1, 8>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, nq);
      else
         hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram_tiled<2, 2>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, nq);
      HIP_CHECK_LAUNCH();
      return;
   }
   dim3 grid((unsigned)gram_splits(N_pad, rows), (unsigned)nq); // (`rows` only sets the number of workgroups: 4 partial planes each)
   switch (b) {
   case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<1>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, rows, nq); break;
   case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<2>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, rows, nq); break;
   case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<3>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, rows, nq); break;
   case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<4>), grid, dim3(256), 0, stream, blocks, W, part, N_pad, rows, nq); break;
   default: throw Error(-1, "gram: block width must be 16, 32, 48 or 64");
   }
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K4 block_gemm: Out = Init + sum_q A_q C_q.  A wave owns 16 rows; lane (i, kq) loads the contiguous
// quarter kq of row i of A_q (b/4 doubles), so a wave reads 16 full rows = 16*b*8 contiguous bytes, and the
// K order is permuted accordingly (k-step t of lane group kq is column kq*b/4 + t).  C is tiny and stays
// in L1/L2.  Reads of a wave's rows all precede its stores, so Out may alias Init or any A_q.
template <int NT>
__global__ __launch_bounds__(256) void k_block_gemm(const double *const *__restrict__ blocks, int nq,
                                                     const double *__restrict__ C, const double *Init, double *Out,
                                                     uint64_t N_pad)
{
   constexpr int b = 16 * NT;
   constexpr int KS = b / 4;
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + wave) * 16;
   if (s0 >= N_pad) return;
   d4 acc[NT];
#pragma unroll
   for (int nt = 0; nt < NT; nt++) {
      if (Init) {
#pragma unroll
         for (int r = 0; r < 4; r++) acc[nt][r] = Init[(s0 + kq + 4 * r) * b + nt * 16 + li];
      } else
         acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
   }
   for (int q = 0; q < nq; q++) {
      const double *A = blocks[q];
      const double *Cq = C + (size_t)q * b * b;
      double areg[KS];
      const d2 *src = reinterpret_cast<const d2 *>(A + (s0 + li) * b + kq * KS);
#pragma unroll
      for (int r = 0; r < KS / 2; r++) {
         const d2 v = src[r];
         areg[2 * r] = v.x;
         areg[2 * r + 1] = v.y;
      }
#pragma unroll
      for (int t = 0; t < KS; t++) {
         const int p = kq * KS + t;
#pragma unroll
         for (int nt = 0; nt < NT; nt++) acc[nt] = FPCA_MFMA(areg[t], Cq[(size_t)p * b + nt * 16 + li], acc[nt]);
      }
   }
#pragma unroll
   for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) Out[(s0 + kq + 4 * r) * b + nt * 16 + li] = acc[nt][r];
}

// The same with the coefficients staged through LDS and TPW row tiles per wave.  In k_block_gemm every MFMA's B operand is an 8-byte
// global load that hits L1 -- twice as many vector-memory instructions for C as for the basis rows the kernel is there to stream.
// Here a workgroup copies QC coefficient blocks into LDS once (kq-group rows padded so that the four lane groups of a ds_read_b64
// fall into different bank halves), every wave runs them against TPW tiles of 16 rows whose loads are issued together, and the
// operand reads are LDS reads.  Reads of a wave's rows all precede its stores: Out may alias Init or any A_q.
// GRAM: the launch also leaves Out'Out -- one partial plane [b][b] per workgroup in gpart, to be summed by reduce_sum: the output
// tile sits in the accumulators in exactly the layout both MFMA operands of W'W want (lane (li, kq), register r = row kq + 4r,
// column li), so the Gram matrix of the block just written costs NT^2 x 4 MFMAs per tile and no memory traffic (the solver's
// re-normalisation pass needs nothing else: it saves a launch that re-reads the block and one round trip to the host).
template <int NT, int QC, int TPW, bool GRAM>
__global__ __launch_bounds__(256) void k_block_gemm_lds(const double *const *__restrict__ blocks, int nq, const double *__restrict__ C,
                                                         const double *Init, double *Out, uint64_t N_pad, double *__restrict__ gpart)
{
   constexpr int b = 16 * NT, KS = b / 4, GRP = KS * b + 16 /* doubles per kq group of one block, padded */, BLK = 4 * GRP;
   __shared__ double Cs[QC * BLK];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int li = lane & 15, kq = lane >> 4;
   uint64_t s0[TPW];
   bool ok[TPW]; // (wave-uniform) a tile past the end computes on tile 0's rows and stores nothing
   d4 acc[TPW][NT];
#pragma unroll
   for (int t = 0; t < TPW; t++) {
      s0[t] = (((uint64_t)blockIdx.x * 4 + wave) * TPW + t) * 16;
      ok[t] = s0[t] < N_pad;
      if (!ok[t]) s0[t] = 0;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
         if (Init && ok[t]) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[t][nt][r] = Init[(s0[t] + kq + 4 * r) * b + nt * 16 + li];
         } else
            acc[t][nt] = (d4){0.0, 0.0, 0.0, 0.0};
      }
   }
   for (int q0 = 0; q0 < nq; q0 += QC) {
      const int nqc = min(QC, nq - q0);
      __syncthreads(); // (the previous stage's operand reads are done)
      for (int idx = threadIdx.x; idx < nqc * (b * b / 2); idx += 256) { // pairs of doubles along c
         const int qi = idx / (b * b / 2), rem = idx - qi * (b * b / 2), p = rem / (b / 2), c2 = rem - p * (b / 2);
         const d2 v = reinterpret_cast<const d2 *>(C + (size_t)(q0 + qi) * b * b)[rem];
         *reinterpret_cast<d2 *>(&Cs[qi * BLK + (p / KS) * GRP + (p % KS) * b + 2 * c2]) = v;
      }
      __syncthreads();
      for (int qi = 0; qi < nqc; qi++) {
         const double *A = blocks[q0 + qi];
         d2 av[TPW][KS / 2];
#pragma unroll
         for (int t = 0; t < TPW; t++) {
            const d2 *src = reinterpret_cast<const d2 *>(A + (s0[t] + li) * b + kq * KS);
#pragma unroll
            for (int r = 0; r < KS / 2; r++) av[t][r] = src[r];
         }
         const double *cq = &Cs[qi * BLK + kq * GRP + li];
#pragma unroll
         for (int tt = 0; tt < KS; tt++) {
            double cv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) cv[nt] = cq[tt * b + nt * 16];
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
               for (int nt = 0; nt < NT; nt++) acc[t][nt] = FPCA_MFMA((tt & 1) ? av[t][tt / 2].y : av[t][tt / 2].x, cv[nt], acc[t][nt]);
         }
      }
   }
#pragma unroll
   for (int t = 0; t < TPW; t++) {
      if (!ok[t]) continue;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
         for (int r = 0; r < 4; r++) Out[(s0[t] + kq + 4 * r) * b + nt * 16 + li] = acc[t][nt][r];
   }
   if (GRAM) {
      d4 g[NT][NT];
#pragma unroll
      for (int pt = 0; pt < NT; pt++)
#pragma unroll
         for (int nt = 0; nt < NT; nt++) g[pt][nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < TPW; t++) {
         if (!ok[t]) continue;
#pragma unroll
         for (int r = 0; r < 4; r++)
#pragma unroll
            for (int pt = 0; pt < NT; pt++)
#pragma unroll
               for (int nt = 0; nt < NT; nt++) g[pt][nt] = FPCA_MFMA(acc[t][pt][r], acc[t][nt][r], g[pt][nt]);
      }
      // the four waves' planes are summed through LDS (fixed order), one plane per workgroup leaves
      __syncthreads(); // (the coefficient stage is done with)
      static_assert(3 * b * b <= QC * BLK, "the coefficient stage must hold three planes");
      if (wave > 0) {
#pragma unroll
         for (int pt = 0; pt < NT; pt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
               for (int r = 0; r < 4; r++) Cs[(wave - 1) * b * b + (pt * 16 + kq + 4 * r) * b + nt * 16 + li] = g[pt][nt][r];
      }
      __syncthreads();
      if (wave == 0) {
         double *out = gpart + (size_t)blockIdx.x * (b * b);
#pragma unroll
         for (int pt = 0; pt < NT; pt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
               for (int r = 0; r < 4; r++) {
                  const int idx = (pt * 16 + kq + 4 * r) * b + nt * 16 + li;
                  out[idx] = ((g[pt][nt][r] + Cs[idx]) + Cs[b * b + idx]) + Cs[2 * b * b + idx];
               }
      }
   }
}

constexpr int BG_TPW = 4; // row tiles per wave of k_block_gemm_lds
int block_gemm_gram_planes(uint64_t N_pad, int b)
{
   return (g_k4_variant && (b == 16 || b == 32) && N_pad >= 64) ? (int)((N_pad / 16 + 4 * BG_TPW - 1) / (4 * BG_TPW)) : 0;
}

void block_gemm(const double *const *blocks, int nq, const double *C, const double *Init, double *Out, uint64_t N_pad,
                int b, hipStream_t stream, double *gram_part)
{
   if (gram_part && !block_gemm_gram_planes(N_pad, b)) throw Error(-1, "block_gemm: no fused Gram for this shape");
   if (g_k4_variant && (b == 16 || b == 32) && N_pad >= 64) {
      constexpr int TPW = BG_TPW;
      dim3 grid((unsigned)((N_pad / 16 + 4 * TPW - 1) / (4 * TPW)));
      if (b == 16) {
         if (gram_part)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm_lds<1, 12, TPW, true>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, gram_part);
         else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm_lds<1, 12, TPW, false>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, nullptr);
      } else {
         if (gram_part)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm_lds<2, 3, TPW, true>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, gram_part);
         else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm_lds<2, 3, TPW, false>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, nullptr);
      }
      HIP_CHECK_LAUNCH();
      return;
   }
   dim3 grid((unsigned)(N_pad / 64));
   switch (b) {
   case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm<1>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad); break;
   case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm<2>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad); break;
   case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm<3>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad); break;
   case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_block_gemm<4>), grid, dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad); break;
   default: throw Error(-1, "block_gemm: block width must be 16, 32, 48 or 64");
   }
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K4 update + Gram in ONE pass over the basis (round 6; 16 columns):  Out = Init + sum_q A_q C_q  and then  G_q = A_q' Out (q < nq),
// G_nq = Out' Out -- the first projection's update and the second projection's Gram of classical Gram-Schmidt twice, which read the
// basis one after the other in rounds 1-5.  The second needs the COMPLETE updated tile, so a wave that owns a row tile alone
// would have to keep every basis tile of it in registers (the nq + 1 accumulators beside them held round 5's attempt to one wave per
// SIMD from 9 blocks on).  Here the four waves of a workgroup share ONE row tile and split the basis blocks (q = wave, wave + 4, ...):
// each wave loads its blocks' tiles once, in the accumulator layout (lane (li, kq), register r = row kq + 4r, column li: what both
// Gram operands want), turns each through a 2 KB LDS scratch into the row layout the update's A operand wants, accumulates ITS part
// of the update; the four parts meet in LDS, every wave adds them (fixed order) to the tile of Init, wave 0 stores Out, and each wave
// runs the Gram MFMAs of its blocks against the updated tile from the registers it still holds.  One partial plane [nq + 1][16][16] per
// workgroup (summed by reduce_sum).
template <int QW /* basis blocks per wave, at most */>
__global__ __launch_bounds__(256) void k_update_gram16(const double *const *__restrict__ blocks, int nq, const double *__restrict__ C,
                                                        const double *Init, double *Out, uint64_t N_pad, int tiles_per_wg,
                                                        double *__restrict__ gpart)
{
   constexpr int b = 16, TS = 18 /* scratch row stride (doubles) */;
   __shared__ __attribute__((aligned(16))) double Ts[4 * 16 * TS]; // one transposition scratch per wave
   __shared__ double Ps[2][4 * 256];                               // the waves' parts of the update (two tiles in flight: one barrier per tile)
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int li = lane & 15, kq = lane >> 4;
   const int nmine = (nq > wave) ? (nq - wave + 3) / 4 : 0; // basis blocks of this wave: q = wave + 4 qi
   const bool own_w = (nq & 3) == wave;                     // ... and Out' Out with the wave that has room for it (index nq >> 2)
   const double *bp[QW];
#pragma unroll
   for (int qi = 0; qi < QW; qi++) bp[qi] = qi < nmine ? blocks[wave + 4 * qi] : nullptr;
   // the coefficients of a wave's blocks do not change from tile to tile: the B operands of its update MFMAs live in registers
   // (C_q[p = 4 kq + t][c = li]) -- no LDS for them, so the workgroups per CU are bounded by registers alone
   double cr[QW][4];
#pragma unroll
   for (int qi = 0; qi < QW; qi++)
#pragma unroll
      for (int t = 0; t < 4; t++) cr[qi][t] = qi < nmine ? C[(size_t)(wave + 4 * qi) * (b * b) + (4 * kq + t) * b + li] : 0.0;
   d4 G[QW + 1];
#pragma unroll
   for (int qi = 0; qi <= QW; qi++) G[qi] = (d4){0.0, 0.0, 0.0, 0.0};
   double *ts = Ts + wave * 16 * TS;
   const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_wg, ntiles = N_pad / 16;
   int pb = 0;
   for (uint64_t tile = t0; tile < t0 + tiles_per_wg && tile < ntiles; tile++, pb ^= 1) {
      const uint64_t s0 = tile * 16;
      d4 v2[QW], w2;
#pragma unroll
      for (int qi = 0; qi < QW; qi++)
         if (qi < nmine) {
#pragma unroll
            for (int r = 0; r < 4; r++) v2[qi][r] = bp[qi][(s0 + kq + 4 * r) * b + li];
         }
#pragma unroll
      for (int r = 0; r < 4; r++) w2[r] = Init ? Init[(s0 + kq + 4 * r) * b + li] : 0.0;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int qi = 0; qi < QW; qi++)
         if (qi < nmine) { // (wave-uniform)
#pragma unroll
            for (int r = 0; r < 4; r++) ts[(kq + 4 * r) * TS + li] = v2[qi][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const d2 a01 = *reinterpret_cast<const d2 *>(&ts[li * TS + 4 * kq]), a23 = *reinterpret_cast<const d2 *>(&ts[li * TS + 4 * kq + 2]);
            acc = FPCA_MFMA(a01.x, cr[qi][0], acc);
            acc = FPCA_MFMA(a01.y, cr[qi][1], acc);
            acc = FPCA_MFMA(a23.x, cr[qi][2], acc);
            acc = FPCA_MFMA(a23.y, cr[qi][3], acc);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // (the scratch is rewritten by the next block)
         }
#pragma unroll
      for (int r = 0; r < 4; r++) Ps[pb][wave * 256 + (kq + 4 * r) * 16 + li] = acc[r];
      __syncthreads(); // (the other buffer is free again: every wave passed this point of the previous tile after reading it)
      d4 w1;
#pragma unroll
      for (int r = 0; r < 4; r++) {
         const int e = (kq + 4 * r) * 16 + li;
         w1[r] = w2[r] + ((Ps[pb][e] + Ps[pb][256 + e]) + (Ps[pb][512 + e] + Ps[pb][768 + e])); // the same order in every wave: identical tiles
      }
      if (wave == 0) {
#pragma unroll
         for (int r = 0; r < 4; r++) Out[(s0 + kq + 4 * r) * b + li] = w1[r];
      }
#pragma unroll
      for (int qi = 0; qi < QW; qi++)
         if (qi < nmine) {
#pragma unroll
            for (int r = 0; r < 4; r++) G[qi] = FPCA_MFMA(v2[qi][r], w1[r], G[qi]);
         }
      if (own_w) {
#pragma unroll
         for (int qi = 0; qi <= QW; qi++)
            if (qi == (nq >> 2)) {
#pragma unroll
               for (int r = 0; r < 4; r++) G[qi] = FPCA_MFMA(w1[r], w1[r], G[qi]);
            }
      }
   }
   double *gp = gpart + (size_t)blockIdx.x * (nq + 1) * (b * b);
#pragma unroll
   for (int qi = 0; qi <= QW; qi++) {
      const int q = wave + 4 * qi;
      if (q < nq || (q == nq && own_w)) {
#pragma unroll
         for (int r = 0; r < 4; r++) gp[(size_t)q * (b * b) + (kq + 4 * r) * b + li] = G[qi][r];
      }
   }
}

// workgroups of the fused launch = partial planes it leaves (0: no fused kernel for this shape)
int update_gram_planes(uint64_t N_pad, int nq, int b)
{
   if (!g_k4_variant || b != 16 || nq < 1 || nq > 28 || N_pad < 16 * 512 || FPCA_TEST_ENV("FPCA_K4_NO_FUSED")) return 0;
   return 512;
}

void update_gram(const double *const *blocks, int nq, const double *C, const double *Init, double *Out, uint64_t N_pad, int b, double *gpart,
                 hipStream_t stream)
{
   const int planes = update_gram_planes(N_pad, nq, b);
   if (!planes) throw Error(-1, "update_gram: no fused kernel for this shape");
   const uint64_t ntiles = N_pad / 16;
   const int tpw = (int)((ntiles + planes - 1) / planes);
   if (nq <= 16)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_gram16<4>), dim3((unsigned)planes), dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, tpw, gpart);
   else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_update_gram16<7>), dim3((unsigned)planes), dim3(256), 0, stream, blocks, nq, C, Init, Out, N_pad, tpw, gpart);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
__global__ void k_fill_random(double *blk, uint64_t N, uint64_t total, int b, uint64_t seed, uint64_t row0)
{
   for (uint64_t il = (uint64_t)blockIdx.x * 256 + threadIdx.x; il < total; il += (uint64_t)gridDim.x * 256) {
      const uint64_t i = il + row0 * b; // element index in the WHOLE block: a row slice gets the values the whole block would
      const uint64_t s = i / b;
      double v = 0.0;
      if (s < N) {
         const uint64_t h = synth::mix64(synth::mix64(seed ^ 0x5851F42D4C957F2Dull) ^ (i * 0x9E3779B97F4A7C15ull));
         v = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      }
      blk[il] = v;
   }
}

// rows [row0, row0 + rows) of the random N x b block with this seed (rows >= N are zero), written to blk[0 .. rows*b)
void fill_random(double *blk, uint64_t N, uint64_t rows, int b, uint64_t seed, hipStream_t stream, uint64_t row0)
{
   const uint64_t total = rows * b;
   if (!total) return;
   uint64_t blocks = (total + 255) / 256;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_fill_random, dim3((unsigned)blocks), dim3(256), 0, stream, blk, N, total, b, seed, row0);
   HIP_CHECK_LAUNCH();
}

// *out_bits = max(*out_bits, bit pattern of max_i |a[i] - scale b[i]|) -- non-negative doubles order like their bit patterns
// (the self-test of the multi-rank exchange compares what the collectives delivered with what they should have)
__global__ void k_max_abs_diff(const double *a, const double *b, double scale, uint64_t n, unsigned long long *out_bits)
{
   double m = 0;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
      const double d = fabs(a[i] - scale * b[i]);
      m = (d > m || d != d) ? (d != d ? __longlong_as_double(0x7ff0000000000000ll) : d) : m; // a NaN counts as +inf
   }
   for (int o = 32; o; o >>= 1) m = fmax(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(m));
}

void max_abs_diff(const double *a, const double *b, double scale, uint64_t n, unsigned long long *out_bits, hipStream_t stream)
{
   if (!n) return;
   const unsigned blocks = (unsigned)std::min<uint64_t>(2048, (n + 255) / 256);
   hipLaunchKernelGGL(k_max_abs_diff, dim3(blocks), dim3(256), 0, stream, a, b, scale, n, out_bits);
   HIP_CHECK_LAUNCH();
}

__global__ void k_block_to_colmajor(const double *blk, uint64_t N, int b, int ncols, double *out, uint64_t ld)
{
   const uint64_t total = N * ncols;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
      const uint64_t s = i / ncols;
      const int c = (int)(i % ncols);
      out[(uint64_t)c * ld + s] = blk[s * b + c];
   }
}

void block_to_colmajor(const double *blk, uint64_t N, int b, int ncols, double *out, uint64_t ld, hipStream_t stream)
{
   uint64_t blocks = (N * ncols + 255) / 256;
   if (blocks == 0) return;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_block_to_colmajor, dim3((unsigned)blocks), dim3(256), 0, stream, blk, N, b, ncols, out, ld);
   HIP_CHECK_LAUNCH();
}

__global__ void k_colmajor_to_block(const double *in, uint64_t ld, uint64_t N, uint64_t N_pad, int b, int ncols,
                                    double *blk)
{
   const uint64_t total = N_pad * b;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
      const uint64_t s = i / b;
      const int c = (int)(i % b);
      blk[i] = (s < N && c < ncols) ? in[(uint64_t)c * ld + s] : 0.0;
   }
}

void colmajor_to_block(const double *in, uint64_t ld, uint64_t N, uint64_t N_pad, int b, int ncols, double *blk,
                       hipStream_t stream)
{
   uint64_t blocks = (N_pad * b + 255) / 256;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_colmajor_to_block, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld, N, N_pad, b, ncols, blk);
   HIP_CHECK_LAUNCH();
}

__global__ void k_t_to_colmajor(const double *T, uint64_t P_g, int b, int ncols, const double *colscale, double *out,
                                uint64_t ld)
{
   const uint64_t total = P_g * ncols;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
      const uint64_t j = i / ncols;
      const int c = (int)(i % ncols);
      const double sc = colscale ? colscale[c] : 1.0;
      out[(uint64_t)c * ld + j] = T[j * b + c] * sc;
   }
}

void t_to_colmajor(const double *T, uint64_t P_g, int b, int ncols, const double *colscale, double *out, uint64_t ld,
                   hipStream_t stream)
{
   uint64_t blocks = (P_g * ncols + 255) / 256;
   if (blocks == 0) return;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_t_to_colmajor, dim3((unsigned)blocks), dim3(256), 0, stream, T, P_g, b, ncols, colscale, out, ld);
   HIP_CHECK_LAUNCH();
}

__global__ void k_colmajor_to_t(const double *in, uint64_t ld, uint64_t P_g, uint64_t P_pad, int b, int ncols, double *T)
{
   const uint64_t total = P_pad * b;
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
      const uint64_t j = i / b;
      const int c = (int)(i % b);
      T[i] = (j < P_g && c < ncols) ? in[(uint64_t)c * ld + j] : 0.0;
   }
}

void colmajor_to_t(const double *in, uint64_t ld, uint64_t P_g, uint64_t P_pad, int b, int ncols, double *T,
                   hipStream_t stream)
{
   uint64_t blocks = (P_pad * b + 255) / 256;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_colmajor_to_t, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld, P_g, P_pad, b, ncols, T);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// synthetic genotypes straight into HBM: one workgroup per SNP record, one 32-bit word (16 samples) per
// thread per iteration; integer-only model of synth.hpp, so the host can reproduce any record bit for bit.
__global__ __launch_bounds__(256) void k_synth_generate(uint8_t *packed, size_t pitch, uint64_t N, uint64_t snp_begin,
                                                         uint64_t seed, int n_pop, uint32_t fst_fp, uint32_t miss_thr, int maf_model,
                                                         int missing_model, uint32_t conc_fp, uint64_t med_q32, uint32_t sig2_fp)
{
   __shared__ uint32_t thr[synth::MAX_POP];
   __shared__ uint64_t bnd[synth::MAX_POP + 1];
   const uint64_t snp = snp_begin + blockIdx.x;
   const uint32_t pj = maf_model == 1 ? synth::snp_freq_rare(seed, snp) : synth::snp_freq(seed, snp);
   if (missing_model == 1) miss_thr = synth::snp_miss_thr_concentrated(seed, snp, conc_fp);
   if (missing_model == 2) miss_thr = synth::snp_miss_thr_lognormal(seed, snp, med_q32, sig2_fp);
   if ((int)threadIdx.x < n_pop) thr[threadIdx.x] = synth::pop_freq(seed, snp, pj, fst_fp, (int)threadIdx.x);
   if ((int)threadIdx.x <= n_pop) bnd[threadIdx.x] = synth::pop_boundary(N, n_pop, (int)threadIdx.x);
   __syncthreads();
   uint32_t *row = reinterpret_cast<uint32_t *>(packed + (size_t)blockIdx.x * pitch);
   const uint32_t nwords = (uint32_t)(pitch / 4);
   for (uint32_t w = threadIdx.x; w < nwords; w += 256) {
      const uint64_t s0 = (uint64_t)w * 16;
      uint32_t out = 0x55555555u; // padding = missing
      if (s0 < N) {
         int lo = 0, hi = n_pop; // population of s0: largest c with bnd[c] <= s0
         while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (bnd[mid] <= s0)
               lo = mid;
            else
               hi = mid;
         }
         int pop = lo;
         out = 0;
         for (int s = 0; s < 16; s++) {
            const uint64_t smp = s0 + s;
            uint32_t code = 1u;
            if (smp < N) {
               while (pop + 1 < n_pop && smp >= bnd[pop + 1]) pop++;
               code = synth::cell_code(seed, snp, smp, thr[pop], miss_thr);
            }
            out |= code << (2 * s);
         }
      }
      row[w] = out;
   }
}

void synth_generate(uint8_t *packed, size_t pitch, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed,
                    int n_pop, uint32_t fst_fp, uint32_t miss_thr, hipStream_t stream, int maf_model, int missing_model, uint32_t conc_fp,
                    uint64_t med_q32, uint32_t sig2_fp)
{
   if (P_g == 0) return;
   hipLaunchKernelGGL(k_synth_generate, dim3((unsigned)P_g), dim3(256), 0, stream, packed, pitch, N, snp_begin, seed,
                      n_pop, fst_fp, miss_thr, maf_model, missing_model, conc_fp, med_q32, sig2_fp);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
__global__ void k_mfma_layout_probe(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/,
                                    double *D /*16x16 row-major*/)
{
   const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
   d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
   acc = FPCA_MFMA(A[li * 4 + kq], B[kq * 16 + li], acc);
   for (int r = 0; r < 4; r++) D[(kq + 4 * r) * 16 + li] = acc[r];
}

void mfma_layout_probe(const double *A, const double *B, double *D, hipStream_t stream)
{
   hipLaunchKernelGGL(k_mfma_layout_probe, dim3(1), dim3(64), 0, stream, A, B, D);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// micro-benchmark: sustained rate of a pure v_mfma_f64_16x16x4_f64 stream (explicit registers through inline asm:
// 8 in-place accumulators v[0:63], operands in v[64:95], no memory traffic).  pattern 0: one A/B pair shared by all
// MFMAs; 1: 4 A x 2 B in GEMM order (the kernels' shape); 2: same, no operand shared by consecutive MFMAs; 3: 8
// distinct A/B pairs.  Measured on MI355X: 64.2-69.4 TFLOP/s with one wave per SIMD, 72.3-74.2 TFLOP/s with two --
// the practical ceiling to read the GEMM kernels' 63-70 TFLOP/s against (datasheet: 78.6).
__global__ __launch_bounds__(256, 1) void k_mfma_peak(double *out, int iters, int pattern)
{
   asm volatile(
      "v_mov_b64 v[0:1], 0\n\t"
      "v_mov_b64 v[2:3], 0\n\t"
      "v_mov_b64 v[4:5], 0\n\t"
      "v_mov_b64 v[6:7], 0\n\t"
      "v_mov_b64 v[8:9], 0\n\t"
      "v_mov_b64 v[10:11], 0\n\t"
      "v_mov_b64 v[12:13], 0\n\t"
      "v_mov_b64 v[14:15], 0\n\t"
      "v_mov_b64 v[16:17], 0\n\t"
      "v_mov_b64 v[18:19], 0\n\t"
      "v_mov_b64 v[20:21], 0\n\t"
      "v_mov_b64 v[22:23], 0\n\t"
      "v_mov_b64 v[24:25], 0\n\t"
      "v_mov_b64 v[26:27], 0\n\t"
      "v_mov_b64 v[28:29], 0\n\t"
      "v_mov_b64 v[30:31], 0\n\t"
      "v_mov_b64 v[32:33], 0\n\t"
      "v_mov_b64 v[34:35], 0\n\t"
      "v_mov_b64 v[36:37], 0\n\t"
      "v_mov_b64 v[38:39], 0\n\t"
      "v_mov_b64 v[40:41], 0\n\t"
      "v_mov_b64 v[42:43], 0\n\t"
      "v_mov_b64 v[44:45], 0\n\t"
      "v_mov_b64 v[46:47], 0\n\t"
      "v_mov_b64 v[48:49], 0\n\t"
      "v_mov_b64 v[50:51], 0\n\t"
      "v_mov_b64 v[52:53], 0\n\t"
      "v_mov_b64 v[54:55], 0\n\t"
      "v_mov_b64 v[56:57], 0\n\t"
      "v_mov_b64 v[58:59], 0\n\t"
      "v_mov_b64 v[60:61], 0\n\t"
      "v_mov_b64 v[62:63], 0\n\t"
      "v_mov_b64 v[64:65], 0\n\t"
      "v_mov_b64 v[66:67], 0\n\t"
      "v_mov_b64 v[68:69], 0\n\t"
      "v_mov_b64 v[70:71], 0\n\t"
      "v_mov_b64 v[72:73], 0\n\t"
      "v_mov_b64 v[74:75], 0\n\t"
      "v_mov_b64 v[76:77], 0\n\t"
      "v_mov_b64 v[78:79], 0\n\t"
      "v_mov_b64 v[80:81], 0\n\t"
      "v_mov_b64 v[82:83], 0\n\t"
      "v_mov_b64 v[84:85], 0\n\t"
      "v_mov_b64 v[86:87], 0\n\t"
      "v_mov_b64 v[88:89], 0\n\t"
      "v_mov_b64 v[90:91], 0\n\t"
      "v_mov_b64 v[92:93], 0\n\t"
      "v_mov_b64 v[94:95], 0\n\t"
      ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
   for (int it = 0; it < iters; it++) {
      if (pattern == 0) {
         asm volatile(
            "v_mfma_f64_16x16x4_f64 v[0:7], v[64:65], v[66:67], v[0:7]\n\t"
            "v_mfma_f64_16x16x4_f64 v[8:15], v[64:65], v[66:67], v[8:15]\n\t"
            "v_mfma_f64_16x16x4_f64 v[16:23], v[64:65], v[66:67], v[16:23]\n\t"
            "v_mfma_f64_16x16x4_f64 v[24:31], v[64:65], v[66:67], v[24:31]\n\t"
            "v_mfma_f64_16x16x4_f64 v[32:39], v[64:65], v[66:67], v[32:39]\n\t"
            "v_mfma_f64_16x16x4_f64 v[40:47], v[64:65], v[66:67], v[40:47]\n\t"
            "v_mfma_f64_16x16x4_f64 v[48:55], v[64:65], v[66:67], v[48:55]\n\t"
            "v_mfma_f64_16x16x4_f64 v[56:63], v[64:65], v[66:67], v[56:63]\n\t"
            ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
      }
      else if (pattern == 1) {
         asm volatile(
            "v_mfma_f64_16x16x4_f64 v[0:7], v[64:65], v[80:81], v[0:7]\n\t"
            "v_mfma_f64_16x16x4_f64 v[8:15], v[64:65], v[84:85], v[8:15]\n\t"
            "v_mfma_f64_16x16x4_f64 v[16:23], v[68:69], v[80:81], v[16:23]\n\t"
            "v_mfma_f64_16x16x4_f64 v[24:31], v[68:69], v[84:85], v[24:31]\n\t"
            "v_mfma_f64_16x16x4_f64 v[32:39], v[72:73], v[80:81], v[32:39]\n\t"
            "v_mfma_f64_16x16x4_f64 v[40:47], v[72:73], v[84:85], v[40:47]\n\t"
            "v_mfma_f64_16x16x4_f64 v[48:55], v[76:77], v[80:81], v[48:55]\n\t"
            "v_mfma_f64_16x16x4_f64 v[56:63], v[76:77], v[84:85], v[56:63]\n\t"
            ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
      }
      else if (pattern == 2) {
         asm volatile(
            "v_mfma_f64_16x16x4_f64 v[0:7], v[64:65], v[80:81], v[0:7]\n\t"
            "v_mfma_f64_16x16x4_f64 v[24:31], v[68:69], v[84:85], v[24:31]\n\t"
            "v_mfma_f64_16x16x4_f64 v[32:39], v[72:73], v[80:81], v[32:39]\n\t"
            "v_mfma_f64_16x16x4_f64 v[56:63], v[76:77], v[84:85], v[56:63]\n\t"
            "v_mfma_f64_16x16x4_f64 v[8:15], v[64:65], v[84:85], v[8:15]\n\t"
            "v_mfma_f64_16x16x4_f64 v[16:23], v[68:69], v[80:81], v[16:23]\n\t"
            "v_mfma_f64_16x16x4_f64 v[40:47], v[72:73], v[84:85], v[40:47]\n\t"
            "v_mfma_f64_16x16x4_f64 v[48:55], v[76:77], v[80:81], v[48:55]\n\t"
            ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
      }
      else if (pattern == 3) {
         asm volatile(
            "v_mfma_f64_16x16x4_f64 v[0:7], v[64:65], v[66:67], v[0:7]\n\t"
            "v_mfma_f64_16x16x4_f64 v[8:15], v[68:69], v[70:71], v[8:15]\n\t"
            "v_mfma_f64_16x16x4_f64 v[16:23], v[72:73], v[74:75], v[16:23]\n\t"
            "v_mfma_f64_16x16x4_f64 v[24:31], v[76:77], v[78:79], v[24:31]\n\t"
            "v_mfma_f64_16x16x4_f64 v[32:39], v[80:81], v[82:83], v[32:39]\n\t"
            "v_mfma_f64_16x16x4_f64 v[40:47], v[84:85], v[86:87], v[40:47]\n\t"
            "v_mfma_f64_16x16x4_f64 v[48:55], v[88:89], v[90:91], v[48:55]\n\t"
            "v_mfma_f64_16x16x4_f64 v[56:63], v[92:93], v[94:95], v[56:63]\n\t"
            ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
      }
   }
   asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
   if (iters < 0) out[0] = 1.0;
}

double mfma_peak_tflops(int waves_per_simd, int iters, int pattern, hipStream_t stream)
{
   double *d = nullptr;
   if (hipMalloc(&d, 8) != hipSuccess) throw Error(-3, "hipMalloc failed");
   const int blocks = 256 * waves_per_simd; // one 256-thread workgroup = one wave per SIMD of a CU
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, stream, d, iters / 10, pattern);
   (void)hipEventRecord(e0, stream);
   hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, stream, d, iters, pattern);
   (void)hipEventRecord(e1, stream);
   (void)hipEventSynchronize(e1);
   float ms = 0;
   (void)hipEventElapsedTime(&ms, e0, e1);
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   (void)hipFree(d);
   const double flops = (double)blocks * 4 /*waves*/ * (double)iters * 8 * 2048.0;
   return flops / (ms * 1e-3) / 1e12;
}

// The same for the other matrix instructions the FP kernels could run on, with compiler-scheduled intrinsics (8 independent
// accumulators, 4 A x 2 B operands in GEMM order, no memory traffic): KIND 0 v_mfma_f32_16x16x4_f32, 1 v_mfma_f32_32x32x2_f32,
// 2 v_mfma_f64_16x16x4_f64.  fill = 0: zero operands (the issue-limited ceiling); else pseudo-random operands in (-1, 1) -- the
// multipliers toggle like the real kernels' and the package power cap decides (round 6: the ceiling `fp32_frac` is read against).
typedef float v16f __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256, 1) void k_mfma_fp_peak(float *out, int iters, uint32_t fill)
{
   uint32_t h = (threadIdx.x + 1u) * 2654435761u ^ fill;
   auto rnd = [&]() {
      h = h * 1664525u + 1013904223u;
      return fill ? (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f : 0.0f;
   };
   if constexpr (KIND == 0) {
      float a[4], b[2];
      for (float &x : a) x = rnd();
      for (float &x : b) x = rnd();
      f4 acc[4][2];
      for (auto &r : acc)
         for (auto &x : r) x = (f4){0, 0, 0, 0};
      for (int it = 0; it < iters; it++) {
#pragma unroll
         for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      float sum = 0;
      for (auto &r : acc)
         for (auto &x : r) sum += x[0] + x[1] + x[2] + x[3];
      if (iters < 0) out[threadIdx.x] = sum;
   } else if constexpr (KIND == 1) {
      float a[2], b[2];
      for (float &x : a) x = rnd();
      for (float &x : b) x = rnd();
      v16f acc[2][2];
      for (auto &r : acc)
         for (auto &x : r)
            for (int q = 0; q < 16; q++) x[q] = 0;
      for (int it = 0; it < iters; it++) {
#pragma unroll
         for (int rep = 0; rep < 2; rep++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
               for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      float sum = 0;
      for (auto &r : acc)
         for (auto &x : r)
            for (int q = 0; q < 16; q++) sum += x[q];
      if (iters < 0) out[threadIdx.x] = sum;
   } else {
      double a[4], b[2];
      for (double &x : a) x = rnd();
      for (double &x : b) x = rnd();
      d4 acc[4][2];
      for (auto &r : acc)
         for (auto &x : r) x = (d4){0, 0, 0, 0};
      for (int it = 0; it < iters; it++) {
#pragma unroll
         for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      double sum = 0;
      for (auto &r : acc)
         for (auto &x : r) sum += x[0] + x[1] + x[2] + x[3];
      if (iters < 0) out[threadIdx.x] = (float)sum;
   }
}

// How much matrix-pipe time do plain VALU instructions cost?  8 independent v_mfma_f32_16x16x4_f32 (32 cycles each) per iteration with
// VPM independent v_add_u32 per MFMA, either interleaved (one MFMA, VPM adds, ...) or in bursts (8 MFMAs, then 8 VPM adds: the shape
// of the GEMM kernels' software-pipelined steps).  LDSR > 0: additionally LDSR conflict-free ds_read_b64 per MFMA whose results are
// waited for one iteration later.  Returns the MFMA rate in TFLOP/s (scripts/mfma_valu_mix.py).
template <int VPM, bool BURST, int LDSR>
__global__ __launch_bounds__(256, 1) void k_mfma_valu_mix(float *out, int iters)
{
   __shared__ float lds[4096];
   for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
   __syncthreads();
   float a[4], b[2];
   for (int i = 0; i < 4; i++) a[i] = 1.0f + i + threadIdx.x * 1e-3f;
   for (int i = 0; i < 2; i++) b[i] = 0.5f + i;
   f4 acc[8];
   for (auto &x : acc) x = (f4){0, 0, 0, 0};
   uint32_t v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
   f2 ld[8];
   for (auto &x : ld) x = (f2){0, 0};
   const uint32_t laddr = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)lds + (threadIdx.x & 63) * 8;
   for (int it = 0; it < iters; it++) {
      if constexpr (BURST) {
#pragma unroll
         for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3] + ld[i].x, b[i >> 2], acc[i], 0, 0, 0);
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 8 * LDSR; i++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[i & 7]) : "v"(laddr), "n"((i & 7) * 512));
#pragma unroll
         for (int i = 0; i < 8 * VPM; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
         if constexpr (LDSR > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
         for (int i = 0; i < 8; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3] + ld[i].x, b[i >> 2], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < LDSR; r++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[(i + 4) & 7]) : "v"(laddr), "n"(((i + 4) & 7) * 512));
#pragma unroll
            for (int j = 0; j < VPM; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[(i + j) & 7]) : "v"(v[(i + j + 1) & 7]));
            if constexpr (LDSR > 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LDSR * 3) : "memory");
            __builtin_amdgcn_sched_barrier(0);
         }
      }
   }
   float sum = 0;
   for (auto &x : acc) sum += x[0] + x[1] + x[2] + x[3];
   for (auto x : v) sum += (float)x;
   if (iters < 0) out[threadIdx.x] = sum;
}

double mfma_valu_mix_tflops(int vpm, bool burst, int ldsr, int waves_per_simd, int iters, hipStream_t stream)
{
   float *d = nullptr;
   if (hipMalloc(&d, 1024) != hipSuccess) throw Error(-3, "hipMalloc failed");
   const int blocks = 256 * waves_per_simd;
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   auto go = [&](int n) {
#define FPCA_MIX(V_, B_, L_)                                                                                  \
   if (vpm == V_ && burst == B_ && ldsr == L_) {                                                             \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mfma_valu_mix<V_, B_, L_>), dim3(blocks), dim3(256), 0, stream, d, n); \
      return;                                                                                                \
   }
      FPCA_MIX(0, false, 0) FPCA_MIX(1, false, 0) FPCA_MIX(2, false, 0) FPCA_MIX(3, false, 0) FPCA_MIX(4, false, 0) FPCA_MIX(6, false, 0) FPCA_MIX(8, false, 0)
      FPCA_MIX(1, true, 0) FPCA_MIX(2, true, 0) FPCA_MIX(3, true, 0) FPCA_MIX(4, true, 0) FPCA_MIX(6, true, 0) FPCA_MIX(8, true, 0)
      FPCA_MIX(0, false, 1) FPCA_MIX(2, false, 1) FPCA_MIX(3, false, 1) FPCA_MIX(0, true, 1) FPCA_MIX(2, true, 1) FPCA_MIX(3, true, 1)
#undef FPCA_MIX
      throw Error(-1, "mfma_valu_mix: variant not instantiated");
   };
   go(iters / 10);
   (void)hipEventRecord(e0, stream);
   go(iters);
   (void)hipEventRecord(e1, stream);
   (void)hipEventSynchronize(e1);
   float ms = 0;
   (void)hipEventElapsedTime(&ms, e0, e1);
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   (void)hipFree(d);
   return (double)blocks * 4 * (double)iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
}

double mfma_fp_peak_tflops(int kind, int waves_per_simd, int iters, uint32_t fill, hipStream_t stream)
{
   float *d = nullptr;
   if (hipMalloc(&d, 1024) != hipSuccess) throw Error(-3, "hipMalloc failed");
   const int blocks = 256 * waves_per_simd;
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   auto go = [&](int n) {
      if (kind == 0)
         hipLaunchKernelGGL(k_mfma_fp_peak<0>, dim3(blocks), dim3(256), 0, stream, d, n, fill);
      else if (kind == 1)
         hipLaunchKernelGGL(k_mfma_fp_peak<1>, dim3(blocks), dim3(256), 0, stream, d, n, fill);
      else
         hipLaunchKernelGGL(k_mfma_fp_peak<2>, dim3(blocks), dim3(256), 0, stream, d, n, fill);
   };
   go(iters / 10);
   (void)hipEventRecord(e0, stream);
   go(iters);
   (void)hipEventRecord(e1, stream);
   (void)hipEventSynchronize(e1);
   float ms = 0;
   (void)hipEventElapsedTime(&ms, e0, e1);
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   (void)hipFree(d);
   const double flops = (double)blocks * 4 * (double)iters * 8 * (kind == 1 ? 4096.0 : 2048.0);
   return flops / (ms * 1e-3) / 1e12;
}

// ------------------------------------------------------------------------------------------------
// diagnostic: where does the dispatcher place the workgroups of a chip-sized grid?  Every workgroup records its
// hardware ids and then spins so that the whole grid is co-resident.  (Measured: 256/512/768/1024 workgroups of 256
// threads land exactly 1/2/3/4 per CU, 32/64/96/128 per XCD.)
__global__ __launch_bounds__(256, 2) void k_census(uint32_t *out, long long spin)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
   if (threadIdx.x == 0) {
      const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
      out[blockIdx.x * 2] = hw;
      out[blockIdx.x * 2 + 1] = xcc;
      smem_raw[0] = (unsigned char)hw;
   }
   const long long t0 = clock64();
   while (clock64() - t0 < spin) {
   }
}

void census(uint32_t *d_out, int nwg, size_t lds_bytes, long long spin, hipStream_t stream)
{
   (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_census), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
   hipLaunchKernelGGL(k_census, dim3(nwg), dim3(256), lds_bytes, stream, d_out, spin);
   HIP_CHECK_LAUNCH();
}

} // namespace kern
} // namespace fpca
