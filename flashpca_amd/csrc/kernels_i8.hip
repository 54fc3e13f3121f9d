// kernels_i8.hip -- exact-integer flavour of the two genotype GEMMs (FPCA_ACCUM_I8(S)).
//
// A standardised genotype is x_ij = m_ij (g_ij - mu_j) / sigma_j with g in {0,1,2} the dosage and m in {0,1} the
// "not missing" flag, so
//     T = X' B = diag(1/sigma) [ (G.M)' B  -  diag(mu) M' B ]           (K2)
//     Y = X T  = (G.M) (T / sigma)  -  M (mu T / sigma)                  (K3)
// where G.M and M are tiny integers.  The fp64 operand (B, resp. T/sigma and mu T/sigma) is split per column into S
// signed 7-bit slices sharing one power-of-two scale (q = d_0/64 + d_1/(64*128) + ..., |d_s| <= 64, remainder
// < 2^-7S dropped), and the products run on v_mfma_i32_32x32x32_i8 with EXACT int32 accumulation (|sum| <= 128 N).
// The only rounding of the whole product is the 2^-7S truncation of the fp64 operand and the final fp64
// recombination -- there is no accumulation error at all -- so S = 8 (56 bits) is fp64-equivalent while the int8
// MFMA runs ~64x faster than the fp64 one for 16x the multiply-adds.
//
// MFMA operand maps (32x32x32 i8): lane l holds 16 int8 of A row i = l&31 and of B column j = l&31 for the K-half
// l>>5; A and B use the same (half, byte) -> k assignment, so the dot products do not depend on it.  C/D register r of
// lane l is D[row = (r&3) + 8 (r>>2) + 4 (l>>5)][col = l&31].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace fpca {
namespace kern {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define HIP_CHECK_LAUNCH()                                                                         \
   do {                                                                                             \
      hipError_t e__ = hipGetLastError();                                                           \
      if (e__ != hipSuccess) throw Error(-3, std::string("kernel launch failed: ") + hipGetErrorString(e__)); \
   } while (0)

// ------------------------------------------------------------------------------------------------
// slicing of an fp64 operand V[rows_pad][b] (row-major, rows = samples for K2, SNPs for K3) into
// Q[S*b][rows_pad] int8 (slice-column sc = s*b + c, contiguous along the rows = the GEMM's K dimension).
// Inside every aligned group of 16 rows the bytes are stored in the order the decoded genotype operand uses:
// position 4q + t holds row q + 4t (see decode in k_gemm_i8).

// per-column max |v * rowscale|; doubles >= 0 order like their bit patterns, so an integer atomicMax is exact
__global__ __launch_bounds__(256) void k_colmax(const double *__restrict__ V, const double *__restrict__ rowscale,
                                                uint64_t rows, int b, unsigned long long *__restrict__ colmax_bits)
{
   const int c = threadIdx.x % b;
   const int r_in = threadIdx.x / b, r_step = 256 / b;
   if (r_in >= r_step) return;
   double m = 0.0;
   for (uint64_t r = (uint64_t)blockIdx.x * r_step + r_in; r < rows; r += (uint64_t)gridDim.x * r_step) {
      double v = V[r * b + c];
      if (rowscale) v *= rowscale[r];
      v = fabs(v);
      if (v > m) m = v; // NaN never wins: a NaN operand would poison the fp64 path as well
   }
   atomicMax(&colmax_bits[c], (unsigned long long)__double_as_longlong(m));
}

// weights: colw[s*b + c] = 2^e_c / 64 / 128^s with 2^e_c > max|column c|  (e_c from frexp; zero column -> e = 0)
__global__ void k_slice_weights(const unsigned long long *__restrict__ colmax_bits, int b, int S, double *__restrict__ colw,
                                double *__restrict__ colinv)
{
   const int c = threadIdx.x;
   if (c >= b) return;
   const double m = __longlong_as_double((long long)colmax_bits[c]);
   int e = 0;
   if (m > 0.0 && isfinite(m)) (void)frexp(m, &e); // m = f 2^e, f in [0.5, 1)  =>  m < 2^e
   colinv[c] = ldexp(1.0, -e);
   double w = ldexp(1.0, e - 6);
   for (int s = 0; s < S; s++) {
      colw[s * b + c] = w;
      w *= 1.0 / 128.0;
   }
}

// one thread = one 16-row group of one column: 16 strided fp64 in, one 16-byte store per slice out
__global__ __launch_bounds__(256) void k_slice(const double *__restrict__ V, const double *__restrict__ rowscale, uint64_t rows_pad,
                                               uint64_t rows, int b, int S, const double *__restrict__ colinv,
                                               int8_t *__restrict__ Q)
{
   const uint64_t groups = rows_pad / 16;
   for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < groups * b; t += (uint64_t)gridDim.x * 256) {
      const int c = (int)(t % b);
      const uint64_t r0 = (t / b) * 16;
      const double sc = colinv[c] * 64.0;
      double q[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
         const uint64_t r = r0 + j;
         double v = (r < rows) ? V[r * b + c] : 0.0;
         if (rowscale && r < rows) v *= rowscale[r];
         q[j] = v * sc; // |q| < 64; exact (power-of-two scaling)
      }
      for (int s = 0; s < S; s++) {
         u4 word = {0u, 0u, 0u, 0u};
#pragma unroll
         for (int j = 0; j < 16; j++) {
            const double d = rint(q[j]);
            q[j] = (q[j] - d) * 128.0; // exact
            word[j & 3] |= ((uint32_t)(int)d & 0xFFu) << (8 * (j >> 2)); // position 4 (j&3) + (j>>2)
         }
         *reinterpret_cast<u4 *>(Q + ((uint64_t)(s * b + c)) * rows_pad + r0) = word;
      }
   }
}

void slice_operand(const double *V, const double *rowscale, uint64_t rows_pad, uint64_t rows, int b, int S, int8_t *Q,
                   double *colw /* [S*b] */, double *scratch /* >= 2*b doubles */, hipStream_t stream)
{
   unsigned long long *bits = reinterpret_cast<unsigned long long *>(scratch);
   double *colinv = scratch + b;
   (void)hipMemsetAsync(bits, 0, sizeof(unsigned long long) * b, stream);
   unsigned blocks = (unsigned)std::min<uint64_t>(1024, rows * b / 256 + 1);
   hipLaunchKernelGGL(k_colmax, dim3(blocks), dim3(256), 0, stream, V, rowscale, rows, b, bits);
   HIP_CHECK_LAUNCH();
   hipLaunchKernelGGL(k_slice_weights, dim3(1), dim3(64), 0, stream, bits, b, S, colw, colinv);
   HIP_CHECK_LAUNCH();
   blocks = (unsigned)std::min<uint64_t>(16384, (rows_pad / 16 * b + 255) / 256);
   hipLaunchKernelGGL(k_slice, dim3(blocks), dim3(256), 0, stream, V, rowscale, rows_pad, rows, b, S, colinv, Q);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K2i / K3i core:  acc[row][sc] = sum_k A[row][k] * Q[sc][k]  for A in {G.M, M}
//   `packed`: 2-bit records, one per output row (K2: the SNP-major stream, K = samples; K3: its sample-major copy,
//   K = SNPs).  Workgroup = 8 waves = 128 rows x 256 slice-columns; wave (wr, wc) owns 64 rows x 64 columns of BOTH
//   integer matrices (2 x 2 x 2 accumulators of 32x32), so one LDS operand read feeds 4 MFMAs (TWO = false, K2: both
//   matrices multiply the same Q) or 2 (TWO = true, K3: Qg = slices of T/sd, Qm = slices of mean T/sd).
//   Per KC-chunk the Q tile(s) [256 sc][KC] and the packed tile [128 rows][KC/4 B] are double-buffered in LDS (one
//   barrier per chunk; row strides KC+16 / KC/4+16 bytes keep the ds_read_b128 of 16 lanes on distinct banks).
//   Decode: a lane's dword w holds 16 codes; (w >> 2q) & 0x03030303 leaves codes q, q+4, q+8, q+12 in the four bytes
//   and v_perm_b32 with the code as selector looks G.M / M up in a 4-byte table -- 4 VALU ops per operand dword
//   pair; the Q bytes were stored in the matching order by k_slice.
//   Output: int32 partials part[split][row][mat][NSC], combined exactly by k_i8_combine.
constexpr int I8_ROWS = 128;
constexpr int I8_COLS = 256;

template <bool TWO>
struct I8Cfg {
   static constexpr int KC = TWO ? 128 : 256;
   static constexpr int LDQ = KC + 16;               // Q tile row stride (bytes)
   static constexpr int LDP = KC / 4 + 16;           // packed tile row stride (bytes)
   static constexpr int NQ = TWO ? 2 : 1;            // Q tiles per stage
   static constexpr int STAGE = NQ * I8_COLS * LDQ + I8_ROWS * LDP; // bytes per LDS stage
   static constexpr int QPIECES = NQ * I8_COLS * (KC / 16) / 512;   // 16-byte pieces per thread per chunk (8)
};

template <bool TWO>
__global__ __launch_bounds__(512, 1) void k_gemm_i8(const uint8_t *__restrict__ packed, size_t pitch,
                                                     const int8_t *__restrict__ Qg, const int8_t *__restrict__ Qm,
                                                     uint64_t k_pad, int nsc_total, int *__restrict__ part, uint64_t rows_pad,
                                                     int chunks_total, int chunks_per_split)
{
   using C = I8Cfg<TWO>;
   constexpr int KC = C::KC, LDQ = C::LDQ, LDP = C::LDP, KS = KC / 32, NP = C::QPIECES;
   static_assert(NP == 8, "staging assumes 8 pieces per thread");
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 31, kh = lane >> 5;
   const int wr = wave >> 2, wc = wave & 3;
   const uint64_t row0 = (uint64_t)blockIdx.x * I8_ROWS;
   const int col0 = blockIdx.z * I8_COLS;
   const int c_begin = blockIdx.y * chunks_per_split;
   int c_end = c_begin + chunks_per_split;
   if (c_end > chunks_total) c_end = chunks_total;

   v16i acc[2][2][2]; // [mat][m][n]
#pragma unroll
   for (int a = 0; a < 2; a++)
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
         for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][m][n][r] = 0;

   // staging assignment: Q piece p = tid + 512 r (r < 8): tile p / (256*KC/16), row, 16-byte segment
   constexpr int SEGS = KC / 16;
   const int8_t *qsrc[NP];
   int qdst[NP];
#pragma unroll
   for (int r = 0; r < NP; r++) {
      const int p = tid + 512 * r;
      const int tile = p / (I8_COLS * SEGS), pp = p % (I8_COLS * SEGS);
      const int row = pp / SEGS, seg = pp % SEGS;
      qsrc[r] = ((TWO && tile) ? Qm : Qg) + (uint64_t)(col0 + row) * k_pad + seg * 16;
      qdst[r] = tile * I8_COLS * LDQ + row * LDQ + seg * 16;
   }
   // packed tile: 128 rows x KC/4 bytes = 128 * KC/64 pieces of 16 bytes  (KC = 256: 512 pieces, KC = 128: 256)
   constexpr int PSEGS = KC / 64;
   const bool p_active = tid < I8_ROWS * PSEGS;
   const int prow = tid / PSEGS, pseg = tid % PSEGS;
   const uint8_t *psrc = packed + (row0 + (p_active ? prow : 0)) * pitch + pseg * 16;
   const int pdst = C::NQ * I8_COLS * LDQ + prow * LDP + pseg * 16;

   u4 qreg[NP], preg;
#define FPCA_I8_LOAD(cc)                                                                      \
   {                                                                                           \
      _Pragma("unroll") for (int r = 0; r < NP; r++) qreg[r] = *reinterpret_cast<const u4 *>(qsrc[r] + (uint64_t)(cc) * KC); \
      if (p_active) preg = *reinterpret_cast<const u4 *>(psrc + (size_t)(cc) * (KC / 4));      \
   }
#define FPCA_I8_STORE(buf)                                                                     \
   {                                                                                           \
      unsigned char *st = smem + (size_t)(buf) * C::STAGE;                                     \
      _Pragma("unroll") for (int r = 0; r < NP; r++) *reinterpret_cast<u4 *>(st + qdst[r]) = qreg[r]; \
      if (p_active) *reinterpret_cast<u4 *>(st + pdst) = preg;                                 \
   }
   if (c_begin < c_end) {
      FPCA_I8_LOAD(c_begin);
      FPCA_I8_STORE(0);
   }
   __syncthreads();

   const uint32_t tabG = 0x00010002u, tabM = 0x01010001u; // byte[code]: code 0 -> (2,1), 1 (missing) -> (0,0), 2 -> (1,1), 3 -> (0,1)
   for (int c = c_begin; c < c_end; c++) {
      const int buf = (c - c_begin) & 1;
      if (c + 1 < c_end) FPCA_I8_LOAD(c + 1);
      const unsigned char *st = smem + (size_t)buf * C::STAGE;
      const unsigned char *sQ = st + (size_t)(wc * 64 + li) * LDQ + kh * (KC / 2);
      const unsigned char *sP = st + C::NQ * I8_COLS * LDQ + (size_t)(wr * 64 + li) * LDP + kh * (KC / 8);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) { // this lane half covers k = (KC/2) kh + 16 ks .. +15 of the chunk
         v4i ag[2], am[2];
#pragma unroll
         for (int m = 0; m < 2; m++) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(sP + (size_t)(32 * m) * LDP + ks * 4);
#pragma unroll
            for (int q = 0; q < 4; q++) {
               const uint32_t sel = (w >> (2 * q)) & 0x03030303u;
               ag[m][q] = (int)__builtin_amdgcn_perm(0u, tabG, sel);
               am[m][q] = (int)__builtin_amdgcn_perm(0u, tabM, sel);
            }
         }
#pragma unroll
         for (int n = 0; n < 2; n++) {
            const v4i bg = *reinterpret_cast<const v4i *>(sQ + (size_t)(32 * n) * LDQ + ks * 16);
            v4i bm = bg;
            if (TWO) bm = *reinterpret_cast<const v4i *>(sQ + (size_t)I8_COLS * LDQ + (size_t)(32 * n) * LDQ + ks * 16);
#pragma unroll
            for (int m = 0; m < 2; m++) {
               acc[0][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ag[m], bg, acc[0][m][n], 0, 0, 0);
               acc[1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(am[m], bm, acc[1][m][n], 0, 0, 0);
            }
         }
      }
      if (c + 1 < c_end) FPCA_I8_STORE(buf ^ 1);
      __syncthreads();
   }
#undef FPCA_I8_LOAD
#undef FPCA_I8_STORE

   int *out = part + ((size_t)blockIdx.y * rows_pad + row0 + wr * 64) * 2 * nsc_total + col0 + wc * 64 + li;
#pragma unroll
   for (int a = 0; a < 2; a++)
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
         for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
               const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * kh;
               out[((size_t)row * 2 + a) * nsc_total + 32 * n] = acc[a][m][n][r];
            }
}

// exact combine of the int32 split-K partials and recombination of the slices:
//   K2 (mean != null): out[row][c] = ( sum_s w[s,c] (G[row][s,c] - mean[row] M[row][s,c]) ) / sd[row]   (0 if sd <= 1e-9)
//   K3               : out[row][c] = sum_s ( wg[s,c] G[row][s,c] - wm[s,c] M[row][s,c] )
__global__ __launch_bounds__(256) void k_i8_combine(const int *__restrict__ part, int nsplit, uint64_t rows_pad, int b, int S,
                                                     int NSC /* row stride of the partials: S*b rounded up to 256 */,
                                                     const double *__restrict__ wg, const double *__restrict__ wm,
                                                     const double *__restrict__ mean, const double *__restrict__ sd,
                                                     double *__restrict__ out)
{
   for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < rows_pad * b; t += (uint64_t)gridDim.x * 256) {
      const uint64_t row = t / b;
      const int c = (int)(t % b);
      double accg = 0.0, accm = 0.0;
      for (int s = S - 1; s >= 0; s--) { // small terms first
         long long g = 0, m = 0;
         for (int k = 0; k < nsplit; k++) {
            const int *p = part + (((size_t)k * rows_pad + row) * 2) * NSC + s * b + c;
            g += p[0];
            m += p[NSC];
         }
         accg += wg[s * b + c] * (double)g;
         accm += wm[s * b + c] * (double)m;
      }
      double v;
      if (mean) {
         const double sdv = sd[row];
         v = (sdv > 1e-9) ? (accg - mean[row] * accm) / sdv : 0.0;
      } else
         v = accg - accm;
      out[t] = v;
   }
}

int gemm_i8_splits(uint64_t rows_pad, uint64_t k_pad, int nsc, bool two)
{
   static const char *env = getenv("FPCA_I8_SPLITS");
   const uint64_t tiles = rows_pad / I8_ROWS * (uint64_t)(nsc / I8_COLS), chunks = k_pad / (two ? 128 : 256);
   if (env && atoi(env) > 0) return (int)std::min<uint64_t>((uint64_t)atoi(env), chunks);
   // one workgroup per CU: aim at >= 4 rounds of 256 workgroups, at least 8 chunks per workgroup
   uint64_t s = (1024 + tiles - 1) / tiles;
   if (s > chunks / 8) s = chunks / 8;
   if (s < 1) s = 1;
   if (s > 16) s = 16;
   return (int)s;
}

int gemm_i8_nsc_pad(int S, int b) { return (S * b + I8_COLS - 1) / I8_COLS * I8_COLS; }

size_t gemm_i8_workspace_ints(uint64_t rows_pad, uint64_t k_pad, int S, int b, bool two)
{
   const int nsc = gemm_i8_nsc_pad(S, b);
   return (size_t)gemm_i8_splits(rows_pad, k_pad, nsc, two) * rows_pad * 2 * (size_t)nsc;
}

void gemm_i8(const uint8_t *packed, size_t pitch, const int8_t *Qg, const int8_t *Qm, const double *wg, const double *wm,
             const double *mean, const double *sd, double *out, int *ws, uint64_t rows_pad, uint64_t k_pad, int b, int S,
             hipStream_t stream)
{
   const int nsc = gemm_i8_nsc_pad(S, b); // Q holds nsc rows; rows >= S*b are zero
   const bool two = (Qg != Qm);
   const int nsplit = gemm_i8_splits(rows_pad, k_pad, nsc, two);
   const int KC = two ? 128 : 256;
   const int chunks_total = (int)(k_pad / KC);
   const int cps = (chunks_total + nsplit - 1) / nsplit;
   static bool attr_set = false;
   if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_i8<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * I8Cfg<false>::STAGE);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_i8<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * I8Cfg<true>::STAGE);
      attr_set = true;
   }
   dim3 grid((unsigned)(rows_pad / I8_ROWS), (unsigned)nsplit, (unsigned)(nsc / I8_COLS));
   if (two)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm_i8<true>), grid, dim3(512), 2 * I8Cfg<true>::STAGE, stream, packed, pitch, Qg, Qm, k_pad, nsc, ws,
                         rows_pad, chunks_total, cps);
   else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm_i8<false>), grid, dim3(512), 2 * I8Cfg<false>::STAGE, stream, packed, pitch, Qg, Qm, k_pad, nsc, ws,
                         rows_pad, chunks_total, cps);
   HIP_CHECK_LAUNCH();
   unsigned blocks = (unsigned)std::min<uint64_t>(8192, (rows_pad * b + 255) / 256);
   hipLaunchKernelGGL(k_i8_combine, dim3(blocks), dim3(256), 0, stream, ws, nsplit, rows_pad, b, S, nsc, wg, wm, mean, sd, out);
   HIP_CHECK_LAUNCH();
}

// per-SNP row scales of the K3 operands: inv_sd = 1/sd (0 for a monomorphic SNP, sd <= 1e-9, like the lookup table of
// data.cpp:300-320), mu_inv_sd = mean/sd
__global__ void k_i8_rowscales(const double *__restrict__ mean, const double *__restrict__ sd, uint64_t P_g, uint64_t P_pad,
                               double *__restrict__ inv_sd, double *__restrict__ mu_inv_sd)
{
   const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
   if (j >= P_pad) return;
   double a = 0.0, m = 0.0;
   if (j < P_g && sd[j] > 1e-9) {
      a = 1.0 / sd[j];
      m = mean[j] / sd[j];
   }
   inv_sd[j] = a;
   mu_inv_sd[j] = m;
}

void i8_rowscales(const double *mean, const double *sd, uint64_t P_g, uint64_t P_pad, double *inv_sd, double *mu_inv_sd,
                  hipStream_t stream)
{
   hipLaunchKernelGGL(k_i8_rowscales, dim3((unsigned)((P_pad + 255) / 256)), dim3(256), 0, stream, mean, sd, P_g, P_pad, inv_sd, mu_inv_sd);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// sample-major copy of the packed stream for K3i: out[sample][snp/4] from in[snp][sample/4]  (2-bit transpose)
//   tile = 64 SNPs x 64 samples through LDS as bytes-per-code
__global__ __launch_bounds__(256) void k_transpose_packed(const uint8_t *__restrict__ in, size_t pitch_in, uint8_t *__restrict__ out,
                                                           size_t pitch_out)
{
   __shared__ uint8_t tile[64][65];
   const uint64_t snp0 = (uint64_t)blockIdx.x * 64, smp0 = (uint64_t)blockIdx.y * 64;
   // read: 64 SNP rows x 16 bytes; thread t -> row t/4, 4 bytes
   {
      const int r = threadIdx.x >> 2, seg = threadIdx.x & 3;
      const uint32_t w = *reinterpret_cast<const uint32_t *>(in + (snp0 + r) * pitch_in + smp0 / 4 + seg * 4);
#pragma unroll
      for (int k = 0; k < 16; k++) tile[r][seg * 16 + k] = (uint8_t)((w >> (2 * k)) & 3u);
   }
   __syncthreads();
   // write: 64 sample rows x 16 bytes (64 SNPs)
   {
      const int r = threadIdx.x >> 2, seg = threadIdx.x & 3;
      uint32_t w = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) w |= (uint32_t)tile[seg * 16 + k][r] << (2 * k);
      *reinterpret_cast<uint32_t *>(out + (smp0 + r) * pitch_out + snp0 / 4 + seg * 4) = w;
   }
}

void transpose_packed(const uint8_t *in, size_t pitch_in, uint64_t N_pad, uint64_t P_pad, uint8_t *out, size_t pitch_out,
                      hipStream_t stream)
{
   dim3 grid((unsigned)(P_pad / 64), (unsigned)(N_pad / 64));
   hipLaunchKernelGGL(k_transpose_packed, grid, dim3(256), 0, stream, in, pitch_in, out, pitch_out);
   HIP_CHECK_LAUNCH();
}

// diagnostic: C(32x32) = A(32x32 int8) * B(32x32 int8)^T-style product through v_mfma_i32_32x32x32_i8 with the operand
// mapping used above: lane l supplies 16 bytes of A row l&31 and of B column l&31 for k = 16 (l>>5) .. +15
__global__ void k_mfma_i8_probe(const int8_t *A /*[32][32] row-major: A[i][k]*/, const int8_t *Bt /*[32][32]: Bt[j][k]*/, int *D)
{
   const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
   v4i a = *reinterpret_cast<const v4i *>(A + li * 32 + kh * 16);
   v4i b = *reinterpret_cast<const v4i *>(Bt + li * 32 + kh * 16);
   v16i acc;
   for (int r = 0; r < 16; r++) acc[r] = 0;
   acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
   for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + li] = acc[r];
}

void mfma_i8_probe(const int8_t *A, const int8_t *Bt, int *D, hipStream_t stream)
{
   hipLaunchKernelGGL(k_mfma_i8_probe, dim3(1), dim3(64), 0, stream, A, Bt, D);
   HIP_CHECK_LAUNCH();
}

} // namespace kern
} // namespace fpca
