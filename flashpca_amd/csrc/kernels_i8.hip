// kernels_i8.hip -- exact-integer flavour of the two genotype GEMMs (FPCA_ACCUM_I8(S)).
//
// A standardised genotype is x_ij = m_ij (g_ij - mu_j) / sigma_j with g in {0,1,2} the dosage and m in {0,1} the
// "not missing" flag, so
//     T = X' B = diag(1/sigma) [ (G.M)' B  -  diag(mu) M' B ]           (K2)
//     Y = X T  = (G.M) (T / sigma)  -  M (mu T / sigma)                  (K3)
// where G.M and M are tiny integers.  The fp64 operand (B, resp. T/sigma and mu T/sigma) is rounded per column to a
// (8S-2)-bit fixed-point number sharing one power-of-two scale (m = rint(v 2^(8S-2-e)), 2^e > max|column|) and cut into
// its S two's-complement bytes (m = sum_i d_i 256^i, d_i in [-128, 127], the top one within +-65): full 8-bit digits,
// so S = 7 keeps 54 bits below the column maximum.  The products run on v_mfma_i32_32x32x32_i8 with EXACT int32
// accumulation (|sum| <= 256 K).  The only rounding of the whole product is the 2^-(8S-2) rounding of the fp64 operand
// and the fp64 recombination of the S exact slice sums (once per split-K part) -- nothing accumulates along K -- so
// S = 7 is fp64-equivalent while the int8 MFMA issues 64x the multiply-adds per cycle of the fp64 one for 14x as many.
//
// MFMA operand maps (32x32x32 i8): lane l holds 16 int8 of A row i = l&31 and of B column j = l&31 for the K-half
// l>>5; A and B use the same (half, byte) -> k assignment, so the dot products do not depend on it.  C/D register r of
// lane l is D[row = (r&3) + 8 (r>>2) + 4 (l>>5)][col = l&31].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

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

// One pass over V can serve two operands that differ only by a per-row factor (K3: T/sd and mean T/sd).
//
// Column maxima travel as the bit patterns of non-negative doubles (they order like integers, so an integer
// atomicMax is exact); the caller zeroes them -- and the column sums -- once per apply.

// Both kinds of accumulator are sharded I8_SHARDS ways by block index: same-address atomics serialise in L2 (~20 ns
// each), and with hundreds of blocks that was longer than the kernels' useful work.  Consumers fold the shards.
// maxbits: [I8_SHARDS][64], colsum: [I8_SHARDS][I8_CS_STRIDE]

// block-level max over the 256/b row slots of each column, then one atomic per column
__device__ __forceinline__ void colmax_commit(double m, int b, double *smax /* [256] */, unsigned long long *bits)
{
   __syncthreads();
   smax[threadIdx.x] = m;
   __syncthreads();
   if ((int)threadIdx.x < b) {
      for (int k = 1; k < 256 / b; k++) m = fmax(m, smax[k * b + threadIdx.x]);
      atomicMax(&bits[(blockIdx.x % I8_SHARDS) * 64 + threadIdx.x], (unsigned long long)__double_as_longlong(m));
   }
}

__device__ __forceinline__ unsigned long long maxbits_fold(const unsigned long long *bits, int c)
{
   unsigned long long m = 0;
#pragma unroll
   for (int k = 0; k < I8_SHARDS; k++) m = bits[k * 64 + c] > m ? bits[k * 64 + c] : m;
   return m;
}
__device__ __forceinline__ long long colsum_fold(const long long *cs, int t)
{
   long long a = 0;
#pragma unroll
   for (int k = 0; k < I8_SHARDS; k++) a += cs[k * I8_CS_STRIDE + t];
   return a;
}

// per-column max |v * rowscale| for one or two row scalings (standalone pass; the hot path gets K3's maxima from the
// K2 combine instead)
template <int NOPS>
__global__ __launch_bounds__(256) void k_colmax(const double *__restrict__ V, uint64_t rows, int b, const double *__restrict__ rs0,
                                                unsigned long long *__restrict__ bits0, const double *__restrict__ rs1,
                                                unsigned long long *__restrict__ bits1)
{
   __shared__ double smax[256];
   const int c = threadIdx.x % b;
   const int r_in = threadIdx.x / b, r_step = 256 / b;
   double m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0}; // 4 independent chains: 4 loads in flight per thread
   if (r_in < r_step) {
      const uint64_t stride = (uint64_t)gridDim.x * r_step;
      for (uint64_t r = (uint64_t)blockIdx.x * r_step + r_in; r < rows; r += 4 * stride) {
#pragma unroll
         for (int u = 0; u < 4; u++) {
            const uint64_t rr = r + u * stride;
            if (rr < rows) {
               const double v = V[rr * b + c];
               const double a = fabs(rs0 ? v * rs0[rr] : v);
               if (a > m0[u]) m0[u] = a; // NaN never wins: a NaN operand would poison the fp64 path as well
               if (NOPS == 2) {
                  const double a1 = fabs(v * rs1[rr]);
                  if (a1 > m1[u]) m1[u] = a1;
               }
            }
         }
      }
   }
   colmax_commit(fmax(fmax(m0[0], m0[1]), fmax(m0[2], m0[3])), b, smax, bits0);
   if (NOPS == 2) colmax_commit(fmax(fmax(m1[0], m1[1]), fmax(m1[2], m1[3])), b, smax, bits1);
}

// scale of column c: 2^e > max|column| (e from frexp; zero column -> e = 0)
__device__ __forceinline__ int slice_exponent(unsigned long long bits)
{
   const double m = __longlong_as_double((long long)bits);
   int e = 0;
   if (m > 0.0 && isfinite(m)) (void)frexp(m, &e); // m = f 2^e, f in [0.5, 1)  =>  m < 2^e
   return e;
}

// one thread = one 16-row group of one column: 16 strided fp64 in, one 16-byte store per slice and operand out; block 0
// also writes the weights colw[s*b + c] = 2^(e_c - (8S-2)) 256^(S-1-s) (slice 0 = top byte; 0 for the padding
// slice-columns); optionally the exact
// integer column sums of every slice (for  M'Q = 1'Q - E'Q, see k_gemm_i8) -- block-reduced, one atomic per
// slice-column per block
template <int NOPS>
__global__ __launch_bounds__(256) void k_slice(const double *__restrict__ V, uint64_t rows_pad, uint64_t rows, int b, int S, int nsc_pad,
                                               SliceOp o0, SliceOp o1)
{
   __shared__ int ssum[NOPS][9][256];
   const uint64_t groups = rows_pad / 16;
   const int c = threadIdx.x % b, gl = threadIdx.x / b, gpb = 256 / b; // column, local group, groups per block
   const SliceOp *ops[2] = {&o0, &o1};
   const int F = 8 * S - 2; // fractional bits below 2^e; |m| < 2^F <= 2^62
   int sh[NOPS];
#pragma unroll
   for (int o = 0; o < NOPS; o++) {
      const int e = slice_exponent(maxbits_fold(ops[o]->maxbits, c));
      sh[o] = F - e;
      if (blockIdx.x == 0) {
         if (gl == 0)
            for (int s = 0; s < S; s++) ops[o]->colw[s * b + c] = ldexp(1.0, e - F + 8 * (S - 1 - s));
         for (int t = S * b + threadIdx.x; t < nsc_pad; t += 256) ops[o]->colw[t] = 0.0;
      }
   }
   for (uint64_t g0 = (uint64_t)blockIdx.x * gpb; g0 < groups; g0 += (uint64_t)gridDim.x * gpb) {
      const uint64_t g = g0 + gl;
      const bool act = gl < gpb && g < groups;
      const uint64_t r0 = g * 16;
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
         const uint64_t r = r0 + j;
         v[j] = (act && r < rows) ? V[r * b + c] : 0.0;
      }
#pragma unroll
      for (int o = 0; o < NOPS; o++) {
         long long m[16];
#pragma unroll
         for (int j = 0; j < 16; j++) {
            const uint64_t r = r0 + j;
            double x = v[j];
            if (ops[o]->rowscale && act && r < rows) x *= ops[o]->rowscale[r];
            m[j] = __double2ll_rn(ldexp(x, sh[o])); // power-of-two scaling is exact; |m| < 2^F
            if (ops[o]->copy64 && act && r < rows) ops[o]->copy64[r * b + c] = x;
            if (ops[o]->copy32 && act && r < rows) ops[o]->copy32[r * b + c] = (float)ldexp(x, sh[o] - F + 1); // |.| < 2: any column fits fp32
         }
         for (int s = S - 1; s >= 0; s--) { // bytes from the low end, carries into the next one
            u4 word = {0u, 0u, 0u, 0u};
            int tot = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
               const int d = (int)(signed char)(m[j] & 0xFF);
               m[j] = (m[j] - d) >> 8;
               tot += d;
               word[j & 3] |= ((uint32_t)d & 0xFFu) << (8 * (j >> 2)); // position 4 (j&3) + (j>>2)
            }
            if (act) *reinterpret_cast<u4 *>(ops[o]->Q + ((uint64_t)(s * b + c)) * rows_pad + r0) = word;
            ssum[o][s][threadIdx.x] = tot;
         }
      }
      if (o0.colsum || (NOPS == 2 && o1.colsum)) {
         __syncthreads();
#pragma unroll
         for (int o = 0; o < NOPS; o++)
            if (ops[o]->colsum)
               for (int t = threadIdx.x; t < S * b; t += 256) {
                  const int s = t / b, cc = t % b;
                  long long a = 0;
                  for (int k = 0; k < gpb; k++) a += ssum[o][s][k * b + cc];
                  if (a)
                     atomicAdd(reinterpret_cast<unsigned long long *>(&ops[o]->colsum[(blockIdx.x % I8_SHARDS) * I8_CS_STRIDE + t]),
                               (unsigned long long)a);
               }
         __syncthreads();
      }
   }
}

int gemm_i8_nsc_pad(int S, int b);

void i8_colmax(const double *V, uint64_t rows, int b, int nops, const SliceOp *ops, hipStream_t stream)
{
   const unsigned blocks = (unsigned)std::min<uint64_t>(1024, rows * b / 1024 + 1);
   if (nops == 2)
      hipLaunchKernelGGL(k_colmax<2>, dim3(blocks), dim3(256), 0, stream, V, rows, b, ops[0].rowscale, ops[0].maxbits, ops[1].rowscale,
                         ops[1].maxbits);
   else
      hipLaunchKernelGGL(k_colmax<1>, dim3(blocks), dim3(256), 0, stream, V, rows, b, ops[0].rowscale, ops[0].maxbits, nullptr, nullptr);
   HIP_CHECK_LAUNCH();
}

void i8_slice(const double *V, uint64_t rows_pad, uint64_t rows, int b, int S, int nops, const SliceOp *ops, hipStream_t stream)
{
   if (S > 8 || b > 64 || nops < 1 || nops > 2) throw Error(-1, "i8_slice: S <= 8, b <= 64, 1 or 2 operands");
   const unsigned blocks = (unsigned)std::min<uint64_t>(4096, (rows_pad / 16 + (256 / b) - 1) / (256 / b));
   const int nsc = gemm_i8_nsc_pad(S, b);
   if (nops == 2)
      hipLaunchKernelGGL(k_slice<2>, dim3(blocks), dim3(256), 0, stream, V, rows_pad, rows, b, S, nsc, ops[0], ops[1]);
   else
      hipLaunchKernelGGL(k_slice<1>, dim3(blocks), dim3(256), 0, stream, V, rows_pad, rows, b, S, nsc, ops[0], ops[0]);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Row-sharded operand exchange (round 6; operator.hip apply_sharded): every rank slices ITS rows of the block and the ranks
// all-gather the SLICES -- S bytes per entry instead of the 8 of the fp64 block -- in a ROW-MAJOR layout Qrm[row][s*b + c]
// (a rank's rows of a chunk are one contiguous piece, so the gathered buffer is simply [all rows][S*b] in global row order).
//   k_maxbits_fold / k_maxbits_set   the column maxima: shards folded for the exchange / the maximum over all ranks' answers put back
//   k_slice_rows                     V[rows][b] fp64 -> Qrm[rows][S*b] with the GLOBAL column scales; thread = (row, 16 columns)
//   k_unpack_slices                  Qrm[rows_pad][S*b] -> Q[s*b + c][rows_pad] in the 16-row-group byte order of k_slice + the exact
//                                    column sums of every slice; thread = (16 rows, slice, 16 columns)
//   k_dequant_rows                   Qrm -> the scaled operand itself for the sparse gathers of the missing-call route: fp32 rows
//                                    m 2^(1-F) (<= 4 slices) or fp64 rows m 2^(e-F); m = the (8S-2)-bit integer the slices spell --
//                                    exactly the operand the GEMM multiplies
__global__ void k_maxbits_fold(const unsigned long long *__restrict__ bits, double *__restrict__ out)
{
   if (threadIdx.x < 64) out[threadIdx.x] = __longlong_as_double((long long)maxbits_fold(bits, threadIdx.x));
}
__global__ void k_maxbits_set(const double *__restrict__ all, int G, unsigned long long *__restrict__ bits)
{
   if (threadIdx.x < 64) {
      double m = 0.0;
      for (int g = 0; g < G; g++) m = fmax(m, all[g * 64 + threadIdx.x]);
      bits[threadIdx.x] = (unsigned long long)__double_as_longlong(m);
      for (int k = 1; k < I8_SHARDS; k++) bits[k * 64 + threadIdx.x] = 0ull;
   }
}

__global__ __launch_bounds__(256) void k_slice_rows(const double *__restrict__ V, uint64_t rows, int b, int S, int nsc_pad,
                                                    const unsigned long long *__restrict__ maxbits, int8_t *__restrict__ Qrm,
                                                    double *__restrict__ colw)
{
   __shared__ int sh_e[64];
   const int F = 8 * S - 2;
   if ((int)threadIdx.x < b) {
      const int e = slice_exponent(maxbits_fold(maxbits, threadIdx.x));
      sh_e[threadIdx.x] = e;
      if (blockIdx.x == 0)
         for (int s = 0; s < S; s++) colw[s * b + threadIdx.x] = ldexp(1.0, e - F + 8 * (S - 1 - s));
   }
   if (blockIdx.x == 0)
      for (int t = S * b + threadIdx.x; t < nsc_pad; t += 256) colw[t] = 0.0;
   __syncthreads();
   const int gpr = b / 16; // 16-column groups per row
   const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, row = t / gpr;
   const int g = (int)(t % gpr);
   if (row >= rows) return;
   long long m[16];
#pragma unroll
   for (int j = 0; j < 16; j++) m[j] = __double2ll_rn(ldexp(V[row * b + 16 * g + j], F - sh_e[16 * g + j]));
   for (int s = S - 1; s >= 0; s--) {
      u4 word = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 16; j++) {
         const int d = (int)(signed char)(m[j] & 0xFF);
         m[j] = (m[j] - d) >> 8;
         word[j >> 2] |= ((uint32_t)d & 0xFFu) << (8 * (j & 3)); // byte j of the piece = column 16 g + j
      }
      *reinterpret_cast<u4 *>(Qrm + row * (uint64_t)(S * b) + s * b + 16 * g) = word;
   }
}

// One workgroup = TRG 16-row groups x all S b columns: the rows of a tile are one contiguous span of the row-major buffer, loaded
// with coalesced 16-byte reads into LDS (row stride S b + 16, 16 more per row group: the 16 lanes of a ds_read_b128 group then hit
// distinct banks); thread (row group fastest, slice, 16-column group) transposes a 16 x 16 byte block in registers and writes one
// 16-byte piece per column -- adjacent threads = adjacent row groups = contiguous bytes of the same operand row.
// The same pass can leave the scaled operand itself for the sparse gathers of the missing-call route (copy32: m 2^(1-F) in fp32, <= 4
// slices; copy64: m 2^(e_c - F)), rows < rows_valid: lane = (row, 4 resp. 2 columns) reads one dword / one halfword per slice from the
// LDS tile (conflict-free at the odd row stride) and writes 16 contiguous bytes -- a wave stores 1 KB runs.
__global__ __launch_bounds__(256) void k_unpack_slices(const int8_t *__restrict__ Qrm, uint64_t rows_pad, int b, int S, int TRG,
                                                        int8_t *__restrict__ Q, long long *__restrict__ colsum,
                                                        const unsigned long long *__restrict__ maxbits, uint64_t rows_valid,
                                                        float *__restrict__ copy32, double *__restrict__ copy64)
{
   __shared__ double sh_scale[64];
   extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
   __shared__ int ssum[512];
   const int SB = S * b, rstr = SB + 16, gstr = 16 * rstr + 16, ncg = SB / 16; // bytes per row / LDS strides / 16-column groups
   const uint64_t rg0 = (uint64_t)blockIdx.x * TRG, groups = rows_pad / 16;
   const int ng = (int)(groups - rg0 < (uint64_t)TRG ? groups - rg0 : TRG);
   const u4 *src = reinterpret_cast<const u4 *>(Qrm + rg0 * 16 * (uint64_t)SB);
   for (int p = threadIdx.x; p < ng * 16 * ncg; p += 256) { // piece p: row p / ncg of the tile, 16-byte piece p % ncg
      const int row = p / ncg, pc = p % ncg;
      *reinterpret_cast<u4 *>(tile + (row >> 4) * gstr + (row & 15) * rstr + pc * 16) = src[p];
   }
   for (int t = threadIdx.x; t < 512; t += 256) ssum[t] = 0;
   if ((copy32 || copy64) && (int)threadIdx.x < b) {
      const int F = 8 * S - 2;
      sh_scale[threadIdx.x] = copy64 ? ldexp(1.0, slice_exponent(maxbits_fold(maxbits, threadIdx.x)) - F) : ldexp(1.0, 1 - F);
   }
   __syncthreads();
   if (copy32) { // thread = (row of the tile, group of 4 columns)
      const int per_row = b / 4;
      for (int t = threadIdx.x; t < ng * 16 * per_row; t += 256) {
         const int row = t / per_row, q = t % per_row;
         const uint64_t grow = rg0 * 16 + row;
         if (grow >= rows_valid) continue;
         const unsigned char *src_row = tile + (row >> 4) * gstr + (row & 15) * rstr + 4 * q;
         long long m[4] = {0, 0, 0, 0};
         for (int sl = 0; sl < S; sl++) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(src_row + sl * b);
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = m[j] * 256 + (long long)(signed char)((w >> (8 * j)) & 0xFFu);
         }
         float4 v;
         v.x = (float)((double)m[0] * sh_scale[4 * q]);
         v.y = (float)((double)m[1] * sh_scale[4 * q + 1]);
         v.z = (float)((double)m[2] * sh_scale[4 * q + 2]);
         v.w = (float)((double)m[3] * sh_scale[4 * q + 3]);
         *reinterpret_cast<float4 *>(copy32 + grow * b + 4 * q) = v;
      }
   }
   if (copy64) { // thread = (row of the tile, pair of columns)
      const int per_row = b / 2;
      for (int t = threadIdx.x; t < ng * 16 * per_row; t += 256) {
         const int row = t / per_row, q = t % per_row;
         const uint64_t grow = rg0 * 16 + row;
         if (grow >= rows_valid) continue;
         const unsigned char *src_row = tile + (row >> 4) * gstr + (row & 15) * rstr + 2 * q;
         long long m0 = 0, m1 = 0;
         for (int sl = 0; sl < S; sl++) {
            const uint32_t w = *reinterpret_cast<const unsigned short *>(src_row + sl * b);
            m0 = m0 * 256 + (long long)(signed char)(w & 0xFFu);
            m1 = m1 * 256 + (long long)(signed char)((w >> 8) & 0xFFu);
         }
         *reinterpret_cast<d2 *>(copy64 + grow * b + 2 * q) = (d2){(double)m0 * sh_scale[2 * q], (double)m1 * sh_scale[2 * q + 1]};
      }
   }
   for (int t = threadIdx.x; t < TRG * ncg; t += 256) {
      const int rg = t % TRG, cg = t / TRG; // column group cg = s * (b / 16) + g covers slice-columns 16 cg .. 16 cg + 15
      if (rg >= ng) continue;
      u4 piece[16];
#pragma unroll
      for (int i = 0; i < 16; i++) piece[i] = *reinterpret_cast<const u4 *>(tile + rg * gstr + i * rstr + cg * 16);
      int tot[16];
#pragma unroll
      for (int j = 0; j < 16; j++) { // column 16 cg + j: byte j of every piece, row i -> word i % 4, byte i / 4 (k_slice's order)
         u4 word = {0u, 0u, 0u, 0u};
         tot[j] = 0;
#pragma unroll
         for (int i = 0; i < 16; i++) {
            const uint32_t byte = (piece[i][j >> 2] >> (8 * (j & 3))) & 0xFFu;
            word[i & 3] |= byte << (8 * (i >> 2));
            tot[j] += (int)(signed char)byte;
         }
         *reinterpret_cast<u4 *>(Q + (uint64_t)(16 * cg + j) * rows_pad + (rg0 + rg) * 16) = word;
      }
      if (colsum) {
#pragma unroll
         for (int j = 0; j < 16; j++) atomicAdd(&ssum[16 * cg + j], tot[j]); // (LDS; |sum| <= 128 x 16 x TRG)
      }
   }
   if (colsum) {
      __syncthreads();
      for (int t = threadIdx.x; t < SB; t += 256)
         if (ssum[t])
            atomicAdd(reinterpret_cast<unsigned long long *>(&colsum[(blockIdx.x % I8_SHARDS) * I8_CS_STRIDE + t]), (unsigned long long)(long long)ssum[t]);
   }
}

__global__ __launch_bounds__(256) void k_dequant_rows(const int8_t *__restrict__ Qrm, uint64_t rows, int b, int S,
                                                       const unsigned long long *__restrict__ maxbits, float *__restrict__ copy32,
                                                       double *__restrict__ copy64)
{
   __shared__ double sh_scale[64];
   const int F = 8 * S - 2, gpr = b / 16;
   if ((int)threadIdx.x < b) sh_scale[threadIdx.x] = copy64 ? ldexp(1.0, slice_exponent(maxbits_fold(maxbits, threadIdx.x)) - F) : ldexp(1.0, 1 - F);
   __syncthreads();
   const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x, row = t / gpr;
   const int g = (int)(t % gpr);
   if (row >= rows) return;
   long long m[16];
#pragma unroll
   for (int j = 0; j < 16; j++) m[j] = 0;
   for (int s = 0; s < S; s++) { // slice 0 = top byte
      const u4 word = *reinterpret_cast<const u4 *>(Qrm + row * (uint64_t)(S * b) + s * b + 16 * g);
#pragma unroll
      for (int j = 0; j < 16; j++) m[j] = m[j] * 256 + (long long)(signed char)((word[j >> 2] >> (8 * (j & 3))) & 0xFFu);
   }
   // power-of-two scales: exact.  copy32: m 2^(1-F), |.| < 2, like k_slice's copy32; copy64: m 2^(e_c - F)
   if (copy32) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
         float4 v;
         v.x = (float)((double)m[4 * q] * sh_scale[16 * g + 4 * q]);
         v.y = (float)((double)m[4 * q + 1] * sh_scale[16 * g + 4 * q + 1]);
         v.z = (float)((double)m[4 * q + 2] * sh_scale[16 * g + 4 * q + 2]);
         v.w = (float)((double)m[4 * q + 3] * sh_scale[16 * g + 4 * q + 3]);
         *reinterpret_cast<float4 *>(copy32 + row * b + 16 * g + 4 * q) = v;
      }
   }
   if (copy64) {
#pragma unroll
      for (int q = 0; q < 8; q++)
         *reinterpret_cast<d2 *>(copy64 + row * b + 16 * g + 2 * q) =
            (d2){(double)m[2 * q] * sh_scale[16 * g + 2 * q], (double)m[2 * q + 1] * sh_scale[16 * g + 2 * q + 1]};
   }
}

void i8_maxbits_fold(const unsigned long long *bits, double *out64, hipStream_t stream)
{
   hipLaunchKernelGGL(k_maxbits_fold, dim3(1), dim3(64), 0, stream, bits, out64);
   HIP_CHECK_LAUNCH();
}
void i8_maxbits_set(const double *all, int G, unsigned long long *bits, hipStream_t stream)
{
   hipLaunchKernelGGL(k_maxbits_set, dim3(1), dim3(64), 0, stream, all, G, bits);
   HIP_CHECK_LAUNCH();
}
void i8_slice_rows(const double *V, uint64_t rows, int b, int S, const SliceOp &op, int8_t *Qrm, hipStream_t stream)
{
   if (S > 8 || b > 64 || b % 16) throw Error(-1, "i8_slice_rows: S <= 8, b in {16, 32, 48, 64}");
   if (!rows) return;
   const uint64_t threads = rows * (b / 16);
   hipLaunchKernelGGL(k_slice_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, V, rows, b, S, gemm_i8_nsc_pad(S, b), op.maxbits, Qrm, op.colw);
   HIP_CHECK_LAUNCH();
}
void i8_unpack_slices(const int8_t *Qrm, uint64_t rows_pad, int b, int S, const SliceOp &op, hipStream_t stream, uint64_t rows_valid, float *copy32,
                      double *copy64)
{
   const int SB = S * b;
   if (SB > 512 || SB % 16) throw Error(-1, "i8_unpack_slices: at most 512 slice-columns");
   const int trg = SB <= 128 ? 32 : SB <= 272 ? 16 : 8; // 16-row groups per workgroup: 41 / 74 / 68 KB of LDS at the largest S b of each class
   const uint64_t groups = rows_pad / 16;
   const size_t lds = (size_t)trg * (16 * (SB + 16) + 16);
   static bool attr_set = false;
   if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_unpack_slices), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_set = true;
   }
   hipLaunchKernelGGL(k_unpack_slices, dim3((unsigned)((groups + trg - 1) / trg)), dim3(256), lds, stream, Qrm, rows_pad, b, S, trg, op.Q, op.colsum, op.maxbits,
                      rows_valid, copy32, copy64);
   HIP_CHECK_LAUNCH();
}
void i8_dequant_rows(const int8_t *Qrm, uint64_t rows, int b, int S, const SliceOp &op, float *copy32, double *copy64, hipStream_t stream)
{
   if (!rows) return;
   const uint64_t threads = rows * (b / 16);
   hipLaunchKernelGGL(k_dequant_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, Qrm, rows, b, S, op.maxbits, copy32, copy64);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// K2i / K3i core:  acc[row][sc] = sum_k A[row][k] * Q[sc][k]  for A in {G.M, M}
//   `packed`: 2-bit records, one per output row (K2: the SNP-major stream, K = samples; K3: its sample-major copy,
//   K = SNPs).  TWO = false (K2): both integer matrices multiply the same operand Q.  TWO = true (K3): G.M multiplies
//   Qg (slices of T/sd), M multiplies Qm (slices of mean T/sd).
//   Workgroup = 4 waves, ONE wave per SIMD with the whole 512-register budget (up to 256 accumulator AGPRs), arranged
//   WR x WC; a wave owns MT x NT tiles of 32x32 of both matrices (2 MT NT <= 16 accumulators).  Operand tiles
//   [WC NT 32 columns][KC k] are double-buffered in LDS (row stride KC + 16 B: the 16 lanes of a ds_read_b128 group hit
//   distinct banks); the packed words go straight from global memory to the registers of the lane that decodes them.
//   Decode: a lane's dword w holds 16 codes; (w >> 2q) & 0x03030303 leaves codes q, q+4, q+8, q+12 in the four bytes
//   and v_perm_b32 with the code as selector looks G.M / M up in a 4-byte table -- 4 VALU ops per operand dword
//   pair; the Q bytes were stored in the matching order by k_slice.
//   Scheduling: with one wave per SIMD only the wave's own instruction order hides latency, and hipcc left alone
//   sinks every LDS read to just before its MFMAs and waits at once (measured: matrix pipe 42 % busy).  The loop body is
//   therefore a fixed pipeline of micro-steps (k-step, m-tile, n-group) fenced by sched_barrier(0), with explicit (asm)
//   loads and hand-counted s_waitcnt: the operand fragments of the next micro-step are read from LDS and its genotype
//   fragments decoded under this one's MFMAs, and the next chunk's global loads (first half of the chunk) and LDS
//   stores (second half) ride in the MFMA shadow instead of a burst at the chunk boundary.
//   Output: fp64 partials part[split * zblocks][row][mat][bw] (slices already recombined), summed by k_i8_combine.
template <int OFF>
__device__ __forceinline__ v4i lds_read16(uint32_t addr) // explicit LDS read: the caller places the s_waitcnt
{
   v4i r;
   asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
   return r;
}
template <int OFF>
__device__ __forceinline__ u4 gload16(const void *sbase, uint32_t voff) // explicit global load, address = sbase + voff + OFF
{
   u4 r;
   asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF));
   return r;
}
// hand-placed waits; the "+v" operands make the consumers of the loaded registers depend on the wait
__device__ __forceinline__ void lds_wait(v4i &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_wait(v4i &a, v4i &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
template <int N>
__device__ __forceinline__ void vm_wait(u4 &a)
{
   asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
   (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
   static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// MODE: how the missing-indicator matrix E is treated, chosen per shard from the genotype-code counts of K1
//   I8_FULL       both matrices, every block                                   (typical data, missing rate >~ 0.03 %)
//   I8_SKIP_EMPTY both matrices, but a 32 x 32 block of E without a missing genotype is skipped (wave-uniform branch): up
//                 to 23 % less time when nearly nothing is missing, ~10 % MORE when nothing can be skipped -- measured
//   I8_NO_MISSING the shard has no missing genotype at all (imputed / 1000-Genomes-like data): E = 0, only G.M is
//                 multiplied, M'Q = 1'Q comes from the column sums alone -- half the MFMAs
enum { I8_FULL = 0, I8_SKIP_EMPTY = 1, I8_NO_MISSING = 2 };

// ABL (builds with -DFPCA_I8_ABLATION only; results are wrong by construction): bit 0 drops the operand staging of the
// main loop, bit 1 the genotype decode, bit 2 the LDS fragment reads, bit 3 the packed-word loads -- what each costs
// (scripts/i8_ablation.py); bits 4 / 5 keep every instruction but make the operand stream / the packed words L2-resident
// (4 chunks re-read): what their misses cost.  Round 2: operand stream 8.68 -> 8.71 ms (nothing), packed words 8.68 ->
// 8.22 ms -- and that is their HBM energy, not their latency: a variant that issued the packed words 1.5-2.5 chunks ahead
// instead of half a chunk (chunks in straight-line pairs, four register sets) ran 9.20 ms against 8.70 on the same box.
// Bit 6 drops the per-chunk barrier: 8.77 -> 8.82 ms, i.e. the barrier and the pipeline refill behind it cost nothing.
// With an all-zero fp64 operand (same instructions, same traffic; scripts/i8_power_probe.py) the same launch takes 7.07 ms
// instead of 9.04: the kernel follows the bare MFMA stream's power curve (3470 -> 4540 TOP/s) -- it is energy, not time.
// HALF: the last of the NT column tiles has only 16 columns (S b = 112 for b = 16, S = 7: 3.5 tiles).  As a 32-wide tile half
// of it would multiply zero padding -- an eighth of the kernel's MFMA passes; instead the 16 columns run on
// v_mfma_i32_16x16x64_i8, which takes TWO 32-k steps at once: the genotype fragments of an even and an odd k-step (lanes =
// 32 rows x 2 k-halves each) are regrouped by v_permlane16_swap into two 16-row x 64-k operands, the slice-column operand is
// one ds_read_b128 per lane with the matching (k-half, k-step) -> quarter order.  A bare stream of the two mixes
// (scripts/mfma_mix_probe.py): 3280 -> 3720 useful TOP/s.
template <bool TWO_, int MT_, int NT_, int WR_, int WC_, int KC_, int G_, int MODE_ = I8_FULL, int ABL_ = 0, bool HALF_ = false>
struct I8Cfg {
   static constexpr bool TWO = TWO_, HALF = HALF_;
   static constexpr int MT = MT_, NT = NT_, WR = WR_, WC = WC_, KC = KC_, G = G_, NQ = TWO ? 2 : 1, MODE = MODE_;
   static constexpr int ABL = ABL_;
   static constexpr int MATS = MODE == I8_NO_MISSING ? 1 : 2;
   static constexpr int NTF = HALF ? NT - 1 : NT;       // full 32-column tiles
   static_assert(!(TWO && MATS == 1), "without E there is only one operand");
   static_assert(WR * WC == 4 && MATS * MT * NT <= 16 && (MT == 1 || G == 1) && G <= NT, "shape");
   static_assert(!HALF || (WC == 1 && MT == 2 && G == 1 && MODE != I8_SKIP_EMPTY && ABL == 0 && (KC / 32) % 2 == 0), "half-tile variant");
   static constexpr int ROWS = WR * MT * 32;            // workgroup rows
   static constexpr int COLS = WC * NT * 32 - (HALF ? 16 : 0); // workgroup columns of each operand
   static constexpr int LDQ = KC + 16;                  // operand tile row stride (bytes)
   // HALF: the 16 rows of the remainder tile are read by ds_read_b128 with lanes l and l + 16 one 16-byte slot apart, which at
   // the odd slot pitch of the full tiles puts two lanes of a 16-lane service group on one slot (LDS bank conflicts 8.7 % of the
   // kernel's LDS cycles, round 4); those rows get an even pitch of 2 slots (mod 16) instead: slot = 2 j + (l >> 4 & 1)
   static constexpr int LDQH = LDQ + (HALF ? 16 : 0);   // row stride of the remainder tile's rows
   static constexpr int QTILE = COLS * LDQ + (HALF ? 16 * 16 : 0); // bytes of one operand tile
   static constexpr int STAGE = NQ * QTILE;
   static constexpr int SEGS = KC / 16, RSTEP = 256 / SEGS; // 16-byte segments per row; rows covered by 256 threads
   static constexpr int NP1 = COLS / RSTEP;             // pieces per thread per operand tile
   static constexpr int NP = NQ * NP1;                  // ... per chunk
   static constexpr int KS = KC / 32;                   // 32-k steps per chunk
   static constexpr int PW = KC / 128;                  // 16-byte packed pieces per lane per m-tile (lane half = KC/2 k)
   static constexpr int NSTEP = KS * MT * G;            // micro-steps per chunk
   static constexpr int H = NSTEP / 2;
   // dynamic LDS: the double-buffered operand tiles; the epilogue reuses it for 4 waves x NT dumped tiles + the weights
   static constexpr int LDS_NEED = (2 * STAGE > 4 * NT * 4096 + 2 * WC * NT * 32 * 8) ? 2 * STAGE : 4 * NT * 4096 + 2 * WC * NT * 32 * 8;
   // The narrowest column block (2 tiles: 136 registers, 34 KB) would fit three workgroups per CU; measured at 500,000 x
   // 100,000 (scripts/r4_narrow_probe.sh, GEMM kernels K2 / K3): three per CU 4.61 / 4.74 ms, two 4.31 / 4.39, one 5.20 / 5.74 --
   // so it asks for enough LDS to be two (the wider blocks are two or one by their registers: 224+ of 512).
   static constexpr int LDS_BYTES = (!TWO && NT <= 2 && LDS_NEED < 56 * 1024) ? 56 * 1024 : LDS_NEED;
   static_assert(COLS % RSTEP == 0 && NSTEP % 2 == 0 && LDS_BYTES <= 160 * 1024, "staging");
   static_assert(!HALF || RSTEP == 16, "the remainder tile is the last staging piece");
};

// returns (MODE == I8_SKIP_EMPTY) whether any lane of the wave holds a missing genotype in this 32-row x 32-k block
// (wave-uniform), else true
template <int MODE>
__device__ __forceinline__ bool i8_decode(uint32_t w, v4i &ag, v4i &am, uint32_t tab1)
{
   // byte[code] of the two integer matrices: G.M (dosage, 0 if missing) and E = 1 - M (missing indicator): code 0 -> (2,0),
   // 1 (missing) -> (0,1), 2 -> (1,0), 3 -> (0,0).  E instead of M because E is almost all zeros: the products vanish and
   // the power-limited matrix pipe clocks higher; M'Q is recovered exactly in the combine as 1'Q - E'Q.
   // (the one-matrix kernel takes its table as an argument: G.M for the genotype products, E for the missing-indicator
   //  products of the SNPs whose missing calls are too many for the sparse route -- gemm_i8 `e_only`)
   const uint32_t tabG = MODE == I8_NO_MISSING ? tab1 : 0x00010002u, tabM = 0x00000100u;
#pragma unroll
   for (int q = 0; q < 4; q++) {
      const uint32_t sel = (w >> (2 * q)) & 0x03030303u;
      ag[q] = (int)__builtin_amdgcn_perm(0u, tabG, sel);
      if (MODE != I8_NO_MISSING) am[q] = (int)__builtin_amdgcn_perm(0u, tabM, sel);
   }
   if (MODE == I8_SKIP_EMPTY)
      return __builtin_amdgcn_ballot_w64(((w & ~(w >> 1)) & 0x55555555u) != 0u) != 0ull; // code 01: low bit set, high bit clear
   return true;
}

template <class C>
__global__ __launch_bounds__(256, 1) void k_gemm_i8(const uint8_t *__restrict__ packed, size_t pitch,
                                                     const int8_t *__restrict__ Qg, const int8_t *__restrict__ Qm,
                                                     uint64_t k_pad, const double *__restrict__ wg, const double *__restrict__ wm, int bw,
                                                     double *__restrict__ part, uint64_t rows_pad, int chunks_total, int zb,
                                                     int nA, int sB, int cpsB, uint64_t rowB0, uint64_t rowsB, uint32_t tab1)
{
   constexpr bool TWO = C::TWO;
   constexpr int MODE = C::MODE, MATS = C::MATS;
   constexpr int MT = C::MT, NT = C::NT, NQ = C::NQ, KC = C::KC, NP = C::NP, NP1 = C::NP1, NSTEP = C::NSTEP, LDQ = C::LDQ,
                 H = C::H, G = C::G, PW = C::PW, NPK = MT * PW, NTF = C::NTF;
   constexpr bool HALF = C::HALF;
   constexpr int AMASK = HALF ? 3 : 1; // decoded genotype fragments kept alive: 2, or 4 (even AND odd k-step of both m-tiles)
   extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int li = lane & 31, kh = lane >> 5;
   const int wr = wave / C::WC, wc = wave % C::WC;
   // XCD-aware 1-D grid (workgroup w runs on XCD w % 8, each XCD has its own L2): the zb column blocks of a row tile
   // re-read the same packed rows, so they are given to the SAME XCD back to back (w and w + 8); the row tiles of one
   // split are dealt round-robin over the XCDs; splits are outermost, so that a round of workgroups streams one K range
   // of the operand in lock-step (nB is a multiple of 8, so the XCD of a phase-B workgroup is still w0 % 8).  (With the plain 3-D grid the column blocks ran a whole grid apart and K3 read the
   // packed stream from HBM twice: 29.9 GB per launch at cfg3 for 12.7 GB algorithmic.)
   //
   // Two-phase split-K.  Tile ids w0 (in the XCD-aware order) below nA -- whole rounds of one workgroup per CU -- run
   // UNSPLIT (phase A: no partial planes beyond plane 0, the whole round streams the operand in lock-step); the
   // remaining nB < #CU.. tiles, which would otherwise be a mostly idle last round, are cut sB ways along K (phase B,
   // workgroup ids nA + split * nB + k).  nA = 0 is plain split-K.
   const int rtiles = (int)(rows_pad / C::ROWS), rtl = (rtiles + 7) / 8, ids = 8 * rtl * zb, nB = ids - nA;
   int w0 = blockIdx.x, split = 0, c_begin = 0, c_end = chunks_total;
   if (w0 >= nA) {
      const int wp = w0 - nA;
      split = wp / nB;
      w0 = nA + wp % nB;
      c_begin = split * cpsB;
      c_end = c_begin + cpsB < chunks_total ? c_begin + cpsB : chunks_total;
   }
   const int xcd = w0 & 7, sidx = w0 >> 3;
   const int zblk = sidx % zb, rt = (sidx / zb) * 8 + xcd;
   if (rt >= rtiles || c_begin >= c_end) return;
   const uint64_t row0 = (uint64_t)rt * C::ROWS;
   const int col0 = zblk * C::COLS;

   v16i acc[MATS][MT][NT]; // [mat][m][n]   (HALF: the last n is never touched and costs nothing)
   v4i acch[MATS][MT][2];  // HALF: the 16-column remainder, rows 16 t .. 16 t + 15 of m-tile m in acch[mat][m][t]
#pragma unroll
   for (int a = 0; a < MATS; a++)
#pragma unroll
      for (int m = 0; m < MT; m++) acch[a][m][0] = acch[a][m][1] = (v4i){0, 0, 0, 0};
#pragma unroll
   for (int a = 0; a < MATS; a++)
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
         for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][m][n][r] = 0;

   // operand staging: piece r of this thread = operand r / NP1, row tid / SEGS + RSTEP (r % NP1), segment tid % SEGS
   const uint32_t qvoff = (uint32_t)((tid / C::SEGS) * k_pad + (tid % C::SEGS) * 16);
   const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
   const int qdst = (tid / C::SEGS) * LDQ + (tid % C::SEGS) * 16;
   // packed words: lane (li, kh) of wave row wr decodes rows wr*32*MT + 32 m + li, k = (KC/2) kh + 16 ks .. of the chunk
   uint32_t pvoff[MT];
#pragma unroll
   for (int m = 0; m < MT; m++) pvoff[m] = (uint32_t)((wr * 32 * MT + 32 * m + li) * pitch + kh * (KC / 8));
   const uint8_t *prow = packed + row0 * pitch;
   const uint32_t aQ0 = lds_base + (uint32_t)(wc * 32 * NT + li) * LDQ + kh * (KC / 2);
   // HALF: lane (column j = lane & 15, quarter qd = lane >> 4) of the 16x16x64 operand reads k-half qd >> 1 of k-step 2 kp + (qd & 1)
   const uint32_t aQh0 = lds_base + (uint32_t)(32 * NTF * LDQ + (lane & 15) * C::LDQH) + ((lane >> 5) & 1) * (KC / 2) + ((lane >> 4) & 1) * 16;

   u4 qreg[NP];
   u4 pk[MT][PW], pkn[MT][PW]; // packed words of the current / next chunk (one dword per k-step)

   auto issue_q = [&](auto rr, int cc) {
      constexpr int r = decltype(rr)::value, o = r / NP1, r1 = r % NP1;
      const int ccq = (C::ABL & 16) ? (cc & 3) : cc; // ablation: the operand stream stays L2-resident (4 chunks, re-read)
      const int8_t *sb = ((TWO && o) ? Qm : Qg) + (uint64_t)(col0 + C::RSTEP * r1) * k_pad + (uint64_t)ccq * KC;
      qreg[r] = gload16<0>(sb, qvoff);
   };
   auto issue_p = [&](u4(&dst)[MT][PW], int cc) {
      const uint8_t *pb = prow + (size_t)((C::ABL & 32) ? (cc & 3) : cc) * (KC / 4); // ablation: packed words L2-resident
#pragma unroll
      for (int m = 0; m < MT; m++) {
         dst[m][0] = gload16<0>(pb, pvoff[m]);
         if constexpr (PW == 2) dst[m][1] = gload16<16>(pb, pvoff[m]);
      }
   };
   auto store_q = [&](auto rr, unsigned char *st) {
      constexpr int r = decltype(rr)::value, o = r / NP1, r1 = r % NP1;
      vm_wait<NP - 1 - r + NPK>(qreg[r]); // loads are issued in the order q[0..NP), p[..]
      // (HALF: the last piece is the remainder tile's 16 rows, at their own pitch)
      const int hoff = (HALF && r1 == NP1 - 1) ? (tid / C::SEGS) * (C::LDQH - LDQ) : 0;
      *reinterpret_cast<u4 *>(st + o * C::QTILE + r1 * C::RSTEP * LDQ + hoff) = qreg[r];
   };

   // prologue: chunk c_begin
   static_for<NP>([&](auto rr) { issue_q(rr, c_begin); });
   issue_p(pk, c_begin);
   static_for<NP>([&](auto rr) { store_q(rr, smem + qdst); });
#pragma unroll
   for (int m = 0; m < MT; m++)
#pragma unroll
      for (int h = 0; h < PW; h++) vm_wait<0>(pk[m][h]);
   __syncthreads();

   for (int c = c_begin; c < c_end; c++) {
      const int buf = (c - c_begin) & 1;
      const int cn = (c + 1 < c_end) ? c + 1 : c; // the last chunk re-stages itself (branch-free pipeline)
      const uint32_t aQ = aQ0 + (uint32_t)buf * C::STAGE, aQ2 = aQ + C::QTILE;
      unsigned char *wst = smem + (size_t)(buf ^ 1) * C::STAGE + qdst;

      // micro-step s -> (ks, m, g); operand fragments are keyed by (ks, g), genotype fragments by (ks, m)
      constexpr int NG = (NT + G - 1) / G; // n-tiles per group (the last group may be shorter)
      v4i bq[2][NQ][NG];
      v4i bqh = {0, 0, 0, 0}, bqh2 = {0, 0, 0, 0}; // HALF: the 16-column operand(s) of the current pair of k-steps
      v4i ag[AMASK + 1], am[AMASK + 1]; // decoded genotype fragments, index = micro-step parity (HALF: (k-step parity, m))
      bool enz[AMASK + 1];             // ... and whether the E fragment has any nonzero at all (wave-uniform)
      const uint32_t aQh = aQh0 + (uint32_t)buf * C::STAGE;
      auto read_b = [&](auto kk, auto gg, auto par) {
         constexpr int ks = decltype(kk)::value, g = decltype(gg)::value, p = decltype(par)::value;
         static_for<NG>([&](auto jj) {
            constexpr int j = decltype(jj)::value, n = g * NG + j;
            if constexpr (n < NTF) {
               bq[p][0][j] = lds_read16<32 * n * LDQ + ks * 16>(aQ);
               if constexpr (TWO) bq[p][NQ - 1][j] = lds_read16<32 * n * LDQ + ks * 16>(aQ2);
            }
         });
         if constexpr (HALF && (ks & 1) == 1) { // both k-steps of the pair in one read
            bqh = lds_read16<(ks >> 1) * 32>(aQh);
            if constexpr (TWO) bqh2 = lds_read16<(ks >> 1) * 32 + C::QTILE>(aQh);
         }
      };
      auto wait_b = [&](auto gg, auto par) {
         constexpr int g = decltype(gg)::value, p = decltype(par)::value;
         static_for<NG>([&](auto jj) {
            constexpr int j = decltype(jj)::value, n = g * NG + j;
            if constexpr (n < NTF) {
               if constexpr (TWO)
                  lds_wait(bq[p][0][j], bq[p][NQ - 1][j]);
               else
                  lds_wait(bq[p][0][j]);
            }
         });
         if constexpr (HALF) {
            if constexpr (TWO)
               lds_wait(bqh, bqh2);
            else
               lds_wait(bqh);
         }
      };
      read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      enz[0] = i8_decode<MODE>(pk[0][0][0], ag[0], am[0], tab1);
      wait_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if constexpr ((C::ABL & 4) != 0) {
         read_b(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
         wait_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      }
      if constexpr ((C::ABL & 2) != 0) enz[1] = i8_decode<MODE>(pk[0][0][1], ag[1], am[1], tab1);
      __builtin_amdgcn_sched_barrier(0);

      static_for<NSTEP>([&](auto ss) {
         constexpr int s = decltype(ss)::value;
         constexpr int ks = s / (MT * G), m = (G == 1) ? s % MT : 0, g = (G == 1) ? 0 : s % G;
         constexpr int bkey = ks * G + g, akey = ks * MT + m;
         constexpr bool last = (s + 1 == NSTEP);
         constexpr int s1 = last ? s : s + 1;
         constexpr int ks1 = s1 / (MT * G), m1 = (G == 1) ? s1 % MT : 0, g1 = (G == 1) ? 0 : s1 % G;
         constexpr int bkey1 = ks1 * G + g1, akey1 = ks1 * MT + m1;
         // --- staging of chunk cn: loads in the first half, packed words at the half-way mark, LDS stores in the second
         if constexpr (s < H && !(C::ABL & 1)) {
            static_for<(s + 1) * NP / H - s * NP / H>([&](auto jj) {
               issue_q(std::integral_constant<int, s * NP / H + decltype(jj)::value>{}, cn);
            });
         }
         if constexpr (s == H - 1 && !(C::ABL & 8)) issue_p(pkn, cn);
         if constexpr (s >= H && !(C::ABL & 1)) {
            static_for<(s - H + 1) * NP / H - (s - H) * NP / H>([&](auto jj) {
               store_q(std::integral_constant<int, (s - H) * NP / H + decltype(jj)::value>{}, wst);
            });
         }
         // --- operand fragments of the next micro-step
         if constexpr (bkey1 != bkey && !(C::ABL & 4))
            read_b(std::integral_constant<int, ks1>{}, std::integral_constant<int, g1>{}, std::integral_constant<int, (bkey1 & 1)>{});
         // --- this micro-step's MFMAs, the next micro-step's decode in their shadow
         static_for<NG>([&](auto jj) {
            constexpr int j = decltype(jj)::value, n = g * NG + j;
            if constexpr (n < NTF) {
               acc[0][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ag[akey & AMASK], bq[bkey & 1][0][j], acc[0][m][n], 0, 0, 0);
               if constexpr (MODE == I8_FULL)
                  acc[1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(am[akey & AMASK], bq[bkey & 1][NQ - 1][j], acc[1][m][n], 0, 0, 0);
            }
            if constexpr (j == 0 && akey1 != akey && !(C::ABL & 2))
               enz[akey1 & AMASK] = i8_decode<MODE>(pk[m1][ks1 >> 2][ks1 & 3], ag[akey1 & AMASK], am[akey1 & AMASK], tab1);
         });
         if constexpr (HALF && (ks & 1) == 1) {
            // the 16 remaining columns over k-steps ks - 1 and ks: row group r of the even / odd fragment = rows 16 (r & 1) ..
            // of k-half r >> 1; v_permlane16_swap exchanges row groups 1, 3 of its first operand with 0, 2 of its second, which
            // leaves [even.0, odd.0, even.2, odd.2] (rows 0..15 in all four k quarters) and [even.1, odd.1, even.3, odd.3]
            v4i t0 = ag[m], t1 = ag[2 + m];
#pragma unroll
            for (int q = 0; q < 4; q++) {
               const auto r = __builtin_amdgcn_permlane16_swap((unsigned)t0[q], (unsigned)t1[q], false, false);
               t0[q] = (int)r[0];
               t1[q] = (int)r[1];
            }
            acch[0][m][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(t0, bqh, acch[0][m][0], 0, 0, 0);
            acch[0][m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(t1, bqh, acch[0][m][1], 0, 0, 0);
            if constexpr (MATS == 2) { // the missing-indicator matrix against its own operand (K3) or the same one (K2)
               v4i u0 = am[m], u1 = am[2 + m];
#pragma unroll
               for (int q = 0; q < 4; q++) {
                  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)u0[q], (unsigned)u1[q], false, false);
                  u0[q] = (int)r[0];
                  u1[q] = (int)r[1];
               }
               acch[1][m][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(u0, TWO ? bqh2 : bqh, acch[1][m][0], 0, 0, 0);
               acch[1][m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(u1, TWO ? bqh2 : bqh, acch[1][m][1], 0, 0, 0);
            }
         }
         if constexpr (MODE == I8_SKIP_EMPTY) {
            if (enz[akey & AMASK]) {
               static_for<NG>([&](auto jj) {
                  constexpr int j = decltype(jj)::value, n = g * NG + j;
                  if constexpr (n < NT)
                     acc[1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(am[akey & AMASK], bq[bkey & 1][NQ - 1][j], acc[1][m][n], 0, 0, 0);
               });
            }
         }
         __builtin_amdgcn_sched_barrier(0);
         if constexpr (bkey1 != bkey && !(C::ABL & 4)) {
            wait_b(std::integral_constant<int, g1>{}, std::integral_constant<int, (bkey1 & 1)>{});
            __builtin_amdgcn_sched_barrier(0);
         }
      });
      if constexpr (!(C::ABL & 8)) {
#pragma unroll
         for (int m = 0; m < MT; m++)
#pragma unroll
            for (int h = 0; h < PW; h++) {
               vm_wait<0>(pkn[m][h]);
               pk[m][h] = pkn[m][h];
            }
      }
      if constexpr (!(C::ABL & 64)) __syncthreads(); // (ablation bit 6: what the per-chunk barrier + pipeline refill costs)
   }

   // epilogue: the slices are recombined here, per split and column block, into fp64 partial sums -- virtual column
   // j = sc mod bw (bw = lcm(32, b): slice-column sc = s*b + c lands on j == c mod b), so a row of partials is
   // 2 x bw doubles instead of 2 x S*b int32.  Every product w * acc is exact (w is a power of two, |acc| < 2^31); only
   // the additions round, last slice (smallest terms) first.
   const int tile0 = (col0 + wc * 32 * NT) / 32;
   // partial planes: plane 0 holds every row, planes 1 .. sB-1 only the rows of the phase-B tiles (rows >= rowB0)
   double *out = (split == 0 ? part + (((size_t)zblk * rows_pad + row0 + wr * 32 * MT) * 2) * bw
                             : part + (size_t)zb * rows_pad * 2 * bw +
                                  ((((size_t)(split - 1) * zb + zblk) * rowsB + (row0 - rowB0) + wr * 32 * MT) * 2) * bw) +
                 li;
   //
   // Register discipline: the 256 accumulators own every AGPR, and reading ONE element of an AGPR-resident 16-vector makes
   // the compiler copy the whole vector to VGPRs, so any epilogue that walks rows across tiles keeps hundreds of extra
   // VGPRs alive and the allocator answers by spilling INSIDE the main loop (measured: 10 scratch stores per chunk,
   // 3.5 GB of spill traffic per launch, matrix pipe 48 % busy).  So the tiles are dumped to LDS one at a time (the operand
   // tiles are dead by now; 4 KB per tile and wave) and the recombination runs from LDS with a handful of registers.
   __syncthreads(); // every wave has finished reading the operand tiles
   constexpr int WREG = NT * 4096; // bytes of a wave's private tile area: [NT][32 rows][32 cols] int32
   double *sW = reinterpret_cast<double *>(smem + 4 * WREG); // weights of this workgroup's slice-columns: [2][COLS]
   constexpr int WCOLS = C::WC * NT * 32; // (HALF: the 16 columns missing from the last tile carry zero weights)
   for (int t = tid; t < WCOLS; t += 256) {
      sW[t] = t < C::COLS ? wg[col0 + t] : 0.0;
      sW[WCOLS + t] = t < C::COLS ? wm[col0 + t] : 0.0;
   }
   __syncthreads();
   static_assert(C::LDS_BYTES >= 4 * WREG + 2 * WCOLS * 8, "epilogue LDS layout");
   int *sT = reinterpret_cast<int *>(smem + wave * WREG);
   const double *sWl = sW + wc * 32 * NT + li;
   const int kb = bw / 32; // a lane's tiles n, n + kb, ... feed the same virtual column 32 ((tile0 + n) % kb) + li
#pragma unroll
   for (int a = 0; a < MATS; a++) // (the E plane of the partials is not written in I8_NO_MISSING mode: the combine knows)
#pragma unroll
      for (int m = 0; m < MT; m++) {
#pragma unroll
         for (int n = 0; n < NTF; n++) {
            const v16i v = acc[a][m][n];
#pragma unroll
            for (int r = 0; r < 16; r++) sT[(n * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + li] = v[r];
            __builtin_amdgcn_sched_barrier(0);
         }
         if constexpr (HALF) { // 16x16 C/D map: register r of lane l = D[4 (l >> 4) + r][l & 15]; columns 16..31 of the tile are zero
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
               for (int r = 0; r < 4; r++) {
                  const int row = 16 * t + 4 * (lane >> 4) + r;
                  sT[(NTF * 32 + row) * 32 + (lane & 15)] = acch[a][m][t][r];
                  sT[(NTF * 32 + row) * 32 + 16 + (lane & 15)] = 0;
               }
         }
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         // lane (li, kh) recombines rows 16 kh .. 16 kh + 15 of column li
         for (int j = 0; j < 16; j++) {
            const int row = 16 * kh + j;
            for (int k = 0; k < kb; k++) {
               double acc64 = 0.0;
               for (int n = NT - 1 - ((NT - 1 - k) % kb); n >= 0; n -= kb) // tiles n == k (mod kb), last slice first
                  acc64 += sWl[a * WCOLS + 32 * n] * (double)sT[(n * 32 + row) * 32 + li];
               out[((size_t)(32 * m + row) * 2 + a) * bw + 32 * ((tile0 + k) % kb)] = acc64;
            }
         }
         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
         __builtin_amdgcn_wave_barrier();
      }
}

// sum of the per-split / per-column-block fp64 partials, fold of the virtual columns (j == c mod b), M'Q = 1'Q - E'Q,
// and the per-row standardisation:
//   K2 (mean != null): out[row][c] = ( G[row][c] - mean[row] M[row][c] ) / sd[row]   (0 if sd <= 1e-9)
//   K3               : out[row][c] =   G[row][c] - M[row][c]
// The K2 flavour can also leave the column maxima of out * rs0 and out * rs1 behind (the two K3 operands), which saves
// the next stage a pass over T.
__global__ __launch_bounds__(256) void k_i8_combine(const double *__restrict__ part, int zb, int rows_tile, int nA, int sB, uint64_t rowB0,
                                                     uint64_t rowsB, uint64_t rows_pad, uint64_t rows_valid, int mats,
                                                     const double *__restrict__ eplane /* [rows_pad][b]: E'Q from the sparse path */,
                                                     int b, int bw, int S,
                                                     const double *__restrict__ wm, const long long *__restrict__ colsum_m /* 1'Qm */,
                                                     const double *__restrict__ mean, const double *__restrict__ sd,
                                                     double *__restrict__ out, const double *__restrict__ rs0,
                                                     unsigned long long *__restrict__ bits0, const double *__restrict__ rs1,
                                                     unsigned long long *__restrict__ bits1)
{
   __shared__ double smax[256], sw[9 * 64], sones[64];
   const int c = threadIdx.x % b, r_in = threadIdx.x / b, r_step = 256 / b;
   double m0 = 0.0, m1 = 0.0;
   // 1'Qm recombined once per block: sum_s w[s,c] colsum[s,c] (shards folded), small terms first
   // (mats == -1: the plain product of ONE integer matrix -- out = sum of its partials, rows >= rows_valid zero)
   for (int t = threadIdx.x; t < S * b; t += 256) sw[t] = (mats < 0 || !colsum_m) ? 0.0 : wm[t] * (double)colsum_fold(colsum_m, t);
   __syncthreads();
   if ((int)threadIdx.x < b) {
      double o = 0.0;
      for (int s = S - 1; s >= 0; s--) o += sw[s * b + threadIdx.x];
      sones[threadIdx.x] = o;
   }
   __syncthreads();
   if (r_in < r_step) {
      const double ones = sones[c];
      for (uint64_t row = (uint64_t)blockIdx.x * r_step + r_in; row < rows_pad; row += (uint64_t)gridDim.x * r_step) {
         double accg = 0.0, acce = 0.0;
         const int rt = (int)(row / rows_tile);
         for (int z = 0; z < zb; z++) {
            const int w0 = ((rt >> 3) * zb + z) * 8 + (rt & 7);
            const int np = w0 < nA ? 1 : sB; // phase-A tiles were not split
            for (int p = 0; p < np; p++) {
               const double *q = p == 0 ? part + (((size_t)z * rows_pad + row) * 2) * bw
                                        : part + (size_t)zb * rows_pad * 2 * bw + ((((size_t)(p - 1) * zb + z) * rowsB + (row - rowB0)) * 2) * bw;
               for (int j = c; j < bw; j += b) {
                  accg += q[j];
                  if (mats == 2) acce += q[bw + j];
               }
            }
         }
         // (mats == 1: E is not in the partials -- it is zero, or comes from the sparse path; the padding rows are all
         // "missing" and must stay zero)
         if (mats == 1) acce = row >= rows_valid ? ones : (eplane ? eplane[row * b + c] : 0.0);
         const double accm = ones - acce;
         double v;
         if (mats < 0)
            v = row < rows_valid ? accg : 0.0;
         else if (mean) {
            const double sdv = sd[row];
            v = (sdv > 1e-9) ? (accg - mean[row] * accm) / sdv : 0.0;
         } else
            v = accg - accm;
         out[row * b + c] = v;
         if (bits0) {
            m0 = fmax(m0, fabs(v * rs0[row]));
            m1 = fmax(m1, fabs(v * rs1[row]));
         }
      }
   }
   if (bits0) {
      colmax_commit(m0, b, smax, bits0);
      colmax_commit(m1, b, smax, bits1);
   }
}

// ---- shape selection ----
// The slice-columns (S*b, rounded up to whole 32-column tiles) are cut into `zb` equal column blocks of NT tiles
// (blockIdx.z); 4 x 1 waves per workgroup either way, KC = 256:
//   K2: wave = 32 rows x NT <= 8 tiles (every packed row is decoded exactly once; any S without padding for b = 32)
//   K3: wave = 64 rows x NT = 3..4 tiles of BOTH operands (two operand tiles per stage -> half the columns per block)
// Measured alternatives at cfg3, S = 8 (K2 / K3 ms): 2 x 2 waves of 64 x 128 | 128 x 64: 19.2 / 22.4; K3 with 32-row waves,
// all tiles and KC = 128: 24.3; K2 with the K3 shape: 18.9 -- versus 18.7 / 20.6 for the shapes kept.
struct I8Shape {
   int nt, zb, rows, cols, kc;
   bool half; // the last tile is the 16-column remainder (v_mfma_i32_16x16x64_i8), cols = 32 nt - 16
   int mt = 2; // 32-row tiles per wave (one-matrix kernel: 2, or 4 for the narrow column blocks)
};

static I8Shape i8_shape(int S, int b, bool two, int mode = I8_FULL)
{
   const int tiles = (S * b + 31) / 32, cap = two ? 4 : 8; // largest instantiated block
   // smallest instantiated block: 2 tiles (few slices of a narrow block: S = 4, b = 16 is 64 slice-columns).  Rounds 2-5 had the
   // two-matrix kernels from 4 (K2) / 3 (K3) tiles up only: the eigensolver's 4-slice passes on data whose missing calls take the dense
   // route multiplied 2 / 1 tiles of zero padding per launch (profiles/r06_missing_routes.txt)
   int lo = 2;
   if (FPCA_TEST_ENV("FPCA_I8_LO_R5")) lo = two ? 3 : (mode == I8_NO_MISSING ? 2 : 4); // (A/B against round 5's shapes)
   I8Shape sh;
   sh.zb = (tiles + cap - 1) / cap;
   sh.nt = std::max(lo, (tiles + sh.zb - 1) / sh.zb);
   // one matrix only (I8_NO_MISSING): the freed accumulators go into a second row tile per wave (64 rows x NT tiles)
   sh.rows = (two || mode == I8_NO_MISSING || (sh.nt <= 3 && !FPCA_TEST_ENV("FPCA_I8_LO_R5") && !FPCA_TEST_ENV("FPCA_I8_NARROW_MT1"))) ? 256 : 128;
   if (!two && mode == I8_NO_MISSING && sh.nt <= 3 && FPCA_TEST_ENV("FPCA_I8_MT4")) { // experiment: 128 rows per wave
      sh.mt = 4;
      sh.rows = 512;
   }
   sh.cols = 32 * sh.nt;
   sh.kc = 256;
   // b = 16 with S = 7 slices: 112 slice-columns = 3.5 tiles -- the one-matrix kernel (the default route up to 0.5 % missing
   // calls) takes the remainder as a half tile
   // ... and so does b = 16 with S = 3 (48 slice-columns = 1.5 tiles: the eigensolver's cheap passes, one-matrix kernel only)
   sh.half = mode != I8_SKIP_EMPTY && sh.zb == 1 && S * b == 32 * sh.nt - 16 && (sh.nt == 4 || (sh.nt == 2 && !two && mode == I8_NO_MISSING));
   if (sh.half) {
      sh.cols -= 16;
      sh.rows = 256; // (every half-tile instantiation is 4 waves x 64 rows, the two-matrix K2 one included)
   }
   return sh;
}

int gemm_i8_nsc_pad(int S, int b)
{
   const I8Shape s2 = i8_shape(S, b, false), s3 = i8_shape(S, b, true); // one padded width serves both kernels
   return std::max(s2.zb * 32 * s2.nt, s3.zb * 32 * s3.nt); // (whole tiles: the zero rows behind S*b carry zero weights)
}

// Work decomposition of one GEMM launch (see k_gemm_i8): one workgroup per CU, so whole rounds of #CU tiles run unsplit
// (phase A) and only the leftover tiles are split along K (phase B); each extra split of phase B costs a partial plane
// for ITS rows only.  Plain split-K (nA = 0) is kept for problems with less than one round of tiles.
// (A stream-K schedule -- one persistent workgroup per CU walking a contiguous range of (tile, chunk) units -- was built
// and measured: it removes the round quantisation too, but it breaks the lock-step in which the workgroups of a round
// stream the same operand chunks through L2 and adds a prologue/epilogue per segment; net: K3 0.561 vs 0.568 ms at cfg2,
// +10 % time at cfg3.  Not kept.)
struct I8Plan {
   int nA, sB, cpsB;
   uint64_t rowB0, rowsB;
   unsigned grid;
};

static I8Plan i8_plan(uint64_t rows_pad, uint64_t k_pad, const I8Shape &sh, int bw)
{
   static const char *env_s = FPCA_TEST_ENV("FPCA_I8_SPLITS"); // force plain split-K with this factor
   static int ncu = 0;
   if (!ncu) {
      hipDeviceProp_t prop;
      int dev = 0;
      (void)hipGetDevice(&dev);
      ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? prop.multiProcessorCount / 8 * 8 : 256;
   }
   const int rtiles = (int)(rows_pad / sh.rows), rtl = (rtiles + 7) / 8, ids = 8 * rtl * sh.zb;
   const int chunks = (int)(k_pad / sh.kc);
   const double t_chunk = 2.2e-6 * (double)sh.rows * sh.cols * sh.kc / (128.0 * 256 * 256); // one workgroup-chunk at ~3 POP/s
   const double t_seg = 5e-6;                                                               // prologue + epilogue of a workgroup
   const double t_row_plane = 2.0 * bw * 8 * 2 * sh.rows / 3.0e12;                          // one tile's partial, written + read
   double best = 1e30;
   int best_nA = 0, best_s = 1;
   const int nA_full = ids / ncu * ncu;
   for (int pass = 0; pass < 2; pass++) {
      const int nA = pass == 0 ? 0 : (nA_full == ids ? ids - ncu : nA_full);
      if (pass == 1 && (nA <= 0 || (env_s && atoi(env_s) > 0))) break;
      const int nB = ids - nA;
      for (int s = 1; s <= 16 && (s == 1 || s * 4 <= chunks); s++) {
         if (pass == 0 && env_s && atoi(env_s) > 0 && s != std::min(atoi(env_s), std::max(chunks / 4, 1))) continue;
         const int cps = (chunks + s - 1) / s;
         const double tA = (double)(nA / ncu) * (chunks * t_chunk + t_seg);
         const double tB = (double)((nB * s + ncu - 1) / ncu) * (cps * t_chunk + t_seg);
         const double t = tA + tB + (s > 1 ? (double)s * nB * t_row_plane : 0.0);
         if (t < best * 0.995) {
            best = t;
            best_nA = nA;
            best_s = s;
         }
      }
   }
   {  // test builds: FPCA_I8_PLAN="nA,s" overrides the decomposition of launches with more than one round of tiles (A/B runs)
      const char *env_p = FPCA_TEST_ENV("FPCA_I8_PLAN");
      int e_nA = 0, e_s = 0;
      if (env_p && std::sscanf(env_p, "%d,%d", &e_nA, &e_s) == 2 && ids >= ncu && e_s >= 1 && e_s * 4 <= chunks) {
         best_nA = std::min(e_nA / (8 * sh.zb) * (8 * sh.zb), ids);
         best_s = best_nA == ids ? 1 : e_s;
      }
   }
   I8Plan p;
   p.nA = best_nA;
   p.cpsB = (chunks + best_s - 1) / best_s;
   p.sB = (chunks + p.cpsB - 1) / p.cpsB; // no empty split
   const int qA = (p.nA / 8) / sh.zb;     // phase-B tiles have row tile >= 8 qA
   p.rowB0 = std::min<uint64_t>((uint64_t)qA * 8 * sh.rows, rows_pad);
   p.rowsB = rows_pad - p.rowB0;
   p.grid = (unsigned)(p.nA + (ids - p.nA) * p.sB);
   static const bool verbose = FPCA_TEST_ENV("FPCA_I8_VERBOSE") != nullptr;
   if (verbose)
      std::fprintf(stderr, "[fpca] int8 GEMM plan: rows %llu K %llu tile %dx%d zb %d -> %d tile ids, %d chunks; %d unsplit + %d tiles x %d splits (%d chunks each), %u workgroups, est %.3f ms\n",
                   (unsigned long long)rows_pad, (unsigned long long)k_pad, sh.rows, sh.cols, sh.zb, ids, chunks, p.nA, ids - p.nA, p.sB, p.cpsB, p.grid,
                   best * 1e3);
   return p;
}

static int i8_bw(int b) // lcm(32, b) for b in {16, 32, 48, 64}
{
   return b == 48 ? 96 : std::max(b, 32);
}

size_t gemm_i8_workspace_doubles(uint64_t rows_pad, uint64_t k_pad, int S, int b, bool two)
{
   size_t need = 0;
   for (int mode : {(int)I8_FULL, (int)I8_SKIP_EMPTY, (int)I8_NO_MISSING}) { // (the shapes differ: only FULL / NO_MISSING have the half tile)
      if (two && mode == I8_NO_MISSING) continue;
      const I8Shape sh = i8_shape(S, b, two, mode);
      const I8Plan p = i8_plan(rows_pad, k_pad, sh, i8_bw(b));
      need = std::max(need, ((size_t)rows_pad + (size_t)(p.sB - 1) * p.rowsB) * sh.zb * 2 * (size_t)i8_bw(b));
   }
   return need;
}

template <class C>
static void launch_i8(const I8Plan &pl, hipStream_t stream, const uint8_t *packed, size_t pitch, const int8_t *Qg, const int8_t *Qm,
                      uint64_t k_pad, const double *wg, const double *wm, int bw, double *ws, uint64_t rows_pad, int chunks_total, int zb, uint32_t tab1)
{
   static bool attr_set = false, attr_set2 = false;
   if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_i8<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      attr_set = true;
   }
   // (test builds: FPCA_I8_LDS_PAD = extra bytes of dynamic LDS per workgroup, i.e. fewer co-resident workgroups per CU)
   static const int lds_pad = FPCA_TEST_ENV("FPCA_I8_LDS_PAD") ? atoi(FPCA_TEST_ENV("FPCA_I8_LDS_PAD")) : 0;
   if (lds_pad && !attr_set2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_i8<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + lds_pad);
      attr_set2 = true;
   }
   hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gemm_i8<C>), dim3(pl.grid), dim3(256), C::LDS_BYTES + lds_pad, stream, packed, pitch, Qg, Qm, k_pad, wg, wm, bw,
                      ws, rows_pad, chunks_total, zb, pl.nA, pl.sB, pl.cpsB, pl.rowB0, pl.rowsB, tab1);
}

void gemm_i8(const uint8_t *packed, size_t pitch, const int8_t *Qg, const int8_t *Qm, const double *wg, const double *wm,
             const long long *colsum_m, const double *mean, const double *sd, double *out, double *ws, uint64_t rows_pad, uint64_t k_pad,
             uint64_t rows_valid, int mode, const double *eplane /* I8_NO_MISSING kernel + E'Q computed elsewhere, or null */, int b,
             int S, const SliceOp *next_ops /* null, or the two operands whose column maxima the combine should leave */,
             hipStream_t stream, hipEvent_t *gemm_events /* null, or 2 events recorded around the GEMM kernel itself */,
             hipEvent_t before_combine /* null, or an event the combine must wait for (eplane produced on another stream) */,
             bool e_only /* mode 2 only: the missing-indicator matrix E alone, out = E Q (no statistics, no column sums) */)
{
   const bool two = (Qg != Qm);
   if (e_only && (mode != I8_NO_MISSING || two)) throw Error(-1, "gemm_i8: e_only goes with the one-matrix kernel");
   const I8Shape sh = i8_shape(S, b, two, mode);
   const int bw = i8_bw(b); // Q holds gemm_i8_nsc_pad(S, b) rows; rows >= S*b are zero and carry zero weights
   const I8Plan pl = i8_plan(rows_pad, k_pad, sh, bw);
   const int chunks_total = (int)(k_pad / sh.kc);
#define FPCA_I8_ARGS pl, stream, packed, pitch, Qg, Qm, k_pad, wg, wm, bw, ws, rows_pad, chunks_total, sh.zb, (e_only ? 0x00000100u : 0x00010002u)
   if (gemm_events) (void)hipEventRecord(gemm_events[0], stream);
   if (two && mode == I8_NO_MISSING) throw Error(-1, "gemm_i8: without missing genotypes both matrices share one operand (pass Qm == Qg)");
#define FPCA_I8_K3(NT_, MODE_) launch_i8<I8Cfg<true, 2, NT_, 4, 1, 256, 1, MODE_>>(FPCA_I8_ARGS)
#define FPCA_I8_K2(NT_, MODE_) launch_i8<I8Cfg<false, (MODE_ == I8_NO_MISSING ? 2 : 1), NT_, 4, 1, 256, (MODE_ == I8_NO_MISSING ? 1 : 2), MODE_>>(FPCA_I8_ARGS)
// (narrow column blocks: 64-row waves for both matrices too -- the accumulators fit, the operand tile is staged for twice the rows)
#define FPCA_I8_K2W(NT_, MODE_) launch_i8<I8Cfg<false, 2, NT_, 4, 1, 256, 1, MODE_>>(FPCA_I8_ARGS)
#define FPCA_I8_K2_NT(MODE_)                                                                                 \
   switch (sh.nt) {                                                                                           \
   case 2: if (sh.rows == 256) FPCA_I8_K2W(2, MODE_); else FPCA_I8_K2(2, MODE_); break;                       \
   case 3: if (sh.rows == 256) FPCA_I8_K2W(3, MODE_); else FPCA_I8_K2(3, MODE_); break;                       \
   case 4: FPCA_I8_K2(4, MODE_); break;                                                                       \
   case 5: FPCA_I8_K2(5, MODE_); break;                                                                       \
   case 6: FPCA_I8_K2(6, MODE_); break;                                                                       \
   case 7: FPCA_I8_K2(7, MODE_); break;                                                                       \
   default: FPCA_I8_K2(8, MODE_); break;                                                                      \
   }
   if (two && sh.half)
      launch_i8<I8Cfg<true, 2, 4, 4, 1, 256, 1, I8_FULL, 0, true>>(FPCA_I8_ARGS);
   else if (!two && sh.half && mode == I8_FULL)
      launch_i8<I8Cfg<false, 2, 4, 4, 1, 256, 1, I8_FULL, 0, true>>(FPCA_I8_ARGS);
   else if (two) {
      if (mode == I8_SKIP_EMPTY) {
         if (sh.nt == 2)
            FPCA_I8_K3(2, I8_SKIP_EMPTY);
         else if (sh.nt == 3)
            FPCA_I8_K3(3, I8_SKIP_EMPTY);
         else
            FPCA_I8_K3(4, I8_SKIP_EMPTY);
      } else {
         if (sh.nt == 2)
            FPCA_I8_K3(2, I8_FULL);
         else if (sh.nt == 3)
            FPCA_I8_K3(3, I8_FULL);
         else
            FPCA_I8_K3(4, I8_FULL);
      }
   } else if (mode == I8_NO_MISSING) {
#ifdef FPCA_I8_ABLATION
      static const char *abl = FPCA_TEST_ENV("FPCA_I8_ABL");
      const int ab = abl ? atoi(abl) : 0;
#define FPCA_I8_AB(A_) launch_i8<I8Cfg<false, 2, 7, 4, 1, 256, 1, I8_NO_MISSING, A_>>(FPCA_I8_ARGS)
      if (ab && sh.nt == 7) {
         switch (ab) {
         case 1: FPCA_I8_AB(1); break;
         case 2: FPCA_I8_AB(2); break;
         case 4: FPCA_I8_AB(4); break;
         case 8: FPCA_I8_AB(8); break;
         case 9: FPCA_I8_AB(9); break;
         case 6: FPCA_I8_AB(6); break;
         case 16: FPCA_I8_AB(16); break;
         case 32: FPCA_I8_AB(32); break;
         case 48: FPCA_I8_AB(48); break;
         case 64: FPCA_I8_AB(64); break;
         case 112: FPCA_I8_AB(112); break;
         default: FPCA_I8_AB(15); break;
         }
      } else if (ab && sh.nt == 2 && sh.mt == 2) { // the 2-tile kernel of the 4-slice passes (profiles/r04_i8_ablation_2tile.txt)
#define FPCA_I8_AB2(A_) launch_i8<I8Cfg<false, 2, 2, 4, 1, 256, 1, I8_NO_MISSING, A_>>(FPCA_I8_ARGS)
         switch (ab) {
         case 1: FPCA_I8_AB2(1); break;
         case 2: FPCA_I8_AB2(2); break;
         case 8: FPCA_I8_AB2(8); break;
         case 16: FPCA_I8_AB2(16); break;
         case 32: FPCA_I8_AB2(32); break;
         case 48: FPCA_I8_AB2(48); break;
         case 64: FPCA_I8_AB2(64); break;
         default: FPCA_I8_AB2(15); break;
         }
      } else
#endif
      if (sh.half && sh.nt == 2)
         launch_i8<I8Cfg<false, 2, 2, 4, 1, 256, 1, I8_NO_MISSING, 0, true>>(FPCA_I8_ARGS);
      else if (sh.half)
         launch_i8<I8Cfg<false, 2, 4, 4, 1, 256, 1, I8_NO_MISSING, 0, true>>(FPCA_I8_ARGS);
      else if (sh.nt == 2 && sh.mt == 4)
         launch_i8<I8Cfg<false, 4, 2, 4, 1, 256, 1, I8_NO_MISSING>>(FPCA_I8_ARGS);
      else if (sh.nt == 3 && sh.mt == 4)
         launch_i8<I8Cfg<false, 4, 3, 4, 1, 256, 1, I8_NO_MISSING>>(FPCA_I8_ARGS);
      else if (sh.nt == 2)
         launch_i8<I8Cfg<false, 2, 2, 4, 1, 256, 1, I8_NO_MISSING>>(FPCA_I8_ARGS);
      else if (sh.nt == 3)
         launch_i8<I8Cfg<false, 2, 3, 4, 1, 256, 1, I8_NO_MISSING>>(FPCA_I8_ARGS);
      else
         FPCA_I8_K2_NT(I8_NO_MISSING)
   } else if (mode == I8_SKIP_EMPTY) {
      FPCA_I8_K2_NT(I8_SKIP_EMPTY)
   } else {
      FPCA_I8_K2_NT(I8_FULL)
   }
#undef FPCA_I8_K2_NT
#undef FPCA_I8_K2W
#undef FPCA_I8_K2
#undef FPCA_I8_K3
#undef FPCA_I8_ARGS
   HIP_CHECK_LAUNCH();
   if (gemm_events) (void)hipEventRecord(gemm_events[1], stream);
   if (before_combine) (void)hipStreamWaitEvent(stream, before_combine, 0);
   const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (rows_pad + (256 / b) - 1) / (256 / b));
   hipLaunchKernelGGL(k_i8_combine, dim3(blocks), dim3(256), 0, stream, ws, sh.zb, sh.rows, pl.nA, pl.sB, pl.rowB0, pl.rowsB, rows_pad, rows_valid, e_only ? -1 : mode == I8_NO_MISSING ? 1 : 2, eplane, b, bw, S, wm, colsum_m,
                      mean, sd, out,
                      next_ops ? next_ops[0].rowscale : nullptr, next_ops ? next_ops[0].maxbits : nullptr,
                      next_ops ? next_ops[1].rowscale : nullptr, next_ops ? next_ops[1].maxbits : nullptr);
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
// Sparse missing indicator.  With a typical array-data missing rate (0.1 %) the E half of the int8 work multiplies a
// matrix that is 99.9 % zeros.  Instead: index lists of the missing calls (per SNP for K2, per sample for K3), built
// once, and  E'B  /  E (mean T / sd)  as gathers of fp64 rows -- 256 bytes per missing call, a few ms where the MFMA
// route took 8-9 -- while the int8 GEMM multiplies G.M alone with the one-matrix kernel.

// missing calls of each 2-bit record among its first `ncols` codes
__global__ __launch_bounds__(256) void k_count_missing(const uint8_t *__restrict__ packed, size_t pitch, uint64_t ncols,
                                                        uint32_t *__restrict__ cnt)
{
   const uint8_t *row = packed + (uint64_t)blockIdx.x * pitch;
   const uint64_t nbytes = (ncols + 3) / 4, nw = nbytes / 4;
   uint32_t n = 0;
   for (uint64_t i = threadIdx.x; i < nw; i += 256) {
      const uint32_t w = reinterpret_cast<const uint32_t *>(row)[i];
      n += __popc(w & ~(w >> 1) & 0x55555555u);
   }
   for (uint64_t i = nw * 4 + threadIdx.x; i < nbytes; i += 256) {
      const uint32_t w = row[i];
      n += __popc(w & ~(w >> 1) & 0x55u);
   }
   // codes beyond ncols in the last byte
   if (threadIdx.x == 0 && (ncols & 3)) {
      const uint32_t w = row[nbytes - 1] >> (2 * (ncols & 3));
      n -= __popc(w & ~(w >> 1) & 0x55u);
   }
   __shared__ uint32_t red[256];
   red[threadIdx.x] = n;
   __syncthreads();
   for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
   }
   if (threadIdx.x == 0) cnt[blockIdx.x] = red[0];
}

// idx[ptr[r] ..] = ascending positions (< ncols) of the missing calls of record r.  The record is walked in tiles of 1024
// dwords (coalesced 16-byte loads), every thread owns four consecutive dwords = 64 codes; a wave scan + 4 wave totals
// place each thread's hits.  (Rows are 128-byte aligned and padded with "missing" codes up to the pitch, so whole
// 16-byte pieces can be read; positions >= ncols are masked off.)
__global__ __launch_bounds__(256) void k_fill_missing(const uint8_t *__restrict__ packed, size_t pitch, uint64_t ncols,
                                                       const uint32_t *__restrict__ ptr, uint32_t *__restrict__ idx)
{
   const u4 *row = reinterpret_cast<const u4 *>(packed + (uint64_t)blockIdx.x * pitch);
   const uint64_t nq = (ncols + 63) / 64; // 16-byte pieces holding valid codes
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   __shared__ uint32_t wsum[4];
   uint32_t base = ptr[blockIdx.x];
   if (ptr[blockIdx.x + 1] == base) return; // nothing to list: a record without a missing call -- or one whose missing calls go the
                                            // dense route (its count was set to zero on purpose, device_ctx.hip ensure_hybrid)
   for (uint64_t q0 = 0; q0 < nq; q0 += 256) {
      const uint64_t q = q0 + threadIdx.x;
      uint32_t m[4] = {0u, 0u, 0u, 0u};
      if (q < nq) {
         const u4 x = row[q];
#pragma unroll
         for (int k = 0; k < 4; k++) {
            m[k] = x[k] & ~(x[k] >> 1) & 0x55555555u;
            const uint64_t first = q * 64 + 16 * k; // first code of this dword
            if (first >= ncols)
               m[k] = 0u;
            else if (ncols - first < 16)
               m[k] &= (1u << (2 * (ncols - first))) - 1u;
         }
      }
      const uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
      uint32_t v = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
         const uint32_t t = __shfl_up(v, o);
         if (lane >= o) v += t;
      }
      if (lane == 63) wsum[wave] = v;
      __syncthreads();
      uint32_t woff = 0, total = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
         if (k < wave) woff += wsum[k];
         total += wsum[k];
      }
      uint32_t pos = base + woff + v - c;
#pragma unroll
      for (int k = 0; k < 4; k++) {
         uint32_t mk = m[k];
         while (mk) {
            const int bit = __ffs(mk) - 1;
            idx[pos++] = (uint32_t)(q * 64 + 16 * k + bit / 2);
            mk &= mk - 1;
         }
      }
      base += total;
      __syncthreads();
   }
}

void count_missing(const uint8_t *packed, size_t pitch, uint64_t ncols, uint64_t nrec, uint32_t *cnt, hipStream_t stream)
{
   if (!nrec) return;
   hipLaunchKernelGGL(k_count_missing, dim3((unsigned)nrec), dim3(256), 0, stream, packed, pitch, ncols, cnt);
   HIP_CHECK_LAUNCH();
}

void fill_missing(const uint8_t *packed, size_t pitch, uint64_t ncols, uint64_t nrec, const uint32_t *ptr, uint32_t *idx, hipStream_t stream)
{
   if (!nrec) return;
   hipLaunchKernelGGL(k_fill_missing, dim3((unsigned)nrec), dim3(256), 0, stream, packed, pitch, ncols, ptr, idx);
   HIP_CHECK_LAUNCH();
}

// out[r][c] = sum over s in list(r) of V[s][c] * (rowscale ? rowscale[s] : 1)   (fp64, list order = ascending s);
// rows r >= nrec are zeroed.  One wave per output row; EPW = 64 / b list entries per step, 4 steps in flight.
template <int B, class VT>
__global__ __launch_bounds__(256) void k_sparse_rows_sum(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                          const VT *__restrict__ V, const double *__restrict__ rowscale, uint64_t nrec,
                                                          uint64_t rows_out, double *__restrict__ out, const double *__restrict__ init,
                                                          const double *__restrict__ colw)
{
   constexpr int EPW = 64 / B;
   const int lane = threadIdx.x & 63, c = lane % B, e0 = lane / B;
   for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows_out; r += (uint64_t)gridDim.x * 4) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      if (r < nrec) {
         const uint32_t p0 = ptr[r], p1 = ptr[r + 1];
         for (uint32_t t = p0 + e0; t < p1; t += 4 * EPW) {
            const uint32_t t1 = t + EPW, t2 = t + 2 * EPW, t3 = t + 3 * EPW;
            const uint32_t s0 = idx[t], s1 = t1 < p1 ? idx[t1] : 0, s2 = t2 < p1 ? idx[t2] : 0, s3 = t3 < p1 ? idx[t3] : 0;
            double v0 = V[(uint64_t)s0 * B + c], v1 = t1 < p1 ? V[(uint64_t)s1 * B + c] : 0.0, v2 = t2 < p1 ? V[(uint64_t)s2 * B + c] : 0.0,
                   v3 = t3 < p1 ? V[(uint64_t)s3 * B + c] : 0.0;
            if (rowscale) {
               v0 *= rowscale[s0];
               v1 *= t1 < p1 ? rowscale[s1] : 0.0;
               v2 *= t2 < p1 ? rowscale[s2] : 0.0;
               v3 *= t3 < p1 ? rowscale[s3] : 0.0;
            }
            a0 += v0;
            a1 += v1;
            a2 += v2;
            a3 += v3;
         }
      }
      double a = (a0 + a1) + (a2 + a3);
#pragma unroll
      for (int o = 32; o >= B; o >>= 1) a += __shfl_down(a, o);
      if (lane < B) {
         if (colw) a *= colw[c] * 32.0; // fp32 rows were stored as x 2^(1 - e_c); colw[c] = 2^(e_c - 6) (top slice's weight)
         out[r * B + c] = init ? init[r * B + c] + a : a;
      }
   }
}

// The same sum with the index list read in coalesced batches of 64 (one entry per lane, broadcast by shuffles) instead of
// one dependent 4-byte load per gathered row: the loop above is a chain idx -> row of V, both at Infinity-Cache latency,
// with four rows in flight per wave; here the rows of a batch are independent of any further index load and eight of them
// are in flight (per lane slot).  The per-row factors of a batch are gathered once, one per lane, the same way.
template <int B, class VT>
__global__ __launch_bounds__(256) void k_sparse_rows_sum_batched(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                                  const VT *__restrict__ V, const double *__restrict__ rowscale,
                                                                  uint64_t nrec, uint64_t rows_out, double *__restrict__ out, const double *__restrict__ init,
                                                                  const double *__restrict__ colw)
{
   constexpr int EPW = 64 / B, U = 8;
   const int lane = threadIdx.x & 63, c = lane % B, e0 = lane / B;
   for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows_out; r += (uint64_t)gridDim.x * 4) {
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = 0.0;
      if (r < nrec) {
         const uint32_t p0 = ptr[r], p1 = ptr[r + 1];
         for (uint32_t t0 = p0; t0 < p1; t0 += 64) {
            const int cnt = (int)(p1 - t0 < 64u ? p1 - t0 : 64u);
            const uint32_t mine = lane < cnt ? idx[t0 + lane] : 0u;
            const double myscale = (rowscale && lane < cnt) ? rowscale[mine] : 1.0;
            for (int u0 = 0; u0 < cnt; u0 += EPW * U) {
#pragma unroll
               for (int u = 0; u < U; u++) {
                  const int e = u0 + u * EPW + e0;
                  const uint32_t srow = (uint32_t)__shfl((int)mine, e & 63);
                  const double sc = __shfl(myscale, e & 63);
                  if (e < cnt) acc[u] += V[(uint64_t)srow * B + c] * sc;
               }
            }
         }
      }
      double a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
      for (int o = 32; o >= B; o >>= 1) a += __shfl_down(a, o);
      if (lane < B) {
         if (colw) a *= colw[c] * 32.0; // fp32 rows were stored as x 2^(1 - e_c); colw[c] = 2^(e_c - 6) (top slice's weight)
         out[r * B + c] = init ? init[r * B + c] + a : a;
      }
   }
}

// Short lists (a dozen entries per row: the samples of a 1/8 SNP shard, small problems): with one wave per row the chain
// ptr -> idx -> rows is three dependent round trips per row and nothing else in flight in that wave -- latency-bound (4 TB/s
// out of an L2-resident operand).  Here 64 / B rows share a wave, B lanes (one per column) each, every group walking its own
// list four entries at a time: four times the rows in flight, no cross-lane reduction.
template <int B, class VT>
__global__ __launch_bounds__(256) void k_sparse_rows_sum_short(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                                const VT *__restrict__ V, uint64_t nrec, uint64_t rows_out,
                                                                double *__restrict__ out, const double *__restrict__ init,
                                                                const double *__restrict__ colw)
{
   constexpr int G = 64 / B;
   const int lane = threadIdx.x & 63, c = lane % B, g = lane / B;
   for (uint64_t r0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G; r0 < rows_out; r0 += (uint64_t)gridDim.x * 4 * G) {
      const uint64_t r = r0 + g;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      uint32_t p0 = 0, p1 = 0;
      if (r < nrec) {
         p0 = ptr[r];
         p1 = ptr[r + 1];
      }
      // the group's index list in batches of B (one coalesced read, an entry per lane, handed round by shuffles): the row reads of
      // a batch do not wait for any further index read.  (Groups of a wave may run a different number of batches: the shuffles
      // are executed by all lanes, the reads are predicated.)
      uint32_t longest = p1 - p0;
#pragma unroll
      for (int o = 32; o >= B; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
      for (uint32_t base = 0; base < longest; base += B) {
         const uint32_t t = p0 + base + c;
         const int cnt = (int)min((uint32_t)B, p1 - p0 > base ? p1 - p0 - base : 0u);
         const uint32_t mine = t < p1 ? idx[t] : 0u;
#pragma unroll
         for (int e = 0; e < B; e += 4) {
            const uint32_t s0 = (uint32_t)__shfl((int)mine, g * B + e), s1 = (uint32_t)__shfl((int)mine, g * B + e + 1),
                           s2 = (uint32_t)__shfl((int)mine, g * B + e + 2), s3 = (uint32_t)__shfl((int)mine, g * B + e + 3);
            if (e < cnt) a0 += (double)V[(uint64_t)s0 * B + c];
            if (e + 1 < cnt) a1 += (double)V[(uint64_t)s1 * B + c];
            if (e + 2 < cnt) a2 += (double)V[(uint64_t)s2 * B + c];
            if (e + 3 < cnt) a3 += (double)V[(uint64_t)s3 * B + c];
         }
      }
      if (r < rows_out) {
         double a = (a0 + a1) + (a2 + a3);
         if (colw) a *= colw[c] * 32.0;
         out[r * B + c] = init ? init[r * B + c] + a : a;
      }
   }
}

template <class VT>
static void sparse_rows_sum_t(const uint32_t *ptr, const uint32_t *idx, const VT *V, const double *rowscale, int b, uint64_t nrec,
                              uint64_t rows_out, double *out, hipStream_t stream, const double *init, bool short_lists, const double *colw, double avg_len)
{
   if (!rows_out) return;
   const unsigned blocks = (unsigned)std::min<uint64_t>(65536, (rows_out + 3) / 4);
   // measured (scripts/ab_gather.sh, cfg3): the batched kernel takes 0.3 ms off the K3 gather (short lists per sample),
   // nothing off the K2 one and costs it 6-50 us at the small sizes -- so K3 takes the batched kernel, K2 the plain one.
   // Both sit at ~7 TB/s out of the Infinity Cache; with the gathered matrix resident in L2 the same kernel reaches 9.4 TB/s
   // (scripts/gather_l2_probe.py), which is all an L2-blocked gather order could win.
   static const int forced = FPCA_TEST_ENV("FPCA_GATHER") ? atoi(FPCA_TEST_ENV("FPCA_GATHER")) : 0; // 1 / 2 force one kernel (A/B)
   // 3: several rows per wave, for lists of a dozen entries (measured on the 1/8 shard of cfg3, 12.5 entries per sample: see DESIGN 3c)
   const int variant = forced ? forced : (avg_len > 0 && avg_len <= 24.0 && b <= 32 && !rowscale) ? 3 : ((rowscale || short_lists) ? 2 : 1);
#define FPCA_GATHER_CASE(B_)                                                                                                    \
   case B_:                                                                                                                     \
      if (variant == 3 && B_ <= 32 && !rowscale) {                                                                              \
         const unsigned blocks3 = (unsigned)std::min<uint64_t>(65536, (rows_out + 4 * (64 / B_) - 1) / (4 * (64 / B_)));        \
         hipLaunchKernelGGL((k_sparse_rows_sum_short<(B_ <= 32 ? B_ : 32), VT>), dim3(blocks3), dim3(256), 0, stream, ptr, idx, V, nrec, rows_out, out, init, colw); \
      } else if (variant == 1)                                                                                                  \
         hipLaunchKernelGGL((k_sparse_rows_sum<B_, VT>), dim3(blocks), dim3(256), 0, stream, ptr, idx, V, rowscale, nrec, rows_out, out, init, colw); \
      else                                                                                                                      \
         hipLaunchKernelGGL((k_sparse_rows_sum_batched<B_, VT>), dim3(blocks), dim3(256), 0, stream, ptr, idx, V, rowscale, nrec, rows_out, out, init, colw); \
      break;
   switch (b) {
      FPCA_GATHER_CASE(16)
      FPCA_GATHER_CASE(32)
      FPCA_GATHER_CASE(64)
   default: throw Error(-1, "sparse_rows_sum: block width must be 16, 32 or 64");
   }
#undef FPCA_GATHER_CASE
   HIP_CHECK_LAUNCH();
}
void sparse_rows_sum(const uint32_t *ptr, const uint32_t *idx, const double *V, const double *rowscale, int b, uint64_t nrec,
                     uint64_t rows_out, double *out, hipStream_t stream, const double *init, bool short_lists, double avg_len)
{
   sparse_rows_sum_t<double>(ptr, idx, V, rowscale, b, nrec, rows_out, out, stream, init, short_lists, nullptr, avg_len);
}
void sparse_rows_sum_f32(const uint32_t *ptr, const uint32_t *idx, const float *V, const double *colw, int b, uint64_t nrec, uint64_t rows_out,
                         double *out, hipStream_t stream, const double *init, bool short_lists, double avg_len)
{
   sparse_rows_sum_t<float>(ptr, idx, V, nullptr, b, nrec, rows_out, out, stream, init, short_lists, colw, avg_len);
}

// ------------------------------------------------------------------------------------------------
// Helpers of the hybrid missing-indicator route (device_ctx.hip ensure_hybrid): the few SNPs whose missing calls are too many
// for the sparse gathers get their indicator matrix E on the matrix cores, as a compacted sub-matrix.
//   gather_packed_rows  dst[r] = src[idx[r]] (records of `pitch` bytes), rows r >= nidx filled with 0xff = "dosage 0, not missing"
//   patch_missing_rows  in the records idx[r] of `packed`: code 01 (missing) -> 11 (dosage 0): G.M is unchanged, E becomes 0 --
//                       the view of the matrix whose remaining missing calls the sparse lists hold
//   scatter_packed_rows packed[idx[r]] = src[r]  (puts the original records back)
//   gather_scaled_rows  dst[r][c] = V[idx[r]][c] * scale[idx[r]], rows >= nidx zero        (fp64, [.][b])
//   scatter_rows        dst[idx[r]][c] = src[r][c]
__global__ __launch_bounds__(256) void k_gather_packed_rows(const uint8_t *__restrict__ src, size_t pitch, const uint32_t *__restrict__ idx,
                                                             uint32_t nidx, uint8_t *__restrict__ dst)
{
   const u4 *s = blockIdx.x < nidx ? reinterpret_cast<const u4 *>(src + (size_t)idx[blockIdx.x] * pitch) : nullptr;
   u4 *d = reinterpret_cast<u4 *>(dst + (size_t)blockIdx.x * pitch);
   const u4 fill = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
   for (size_t i = threadIdx.x; i < pitch / 16; i += 256) d[i] = s ? s[i] : fill;
}
__global__ __launch_bounds__(256) void k_patch_missing_rows(uint8_t *__restrict__ packed, size_t pitch, const uint32_t *__restrict__ idx)
{
   u4 *row = reinterpret_cast<u4 *>(packed + (size_t)idx[blockIdx.x] * pitch);
   for (size_t i = threadIdx.x; i < pitch / 16; i += 256) {
      u4 x = row[i];
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] |= (x[k] & ~(x[k] >> 1) & 0x55555555u) << 1;
      row[i] = x;
   }
}
__global__ __launch_bounds__(256) void k_scatter_packed_rows(const uint8_t *__restrict__ src, size_t pitch, const uint32_t *__restrict__ idx,
                                                              uint8_t *__restrict__ packed)
{
   const u4 *s = reinterpret_cast<const u4 *>(src + (size_t)blockIdx.x * pitch);
   u4 *d = reinterpret_cast<u4 *>(packed + (size_t)idx[blockIdx.x] * pitch);
   for (size_t i = threadIdx.x; i < pitch / 16; i += 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_gather_scaled_rows(const double *__restrict__ V, const double *__restrict__ scale,
                                                             const uint32_t *__restrict__ idx, uint32_t nidx, uint64_t rows_out, int b,
                                                             double *__restrict__ dst)
{
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < rows_out * b; i += (uint64_t)gridDim.x * 256) {
      const uint64_t r = i / b;
      const int c = (int)(i % b);
      double v = 0.0;
      if (r < nidx) {
         const uint32_t j = idx[r];
         v = V[(uint64_t)j * b + c] * (scale ? scale[j] : 1.0);
      }
      dst[i] = v;
   }
}
__global__ __launch_bounds__(256) void k_scatter_rows(const double *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t nidx, int b,
                                                       double *__restrict__ dst)
{
   for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (uint64_t)nidx * b; i += (uint64_t)gridDim.x * 256) {
      const uint64_t r = i / b;
      dst[(uint64_t)idx[r] * b + (i % b)] = src[i];
   }
}
void gather_packed_rows(const uint8_t *src, size_t pitch, const uint32_t *idx, uint32_t nidx, uint32_t rows_out, uint8_t *dst, hipStream_t stream)
{
   if (!rows_out) return;
   hipLaunchKernelGGL(k_gather_packed_rows, dim3(rows_out), dim3(256), 0, stream, src, pitch, idx, nidx, dst);
   HIP_CHECK_LAUNCH();
}
void patch_missing_rows(uint8_t *packed, size_t pitch, const uint32_t *idx, uint32_t nidx, hipStream_t stream)
{
   if (!nidx) return;
   hipLaunchKernelGGL(k_patch_missing_rows, dim3(nidx), dim3(256), 0, stream, packed, pitch, idx);
   HIP_CHECK_LAUNCH();
}
void scatter_packed_rows(const uint8_t *src, size_t pitch, const uint32_t *idx, uint32_t nidx, uint8_t *packed, hipStream_t stream)
{
   if (!nidx) return;
   hipLaunchKernelGGL(k_scatter_packed_rows, dim3(nidx), dim3(256), 0, stream, src, pitch, idx, packed);
   HIP_CHECK_LAUNCH();
}
void gather_scaled_rows(const double *V, const double *scale, const uint32_t *idx, uint32_t nidx, uint64_t rows_out, int b, double *dst,
                        hipStream_t stream)
{
   if (!rows_out) return;
   const unsigned blocks = (unsigned)std::min<uint64_t>(4096, (rows_out * b + 255) / 256);
   hipLaunchKernelGGL(k_gather_scaled_rows, dim3(blocks), dim3(256), 0, stream, V, scale, idx, nidx, rows_out, b, dst);
   HIP_CHECK_LAUNCH();
}
void scatter_rows(const double *src, const uint32_t *idx, uint32_t nidx, int b, double *dst, hipStream_t stream)
{
   if (!nidx) return;
   const unsigned blocks = (unsigned)std::min<uint64_t>(4096, ((uint64_t)nidx * b + 255) / 256);
   hipLaunchKernelGGL(k_scatter_rows, dim3(blocks), dim3(256), 0, stream, src, idx, nidx, b, dst);
   HIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// sample-major copy of the packed stream for K3i: out[sample][snp/4] from in[snp][sample/4]  (2-bit transpose)
//   tile = 256 SNPs x 256 samples.  A thread loads the 16 x 16 block (16 SNP rows, one dword = 16 samples each; the 16
//   lanes of a sample-word run read 64 contiguous bytes of a row), transposes it in registers with the four
//   butterfly stages of a bit-matrix transpose on 2-bit elements, and the 256 blocks are regrouped through LDS so that
//   every thread stores 64 contiguous bytes (256 SNPs) of one sample row.
__global__ __launch_bounds__(256) void k_transpose_packed(const uint8_t *__restrict__ in, size_t pitch_in, uint8_t *__restrict__ out,
                                                           size_t pitch_out)
{
   __shared__ uint32_t tile[256][17];
   const uint64_t snp0 = (uint64_t)blockIdx.x * 256, smp0 = (uint64_t)blockIdx.y * 256;
   const int bs = threadIdx.x & 15, bp = threadIdx.x >> 4;
   uint32_t w[16];
#pragma unroll
   for (int i = 0; i < 16; i++) w[i] = *reinterpret_cast<const uint32_t *>(in + (snp0 + 16 * bp + i) * pitch_in + smp0 / 4 + 4 * bs);
   // w[i] holds (SNP i, samples 0..15); swap the off-diagonal sub-blocks at element distances 8, 4, 2, 1
#define FPCA_T2_STAGE(S_, MASK_)                                                    \
   _Pragma("unroll") for (int r = 0; r < 16; r++) if (!(r & S_))                    \
   {                                                                                \
      const uint32_t t = ((w[r] >> (2 * S_)) ^ w[r + S_]) & MASK_;                  \
      w[r + S_] ^= t;                                                               \
      w[r] ^= t << (2 * S_);                                                        \
   }
   FPCA_T2_STAGE(8, 0x0000FFFFu)
   FPCA_T2_STAGE(4, 0x00FF00FFu)
   FPCA_T2_STAGE(2, 0x0F0F0F0Fu)
   FPCA_T2_STAGE(1, 0x33333333u)
#undef FPCA_T2_STAGE
   // now w[j] holds (sample j, SNPs 0..15) of the block
#pragma unroll
   for (int j = 0; j < 16; j++) tile[16 * bs + j][bp] = w[j];
   __syncthreads();
   {
      const int r = threadIdx.x;
      u4 *dst = reinterpret_cast<u4 *>(out + (smp0 + r) * pitch_out + snp0 / 4);
#pragma unroll
      for (int q = 0; q < 4; q++) dst[q] = (u4){tile[r][4 * q], tile[r][4 * q + 1], tile[r][4 * q + 2], tile[r][4 * q + 3]};
   }
}

void transpose_packed(const uint8_t *in, size_t pitch_in, uint64_t N_pad, uint64_t P_pad, uint8_t *out, size_t pitch_out,
                      hipStream_t stream)
{
   if (N_pad % 256 || P_pad % 256) throw Error(-1, "transpose_packed: padded sizes must be multiples of 256");
   dim3 grid((unsigned)(P_pad / 256), (unsigned)(N_pad / 256));
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

// issue-rate ceiling of v_mfma_i32_32x32x32_i8: 8 independent accumulators per wave, explicit registers (the compiler
// must not reorder or shuffle them), operands = `fill` pattern (0: zeros, else pseudo-random bytes -- the rate is
// power-limited, so the operand toggling matters)
__global__ __launch_bounds__(256, 1) void k_mfma_i8_peak(int *out, int iters, uint32_t fill)
{
   const uint32_t l = threadIdx.x * 2654435761u;
   const uint32_t f0 = fill ? (fill ^ l) * 0x9E3779B1u : 0u, f1 = fill ? (f0 >> 3) * 0x85EBCA6Bu : 0u, f2 = fill ? (f1 >> 5) * 0xC2B2AE35u : 0u,
                  f3 = fill ? (f2 >> 7) * 0x27D4EB2Fu : 0u;
   asm volatile(
      "v_mov_b32 v0, 0\n\t"
      "v_mov_b32 v1, 0\n\t"
      "v_mov_b32 v2, 0\n\t"
      "v_mov_b32 v3, 0\n\t"
      "v_mov_b32 v4, 0\n\t"
      "v_mov_b32 v5, 0\n\t"
      "v_mov_b32 v6, 0\n\t"
      "v_mov_b32 v7, 0\n\t"
      "v_mov_b32 v8, 0\n\t"
      "v_mov_b32 v9, 0\n\t"
      "v_mov_b32 v10, 0\n\t"
      "v_mov_b32 v11, 0\n\t"
      "v_mov_b32 v12, 0\n\t"
      "v_mov_b32 v13, 0\n\t"
      "v_mov_b32 v14, 0\n\t"
      "v_mov_b32 v15, 0\n\t"
      "v_mov_b32 v16, 0\n\t"
      "v_mov_b32 v17, 0\n\t"
      "v_mov_b32 v18, 0\n\t"
      "v_mov_b32 v19, 0\n\t"
      "v_mov_b32 v20, 0\n\t"
      "v_mov_b32 v21, 0\n\t"
      "v_mov_b32 v22, 0\n\t"
      "v_mov_b32 v23, 0\n\t"
      "v_mov_b32 v24, 0\n\t"
      "v_mov_b32 v25, 0\n\t"
      "v_mov_b32 v26, 0\n\t"
      "v_mov_b32 v27, 0\n\t"
      "v_mov_b32 v28, 0\n\t"
      "v_mov_b32 v29, 0\n\t"
      "v_mov_b32 v30, 0\n\t"
      "v_mov_b32 v31, 0\n\t"
      "v_mov_b32 v32, 0\n\t"
      "v_mov_b32 v33, 0\n\t"
      "v_mov_b32 v34, 0\n\t"
      "v_mov_b32 v35, 0\n\t"
      "v_mov_b32 v36, 0\n\t"
      "v_mov_b32 v37, 0\n\t"
      "v_mov_b32 v38, 0\n\t"
      "v_mov_b32 v39, 0\n\t"
      "v_mov_b32 v40, 0\n\t"
      "v_mov_b32 v41, 0\n\t"
      "v_mov_b32 v42, 0\n\t"
      "v_mov_b32 v43, 0\n\t"
      "v_mov_b32 v44, 0\n\t"
      "v_mov_b32 v45, 0\n\t"
      "v_mov_b32 v46, 0\n\t"
      "v_mov_b32 v47, 0\n\t"
      "v_mov_b32 v48, 0\n\t"
      "v_mov_b32 v49, 0\n\t"
      "v_mov_b32 v50, 0\n\t"
      "v_mov_b32 v51, 0\n\t"
      "v_mov_b32 v52, 0\n\t"
      "v_mov_b32 v53, 0\n\t"
      "v_mov_b32 v54, 0\n\t"
      "v_mov_b32 v55, 0\n\t"
      "v_mov_b32 v56, 0\n\t"
      "v_mov_b32 v57, 0\n\t"
      "v_mov_b32 v58, 0\n\t"
      "v_mov_b32 v59, 0\n\t"
      "v_mov_b32 v60, 0\n\t"
      "v_mov_b32 v61, 0\n\t"
      "v_mov_b32 v62, 0\n\t"
      "v_mov_b32 v63, 0\n\t"
      "v_mov_b32 v64, 0\n\t"
      "v_mov_b32 v65, 0\n\t"
      "v_mov_b32 v66, 0\n\t"
      "v_mov_b32 v67, 0\n\t"
      "v_mov_b32 v68, 0\n\t"
      "v_mov_b32 v69, 0\n\t"
      "v_mov_b32 v70, 0\n\t"
      "v_mov_b32 v71, 0\n\t"
      "v_mov_b32 v72, 0\n\t"
      "v_mov_b32 v73, 0\n\t"
      "v_mov_b32 v74, 0\n\t"
      "v_mov_b32 v75, 0\n\t"
      "v_mov_b32 v76, 0\n\t"
      "v_mov_b32 v77, 0\n\t"
      "v_mov_b32 v78, 0\n\t"
      "v_mov_b32 v79, 0\n\t"
      "v_mov_b32 v80, 0\n\t"
      "v_mov_b32 v81, 0\n\t"
      "v_mov_b32 v82, 0\n\t"
      "v_mov_b32 v83, 0\n\t"
      "v_mov_b32 v84, 0\n\t"
      "v_mov_b32 v85, 0\n\t"
      "v_mov_b32 v86, 0\n\t"
      "v_mov_b32 v87, 0\n\t"
      "v_mov_b32 v88, 0\n\t"
      "v_mov_b32 v89, 0\n\t"
      "v_mov_b32 v90, 0\n\t"
      "v_mov_b32 v91, 0\n\t"
      "v_mov_b32 v92, 0\n\t"
      "v_mov_b32 v93, 0\n\t"
      "v_mov_b32 v94, 0\n\t"
      "v_mov_b32 v95, 0\n\t"
      "v_mov_b32 v96, 0\n\t"
      "v_mov_b32 v97, 0\n\t"
      "v_mov_b32 v98, 0\n\t"
      "v_mov_b32 v99, 0\n\t"
      "v_mov_b32 v100, 0\n\t"
      "v_mov_b32 v101, 0\n\t"
      "v_mov_b32 v102, 0\n\t"
      "v_mov_b32 v103, 0\n\t"
      "v_mov_b32 v104, 0\n\t"
      "v_mov_b32 v105, 0\n\t"
      "v_mov_b32 v106, 0\n\t"
      "v_mov_b32 v107, 0\n\t"
      "v_mov_b32 v108, 0\n\t"
      "v_mov_b32 v109, 0\n\t"
      "v_mov_b32 v110, 0\n\t"
      "v_mov_b32 v111, 0\n\t"
      "v_mov_b32 v112, 0\n\t"
      "v_mov_b32 v113, 0\n\t"
      "v_mov_b32 v114, 0\n\t"
      "v_mov_b32 v115, 0\n\t"
      "v_mov_b32 v116, 0\n\t"
      "v_mov_b32 v117, 0\n\t"
      "v_mov_b32 v118, 0\n\t"
      "v_mov_b32 v119, 0\n\t"
      "v_mov_b32 v120, 0\n\t"
      "v_mov_b32 v121, 0\n\t"
      "v_mov_b32 v122, 0\n\t"
      "v_mov_b32 v123, 0\n\t"
      "v_mov_b32 v124, 0\n\t"
      "v_mov_b32 v125, 0\n\t"
      "v_mov_b32 v126, 0\n\t"
      "v_mov_b32 v127, 0\n\t"
      ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
   asm volatile(
      "v_mov_b32 v128, %0\n\t"
      "v_mov_b32 v129, %1\n\t"
      "v_mov_b32 v130, %2\n\t"
      "v_mov_b32 v131, %3\n\t"
      "v_mov_b32 v132, %0\n\t"
      "v_mov_b32 v133, %1\n\t"
      "v_mov_b32 v134, %2\n\t"
      "v_mov_b32 v135, %3\n\t"
      "v_mov_b32 v136, %0\n\t"
      "v_mov_b32 v137, %1\n\t"
      "v_mov_b32 v138, %2\n\t"
      "v_mov_b32 v139, %3\n\t"
      "v_mov_b32 v140, %0\n\t"
      "v_mov_b32 v141, %1\n\t"
      "v_mov_b32 v142, %2\n\t"
      "v_mov_b32 v143, %3\n\t"
      :: "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
   for (int it = 0; it < iters; it++) {
      asm volatile(
         "v_mfma_i32_32x32x32_i8 v[0:15], v[128:131], v[136:139], v[0:15]\n\t"
         "v_mfma_i32_32x32x32_i8 v[16:31], v[132:135], v[136:139], v[16:31]\n\t"
         "v_mfma_i32_32x32x32_i8 v[32:47], v[128:131], v[140:143], v[32:47]\n\t"
         "v_mfma_i32_32x32x32_i8 v[48:63], v[132:135], v[140:143], v[48:63]\n\t"
         "v_mfma_i32_32x32x32_i8 v[64:79], v[128:131], v[136:139], v[64:79]\n\t"
         "v_mfma_i32_32x32x32_i8 v[80:95], v[132:135], v[136:139], v[80:95]\n\t"
         "v_mfma_i32_32x32x32_i8 v[96:111], v[128:131], v[140:143], v[96:111]\n\t"
         "v_mfma_i32_32x32x32_i8 v[112:127], v[132:135], v[140:143], v[112:127]\n\t"
         ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
   }
   int r;
   asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v0" : "=v"(r)::"v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
   if (r == 0x7fffffff) out[threadIdx.x] = r;
}

// diagnostic: what the 16-column remainder of the b = 16 GEMM costs on the matrix pipe.  One loop iteration = the MFMAs of a
// wave (64 rows) for two 32-k steps against S b = 112 slice-columns, random operands:
//   MIX = 0: 16 x v_mfma_i32_32x32x32_i8 (4 column tiles; in the 4th, columns 16..31 of the operand are zero padding)
//   MIX = 1: 12 x 32x32x32 (3 full tiles) + 4 x v_mfma_i32_16x16x64_i8 (the 16 remaining columns, both k-steps at once)
// Returns useful TOP/s (2 x 64 x 112 x 64 operations per iteration).
template <int MIX>
__global__ __launch_bounds__(256, 1) void k_mfma_i8_mix(int *out, int iters, uint32_t seed)
{
   const uint32_t l = threadIdx.x & 63;
   v4i a[4], b[4], bh[4];
   for (int t = 0; t < 4; t++)
      for (int q = 0; q < 4; q++) {
         uint32_t h = (seed + 0x9E3779B9u * (t * 4 + q + 1)) ^ (l * 0x85EBCA6Bu);
         h ^= h >> 15;
         h *= 0x2C1B3C6Du;
         h ^= h >> 12;
         a[t][q] = (int)(h & 0x02020202u) | (int)((h >> 8) & 0x01010101u); // genotype-like bytes 0..3
         b[t][q] = (int)(h * 0x9E3779B1u);                                // full-range bytes
         bh[t][q] = ((l & 31) >= 16) ? 0 : b[t][q];                       // a column tile whose upper 16 columns are padding
      }
   v16i acc[16];
   v4i acch[4];
   for (int t = 0; t < 16; t++)
      for (int r = 0; r < 16; r++) acc[t][r] = 0;
   for (int t = 0; t < 4; t++) acch[t] = (v4i){0, 0, 0, 0};
   for (int it = 0; it < iters; it++) {
      if (MIX == 0) {
#pragma unroll
         for (int t = 0; t < 16; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 3], (t & 3) == 3 ? bh[t >> 2] : b[t & 3], acc[t], 0, 0, 0);
      } else {
#pragma unroll
         for (int t = 0; t < 12; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 3], b[t % 3], acc[t], 0, 0, 0);
#pragma unroll
         for (int t = 0; t < 4; t++) acch[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], b[t], acch[t], 0, 0, 0);
      }
      asm volatile("" ::: "memory");
   }
   int r = 0;
   for (int t = 0; t < 16; t++) r ^= acc[t][t & 15];
   for (int t = 0; t < 4; t++) r ^= acch[t][t];
   if (r == 0x7fffffff) out[threadIdx.x] = r;
}

double mfma_i8_mix_tops(int mix, int iters, hipStream_t stream)
{
   int *d = nullptr;
   (void)hipMalloc(&d, 4096);
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   auto go = [&](int n) {
      if (mix)
         hipLaunchKernelGGL(k_mfma_i8_mix<1>, dim3(256), dim3(256), 0, stream, d, n, 0x1234567u);
      else
         hipLaunchKernelGGL(k_mfma_i8_mix<0>, dim3(256), dim3(256), 0, stream, d, n, 0x1234567u);
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
   return 256.0 * 4 * (double)iters * 2.0 * 64 * 112 * 64 / (ms * 1e-3) / 1e12;
}

double mfma_i8_peak_tops(int waves_per_simd, int iters, uint32_t fill, hipStream_t stream)
{
   int *d = nullptr;
   (void)hipMalloc(&d, 4096);
   const int blocks = 256 * waves_per_simd;
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   hipLaunchKernelGGL(k_mfma_i8_peak, dim3(blocks), dim3(256), 0, stream, d, iters / 10, fill);
   (void)hipEventRecord(e0, stream);
   hipLaunchKernelGGL(k_mfma_i8_peak, dim3(blocks), dim3(256), 0, stream, d, iters, fill);
   (void)hipEventRecord(e1, stream);
   (void)hipEventSynchronize(e1);
   float ms = 0;
   (void)hipEventElapsedTime(&ms, e0, e1);
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   (void)hipFree(d);
   const double ops = (double)blocks * 4 /*waves*/ * (double)iters * 8 * 65536.0;
   return ops / (ms * 1e-3) / 1e12;
}

} // namespace kern
} // namespace fpca
