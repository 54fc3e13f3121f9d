// mx_probe.hip -- is v_mfma_scale_f32_32x32x64_f8f6f4 usable for the exact-integer GEMMs?  (round-1 exploration, not product)
//   A = genotypes as fp4 (E2M1: dosage/2 in {0, 0.5, 1}), B = slices of the fp64 operand as fp6 (E2M3 sign-magnitude
//   digits t/8, |t| <= 15), scales = 1, fp32 accumulation: every product and partial sum is an integer multiple of 1/16,
//   so the result is exact while |sum * 16| < 2^24.
// 1. exactness: one wave accumulates K = 64 * steps with all-positive digits (sums near the 2^24 limit) and with random
//    signs, compared with integer arithmetic on the host, in the natural packing order (element i of a lane's 32 at bits
//    [4i, 4i+4) / [6i, 6i+6)).
// 2. rate: 1 wave per SIMD, 8 independent accumulators, realistic operands; TOP/s = 2*32*32*64 per instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o flashpca_amd/_build/mx_probe scripts/mx_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x)                                                                                                          \
   do {                                                                                                                \
      hipError_t e_ = (x);                                                                                             \
      if (e_ != hipSuccess) {                                                                                          \
         std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                  \
         std::exit(1);                                                                                                 \
      }                                                                                                                \
   } while (0)

constexpr int FMT_FP6 = 2, FMT_FP4 = 4;
constexpr int SCALE_ONE = 0x7F7F7F7F; // E8M0 127 = 2^0 in every byte

// A: [steps][64 lanes][4 dwords]  B: [steps][64 lanes][6 dwords]  D: [64 lanes][16]
__global__ void k_exact(const uint32_t *A, const uint32_t *B, float *D, int steps)
{
   const int l = threadIdx.x;
   v16f acc = {0};
   for (int s = 0; s < steps; s++) {
      v8i a = {0}, b = {0};
      for (int i = 0; i < 4; i++) a[i] = (int)A[((size_t)s * 64 + l) * 4 + i];
      for (int i = 0; i < 6; i++) b[i] = (int)B[((size_t)s * 64 + l) * 6 + i];
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT_FP4, FMT_FP6, 0, SCALE_ONE, 0, SCALE_ONE);
   }
   for (int r = 0; r < 16; r++) D[l * 16 + r] = acc[r];
}

template <int FA, int FB> __global__ __launch_bounds__(256, 1) void k_rate(float *out, int iters, uint32_t seed, int realistic)
{
   uint32_t x = (threadIdx.x + 1) * 2654435761u ^ seed;
   auto rnd = [&]() {
      x ^= x << 13;
      x ^= x >> 17;
      x ^= x << 5;
      return x;
   };
   v8i a[2], b[4];
   for (int j = 0; j < 2; j++)
      for (int i = 0; i < 8; i++) {
         uint32_t w = rnd();
         if (realistic && FA == FMT_FP4) w = (w & (w >> 1) & 0x11111111u) | ((w >> 2) & ~w & 0x22222222u); // nibbles in {0,1,2}
         a[j][i] = (int)w;
      }
   for (int j = 0; j < 4; j++)
      for (int i = 0; i < 8; i++) b[j][i] = (int)rnd();
   v16f acc[8];
   for (int j = 0; j < 8; j++) acc[j] = (v16f){0};
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 8; j++)
         acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j & 1], b[j >> 1], acc[j], FA, FB, 0, SCALE_ONE, 0, SCALE_ONE);
   }
   float s = 0;
   for (int j = 0; j < 8; j++)
      for (int r = 0; r < 16; r++) s += acc[j][r];
   if (s == 12345.678f) out[0] = s;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void k_rate_i8(int *out, int iters, uint32_t seed, int realistic)
{
   uint32_t x = (threadIdx.x + 1) * 2654435761u ^ seed;
   auto rnd = [&]() {
      x ^= x << 13;
      x ^= x >> 17;
      x ^= x << 5;
      return x;
   };
   v4i a[2], b[4];
   for (int j = 0; j < 2; j++)
      for (int i = 0; i < 4; i++) {
         uint32_t w = rnd();
         if (realistic) w = (w & (w >> 1) & 0x01010101u) | ((w >> 2) & ~w & 0x02020202u); // bytes in {0,1,2}
         a[j][i] = (int)w;
      }
   for (int j = 0; j < 4; j++)
      for (int i = 0; i < 4; i++) b[j][i] = (int)(rnd() & 0x7F7F7F7Fu) - 0x40404040;
   v16i acc[8];
   for (int j = 0; j < 8; j++) acc[j] = (v16i){0};
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j & 1], b[j >> 1], acc[j], 0, 0, 0);
   }
   int s = 0;
   for (int j = 0; j < 8; j++)
      for (int r = 0; r < 16; r++) s += acc[j][r];
   if (s == 123456789) out[0] = s;
}

typedef int v4iacc __attribute__((ext_vector_type(4)));
// the 16x16x64 shape of the same int8 MFMA: a quarter of the accumulator traffic per instruction, half per op
__global__ __launch_bounds__(256, 1) void k_rate_i8_16(int *out, int iters, uint32_t seed, int realistic)
{
   uint32_t x = (threadIdx.x + 1) * 2654435761u ^ seed;
   auto rnd = [&]() {
      x ^= x << 13;
      x ^= x >> 17;
      x ^= x << 5;
      return x;
   };
   v4i a[4], b[4];
   for (int j = 0; j < 4; j++)
      for (int i = 0; i < 4; i++) {
         uint32_t w = rnd();
         if (realistic) w = (w & (w >> 1) & 0x01010101u) | ((w >> 2) & ~w & 0x02020202u);
         a[j][i] = (int)w;
         b[j][i] = (int)(rnd() & 0x7F7F7F7Fu) - 0x40404040;
      }
   v4iacc acc[16];
   for (int j = 0; j < 16; j++) acc[j] = (v4iacc){0, 0, 0, 0};
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 16; j++) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
   }
   int s = 0;
   for (int j = 0; j < 16; j++)
      for (int r = 0; r < 4; r++) s += acc[j][r];
   if (s == 123456789) out[0] = s;
}

template <typename K, typename... Args> static double time_ms(K kern, int blocks, Args... args)
{
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0));
   CK(hipEventCreate(&e1));
   CK(hipEventRecord(e0, 0));
   hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
   CK(hipEventRecord(e1, 0));
   CK(hipEventSynchronize(e1));
   float ms = 0;
   CK(hipEventElapsedTime(&ms, e0, e1));
   return ms;
}

int main()
{
   // ---- 1. exactness --------------------------------------------------------------------------------------------
   for (int mode = 0; mode < 2; mode++) {
      const int steps = mode == 0 ? 8192 : 4096; // K = 524288 / 262144
      std::vector<uint32_t> A((size_t)steps * 64 * 4, 0), B((size_t)steps * 64 * 6, 0);
      std::vector<int8_t> ga((size_t)steps * 64 * 32), tb((size_t)steps * 64 * 32); // [step][lane][32]
      uint64_t st = 88172645463325252ull + mode;
      auto rnd = [&]() {
         st ^= st << 13;
         st ^= st >> 7;
         st ^= st << 17;
         return (uint32_t)(st >> 11);
      };
      for (size_t i = 0; i < ga.size(); i++) {
         const uint32_t r = rnd();
         ga[i] = (int8_t)(r % 3);                                            // dosage 0,1,2 -> fp4 code 0, 1 (0.5), 2 (1.0)
         int t = mode == 0 ? 8 + (int)((r >> 8) % 8) : (int)((r >> 8) % 31) - 15; // digits: 8..15 / -15..15
         tb[i] = (int8_t)t;
      }
      for (size_t sl = 0; sl < (size_t)steps * 64; sl++)
         for (int i = 0; i < 32; i++) {
            const uint32_t ca = (uint32_t)ga[sl * 32 + i];
            A[sl * 4 + (4 * i) / 32] |= ca << ((4 * i) % 32);
            const int t = tb[sl * 32 + i];
            const uint32_t cb = (t < 0 ? 32u : 0u) | (uint32_t)std::abs(t);
            const int bit = 6 * i;
            B[sl * 6 + bit / 32] |= cb << (bit % 32);
            if (bit % 32 > 26) B[sl * 6 + bit / 32 + 1] |= cb >> (32 - bit % 32);
         }
      uint32_t *dA, *dB;
      float *dD;
      CK(hipMalloc(&dA, A.size() * 4));
      CK(hipMalloc(&dB, B.size() * 4));
      CK(hipMalloc(&dD, 64 * 16 * 4));
      CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
      std::vector<float> D(64 * 16);
      CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
      // host: A row m = lane & 31 holds K-half lane >> 5; B column n likewise; D[row][col]: col = lane & 31,
      // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
      long bad = 0;
      double maxabs = 0;
      for (int l = 0; l < 64; l++)
         for (int r = 0; r < 16; r++) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            long sum = 0;
            for (int s = 0; s < steps; s++)
               for (int h = 0; h < 2; h++) {
                  const int8_t *pa = &ga[((size_t)s * 64 + h * 32 + row) * 32], *pb = &tb[((size_t)s * 64 + h * 32 + col) * 32];
                  for (int i = 0; i < 32; i++) sum += (long)pa[i] * pb[i];
               }
            const double want = (double)sum / 16.0;
            if ((double)D[l * 16 + r] != want) bad++;
            if (std::abs(want * 16) > maxabs) maxabs = std::abs(want * 16);
         }
      std::printf("exactness mode %d: K = %d, max |integer sum| = %.0f (2^24 = 16777216), mismatches %ld / 1024\n", mode, steps * 64,
                  maxabs, bad);
      CK(hipFree(dA));
      CK(hipFree(dB));
      CK(hipFree(dD));
   }
   // ---- 2. rates ---------------------------------------------------------------------------------------------------
   float *d;
   CK(hipMalloc(&d, 64));
   const int blocks = 256, iters = 200000;
   for (int realistic = 0; realistic < 2; realistic++) {
      (void)time_ms(k_rate<FMT_FP4, FMT_FP6>, blocks, d, iters / 10, 1u, realistic);
      const double m46 = time_ms(k_rate<FMT_FP4, FMT_FP6>, blocks, d, iters, 1u, realistic);
      const double m44 = time_ms(k_rate<FMT_FP4, FMT_FP4>, blocks, d, iters, 1u, realistic);
      const double m66 = time_ms(k_rate<FMT_FP6, FMT_FP6>, blocks, d, iters, 1u, realistic);
      const double m48 = time_ms(k_rate<FMT_FP4, 0>, blocks, d, iters, 1u, realistic);
      const double mi8 = time_ms(k_rate_i8, blocks, (int *)d, iters, 1u, realistic);
      const double mi16 = time_ms(k_rate_i8_16, blocks, (int *)d, iters, 1u, realistic);
      const double ops = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 64;
      std::printf("%s A operand: fp4 x fp6 %.0f TOP/s | fp4 x fp4 %.0f | fp6 x fp6 %.0f | fp4 x fp8 %.0f | i8 32x32x32 %.0f TOP/s\n",
                  realistic ? "genotype-like" : "random", ops / (m46 * 1e-3) / 1e12, ops / (m44 * 1e-3) / 1e12, ops / (m66 * 1e-3) / 1e12,
                  ops / (m48 * 1e-3) / 1e12, ops / 2 / (mi8 * 1e-3) / 1e12);
      std::printf("   i8 16x16x64 %.0f TOP/s\n", (double)blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 64 / (mi16 * 1e-3) / 1e12);
   }
   return 0;
}
