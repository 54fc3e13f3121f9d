// synth.hpp -- deterministic synthetic genotype model shared by host and device code.
//
// SURVEY.md section 8(d) asks for a structured (Balding-Nichols-like) population model so that the top-k
// eigenvalues separate from the bulk, generated with a counter-based RNG so that any SNP range can be
// produced independently on any rank.  Everything here is INTEGER arithmetic (16.16 fixed point), so the
// host and the GPU produce bit-identical matrices with no dependence on libm or FMA contraction:
//
//   populations      n_pop sub-populations with unequal sizes: weight of population c is (c+1), samples
//                    are assigned contiguously (boundaries[c] = N * c(c+1)/2 / (n_pop(n_pop+1)/2))
//   ancestral freq   p_j  = 0.05 + 0.90 * u16(j)                         (16.16 fixed point)
//   population freq  p_jc = clamp(p_j + sqrt(fst p_j (1-p_j)) * z_jc, 0.01, 0.99), z_jc = Irwin-Hall(12)
//                    sum of twelve 16-bit uniforms, recentred (mean 0, variance 1): the normal approximation
//                    of the Balding-Nichols Beta with the same mean and variance
//   genotype         g = [u1 < p_jc] + [u2 < p_jc]  (dosage of allele A1), missing when u3 < missing_rate
//   PLINK code       g=2 -> 00, g=1 -> 10, g=0 -> 11, missing -> 01; sample 4i+s in bits 2s..2s+1 of byte i
//   RNG              64-bit finaliser hash of the counter (stream, snp, index) xor seed: 4 x 16-bit uniforms
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FPCA_HD __host__ __device__ inline
#else
#define FPCA_HD inline
#endif

namespace fpca {
namespace synth {

constexpr int MAX_POP = 64;

FPCA_HD uint64_t mix64(uint64_t x)
{
   x ^= x >> 30;
   x *= 0xBF58476D1CE4E5B9ull;
   x ^= x >> 27;
   x *= 0x94D049BB133111EBull;
   x ^= x >> 31;
   return x;
}

// counter -> 64 random bits; stream separates the three uses (snp freq, population freq, genotype)
FPCA_HD uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t snp, uint64_t idx)
{
   uint64_t x = seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull;
   x = mix64(x ^ (snp * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull));
   return mix64(x ^ (idx * 0x9FB21C651E98DF25ull + 0x2545F4914F6CDD1Dull));
}

FPCA_HD uint32_t isqrt64(uint64_t v)
{
   // integer square root (floor), bit-by-bit; v < 2^48 here
   uint64_t r = 0, bit = 1ull << 46;
   while (bit > v) bit >>= 2;
   while (bit) {
      if (v >= r + bit) {
         v -= r + bit;
         r = (r >> 1) + bit;
      } else
         r >>= 1;
      bit >>= 2;
   }
   return (uint32_t)r;
}

// ancestral allele frequency of SNP j in 16.16 fixed point, in [0.05, 0.95)
FPCA_HD uint32_t snp_freq(uint64_t seed, uint64_t snp)
{
   uint32_t u = (uint32_t)(rnd64(seed, 1, snp, 0) >> 48); // 16 bits
   return 3277u + ((u * 58982u) >> 16);
}

// ---- "realistic" profile (round 4): what array / sequencing genotypes look like and the uniform model above does not ------
//   allele frequencies  a rare-variant spectrum: minor-allele frequency 0.001 + 0.499 u^3 (u uniform): half of the SNPs below
//                       6 %, an eighth below 0.2 % -- per-SNP sd = sqrt(2p(1-p)) spans 0.045 .. 0.71 (16x; the uniform
//                       model's 0.31 .. 0.71 is 2.3x), SNPs that are monomorphic in a small sample occur (sd <= 1e-9 -> zero
//                       column, data.cpp:299-320)
//   missing calls       concentrated in few SNPs, as failed assays are: a fraction conc_frac of the SNPs (default 5 %) lose
//                       10-30 % of their calls, every other SNP at most 0.1 % -- ~1 % overall, but 95 % of the SNPs are
//                       (nearly) complete
FPCA_HD uint32_t snp_freq_rare(uint64_t seed, uint64_t snp)
{
   const uint64_t u = rnd64(seed, 1, snp, 0) >> 48; // 16 bits
   const uint64_t u3 = (((u * u) >> 16) * u) >> 16;  // u^3 in 0.16
   return 66u + (uint32_t)((u3 * 32702u) >> 16);     // 16.16: [0.001, 0.5)
}
// per-SNP missing-call threshold (16-bit uniform < thr) of the concentrated model; conc_fp = fraction of affected SNPs in 0.16
FPCA_HD uint32_t snp_miss_thr_concentrated(uint64_t seed, uint64_t snp, uint32_t conc_fp)
{
   const uint64_t h = rnd64(seed, 4, snp, 0);
   const uint32_t pick = (uint32_t)(h & 0xFFFF), lvl = (uint32_t)((h >> 16) & 0xFFFF);
   if (pick < conc_fp) return 6554u + ((lvl * 13107u) >> 16); // 10 % .. 30 %
   return (lvl * 66u) >> 16;                                  // 0 .. 0.1 %
}

// missing_model 2 (round 5): per-SNP missing-call rates from a LOG-NORMAL distribution -- where most real arrays sit, between "every
// SNP the same rate" (model 0) and "5 % of the SNPs hold nearly all of it" (model 1): rate_j = median * 2^(sig2 z_j), z_j the
// Irwin-Hall normal of pop_freq, capped at 0.9.  med_q32 = median rate in 0.32 fixed point, sig2_fp = sigma / ln 2 in 16.16;
// 2^f for the fractional part by a cubic whose coefficients sum to one (2^0 = 1 and 2^1 = 2 exactly, 2e-4 relative in between).
FPCA_HD uint32_t snp_miss_thr_lognormal(uint64_t seed, uint64_t snp, uint64_t med_q32, uint32_t sig2_fp)
{
   int64_t zsum = 0;
   for (int r = 0; r < 3; r++) {
      const uint64_t h = rnd64(seed, 5, snp, (uint64_t)r);
      zsum += (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
   }
   const int64_t z = zsum - 6 * 65535;               // mean 0, std 1.0 in 16.16
   const int64_t x = ((int64_t)sig2_fp * z) >> 16;   // exponent of 2 in 16.16 (arithmetic shift: floor)
   const int64_t e = x >> 16;
   const uint64_t f = (uint64_t)(x & 0xFFFF);
   const uint64_t m = 65536u + ((f * (45426u + ((f * (15743u + ((f * 4367u) >> 16))) >> 16))) >> 16); // 2^f in 16.16
   uint64_t rate = (med_q32 * m) >> 16;              // 0.32
   if (e >= 0)
      rate = e >= 20 ? ~0ull : rate << e;
   else
      rate = e <= -40 ? 0 : rate >> (-e);
   const uint64_t thr = rate >> 16;                  // 16-bit uniform < thr
   return (uint32_t)(thr > 58982u ? 58982u : thr);   // at most 90 %
}

// frequency of SNP j in population c (16.16 fixed point threshold for 16-bit uniforms)
FPCA_HD uint32_t pop_freq(uint64_t seed, uint64_t snp, uint32_t pj, uint32_t fst_fp /*16.16*/, int c)
{
   // sd = sqrt(fst * p (1-p)) in 16.16: sqrt( fst_fp * pj * (65536-pj) / 65536 ) has 16 fractional bits
   uint64_t var = ((uint64_t)fst_fp * pj * (65536u - pj)) >> 16; // 32 fractional bits
   uint32_t sd = isqrt64(var);                                    // 16 fractional bits
   int64_t zsum = 0;
   for (int r = 0; r < 3; r++) {
      uint64_t h = rnd64(seed, 2, snp, (uint64_t)c * 4 + r);
      zsum += (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
   }
   int64_t z = zsum - 6 * 65535; // mean 0, std 65536 (i.e. 1.0 in 16.16)
   int64_t p = (int64_t)pj + (((int64_t)sd * z) >> 16);
   const int64_t lo = pj < 3277u ? 7 : 655; // (rare-variant spectrum: down to 0.0001 instead of 0.01)
   if (p < lo) p = lo;
   if (p > 64880) p = 64880;
   return (uint32_t)p;
}

// first sample of population c (c = 0..n_pop); weight of population c is (c+1)
FPCA_HD uint64_t pop_boundary(uint64_t N, int n_pop, int c)
{
   uint64_t tot = (uint64_t)n_pop * (n_pop + 1) / 2;
   uint64_t cum = (uint64_t)c * (c + 1) / 2;
   return (N * cum) / tot; // N < 2^40, cum <= 2080: no overflow
}

// PLINK 2-bit code of (snp, sample) given the population threshold
FPCA_HD uint32_t cell_code(uint64_t seed, uint64_t snp, uint64_t sample, uint32_t thr, uint32_t miss_thr)
{
   uint64_t h = rnd64(seed, 3, snp, sample);
   uint32_t u1 = (uint32_t)(h & 0xFFFF), u2 = (uint32_t)((h >> 16) & 0xFFFF), u3 = (uint32_t)((h >> 32) & 0xFFFF);
   if (u3 < miss_thr) return 1u; // 01 = missing
   uint32_t g = (u1 < thr) + (u2 < thr);
   return g == 2 ? 0u : (g == 1 ? 2u : 3u);
}

} // namespace synth
} // namespace fpca
