// tile_layout_probe.hip -- would a TILED layout of the packed genotype matrix buy the int8 GEMMs anything?  (round-4 exploration,
// not product)
//   The GEMM workgroup reads, per K chunk, 64 bytes of each of its 256 records: 256 pieces of 64 B, one per record, a record
//   pitch apart (125 KB for the SNP-major stream at 500,000 samples, 25 KB for the sample-major copy at 100,000 SNPs).  A tiled
//   copy would make that 16 KB contiguous.  This probe moves the same bytes with the same workgroup shape and co-residency
//   (two workgroups per CU: 61 KB of LDS each) and no arithmetic, in both address patterns: what the memory system alone
//   delivers.  If the strided pattern already streams well above the ~3 TB/s the cheap (4-slice) GEMM reads at, the layout is
//   not what holds that kernel back.
// build: hipcc --offload-arch=gfx950 -O3 -o flashpca_amd/_build/tile_layout_probe scripts/tile_layout_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                          \
   do {                                                                                                                \
      hipError_t e_ = (x);                                                                                             \
      if (e_ != hipSuccess) {                                                                                          \
         std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                  \
         std::exit(1);                                                                                                 \
      }                                                                                                                \
   } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// one workgroup = one tile of 256 records, all K chunks in order; TILED: chunk c of tile t is the 16 KB at (t * chunks + c) * 16 KB
template <bool TILED>
__global__ __launch_bounds__(256) void k_stream(const uint8_t *__restrict__ base, size_t pitch, int chunks, uint32_t *__restrict__ sink)
{
   extern __shared__ uint8_t lds[]; // (only there to limit the co-residency to what the GEMM has)
   const int t = threadIdx.x;
   const size_t tile = blockIdx.x;
   u4 acc = {0u, 0u, 0u, 0u};
   for (int c = 0; c < chunks; c += 2) {
      u4 v[8];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
         for (int i = 0; i < 4; i++) {
            const int cc = c + u < chunks ? c + u : chunks - 1;
            const uint8_t *p = TILED ? base + ((tile * (size_t)chunks + cc) * 1024 + (size_t)(i * 256 + t)) * 16
                                     : base + (tile * 256 + (size_t)(t / 4 + 64 * i)) * pitch + (size_t)cc * 64 + (size_t)(t % 4) * 16;
            v[u * 4 + i] = *reinterpret_cast<const u4 *>(p);
         }
#pragma unroll
      for (int j = 0; j < 8; j++) acc ^= v[j];
   }
   if (lds[0] == 77) acc.x++; // (keeps the LDS allocation alive)
   if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

int main()
{
   struct Case {
      const char *name;
      size_t records, pitch;
   } cases[] = {{"K2 (SNP-major: 100,096 records of 125,056 B)", 100096, 125056}, {"K3 (sample-major: 500,224 records of 25,024 B)", 500224, 25024}};
   uint32_t *sink = nullptr;
   CK(hipMalloc(&sink, 4096 * sizeof(uint32_t)));
   for (const Case &cs : cases) {
      const size_t bytes = cs.records * cs.pitch;
      uint8_t *buf = nullptr;
      CK(hipMalloc(&buf, bytes + 65536));
      CK(hipMemset(buf, 0x5a, bytes + 65536));
      const int tiles = (int)(cs.records / 256), chunks = (int)(cs.pitch / 64);
      for (int tiled = 0; tiled < 2; tiled++)
         for (size_t lds : {(size_t)61 * 1024, (size_t)16 * 1024}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
               CK(hipEventRecord(e0));
               if (tiled)
                  hipLaunchKernelGGL(k_stream<true>, dim3(tiles), dim3(256), lds, 0, buf, cs.pitch, chunks, sink);
               else
                  hipLaunchKernelGGL(k_stream<false>, dim3(tiles), dim3(256), lds, 0, buf, cs.pitch, chunks, sink);
               CK(hipEventRecord(e1));
               CK(hipEventSynchronize(e1));
               float ms = 0;
               CK(hipEventElapsedTime(&ms, e0, e1));
               if (rep > 0 && ms < best) best = ms;
            }
            std::printf("%-50s %-8s %2zu KB LDS per workgroup: %7.3f ms  %6.2f TB/s\n", cs.name, tiled ? "tiled" : "strided", lds / 1024, best,
                        (double)tiles * chunks * 16384.0 / (best * 1e-3) / 1e12);
         }
      CK(hipFree(buf));
   }
   return 0;
}
