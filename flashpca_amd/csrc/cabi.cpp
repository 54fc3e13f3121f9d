// cabi.cpp -- the extern "C" entry points of include/fpca.h (libfpca.so): argument checks, error codes, and the two drivers
// fpca_pca (RandomPCA::pca_fast, randompca.cpp:168-218) and fpca_check (randompca.cpp:663-703).  Every function cites the
// reference interface it replaces in the header; the work is done in the translation units listed in ctx.hpp.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

#include "ctx.hpp"
#include "hip_backend.hpp"
#include "pca_driver.hpp"
#include "synth.hpp"

using namespace fpca;

namespace fpca {
const char *last_error_cstr();
}

// =====================================================================================================
extern "C" {

const char *fpca_last_error(void) { return fpca::last_error_cstr(); }
const char *fpca_version(void) { return FPCA_VERSION; }
int fpca_abi_version(void) { return FPCA_ABI_VERSION; }

int fpca_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return -1;
   return n;
}

int fpca_warmup(int device)
{
   return guarded([&] {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(FPCA_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
      if (device < 0 || device >= ndev) throw Error(FPCA_ENODEVICE, "device index out of range");
      HIP_CHECK(hipSetDevice(device));
      HIP_CHECK(hipFree(nullptr)); // creates the primary context
      void *p = nullptr;          // first allocation + first pinned allocation: the runtime sets its pools up here
      HIP_CHECK(hipMalloc(&p, 1 << 20));
      HIP_CHECK(hipFree(p));
      HIP_CHECK(hipHostMalloc(&p, 1 << 20, hipHostMallocDefault));
      HIP_CHECK(hipHostFree(p));
   });
}

int fpca_device_name(int device, char *buf, int buflen)
{
   return guarded([&] {
      hipDeviceProp_t prop;
      HIP_CHECK(hipGetDeviceProperties(&prop, device));
      std::snprintf(buf, buflen, "%s (%s, %d CUs, %.1f GiB)", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                    (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
   });
}

int fpca_create(fpca_ctx **out, const uint8_t *packed, uint64_t N, uint64_t P_g, int stand_method, int device, int accum)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!packed && P_g > 0) throw Error(FPCA_EINVAL, "packed is NULL");
      ctx_alloc_common(c, N, P_g, stand_method, device, accum);
      if (P_g > 0)
         HIP_CHECK(hipMemcpy2DAsync(c->d_packed, c->pitch, packed, c->np, c->np, P_g, hipMemcpyHostToDevice, c->stream));
      ctx_finish_upload(c);
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

namespace {
// One pread stream tops out near 10 GB/s from the page cache (it is a single-threaded memcpy), a fifth of what the
// PCIe link takes; 16 threads reading disjoint slices of the chunk reach 17-23 GB/s (measured on the overlay file system of the test box).
bool parallel_pread(int fd, uint8_t *buf, uint64_t want, off_t off)
{
   static const unsigned hw = usable_cpus(); // (the CPUs this process may run on at once, not the host's hardware threads)
   static const int cap = getenv("FPCA_READ_THREADS") ? std::max(1, atoi(getenv("FPCA_READ_THREADS"))) : 16;
   const int nt = want < (8u << 20) ? 1 : (int)std::min<unsigned>((unsigned)cap, hw ? hw : 1);
   std::atomic<bool> ok(true);
   auto work = [&](int t) {
      const uint64_t b0 = want * t / nt, b1 = want * (t + 1) / nt;
      uint64_t got = b0;
      while (got < b1) {
         const ssize_t k = pread(fd, buf + got, b1 - got, off + (off_t)got);
         if (k <= 0) {
            ok = false;
            return;
         }
         got += (uint64_t)k;
      }
   };
   if (nt == 1) {
      work(0);
      return ok;
   }
   std::vector<std::thread> th;
   for (int t = 1; t < nt; t++) th.emplace_back(work, t);
   work(0);
   for (auto &x : th) x.join();
   return ok;
}
} // namespace

int fpca_create_from_bed(fpca_ctx **out, const char *bed_path, uint64_t N, uint64_t snp_begin, uint64_t P_g,
                         int stand_method, int device, int accum, uint64_t *P_total)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int fd = -1;
   int rc = guarded([&] {
      if (N == 0) throw Error(FPCA_EINVAL, "N must be > 0");
      fd = open(bed_path, O_RDONLY);
      if (fd < 0) // data.cpp:156-161
         throw Error(FPCA_EIO, std::string("[Data::read_bed] Error reading file ") + bed_path + ", error " + strerror(errno));
      struct stat st;
      if (fstat(fd, &st) != 0 || st.st_size < 3) throw Error(FPCA_EIO, std::string("cannot stat ") + bed_path);
      // The reference skips the three header bytes unseen (data.cpp:218: seekg(3 + ...)); a sample-major file (third byte 0)
      // or something that is not a .bed at all would be decoded as garbage without a word.  Checked here.
      unsigned char magic[3] = {0, 0, 0};
      if (pread(fd, magic, 3, 0) != 3) throw Error(FPCA_EIO, std::string("[Data::read_bed] Error reading file ") + bed_path);
      if (magic[0] != 0x6c || magic[1] != 0x1b)
         throw Error(FPCA_EIO, std::string(bed_path) + " is not a PLINK .bed file (it does not start with the magic bytes 6c 1b)");
      if (magic[2] != 0x01)
         throw Error(FPCA_EIO, std::string(bed_path) + (magic[2] == 0x00 ? " is a sample-major .bed (header 6c 1b 00)" : " has an unknown .bed mode byte") +
                                   "; only SNP-major files (header 6c 1b 01) are supported -- convert with plink --make-bed");
      const uint64_t len = (uint64_t)st.st_size - 3; // data.cpp:165
      const uint64_t np = (N + 3) / 4;               // data.cpp:168
      const uint64_t nsnps = len / np;               // data.cpp:170 (integer division; .bim is not consulted)
      if (P_total) *P_total = nsnps;
      if (snp_begin > nsnps) throw Error(FPCA_EINVAL, "snp_begin beyond the end of the file");
      uint64_t pg = P_g ? P_g : nsnps - snp_begin;
      if (snp_begin + pg > nsnps) throw Error(FPCA_EINVAL, "SNP range beyond the end of the file");
      const bool timing = std::getenv("FPCA_TIMING") != nullptr;
      auto tl = std::chrono::steady_clock::now();
      auto lap = [&](const char *what) {
         const auto now = std::chrono::steady_clock::now();
         if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", what, std::chrono::duration<double>(now - tl).count() * 1e3);
         tl = now;
      };
      ctx_alloc_common(c, N, pg, stand_method, device, accum);
      lap("device init + allocations");
      c->P_total = nsnps;
      // stream the shard: contiguous byte range [3 + np*begin, 3 + np*(begin+pg)) of the file -> parallel pread into one of
      // two pinned bounce buffers -> 1-D H2D copy into a device staging buffer -> repitch kernel into the resident matrix;
      // the read of chunk i+1 overlaps the copy of chunk i
      const uint64_t rows_per_chunk = std::max<uint64_t>(1, (64ull << 20) / np);
      uint8_t *bounce[2] = {nullptr, nullptr}, *dstage[2] = {nullptr, nullptr};
      hipEvent_t done[2] = {nullptr, nullptr};
      auto cleanup = [&] {
         for (int i = 0; i < 2; i++) {
            if (bounce[i]) (void)hipHostFree(bounce[i]);
            if (dstage[i]) (void)hipFree(dstage[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
         }
      };
      try {
         int slot = 0;
         for (uint64_t r0 = 0; r0 < pg; r0 += rows_per_chunk, slot ^= 1) {
            const uint64_t nr = std::min(rows_per_chunk, pg - r0);
            if (!bounce[slot]) { // the second slot's 64 MB of pinned memory (14 ms to allocate) come while the first chunk is on the wire
               HIP_CHECK(hipHostMalloc(&bounce[slot], std::min(rows_per_chunk, pg) * np, hipHostMallocDefault));
               HIP_CHECK(hipMalloc(&dstage[slot], std::min(rows_per_chunk, pg) * np));
               HIP_CHECK(hipEventCreate(&done[slot]));
            }
            HIP_CHECK(hipEventSynchronize(done[slot])); // the previous copy out of this slot has finished
            const uint64_t want = nr * np;
            const off_t off = (off_t)(3 + np * (snp_begin + r0)); // data.cpp:218
            if (!parallel_pread(fd, bounce[slot], want, off)) throw Error(FPCA_EIO, std::string("short read from ") + bed_path);
            HIP_CHECK(hipMemcpyAsync(dstage[slot], bounce[slot], want, hipMemcpyHostToDevice, c->stream));
            kern::repitch(dstage[slot], np, nr, c->d_packed + r0 * c->pitch, c->pitch, c->stream);
            HIP_CHECK(hipEventRecord(done[slot], c->stream));
         }
         HIP_CHECK(hipStreamSynchronize(c->stream));
      } catch (...) {
         cleanup();
         throw;
      }
      cleanup();
      ctx_finish_upload(c);
      lap(".bed -> HBM");
   });
   if (fd >= 0) close(fd);
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

int fpca_create_synthetic(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed, int n_pop,
                          double fst, double missing_rate, int stand_method, int device, int accum)
{
   fpca_synth_model m;
   std::memset(&m, 0, sizeof(m));
   m.n_pop = n_pop;
   m.fst = fst;
   m.missing_rate = missing_rate;
   return fpca_create_synthetic_model(out, N, snp_begin, P_g, seed, &m, stand_method, device, accum);
}

int fpca_create_synthetic_model(fpca_ctx **out, uint64_t N, uint64_t snp_begin, uint64_t P_g, uint64_t seed, const fpca_synth_model *model,
                                int stand_method, int device, int accum)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!model) throw Error(FPCA_EINVAL, "model is NULL");
      const int n_pop = model->n_pop;
      const double fst = model->fst, missing_rate = model->missing_rate;
      if (n_pop < 1 || n_pop > synth::MAX_POP) throw Error(FPCA_EINVAL, "n_pop must be in 1..64");
      if (!(fst >= 0 && fst < 1) || !(missing_rate >= 0 && missing_rate < 1)) throw Error(FPCA_EINVAL, "fst / missing_rate out of range");
      if ((model->maf_model | 1) != 1 || model->missing_model < 0 || model->missing_model > 2 || !(model->conc_frac >= 0 && model->conc_frac <= 1))
         throw Error(FPCA_EINVAL, "maf_model must be 0 or 1, missing_model 0, 1 or 2, conc_frac in [0, 1]");
      if (model->missing_model == 2 && !(model->lognormal_sigma >= 0 && model->lognormal_sigma <= 4))
         throw Error(FPCA_EINVAL, "lognormal_sigma must be in [0, 4]");
      ctx_alloc_common(c, N, P_g, stand_method, device, accum);
      const uint32_t fst_fp = (uint32_t)std::llround(fst * 65536.0);
      const uint32_t miss_thr = (uint32_t)std::llround(missing_rate * 65536.0);
      // log-normal rates: the median that gives the asked-for mean, in 0.32 fixed point; sigma as a power of two (host libm only sets
      // these two integers: the matrix itself is integer arithmetic, identical on every host and device)
      const double sig = model->missing_model == 2 ? model->lognormal_sigma : 0.0;
      const uint64_t med_q32 = (uint64_t)std::llround(std::min(missing_rate * std::exp(-0.5 * sig * sig), 0.9) * 4294967296.0);
      const uint32_t sig2_fp = (uint32_t)std::llround(sig / std::log(2.0) * 65536.0);
      kern::synth_generate(c->d_packed, c->pitch, N, snp_begin, P_g, seed, n_pop, fst_fp, miss_thr, c->stream, model->maf_model, model->missing_model,
                           (uint32_t)std::llround(model->conc_frac * 65536.0), med_q32, sig2_fp);
      HIP_CHECK(hipStreamSynchronize(c->stream));
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

int fpca_create_dense(fpca_ctx **out, const double *X, int64_t ldx, uint64_t N, uint64_t P_g, int stand_method, int device)
{
   if (!out) return FPCA_EINVAL;
   *out = nullptr;
   fpca_ctx *c = new fpca_ctx();
   int rc = guarded([&] {
      if (!X || ldx < (int64_t)N || P_g == 0) throw Error(FPCA_EINVAL, "bad argument to fpca_create_dense");
      ctx_alloc_common(c, N, P_g, stand_method, device, FPCA_ACCUM_FP64, true);
      // column-major N x P on the host == row-major [P][N] : one strided copy into the padded [P_pad][N_pad] image
      HIP_CHECK(hipMemcpy2DAsync(c->d_Xd, c->N_pad * sizeof(double), X, (size_t)ldx * sizeof(double), N * sizeof(double), P_g,
                                 hipMemcpyHostToDevice, c->stream));
      kern::dense_standardise(c->d_Xd, c->N_pad, N, P_g, stand_method, c->d_mean, c->d_sd, c->d_sumsq, c->stream);
      std::vector<double> ss(P_g);
      HIP_CHECK(hipMemcpyAsync(ss.data(), c->d_sumsq, P_g * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_CHECK(hipStreamSynchronize(c->stream));
      double tot = 0;
      for (size_t i0 = 0; i0 < ss.size(); i0 += 1024) {
         double sblk = 0;
         for (size_t i = i0; i < std::min(ss.size(), i0 + 1024); i++) sblk += ss[i];
         tot += sblk;
      }
      c->trace_local = tot; // randompca.cpp:154: sum X^2 of the standardised matrix
      c->stats_done = true;
   });
   if (rc != FPCA_OK) {
      ctx_free(c);
      return rc;
   }
   *out = c;
   return FPCA_OK;
}

void fpca_destroy(fpca_ctx *ctx) { ctx_free(ctx); }

uint64_t fpca_nsamples(const fpca_ctx *ctx) { return ctx ? ctx->N : 0; }
uint64_t fpca_nsnps(const fpca_ctx *ctx) { return ctx ? ctx->P_g : 0; }
uint64_t fpca_block_rows(const fpca_ctx *ctx) { return ctx ? ctx->N_pad : 0; }
void *fpca_stream(fpca_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int fpca_synchronize(fpca_ctx *ctx)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      HIP_CHECK(hipDeviceSynchronize());
   });
}

int fpca_download_packed(fpca_ctx *ctx, uint8_t *out)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      if (ctx->dense) throw Error(FPCA_EINVAL, "this context holds a dense matrix, not a packed stream");
      if (ctx->P_g == 0) return;
      HIP_CHECK(hipMemcpy2D(out, ctx->np, ctx->d_packed, ctx->pitch, ctx->np, ctx->P_g, hipMemcpyDeviceToHost));
      // the pad bits of the last byte were rewritten to "missing" on upload; PLINK writes them as 0
      if (ctx->N % 4) {
         const uint8_t keep = (uint8_t)((1u << (2 * (ctx->N % 4))) - 1u);
         for (uint64_t j = 0; j < ctx->P_g; j++) out[j * ctx->np + ctx->np - 1] &= keep;
      }
   });
}

int fpca_stats(fpca_ctx *ctx, double *mean_sd, double *trace_out)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      if (mean_sd && ctx->P_g) {
         HIP_CHECK(hipMemcpy(mean_sd, ctx->d_mean, ctx->P_g * sizeof(double), hipMemcpyDeviceToHost));
         HIP_CHECK(hipMemcpy(mean_sd + ctx->P_g, ctx->d_sd, ctx->P_g * sizeof(double), hipMemcpyDeviceToHost));
      }
      if (trace_out) *trace_out = ctx->trace_local;
   });
}

int fpca_set_meansd(fpca_ctx *ctx, const double *mean_sd)
{
   return guarded([&] {
      HIP_CHECK(hipSetDevice(ctx->device));
      if (!mean_sd) throw Error(FPCA_EINVAL, "mean_sd is NULL");
      if (ctx->dense) throw Error(FPCA_EINVAL, "preloaded mean/sd applies to packed genotypes only");
      HIP_CHECK(hipMemcpy(ctx->d_mean, mean_sd, ctx->P_g * sizeof(double), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(ctx->d_sd, mean_sd + ctx->P_g, ctx->P_g * sizeof(double), hipMemcpyHostToDevice));
      kern::lut_from_meansd(ctx->d_mean, ctx->d_sd, ctx->P_g, ctx->d_lut, ctx->stream);
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      ctx->i8_scales_done = false;
      ctx->stats_done = true; // trace of the preloaded standardisation is not defined by the reference path
      ctx->trace_local = 0;
   });
}

int fpca_accum(const fpca_ctx *ctx) { return ctx ? ctx->accum : FPCA_EINVAL; }

int fpca_missing_mode(fpca_ctx *ctx, int b)
{
   int mode = -1;
   int rc = guarded([&] {
      if (!ctx || b < 1 || b > MAX_BLOCKVEC) throw Error(FPCA_EINVAL, "bad argument to fpca_missing_mode");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_stats(ctx);
      mode = ctx->i8_S ? i8_mode(ctx, (int)round_up((uint64_t)b, 16)) : -1;
   });
   return rc == FPCA_OK ? mode : rc;
}

int fpca_allreduce_chunks(fpca_ctx *ctx) { return ctx ? ar_chunks(ctx) : FPCA_EINVAL; }

// ---- operator, host pointers -------------------------------------------------------------------------
int fpca_apply_xxt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *Y, int64_t ldy)
{
   return guarded([&] {
      if (!ctx || !B || !Y || b < 1 || ldb < (int64_t)ctx->N || ldy < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xxt");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)ctx->N * nc);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->N * sizeof(double), B + (size_t)c0 * ldb, (size_t)ldb * sizeof(double),
                                    ctx->N * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_block(ctx->d_stage, ctx->N, ctx->N, ctx->N_pad, bw, nc, ctx->d_io_a, ctx->stream);
         apply_xxt_dev(ctx, ctx->d_io_a, bw, ctx->d_io_b, ctx->stream, nullptr);
         kern::block_to_colmajor(ctx->d_io_b, ctx->N, bw, nc, ctx->d_stage, ctx->N, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->N, nc, Y + (size_t)c0 * ldy, ldy, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_xt(fpca_ctx *ctx, const double *B, int64_t ldb, int b, double *T, int64_t ldt)
{
   return guarded([&] {
      if (!ctx || !B || !T || b < 1 || ldb < (int64_t)ctx->N || ldt < (int64_t)ctx->P_g) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xt");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * nc);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->N * sizeof(double), B + (size_t)c0 * ldb, (size_t)ldb * sizeof(double),
                                    ctx->N * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_block(ctx->d_stage, ctx->N, ctx->N, ctx->N_pad, bw, nc, ctx->d_io_a, ctx->stream);
         xt_dev(ctx, ctx->d_io_a, bw, ctx->stream);
         kern::t_to_colmajor(ctx->d_T, ctx->P_g, bw, nc, nullptr, ctx->d_stage, ctx->P_g, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->P_g, nc, T + (size_t)c0 * ldt, ldt, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_x(fpca_ctx *ctx, const double *T, int64_t ldt, int b, double *Y, int64_t ldy)
{
   return guarded([&] {
      if (!ctx || !T || !Y || b < 1 || ldt < (int64_t)ctx->P_g || ldy < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_x");
      HIP_CHECK(hipSetDevice(ctx->device));
      ensure_io(ctx);
      for (int c0 = 0; c0 < b; c0 += MAX_BLOCKVEC) {
         const int nc = std::min(MAX_BLOCKVEC, b - c0), bw = pad16(nc);
         ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * nc);
         ctx->ensure(ctx->d_T, ctx->T_cap, (size_t)ctx->P_pad * MAX_BLOCKVEC);
         HIP_CHECK(hipMemcpy2DAsync(ctx->d_stage, ctx->P_g * sizeof(double), T + (size_t)c0 * ldt, (size_t)ldt * sizeof(double),
                                    ctx->P_g * sizeof(double), nc, hipMemcpyHostToDevice, ctx->stream));
         kern::colmajor_to_t(ctx->d_stage, ctx->P_g, ctx->P_g, ctx->P_pad, bw, nc, ctx->d_T, ctx->stream);
         x_dev(ctx, bw, ctx->d_io_b, ctx->stream);
         kern::block_to_colmajor(ctx->d_io_b, ctx->N, bw, nc, ctx->d_stage, ctx->N, ctx->stream);
         staged_download(ctx, ctx->d_stage, ctx->N, nc, Y + (size_t)c0 * ldy, ldy, nullptr, 0, nullptr); // (synchronises)
      }
   });
}

int fpca_apply_xxt_dev(fpca_ctx *ctx, const double *dB, int b, double *dY, void *stream)
{
   return guarded([&] {
      if (!ctx || !dB || !dY) throw Error(FPCA_EINVAL, "bad argument to fpca_apply_xxt_dev");
      if (b != 16 && b != 32 && b != 48 && b != 64) throw Error(FPCA_EINVAL, "device blocks must be 16, 32, 48 or 64 wide");
      HIP_CHECK(hipSetDevice(ctx->device));
      hipEvent_t *ev = nullptr;
      if (ctx->prof_on && ctx->prof_calls++ % ctx->prof_stride == 0 && (size_t)(ctx->prof_used + 1) * 8 <= ctx->prof_ev.size())
         ev = &ctx->prof_ev[(size_t)ctx->prof_used++ * 8];
      apply_xxt_dev(ctx, dB, b, dY, stream ? (hipStream_t)stream : ctx->stream, ev);
   });
}

// ---- multi-GPU ---------------------------------------------------------------------------------------
int fpca_comm_unique_id(uint8_t id[FPCA_UNIQUE_ID_BYTES])
{
   return guarded([&] {
      static_assert(sizeof(ncclUniqueId) == FPCA_UNIQUE_ID_BYTES, "ncclUniqueId size");
      ncclUniqueId u;
      RCCL_CHECK(rccl().GetUniqueId(&u));
      std::memcpy(id, &u, sizeof(u));
   });
}

int fpca_comm_init_rank(fpca_ctx *ctx, int nranks, int rank, const uint8_t id[FPCA_UNIQUE_ID_BYTES])
{
   return guarded([&] {
      if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) throw Error(FPCA_EINVAL, "bad argument to fpca_comm_init_rank");
      HIP_CHECK(hipSetDevice(ctx->device));
      ncclUniqueId u;
      std::memcpy(&u, id, sizeof(u));
      RCCL_CHECK(rccl().CommInitRank(&ctx->comm, nranks, u, rank));
      ctx->nranks = nranks;
      ctx->rank = rank;
      ctx->rank_known = true;
      ctx->comm_dead = false;
      reset_exchange_state(ctx);
      ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0; // (the row chunks of the multi-rank K3 enter the workspace size)
      comm_streams(ctx);
      // self-test: sum of (rank+1) over ranks must be n(n+1)/2 on every rank
      double v = rank + 1.0;
      HIP_CHECK(hipMemcpyAsync(ctx->d_small, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ctx->allreduce(ctx->d_small, 1, ctx->stream);
      HIP_CHECK(hipMemcpyAsync(&v, ctx->d_small, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (v != nranks * (nranks + 1) / 2.0) throw Error(FPCA_ECOMM, "RCCL all-reduce self-test returned a wrong sum");
   });
}

int fpca_set_allreduce(fpca_ctx *ctx, fpca_allreduce_fn fn, void *user)
{
   if (!ctx) return FPCA_EINVAL;
   ctx->ar_fn = fn;
   ctx->ar_user = user;
   if (fn) ctx->comm_dead = false; // (a new transport)
   reset_exchange_state(ctx);
   return FPCA_OK;
}

int fpca_set_collectives(fpca_ctx *ctx, fpca_allgather_fn ag, fpca_reducescatter_fn rs, void *user)
{
   return guarded([&] {
      if (!ctx || (ag == nullptr) != (rs == nullptr)) throw Error(FPCA_EINVAL, "bad argument to fpca_set_collectives");
      HIP_CHECK(hipSetDevice(ctx->device));
      ctx->ag_fn = ag;
      ctx->rs_fn = rs;
      ctx->coll_user = user;
      reset_exchange_state(ctx);
      if (ag) comm_streams(ctx); // the stream on which a chunk's reduce-scatter runs while K3 computes the next chunk
      ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0;
   });
}

int fpca_set_rank(fpca_ctx *ctx, int nranks, int rank)
{
   if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return FPCA_EINVAL;
   ctx->nranks = nranks;
   ctx->rank = rank;
   ctx->rank_known = true;
   reset_exchange_state(ctx);
   ctx->i8_ws_for_S = ctx->i8_ws_for_b = 0;
   return FPCA_OK;
}

int fpca_collective_stats(const fpca_ctx *ctx, uint64_t *calls, uint64_t *bytes)
{
   if (!ctx) return FPCA_EINVAL;
   if (calls) *calls = ctx->coll_calls;
   if (bytes) *bytes = ctx->coll_bytes;
   return FPCA_OK;
}

int fpca_set_total_snps(fpca_ctx *ctx, uint64_t P_total)
{
   if (!ctx || P_total < ctx->P_g) return FPCA_EINVAL;
   ctx->P_total = P_total;
   return FPCA_OK;
}

// ---- driver -------------------------------------------------------------------------------------------
void fpca_pca_init_opts(fpca_pca_opts *o, size_t opts_size, size_t info_size)
{
   if (!o || opts_size < 2 * sizeof(uint32_t)) return;
   fpca_pca_opts d;
   std::memset(&d, 0, sizeof(d));
   d.ndim = 10;      // flashpca.cpp:325
   d.blockvec = 0;
   d.maxiter = 500;  // flashpca.cpp:426
   d.tol = 1e-6;     // flashpca.cpp:440
   d.divisor = FPCA_DIVISOR_P; // flashpca.cpp:484
   d.seed = 1;       // flashpca.cpp:276
   // the CALLER's sizes: a caller built against another revision of fpca.h is refused by fpca_pca instead of being overrun here
   d.struct_size = (uint32_t)opts_size;
   d.info_size = (uint32_t)info_size;
   std::memcpy(o, &d, std::min(opts_size, sizeof(d)));
}


int fpca_pca_row_ranges(fpca_ctx *ctx, const fpca_pca_opts *opts, uint64_t *ranges, int max_ranges)
{
   int n = 0;
   int rc = guarded([&] {
      if (!ctx || !opts || (max_ranges > 0 && !ranges)) throw Error(FPCA_EINVAL, "bad argument to fpca_pca_row_ranges");
      // the layout the LAST fpca_pca of this context ended on (a demotion changes it), else the one the options ask for
      const int path = ctx->last_solver_path;
      const bool repl = opts->replicated_solver != 0 || path == FPCA_SOLVER_REPLICATED_SELFTEST || path == FPCA_SOLVER_REPLICATED_FAILURE;
      const HipBackend::Layout lay = HipBackend::plan_layout(ctx, repl);
      const RowShard sh = HipBackend::plan_shard(ctx, lay);
      const std::vector<std::pair<uint64_t, uint64_t>> r = HipBackend::rows_of(ctx, sh);
      for (const auto &x : r) {
         if (n < max_ranges) {
            ranges[2 * n] = x.first;
            ranges[2 * n + 1] = x.second;
         }
         n++;
      }
   });
   return rc == FPCA_OK ? n : rc;
}

int fpca_pca(fpca_ctx *ctx, const fpca_pca_opts *opts, double *U, double *d, double *Px, double *pve, double *V,
             double *mean_sd, fpca_pca_info *info)
{
   int solver_rc = FPCA_OK;
   int rc = guarded([&] {
      if (!opts) throw Error(FPCA_EINVAL, "bad argument to fpca_pca");
      if (opts->struct_size != sizeof(fpca_pca_opts) || opts->info_size != sizeof(fpca_pca_info))
         throw Error(FPCA_EINVAL, "fpca_pca_opts carries the struct sizes " + std::to_string(opts->struct_size) + " / " + std::to_string(opts->info_size) +
                                      ", this library expects " + std::to_string(sizeof(fpca_pca_opts)) + " / " + std::to_string(sizeof(fpca_pca_info)) +
                                      ": the caller was built against another include/fpca.h (or did not use FPCA_PCA_OPTS_INIT)");
      if (!ctx) throw Error(FPCA_EINVAL, "bad argument to fpca_pca");
      HIP_CHECK(hipSetDevice(ctx->device));
      const int k = opts->ndim;
      // Spectra's requirement nev < ncv = 2 nev + 1 <= n, enforced by the reference CLI (flashpca.cpp:623-633)
      const uint64_t lim = std::min(ctx->N, ctx->P_total);
      const uint64_t max_dim = lim >= 1 ? (lim - 1) / 2 : 0;
      if (k < 1 || (uint64_t)k > max_dim)
         throw Error(FPCA_EINVAL, "You asked for " + std::to_string(k) + " dimensions, but only " + std::to_string(max_dim) + " allowed");
      const int b = choose_blockvec(k, opts->blockvec); // automatic: 16 (32 / 64 for ndim > 64 / > 128), never 48
      const bool timing = std::getenv("FPCA_TIMING") != nullptr;
      auto tp0 = std::chrono::steady_clock::now();
      auto lap = [&](const char *what) {
         const auto now = std::chrono::steady_clock::now();
         if (timing) std::fprintf(stderr, "[fpca] %-28s %8.3f ms\n", what, std::chrono::duration<double>(now - tp0).count() * 1e3);
         tp0 = now;
      };
      if (opts->cheap_slices != 0 && (opts->cheap_slices < 3 || opts->cheap_slices > 7)) throw Error(FPCA_EINVAL, "cheap_slices must be 0 or 3..7");
      const int cheap_S = opts->mixed < 0 ? 0 : (opts->cheap_slices ? opts->cheap_slices : 4);

      // ---- which solver layout (multi-GPU) ---------------------------------------------------------------------------------
      // Default with several ranks: row-sharded (all-gather -> K2, K3 -> reduce-scatter).  Its exchange checks itself the first
      // time a layout is used, and the verdict is made common to all ranks by one plain all-reduce of a flag: if ANY rank saw a
      // wrong answer or a transport error, EVERY rank continues with the replicated solver -- north_star's literal scheme, one
      // all-reduce of the N x b product per apply (svdwide.cpp:48-62 summed over ranks), nothing else on the wire.
      bool replicated = opts->replicated_solver != 0;
      int path = FPCA_SOLVER_SINGLE;
      std::string demoted_why;
      {
         HipBackend::Layout lay = HipBackend::plan_layout(ctx, replicated);
         if (lay == HipBackend::ROWSHARD) {
            const RowShard sh = HipBackend::plan_shard(ctx, lay);
            const long key = ((long)sh.G * 8 + sh.nch) * 128 + b;
            const bool test_it = sh.G > 1 || FPCA_TEST_ENV("FPCA_DEBUG_SELFTEST_FAIL") != nullptr;
            if (ctx->exchange_failed == key)
               replicated = true, path = ctx->exchange_failed_path;
            else if (test_it && ctx->exchange_tested != key) {
               const std::string why = exchange_selftest(ctx, sh, b);
               const double bad = agree_sum(ctx, why.empty() ? 0.0 : 1.0);
               if (bad > 0) {
                  ctx->exchange_failed = key;
                  ctx->exchange_failed_path = FPCA_SOLVER_REPLICATED_SELFTEST;
                  replicated = true;
                  path = FPCA_SOLVER_REPLICATED_SELFTEST;
                  demoted_why = why.empty() ? "another rank reported the failure" : why;
                  std::fprintf(stderr, "[fpca] rank %d: the self-test of the row-sharded exchange failed on %d of %d rank(s) (%s); every rank continues with the "
                                       "replicated solver (one all-reduce of the N x b product per block apply)\n",
                               ctx->rank, (int)bad, sh.G, demoted_why.c_str());
               } else
                  ctx->exchange_tested = key;
            }
            if (!replicated) path = FPCA_SOLVER_ROWSHARD;
         } else if (lay == HipBackend::REPLICATED)
            path = FPCA_SOLVER_REPLICATED;
      }
      lap("solver layout");

      PcaOutputs out;
      out.U = U;
      out.d = d;
      out.Px = Px;
      out.pve = pve;
      out.partial_rows = opts->partial_rows != 0;
      std::vector<double> dloc(k);
      if (!out.d) out.d = dloc.data();

      // one solve + the loadings on a given layout; FPCA_ECOMM out of the row-sharded one is caught by the caller below
      auto solve = [&](bool repl) {
         HipBackend be(ctx, b, repl, cheap_S);
         lap("backend setup");
         std::vector<int> ritz;
         double div = 1;
         solver_rc = run_pca(be, *opts, ctx->P_total, out, info, &ritz, &div);
         if (info) {
            info->cheap_slices = info->cheap_applies > 0 ? be.cheap_slices() : 0;
            info->seconds_exact = be.seconds_exact();
         }
         lap("run_pca");
         const auto tpost = std::chrono::steady_clock::now();
         if (opts->do_loadings && V) {
            // randompca.cpp:191-204: V[:, j] = X' u_j / sqrt(d_j) / sqrt(div); one K2 pass for all k columns
            // (b eigenvectors per Ritz block; ndim > b takes several)
            ctx->ensure(ctx->d_stage, ctx->stage_cap, (size_t)std::max(ctx->N, ctx->P_g) * std::min(k, b));
            std::vector<double> sc(b);
            for (int j0 = 0, q = 0; j0 < k; j0 += b, q++) {
               const int nc = std::min(b, k - j0);
               xt_dev(ctx, be.full_ptr(ritz[q]), b, ctx->stream); // (row-sharded solver: gathers the rows of the block from all ranks)
               std::fill(sc.begin(), sc.end(), 0.0);
               for (int j = 0; j < nc; j++) sc[j] = (1.0 / std::sqrt(out.d[j0 + j])) / std::sqrt(div);
               HIP_CHECK(hipMemcpyAsync(ctx->d_small, sc.data(), b * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
               kern::t_to_colmajor(ctx->d_T, ctx->P_g, b, nc, ctx->d_small, ctx->d_stage, ctx->P_g, ctx->stream);
               staged_download(ctx, ctx->d_stage, ctx->P_g, nc, V + (size_t)j0 * ctx->P_g, (int64_t)ctx->P_g, nullptr, 0, nullptr); // (synchronises:
                                                                                                  // sc / d_stage are reused by the next block)
            }
         }
         for (int h : ritz) be.free_block(h);
         if (info) {
            info->seconds_post = std::chrono::duration<double>(std::chrono::steady_clock::now() - tpost).count();
            info->seconds_total += info->seconds_post;
         }
      };

      if (path != FPCA_SOLVER_ROWSHARD)
         solve(replicated);
      else {
         // A collective of the row-sharded path that reports a failure (RCCL error return, a caller's transport returning
         // non-zero -- by contract on every rank, the collective having failed as a whole) ends that solve; the ranks agree on it
         // and start over on the replicated solver.  What CANNOT be caught this way is a rank that never returns from a collective
         // (a peer died): that is the launcher's business (flashpca --gpus: SIGCHLD / alarm, cli_main.cpp).
         std::string why;
         try {
            solve(false);
         } catch (const Error &e) {
            if (e.code != FPCA_ECOMM) throw;
            why = e.what();
            // (the collectives this rank enqueued before the failure may wait for peers that are elsewhere: bounded)
            if (!(bounded_sync(ctx, ctx->stream) && (!ctx->comm_stream || bounded_sync(ctx, ctx->comm_stream))))
               abandon_comm(ctx, why + "; the collectives enqueued before the failure never completed");
            (void)hipGetLastError();
         }
         const double bad = agree_sum(ctx, why.empty() ? 0.0 : 1.0);
         if (bad > 0) {
            path = FPCA_SOLVER_REPLICATED_FAILURE;
            demoted_why = why.empty() ? "another rank reported the failure" : why;
            std::fprintf(stderr, "[fpca] rank %d: a collective of the row-sharded solver failed on %d rank(s) (%s); every rank starts over with the replicated "
                                 "solver (one all-reduce of the N x b product per block apply)\n",
                         ctx->rank, (int)bad, demoted_why.c_str());
            const RowShard sh = HipBackend::plan_shard(ctx, HipBackend::ROWSHARD);
            ctx->exchange_failed = ((long)sh.G * 8 + sh.nch) * 128 + b; // (not tried again on this context)
            ctx->exchange_failed_path = FPCA_SOLVER_REPLICATED_FAILURE;
            // the unwound row-sharded backend has returned its slice-sized blocks to the pool, where the replicated retry (whole
            // blocks) cannot reuse them; at the largest inputs they are the memory the retry needs
            for (auto &pb : ctx->block_pool) (void)hipFree(pb.second);
            ctx->block_pool.clear();
            if (ctx->d_full_in) (void)hipFree(ctx->d_full_in);
            if (ctx->d_full_out) (void)hipFree(ctx->d_full_out);
            ctx->d_full_in = ctx->d_full_out = nullptr;
            ctx->full_in_cap = ctx->full_out_cap = 0;
            solve(true);
         }
      }
      ctx->last_solver_path = path;
      if (info) info->solver_path = path;
      if (mean_sd && ctx->P_g) {
         const auto tms = std::chrono::steady_clock::now();
         staged_download(ctx, ctx->d_mean, ctx->P_g, 1, mean_sd, (int64_t)ctx->P_g, nullptr, 0, nullptr);
         staged_download(ctx, ctx->d_sd, ctx->P_g, 1, mean_sd + ctx->P_g, (int64_t)ctx->P_g, nullptr, 0, nullptr);
         if (info) {
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - tms).count();
            info->seconds_post += t;
            info->seconds_total += t;
         }
      }
      lap("loadings, mean/sd");
   });
   if (std::getenv("FPCA_TIMING")) std::fprintf(stderr, "[fpca] %-28s (backend teardown follows)\n", "fpca_pca body done");
   if (rc != FPCA_OK) return rc;
   if (solver_rc == FPCA_ENOTCONVERGED) set_last_error("eigen-decomposition was not successful (not converged within maxiter)");
   return solver_rc;
}

int fpca_check(fpca_ctx *ctx, const double *evec, int64_t ldu, const double *eval, int k, int divisor, double *err,
               double *mse, double *rmse)
{
   return guarded([&] {
      if (!ctx || !evec || !eval || k < 1 || ldu < (int64_t)ctx->N) throw Error(FPCA_EINVAL, "bad argument to fpca_check");
      const uint64_t N = ctx->N;
      double div = 1; // randompca.cpp:676-680
      if (divisor == FPCA_DIVISOR_N1)
         div = (double)N - 1;
      else if (divisor == FPCA_DIVISOR_P)
         div = (double)ctx->P_total;
      std::vector<double> Y((size_t)N * k);
      int rc = fpca_apply_xxt(ctx, evec, ldu, k, Y.data(), (int64_t)N);
      if (rc != FPCA_OK) throw Error(rc, fpca_last_error());
      double tot = 0;
      for (int j = 0; j < k; j++) {
         double s = 0;
         for (uint64_t i = 0; i < N; i++) {
            const double e = Y[i + (size_t)j * N] / div - evec[i + (size_t)j * ldu] * eval[j];
            s += e * e;
         }
         if (err) err[j] = s;
         tot += s;
      }
      const double m = tot / ((double)N * k); // randompca.cpp:694
      if (mse) *mse = m;
      if (rmse) *rmse = std::sqrt(m);
   });
}

} // extern "C"
