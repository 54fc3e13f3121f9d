// context.hip -- life cycle of a device context (include/fpca.h "Data"): HBM allocation of the resident packed matrix and
// its per-SNP tables, K1 statistics, teardown.  Replaces Data::get_size / prepare (data.cpp:150-206) and the first-visit branch
// of Data::read_snp_block (data.cpp:257-322).
#include <algorithm>
#include <chrono>
#include <cmath>

#include "ctx.hpp"

namespace fpca {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const char *last_error_cstr() { return g_last_error.c_str(); }

} // namespace fpca

using namespace fpca;

void fpca_ctx::ensure(double *&p, size_t &cap, size_t need)
{
   if (need <= cap) return;
   if (p) HIP_CHECK(hipFree(p));
   p = nullptr;
   cap = 0;
   HIP_CHECK(hipMalloc(&p, need * sizeof(double)));
   cap = need;
}

namespace fpca {

void ctx_alloc_common(fpca_ctx *c, uint64_t N, uint64_t P_g, int stand, int device, int accum, bool dense)
{
   if (N == 0) throw Error(FPCA_EINVAL, "N must be > 0");
   if (!dense && stand != FPCA_STANDARDISE_BINOM && stand != FPCA_STANDARDISE_BINOM2)
      throw Error(FPCA_EINVAL, "unknown standardisation method: " + std::to_string(stand)); // data.cpp:283-288
   if (dense && (stand < FPCA_STANDARDISE_NONE || stand > FPCA_STANDARDISE_CENTER))
      throw Error(FPCA_EINVAL, "unknown standardization method"); // util.cpp:183
   if (accum == FPCA_ACCUM_AUTO) {
      // the exact-integer path when it applies (2-bit input, int32-safe sizes), else the fp64 MFMA path
      const bool fits = !dense && std::max<uint64_t>(N, P_g) <= (uint64_t)8000000;
      accum = fits ? FPCA_ACCUM_I8(7) : FPCA_ACCUM_FP64;
      c->i8_auto = fits;
   }
   const bool i8 = accum >= FPCA_ACCUM_I8(2) && accum <= FPCA_ACCUM_I8(8);
   if (accum != FPCA_ACCUM_FP64 && accum != FPCA_ACCUM_FP32 && !i8)
      throw Error(FPCA_EINVAL, "accum must be FPCA_ACCUM_AUTO, FPCA_ACCUM_FP64, FPCA_ACCUM_FP32 or FPCA_ACCUM_I8(2..8)");
   if (i8 && dense) throw Error(FPCA_EINVAL, "the int8-sliced mode needs 2-bit genotype input");
   const bool timing = std::getenv("FPCA_TIMING") != nullptr;
   auto tl = std::chrono::steady_clock::now();
   auto lap = [&](const char *what) {
      const auto now = std::chrono::steady_clock::now();
      if (timing) std::fprintf(stderr, "[fpca]   %-26s %8.3f ms\n", what, std::chrono::duration<double>(now - tl).count() * 1e3);
      tl = now;
   };
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw Error(FPCA_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
   lap("hipGetDeviceCount");
   if (device < 0 || device >= ndev) throw Error(FPCA_ENODEVICE, "device index out of range");
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, device) != hipSuccess) throw Error(FPCA_ENODEVICE, "hipGetDeviceProperties failed");
   if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw Error(FPCA_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
   if (hipSetDevice(device) != hipSuccess) throw Error(FPCA_ENODEVICE, "hipSetDevice failed");
   c->device = device;
   c->N = N;
   c->P_g = P_g;
   c->P_total = P_g;
   c->np = (N + 3) / 4;
   c->pitch = (size_t)round_up(c->np, ROW_ALIGN);
   c->N_pad = (uint64_t)c->pitch * 4;
   c->P_pad = round_up(std::max<uint64_t>(P_g, 1), SNP_ALIGN);
   c->stand = stand;
   c->accum = accum;
   c->i8_S = i8 ? accum - FPCA_ACCUM_I8(0) : 0;
   c->i8_S_req = c->i8_S;
   c->dense = dense;
   lap("device properties, hipSetDevice");
   HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
   lap("stream");
   if (dense) {
      HIP_CHECK(hipMalloc(&c->d_Xd, (size_t)c->P_pad * c->N_pad * sizeof(double)));
      HIP_CHECK(hipMemsetAsync(c->d_Xd, 0, (size_t)c->P_pad * c->N_pad * sizeof(double), c->stream));
   } else {
      // The matrix is RESIDENT (the reference streams any size from disk, svdwide.h:57-68): if it does not fit, say what would
      const size_t need = c->pitch * c->P_pad;
      const hipError_t e = hipMalloc(&c->d_packed, need);
      if (e == hipErrorOutOfMemory) {
         (void)hipGetLastError();
         size_t fr = 0, tot = 0;
         (void)hipMemGetInfo(&fr, &tot);
         const double gb = 1.0 / (1024.0 * 1024.0 * 1024.0);
         const int g1 = (int)std::ceil((double)need * 1.05 / std::max<double>((double)fr, 1.0)), g2 = (int)std::ceil((double)need * 2.1 / std::max<double>((double)fr, 1.0));
         char msg[640];
         std::snprintf(msg, sizeof(msg),
                       "the packed genotypes of this shard need %.1f GiB of device memory (%llu samples x %llu SNPs at 2 bits), %.1f of %.1f GiB are free "
                       "on device %d.  What fits: the SNPs sharded over at least %d GPUs (--gpus %d; %d for the default exact-integer arithmetic, "
                       "which keeps a second, sample-major copy -- --accum fp64 does not)",
                       (double)need * gb, (unsigned long long)N, (unsigned long long)P_g, (double)fr * gb, (double)tot * gb, device, std::max(g1, 2),
                       std::max(g1, 2), std::max(g2, 2));
         throw Error(FPCA_ENOMEM, msg);
      }
      if (e != hipSuccess) throw Error(FPCA_EHIP, std::string("hipMalloc of the packed genotypes failed: ") + hipGetErrorString(e));
      HIP_CHECK(hipMemsetAsync(c->d_packed, PAD_BYTE, c->pitch * c->P_pad, c->stream));
   }
   lap("hipMalloc packed + memset");
   HIP_CHECK(hipMalloc(&c->d_lut, c->P_pad * 4 * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_mean, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_sd, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMalloc(&c->d_sumsq, c->P_pad * sizeof(double)));
   HIP_CHECK(hipMemsetAsync(c->d_lut, 0, c->P_pad * 4 * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_mean, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_sd, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMemsetAsync(c->d_sumsq, 0, c->P_pad * sizeof(double), c->stream));
   HIP_CHECK(hipMalloc(&c->d_small, 4096 * sizeof(double)));
}

void ctx_finish_upload(fpca_ctx *c)
{
   kern::fix_last_byte(c->d_packed, c->pitch, c->np, (int)(c->N % 4), c->P_g, c->stream);
   HIP_CHECK(hipStreamSynchronize(c->stream));
}

void ctx_free(fpca_ctx *c)
{
   if (!c) return;
   (void)hipSetDevice(c->device);
   if (c->comm) {
      try {
         rccl().CommDestroy(c->comm);
      } catch (...) {
      }
   }
   void *ptrs[] = {c->d_Xd, c->d_packed, c->d_lut, c->d_mean, c->d_sd,   c->d_sumsq, c->d_T,
                   c->d_part,   c->d_stage, c->d_io_a, c->d_io_b, c->d_small, c->d_packedT, c->d_inv_sd, c->d_mu_inv_sd,
                   c->d_i8w, c->d_Qb, c->d_Qg, c->d_Qm, c->d_i8ws, c->d_snp_ptr, c->d_snp_idx, c->d_smp_ptr, c->d_smp_idx, c->d_eplane,
                   c->d_full_in, c->d_full_out, c->d_qrm_loc, c->d_qrm_full, c->d_xmeta, c->d_hyb_idx, c->d_packedE, c->d_packedET, c->d_hyb_T, c->d_hyb_plane, c->d_Qd};
   for (void *p : ptrs)
      if (p) (void)hipFree(p);
   for (auto &pb : c->block_pool) (void)hipFree(pb.second);
   if (c->be_ptrs) (void)hipFree(c->be_ptrs);
   if (c->be_C) (void)hipFree(c->be_C);
   if (c->be_gpart) (void)hipFree(c->be_gpart);
   if (c->be_pin) (void)hipHostFree(c->be_pin);
   if (c->dl_pin) (void)hipHostFree(c->dl_pin);
   for (hipEvent_t e : c->dl_ev)
      if (e) (void)hipEventDestroy(e);
   for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
   for (hipEvent_t e : c->ev_chunk)
      if (e) (void)hipEventDestroy(e);
   if (c->ev_comm_done) (void)hipEventDestroy(c->ev_comm_done);
   if (c->ev_aux_go) (void)hipEventDestroy(c->ev_aux_go);
   if (c->ev_aux_done) (void)hipEventDestroy(c->ev_aux_done);
   if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
   if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
   if (c->stream) (void)hipStreamDestroy(c->stream);
   delete c;
}

void ensure_stats(fpca_ctx *c)
{
   if (c->stats_done) return;
   HIP_CHECK(hipSetDevice(c->device));
   uint32_t *d_nmiss = nullptr;
   std::vector<uint32_t> nm(c->P_g);
   if (c->P_g) HIP_CHECK(hipMalloc(&d_nmiss, c->P_g * sizeof(uint32_t)));
   kern::bed_stats(c->d_packed, c->pitch, c->N, c->P_g, c->stand, c->d_lut, c->d_mean, c->d_sd, c->d_sumsq, d_nmiss, c->stream);
   std::vector<double> ss(c->P_g);
   HIP_CHECK(hipMemcpyAsync(ss.data(), c->d_sumsq, c->P_g * sizeof(double), hipMemcpyDeviceToHost, c->stream));
   if (c->P_g) HIP_CHECK(hipMemcpyAsync(nm.data(), d_nmiss, c->P_g * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
   HIP_CHECK(hipStreamSynchronize(c->stream));
   if (d_nmiss) (void)hipFree(d_nmiss);
   c->n_missing = 0;
   for (uint32_t v : nm) c->n_missing += v;
   c->missing_known = true;
   c->h_nmiss.swap(nm);
   // pairwise-ish (blocked) summation for a reproducible, accurate trace
   double tot = 0;
   for (size_t i0 = 0; i0 < ss.size(); i0 += 1024) {
      double s = 0;
      const size_t i1 = std::min(ss.size(), i0 + 1024);
      for (size_t i = i0; i < i1; i++) s += ss[i];
      tot += s;
   }
   c->trace_local = tot;
   c->stats_done = true;
}

void ensure_io(fpca_ctx *c)
{
   if (!c->d_io_a) HIP_CHECK(hipMalloc(&c->d_io_a, (size_t)c->N_pad * MAX_BLOCKVEC * sizeof(double)));
   if (!c->d_io_b) HIP_CHECK(hipMalloc(&c->d_io_b, (size_t)c->N_pad * MAX_BLOCKVEC * sizeof(double)));
}

} // namespace fpca
