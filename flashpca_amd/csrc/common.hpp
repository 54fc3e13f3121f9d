// common.hpp -- shared constants and error plumbing for libfpca (MI355X / gfx950 only).
#pragma once
#include <sched.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>

namespace fpca {

// ---- HBM layout constants (see DESIGN.md "Data layout in HBM") -----------------------------------
// packed rows are re-pitched to a multiple of ROW_ALIGN bytes so that every SNP record starts on a
// 128-byte line and a 512-sample tile of one record is exactly one line
constexpr int ROW_ALIGN = 128;                 // bytes
constexpr int SAMPLE_ALIGN = ROW_ALIGN * 4;    // 512 samples: N_pad granularity
constexpr int SNP_ALIGN = 256;                 // P_pad granularity (K2 SNP tile)
constexpr uint8_t PAD_BYTE = 0x55;             // four "01" (missing) codes: standardises to 0

constexpr int XT_TILE = 256; // K2: SNPs per workgroup (4 waves x 4 m-tiles x 16)
constexpr int X_KC = 64;     // K3: SNPs per LDS chunk

constexpr int MAX_BLOCKVEC = 64; // widest block the kernels are instantiated for (NT <= 4)

inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// Test and diagnostic switches -- forced missing-indicator modes and split plans, allocation-failure injection, the
// host-memory test transport and the failure injection of the CLI launcher -- exist only in builds compiled with
// -DFPCA_TEST_HOOKS (flashpca_amd/_build/testhooks/, used by tests/ and scripts/).  In the shipped library and CLI the
// macro is a null pointer: no environment variable can change what they compute, and the names are not in the binaries.
#ifdef FPCA_TEST_HOOKS
#define FPCA_TEST_ENV(name) std::getenv(name)
#else
#define FPCA_TEST_ENV(name) (static_cast<const char *>(nullptr))
#endif

struct Error : std::runtime_error {
   int code;
   Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &msg);

// CPUs this process may actually run on at once: hardware threads, affinity mask and cgroup CPU quota (a container on a
// 256-thread host may own 16 of them; 256 threads against that quota only queue up).  Sizes every helper-thread pool of
// the library (pinned-download scatter, .bed readers) and of the CLI (text writers).
inline unsigned usable_cpus()
{
   unsigned n = std::thread::hardware_concurrency();
   if (n == 0) n = 1;
   cpu_set_t set;
   if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const unsigned a = (unsigned)CPU_COUNT(&set);
      if (a >= 1 && a < n) n = a;
   }
   {
      std::ifstream f("/sys/fs/cgroup/cpu.max"); // cgroup v2: "<quota|max> <period>"
      std::string q;
      long long per = 0;
      if (f >> q >> per && q != "max" && per > 0) {
         const long long c = std::atoll(q.c_str()) / per;
         if (c >= 1 && (unsigned)c < n) n = (unsigned)c;
      }
   }
   {
      std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us"); // cgroup v1
      long long q = 0, per = 0;
      if (fq >> q && fp >> per && q > 0 && per > 0) {
         const long long c = q / per;
         if (c >= 1 && (unsigned)c < n) n = (unsigned)c;
      }
   }
   return n;
}

} // namespace fpca
