// cli_main.cpp -- `flashpca`, drop-in for the reference CLI's PCA modes (flashpca.cpp:30-895) on MI355X.
//
// Same flags (flashpca.cpp:41-92), same defaults (ndim 10, standx binom2, div p, tol 1e-6, maxiter 500, precision 7,
// suffix .txt), same stdout milestones and the same output files/format (eigenvalues / eigenvectors / pcs / pve
// [/ loadings / meansd], flashpca.cpp:755-878).  Host C++ only: all arithmetic goes through the C ABI of libfpca.so
// (include/fpca.h); there is no CPU compute path.  Modes outside the PCA hot path (--scca, --ucca) are refused.
// New, MI355X-specific flags: --device, --blockvec, --maxblocks, --accum.  --memory/--blocksize/--batch/--numthreads are
// accepted for compatibility; the packed matrix is always fully resident in HBM so they have no effect.
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <map>
#include <memory>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <atomic>
#include <csignal>
#include <sched.h>
#include <sys/prctl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "../../include/fpca.h"
#include "common.hpp"
#include "plink_io.hpp"

#define FLASHPCA_VERSION "2.1-mi355x (" FPCA_VERSION ")"

namespace {

bool show_timestamp = true;

std::string timestamp() // util.cpp:270-283
{
   if (!show_timestamp) return "";
   time_t t = time(nullptr);
   char *s = asctime(localtime(&t));
   s[strlen(s) - 1] = '\0';
   return std::string("[") + s + "] ";
}

struct OptSpec {
   const char *name;
   char shortname;
   bool has_value;
   const char *help;
   bool ext = false; // an option this build adds: matched by its full name only, so that every abbreviation the reference
                     // accepts (boost::program_options guesses unambiguous prefixes, flashpca.cpp:97) still means what it meant
};

const OptSpec OPTS[] = {
   {"help", 0, false, "produce help message"},
   {"scca", 0, false, "perform sparse canonical correlation analysis (SCCA) [not supported by this build]"},
   {"ucca", 0, false, "perform per-SNP canonical correlation analysis [not supported by this build]"},
   {"project", 'p', false, "project new samples onto existing principal components"},
   {"batch", 0, false, "load all genotypes into RAM at once (no effect: the packed matrix is always resident in HBM)"},
   {"memory", 'm', true, "size of block, in MB (no effect)"},
   {"blocksize", 'b', true, "size of block for, in number of SNPs (no effect)"},
   {"numthreads", 'n', true, "set number of OpenMP threads (no effect)"},
   {"seed", 0, true, "set random seed"},
   {"bed", 0, true, "PLINK bed file"},
   {"bim", 0, true, "PLINK bim file"},
   {"fam", 0, true, "PLINK fam file"},
   {"pheno", 0, true, "PLINK phenotype file"},
   {"bfile", 0, true, "PLINK root name"},
   {"ndim", 'd', true, "number of PCs to output"},
   {"standx", 's', true, "standardization method for genotypes [binom2 | binom]"},
   {"standy", 0, true, "standardization method for phenotypes (CCA only; ignored)"},
   {"div", 0, true, "whether to divide the eigenvalues by p, n - 1, or don't divide [p | n1 | none]"},
   {"outpc", 0, true, "PC output file"},
   {"outpcx", 0, true, "X PC output file, for CCA (ignored)"},
   {"outpcy", 0, true, "Y PC output file, for CCA (ignored)"},
   {"outvec", 0, true, "eigenvector output file"},
   {"outload", 0, true, "SNP loadings"},
   {"outvecx", 0, true, "X eigenvector output file, for CCA (ignored)"},
   {"outvecy", 0, true, "Y eigenvector output file, for CCA (ignored)"},
   {"outval", 0, true, "Eigenvalue output file"},
   {"outpve", 0, true, "proportion of variance explained output file"},
   {"outmeansd", 0, true, "mean+SD (used to standardize SNPs) output file"},
   {"outproj", 0, true, "PCA projection output file"},
   {"inload", 0, true, "SNP loadings input file"},
   {"inmeansd", 0, true, "mean+SD (used to standardize SNPs) input file"},
   {"inmaf", 0, true, "MAF input file"},
   {"verbose", 'v', false, "verbose"},
   {"tol", 0, true, "tolerance for PCA iterations"},
   {"lambda1", 0, true, "1st penalty for CCA/SCCA (ignored)"},
   {"lambda2", 0, true, "2nd penalty for CCA/SCCA (ignored)"},
   {"maxiter", 0, true, "maximum number of iterations: restarts of the reference's 2 ndim + 1 vector Lanczos factorisation, i.e. a budget of 2 ndim + 1 + maxiter (ndim + 1) operator applications"},
   {"debug", 0, false, "debug (no effect)"},
   {"suffix", 'f', true, "suffix for all output files"},
   {"check", 'c', false, "check eigenvalues/eigenvectors"},
   {"precision", 0, true, "digits of precision for output"},
   {"notime", 0, false, "don't print timestamp in output"},
   {"save-vinit", 0, false, "saves the initial v eigenvector for SCCA (no effect)"},
   {"version", 0, false, "version"},
   {"device", 0, true, "HIP device index [0] (with --gpus G: the first of G consecutive devices)", true},
   {"gpus", 0, true, "number of GPUs for PCA [1]: the SNPs are split into that many contiguous shards, one process per GPU, partial products summed over RCCL", true},
   {"solver", 0, true, "with --gpus: how the eigensolver's sample-sized work is laid out [rowshard | replicated]: rowshard (default) = every GPU keeps and orthogonalises 1/G of the rows of the Krylov basis (all-gather -> products -> reduce-scatter per pass); replicated = every GPU keeps the whole basis, ONE all-reduce of the N x b product per pass and nothing else on the wire.  rowshard checks its exchange once and falls back to replicated by itself if the check fails", true},
   {"blockvec", 0, true, "block width of the eigensolver: 16, 32, 48 or 64 [16; 32 / 64 for ndim > 64 / > 128]", true},
   {"maxblocks", 0, true, "basis cap (in blocks) before a thick restart [automatic]", true},
   {"passes", 0, true, "arithmetic of the eigensolver's passes in the exact-integer modes [mixed | exact]: mixed (default) = a solve that needs many passes makes most of them on 4 byte slices of the fp64 operand and puts the Ritz vectors through the exact operator before it declares convergence; exact = every pass on all slices", true},
   {"accum", 0, true, "arithmetic of the two genotype GEMMs [auto | fp64 | fp32 | i8 | i8xS]: i8 = exact-integer int8 MFMA on S = 7 (i8xS: S = 2..8) byte slices of the fp64 operand, results equal to fp64; fp32 = fp32 MFMA products, fp64 long accumulation; auto (default) = i8, or fp64 if the int8 buffers do not fit", true},
};

// Long options like po::parse_command_line with its default style (flashpca.cpp:97; allow_guessing is part of
// command_line_style::default_style): the full name wins; otherwise an abbreviation that is a prefix of exactly one of the
// reference's options selects it (--nd 10, --outl f), and one that fits several is refused with boost's "ambiguous" error.
const OptSpec *find_long(const std::string &n, const std::string &as_typed)
{
   for (const auto &o : OPTS)
      if (n == o.name) return &o;
   std::vector<const OptSpec *> hits;
   if (!n.empty())
      for (const auto &o : OPTS)
         if (!o.ext && std::string(o.name).compare(0, n.size(), n) == 0) hits.push_back(&o);
   if (hits.size() == 1) return hits[0];
   if (hits.empty()) throw std::runtime_error("unrecognised option '" + as_typed + "'");
   std::string msg = "option '--" + n + "' is ambiguous and matches ";
   for (size_t i = 0; i < hits.size(); i++) {
      if (i) msg += i + 1 == hits.size() ? (hits.size() > 2 ? ", and " : " and ") : ", ";
      msg += std::string("'--") + hits[i]->name + "'";
   }
   throw std::runtime_error(msg);
}
const OptSpec *find_short(char c)
{
   for (const auto &o : OPTS)
      if (o.shortname && o.shortname == c) return &o;
   return nullptr;
}

typedef std::map<std::string, std::string> VarMap;

// throws std::runtime_error like boost::program_options does on malformed command lines
VarMap parse_command_line(int argc, char *argv[])
{
   VarMap vm;
   for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      const OptSpec *o = nullptr;
      std::string val;
      bool have_val = false;
      if (a.rfind("--", 0) == 0) {
         std::string body = a.substr(2);
         size_t eq = body.find('=');
         if (eq != std::string::npos) {
            val = body.substr(eq + 1);
            have_val = true;
            body = body.substr(0, eq);
         }
         o = find_long(body, a);
      } else if (a.size() >= 2 && a[0] == '-') {
         o = find_short(a[1]);
         if (!o) throw std::runtime_error("unrecognised option '" + a + "'");
         if (a.size() > 2) {
            val = a.substr(2);
            have_val = true;
         }
      } else
         throw std::runtime_error("too many positional options have been specified on the command line");
      if (o->has_value) {
         if (!have_val) {
            if (i + 1 >= argc) throw std::runtime_error(std::string("the required argument for option '--") + o->name + "' is missing");
            val = argv[++i];
         }
         vm[o->name] = val;
      } else {
         if (have_val) throw std::runtime_error(std::string("option '--") + o->name + "' does not take any arguments");
         vm[o->name] = "";
      }
   }
   return vm;
}

long to_long(const VarMap &vm, const char *name)
{
   const std::string &s = vm.at(name);
   char *end = nullptr;
   errno = 0;
   long v = std::strtol(s.c_str(), &end, 10);
   if (*end != '\0' || errno != 0 || s.empty()) throw std::runtime_error(std::string("the argument ('") + s + "') for option '--" + name + "' is invalid");
   return v;
}
double to_double(const VarMap &vm, const char *name)
{
   const std::string &s = vm.at(name);
   char *end = nullptr;
   errno = 0;
   double v = std::strtod(s.c_str(), &end);
   if (*end != '\0' || errno != 0 || s.empty()) throw std::runtime_error(std::string("the argument ('") + s + "') for option '--" + name + "' is invalid");
   return v;
}

void print_help()
{
   std::cerr << "Options:" << std::endl;
   for (const auto &o : OPTS) {
      std::string left = "  ";
      if (o.shortname) left += std::string("-") + o.shortname + " [ --" + o.name + " ]";
      else left += std::string("--") + o.name;
      if (o.has_value) left += " arg";
      while (left.size() < 30) left += ' ';
      std::cerr << left << " " << o.help << std::endl;
   }
   std::cerr << std::endl;
}

void fpca_ok(int rc)
{
   if (rc != FPCA_OK) throw std::runtime_error(fpca_last_error());
}

} // namespace

// ---- --gpus G: one process per GPU -------------------------------------------------------------------------------------
// The parent parses the command line and the .fam/.bim, maps one shared region, and forks G - 1 children BEFORE anything
// touches HIP; every process (the parent is rank 0) opens its contiguous SNP shard of the .bed on its own device, joins
// the RCCL communicator (id made by rank 0, handed over through the shared region) and runs the same fpca_pca -- the host
// algebra is replicated and deterministic, the only data-path exchange is the all-reduce inside the block apply
// (DESIGN section 5).  Eigenvectors / eigenvalues are identical on every rank; the loadings and mean/sd rows of each shard
// are deposited in the shared region and rank 0 writes every file.
static_assert(std::atomic<int>::is_always_lock_free, "the SIGCHLD handler touches these atomics: they must be lock-free");
struct MultiShared {
   std::atomic<int> created, failed, id_ready, bar_count, bar_sense;
   std::atomic<int> op_kind[64];             // test transport: the call every rank is in (shm_same_call)
   std::atomic<unsigned long long> op_count[64];
   uint8_t id[FPCA_UNIQUE_ID_BYTES];
   char msg[512];
};

struct Multi {
   int ngpus = 1, rank = 0;
   MultiShared *sh = nullptr;
   double *V = nullptr, *meansd = nullptr; // P x k and P x 2, column-major, in the shared region
   double *U = nullptr, *Px = nullptr;     // N x k each, column-major, in the shared region: every rank writes its own rows
   double *slots = nullptr;                // FPCA_CLI_TEST_TRANSPORT=shm only: G x slot_cap doubles
   size_t slot_cap = 0;
   bool test_transport = false;
   bool test_collectives = false; // ... with all-gather / reduce-scatter of its own (shm2)
};

// set in the children of a --gpus run: an exception there must not fall through to rank 0's output code
static int g_child_rank = 0;
static MultiShared *g_shared = nullptr;

// rank 0's view of its children.  A child that dies on its own (segfault, OOM kill, an uncaught exit) can never reach the
// next rendezvous, and rank 0 may itself be blocked inside an RCCL collective waiting for it -- so the parent watches
// SIGCHLD: an abnormal child exit that nobody announced in the shared region ends the whole run at once.
static pid_t g_children[64];
static int g_nchildren = 0;
static volatile sig_atomic_t g_child_done[64];
static volatile sig_atomic_t g_quiesce = 0; // set while rank 0 itself winds the children down

static void kill_children()
{
   for (int i = 0; i < g_nchildren; i++)
      if (!g_child_done[i]) (void)kill(g_children[i], SIGKILL);
}

static void on_sigchld(int)
{
   const int saved = errno;
   for (int i = 0; i < g_nchildren; i++) {
      if (g_child_done[i]) continue;
      int st = 0;
      if (waitpid(g_children[i], &st, WNOHANG) != g_children[i]) continue;
      g_child_done[i] = 1;
      const bool bad = WIFSIGNALED(st) || (WIFEXITED(st) && WEXITSTATUS(st) != 0);
      if (bad && !g_quiesce && g_shared && g_shared->failed.load() == 0) {
         g_shared->failed.fetch_add(1);
         static const char msg[] = "Exception: a GPU rank of the --gpus run died unexpectedly\nTerminating\n";
         (void)!write(2, msg, sizeof(msg) - 1);
         kill_children();
         _exit(EXIT_FAILURE);
      }
      // A rank that ANNOUNCED its failure and left: normally everybody meets at the next rendezvous and rank 0 reports the
      // message -- unless the others (rank 0 included) sit inside a collective that the leaver will never join.  Give the
      // orderly path five seconds, then end the run from the alarm.
      if (bad && !g_quiesce) alarm(5);
   }
   errno = saved;
}

static void on_sigalrm(int)
{
   if (g_quiesce) return;
   static const char head[] = "Exception: ";
   static const char tail[] = " (the other ranks were still inside a collective)\nTerminating\n";
   (void)!write(2, head, sizeof(head) - 1);
   if (g_shared) (void)!write(2, g_shared->msg, strnlen(g_shared->msg, sizeof(g_shared->msg)));
   (void)!write(2, tail, sizeof(tail) - 1);
   kill_children();
   _exit(EXIT_FAILURE);
}

// rank 0: wait for every child that has not been reaped yet (the handler may reap them first: ECHILD is fine)
static void wait_children()
{
   for (int i = 0; i < g_nchildren; i++) {
      if (g_child_done[i]) continue;
      (void)waitpid(g_children[i], nullptr, 0);
      g_child_done[i] = 1;
   }
}

// rank 0 cannot go on (exception, early return after the fork): tell the children through the shared region, give them two
// seconds to leave at their next rendezvous, then kill what is left (a child inside an RCCL collective never gets there)
static void abandon_children()
{
   if (g_child_rank > 0 || g_nchildren == 0) return;
   g_quiesce = 1;
   if (g_shared) g_shared->failed.fetch_add(1);
   for (int t = 0; t < 200; t++) {
      bool all = true;
      for (int i = 0; i < g_nchildren; i++) {
         if (g_child_done[i]) continue;
         if (waitpid(g_children[i], nullptr, WNOHANG) != 0) // reaped here, or already by the handler (ECHILD)
            g_child_done[i] = 1;
         else
            all = false;
      }
      if (all) return;
      usleep(10000);
   }
   kill_children();
   wait_children();
}

void multi_fail(Multi &m, const std::string &why)
{
   if (!m.sh) return;
   if (m.sh->failed.fetch_add(1) == 0) std::snprintf(m.sh->msg, sizeof(m.sh->msg), "rank %d: %s", m.rank, why.c_str());
}

// all ranks arrive, or somebody failed (returns false)
bool multi_barrier(Multi &m)
{
   MultiShared *sh = m.sh;
   const int sense = sh->bar_sense.load();
   if (sh->bar_count.fetch_add(1) + 1 == m.ngpus) {
      sh->bar_count.store(0);
      sh->bar_sense.store(sense ^ 1);
      return sh->failed.load() == 0;
   }
   while (sh->bar_sense.load() == sense) {
      if (sh->failed.load()) return false;
      sched_yield();
   }
   return sh->failed.load() == 0;
}

// Test transport (FPCA_CLI_TEST_TRANSPORT=shm; only in builds with -DFPCA_TEST_HOOKS, i.e. _build/testhooks/flashpca):
// every rank on the SAME device, the sum staged through host shared memory in rank order -- exercises the launcher, the
// sharding and the gather of the outputs on a one-GPU box, where RCCL refuses two ranks on one device.  The shipped CLI
// has no such path: its only transport is RCCL.
#ifdef FPCA_TEST_HOOKS
// the hook contract of fpca.h: a collective fails on every rank or on none.  Every rank announces (call kind, count) before the
// first rendezvous and checks after it that all ranks are in the SAME call -- ranks out of step (one of them took an error path
// the others did not) all see the mismatch and all return non-zero.
static bool shm_same_call(Multi &m, int kind, uint64_t count)
{
   m.sh->op_kind[m.rank].store(kind);
   m.sh->op_count[m.rank].store(count);
   if (!multi_barrier(m)) return false;
   for (int r = 0; r < m.ngpus; r++)
      if (m.sh->op_kind[r].load() != kind || m.sh->op_count[r].load() != count) {
         std::fprintf(stderr, "[fpca-cli] rank %d: the ranks are not in the same collective (rank %d: kind %d count %llu; here kind %d count %llu)\n", m.rank, r,
                      m.sh->op_kind[r].load(), (unsigned long long)m.sh->op_count[r].load(), kind, (unsigned long long)count);
         return false;
      }
   return true;
}
int shm_allreduce(void *user, double *dbuf, uint64_t count, void *stream)
{
   Multi &m = *static_cast<Multi *>(user);
   if (count > m.slot_cap) return -1;
   if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
   if (hipMemcpy(m.slots + (size_t)m.rank * m.slot_cap, dbuf, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
   if (!shm_same_call(m, 1, count)) return -1;
   std::vector<double> sum(count, 0.0);
   for (int r = 0; r < m.ngpus; r++) {
      const double *p = m.slots + (size_t)r * m.slot_cap;
      for (uint64_t i = 0; i < count; i++) sum[i] += p[i];
   }
   if (!multi_barrier(m)) return -1; // nobody overwrites a slot before everyone has read it
   return hipMemcpy(dbuf, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
// FPCA_CLI_TEST_TRANSPORT=shm2: all-gather and reduce-scatter of their own as well (fpca_set_collectives), so that the
// row-sharded solver runs the call sequence it runs over RCCL -- per row chunk, on the communication stream -- on one device
int shm_allgather(void *user, const double *send, double *recv, uint64_t count, void *stream)
{
   Multi &m = *static_cast<Multi *>(user);
   if (count > m.slot_cap) return -1;
   if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
   if (hipMemcpy(m.slots + (size_t)m.rank * m.slot_cap, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
   if (!shm_same_call(m, 2, count)) return -1;
   for (int r = 0; r < m.ngpus; r++)
      if (hipMemcpy(recv + (size_t)r * count, m.slots + (size_t)r * m.slot_cap, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return -1;
   return multi_barrier(m) ? 0 : -1;
}
int shm_reducescatter(void *user, const double *send, double *recv, uint64_t count, void *stream)
{
   Multi &m = *static_cast<Multi *>(user);
   if (count * (uint64_t)m.ngpus > m.slot_cap) return -1;
   if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
   if (hipMemcpy(m.slots + (size_t)m.rank * m.slot_cap, send, count * m.ngpus * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
   if (!shm_same_call(m, 3, count)) return -1;
   std::vector<double> sum(count, 0.0);
   for (int r = 0; r < m.ngpus; r++) {
      const double *p = m.slots + (size_t)r * m.slot_cap + (size_t)m.rank * count;
      for (uint64_t i = 0; i < count; i++) sum[i] += p[i];
   }
   if (!multi_barrier(m)) return -1;
   return hipMemcpy(recv, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
#endif

#ifdef FPCA_TEST_HOOKS
#include <execinfo.h>
static void on_segv(int)
{
   void *bt[64];
   const int n = backtrace(bt, 64);
   backtrace_symbols_fd(bt, n, 2);
   _exit(139);
}
#endif

int main(int argc, char *argv[])
{
#ifdef FPCA_TEST_HOOKS
   signal(SIGSEGV, on_segv);
#endif
   VarMap vm;
   try {
      vm = parse_command_line(argc, argv);
   } catch (std::exception &e) {
      // flashpca.cpp:100-106 (exit status is EXIT_SUCCESS there too)
      std::cerr << e.what() << std::endl << "Use --help to get more help" << std::endl;
      return EXIT_SUCCESS;
   }
   auto has = [&](const char *n) { return vm.count(n) > 0; };

   show_timestamp = !has("notime");
   const bool verbose = has("verbose");

   std::cout << timestamp() << "arguments: flashpca ";
   for (int i = 0; i < argc; i++) std::cout << argv[i] << " ";
   std::cout << std::endl;

   if (has("version")) {
      std::cerr << "flashpca " << FLASHPCA_VERSION << std::endl;
      std::cerr << "MI355X-native implementation of the flashpca 2.1 PCA path (command line after Gad Abraham's flashpca)." << std::endl << std::endl;
      return EXIT_SUCCESS;
   }
   if (has("help")) {
      std::cerr << "flashpca " << FLASHPCA_VERSION << std::endl;
      print_help();
      return EXIT_SUCCESS;
   }

   // ---- mode selection (flashpca.cpp:136-228) ------------------------------------------------------------
   enum { MODE_PCA, MODE_CHECK, MODE_PROJECT } mode = MODE_PCA;
   const char *modes[] = {"ucca", "scca", "check", "project"};
   for (const char *m1 : modes)
      for (const char *m2 : modes)
         if (std::string(m1) < m2 && has(m1) && has(m2)) {
            std::cerr << "Error: conflicting modes requested: --" << m1 << ", --" << m2 << std::endl << "Use --help to get more help" << std::endl;
            return EXIT_FAILURE;
         }
   if (has("scca") || has("ucca")) {
      std::cerr << "Error: --scca / --ucca are outside the PCA path this build implements" << std::endl;
      return EXIT_FAILURE;
   }
   if (has("check")) mode = MODE_CHECK;
   else if (has("project")) {
      mode = MODE_PROJECT;
      if (!has("inload")) {
         std::cerr << "Error: SNP-loadings must be specified using --inload" << std::endl;
         return EXIT_FAILURE;
      }
      if (!has("inmaf") && !has("inmeansd")) {
         std::cerr << "Error: one of MAF or mean/stdev must be specified using  --inmaf or --inmeansd, respectively" << std::endl;
         return EXIT_FAILURE;
      }
   }

   try {
      if (has("memory") && to_long(vm, "memory") < 1) {
         std::cerr << "Error: memory (MB) must be >=1" << std::endl;
         return EXIT_FAILURE;
      }
      if (has("blocksize")) {
         if (has("memory")) {
            std::cerr << "Error: cannot specify both --memory and --blocksize at the same time" << std::endl;
            return EXIT_FAILURE;
         }
         if (to_long(vm, "blocksize") < 1) {
            std::cerr << "Error: blocksize must be >=1" << std::endl;
            return EXIT_FAILURE;
         }
      }
      if (has("numthreads")) (void)to_long(vm, "numthreads");
      long seed = has("seed") ? to_long(vm, "seed") : 1L;

      std::string fam_file, geno_file, bim_file;
      if (has("bfile")) {
         geno_file = vm["bfile"] + ".bed";
         bim_file = vm["bfile"] + ".bim";
         fam_file = vm["bfile"] + ".fam";
      } else if (has("bed") && has("bim") && has("fam")) {
         geno_file = vm["bed"];
         bim_file = vm["bim"];
         fam_file = vm["fam"];
      } else {
         std::cerr << "Error: you must specify either --bfile or --bed / --fam / --bim" << std::endl << "Use --help to get more help" << std::endl;
         return EXIT_FAILURE;
      }

      int n_dim = 10;
      if (has("ndim")) {
         n_dim = (int)to_long(vm, "ndim");
         if (n_dim < 1) {
            std::cerr << "Error: --ndim can't be less than 1" << std::endl;
            return EXIT_FAILURE;
         }
      }
      int stand_method_x = FPCA_STANDARDISE_BINOM2;
      if (has("standx")) {
         const std::string m = vm["standx"];
         if (m == "binom") stand_method_x = FPCA_STANDARDISE_BINOM;
         else if (m == "binom2") stand_method_x = FPCA_STANDARDISE_BINOM2;
         else {
            std::cerr << "Error: unknown standardization method (--standx): " << m << std::endl;
            return EXIT_FAILURE;
         }
      }
      std::string suffix = has("suffix") ? vm["suffix"] : ".txt";
      std::string pcfile = has("outpc") ? vm["outpc"] : "pcs" + suffix;
      std::string eigvecfile = has("outvec") ? vm["outvec"] : "eigenvectors" + suffix;
      std::string eigvalfile = has("outval") ? vm["outval"] : "eigenvalues" + suffix;
      std::string eigpvefile = has("outpve") ? vm["outpve"] : "pve" + suffix;
      std::string meansdfile = has("outmeansd") ? vm["outmeansd"] : "meansd" + suffix;
      const bool save_meansd = has("outmeansd");
      std::string projfile = has("outproj") ? vm["outproj"] : "projection" + suffix;

      int maxiter = 500;
      if (has("maxiter")) {
         maxiter = (int)to_long(vm, "maxiter");
         if (maxiter <= 0) {
            std::cerr << "Error: --maxiter can't be less than 1" << std::endl;
            return EXIT_FAILURE;
         }
      }
      double tol = 1e-6;
      if (has("tol")) {
         tol = to_double(vm, "tol");
         if (tol <= 0) {
            std::cerr << "Error: --tol can't be zero or negative" << std::endl;
            return EXIT_FAILURE;
         }
      }
      const bool do_loadings = has("outload");
      const std::string loadingsfile = do_loadings ? vm["outload"] : "";
      int divisor = FPCA_DIVISOR_P;
      if (has("div")) {
         const std::string m = vm["div"];
         if (m == "none") divisor = FPCA_DIVISOR_NONE;
         else if (m == "n1") divisor = FPCA_DIVISOR_N1;
         else if (m == "p") divisor = FPCA_DIVISOR_P;
         else {
            std::cerr << "Error: unknown divisor (--div): " << m << std::endl;
            return EXIT_FAILURE;
         }
      }
      std::string in_meansd_file, in_maf_file, in_load_file;
      if (has("inmeansd")) {
         if (has("inmaf")) {
            std::cerr << "Error: conflicting options requested --inmeansd, --inmaf" << std::endl;
            return EXIT_FAILURE;
         }
         in_meansd_file = vm["inmeansd"];
         if (in_meansd_file.empty()) {
            std::cerr << "Error: no file specified for --inmeansd" << std::endl;
            return EXIT_FAILURE;
         }
      } else if (has("inmaf")) {
         in_maf_file = vm["inmaf"];
         if (in_maf_file.empty()) {
            std::cerr << "Error: no file specified for --inmaf" << std::endl;
            return EXIT_FAILURE;
         }
      }
      if (has("inload")) {
         in_load_file = vm["inload"];
         if (in_load_file.empty()) {
            std::cerr << "Error: no file specified for --inload" << std::endl;
            return EXIT_FAILURE;
         }
      }
      int precision = 7;
      if (has("precision")) {
         precision = (int)to_long(vm, "precision");
         if (precision <= 1) {
            std::cerr << "Error: output --precision too low" << std::endl;
            return EXIT_FAILURE;
         }
      }
      const int device = has("device") ? (int)to_long(vm, "device") : 0;
      const int ngpus = has("gpus") ? (int)to_long(vm, "gpus") : 1;
      if (ngpus < 1 || ngpus > 64) {
         std::cerr << "Error: --gpus must be between 1 and 64" << std::endl;
         return EXIT_FAILURE;
      }
      if (ngpus > 1 && mode != MODE_PCA) {
         std::cerr << "Error: --gpus applies to PCA only (--check and --project run on one GPU)" << std::endl;
         return EXIT_FAILURE;
      }
      const int blockvec = has("blockvec") ? (int)to_long(vm, "blockvec") : 0;
      const int maxblocks = has("maxblocks") ? (int)to_long(vm, "maxblocks") : 0;
      int accum = FPCA_ACCUM_AUTO;
      if (has("accum")) {
         const std::string m = vm["accum"];
         if (m == "auto") accum = FPCA_ACCUM_AUTO;
         else if (m == "fp64") accum = FPCA_ACCUM_FP64;
         else if (m == "fp32") accum = FPCA_ACCUM_FP32;
         else if (m == "i8") accum = FPCA_ACCUM_I8(7);
         else if (m.size() == 4 && m.compare(0, 3, "i8x") == 0 && m[3] >= '2' && m[3] <= '8') accum = FPCA_ACCUM_I8(m[3] - '0');
         else {
            std::cerr << "Error: unknown accumulate mode (--accum): " << m << std::endl;
            return EXIT_FAILURE;
         }
      }

      int replicated_solver = 0;
      if (has("solver")) {
         const std::string m = vm["solver"];
         if (m == "replicated") replicated_solver = 1;
         else if (m != "rowshard") {
            std::cerr << "Error: unknown --solver layout (rowshard | replicated): " << m << std::endl;
            return EXIT_FAILURE;
         }
      }
      int mixed = 0;
      if (has("passes")) {
         const std::string m = vm["passes"];
         if (m == "mixed") mixed = 1;
         else if (m == "exact") mixed = -1;
         else {
            std::cerr << "Error: unknown --passes mode (mixed | exact): " << m << std::endl;
            return EXIT_FAILURE;
         }
      }

      // ---- end of command line parsing -------------------------------------------------------------------
      std::cout << timestamp() << "Start flashpca (version " << FLASHPCA_VERSION << ")" << std::endl;
      verbose && std::cout << timestamp() << "seed: " << seed << std::endl;

      // FPCA_TIMING=1: wall-clock of each phase on stderr
      const bool phase_timing = std::getenv("FPCA_TIMING") != nullptr;
      bool quiet = false; // ranks > 0 of a --gpus run
      auto phase = [&, last = std::chrono::steady_clock::now()](const char *what) mutable {
         const auto now = std::chrono::steady_clock::now();
         if (phase_timing && !quiet) std::fprintf(stderr, "[fpca-cli] %-32s %8.3f ms\n", what, std::chrono::duration<double>(now - last).count() * 1e3);
         last = now;
      };
      // One GPU: the HIP runtime starts up (~0.1 s) on a helper thread while this one reads the text files, and the .bim is
      // parsed on another while the .fam is (only N, from the .fam, is needed before the upload can start).  With --gpus the
      // parent must not touch HIP, nor hold threads, before it forks: everything stays on this thread.
      struct Joiner { // (joins on every way out of the try block, exceptions included)
         std::vector<std::thread> th;
         ~Joiner()
         {
            for (auto &t : th)
               if (t.joinable()) t.join();
         }
      } helpers;
      helpers.th.reserve(8); // (threads are referred to by address below)
      if (ngpus == 1) helpers.th.emplace_back([device] { (void)fpca_warmup(device); }); // (errors resurface in fpca_create_from_bed)
      std::vector<std::string> snp_ids, ref_alleles, alt_alleles, fam_ids, indiv_ids;
      std::exception_ptr bim_error;
      auto parse_bim = [&] {
         try {
            fpca::read_plink_bim(bim_file, snp_ids, ref_alleles, alt_alleles);
         } catch (...) {
            bim_error = std::current_exception();
         }
      };
      std::thread bim_thread;
      if (ngpus == 1) bim_thread = std::thread(parse_bim);
      // N = number of rows of the .fam whose 6th column parses as a number (flashpca.cpp:589 -> data.cpp:408-413), and the
      // two id columns (read_plink_fam, flashpca.cpp:591) from the same pass over the file
      uint64_t N = 0;
      try {
         N = fpca::read_fam(fam_file, fam_ids, indiv_ids);
      } catch (...) {
         if (bim_thread.joinable()) bim_thread.join();
         throw;
      }
      if (ngpus == 1)
         bim_thread.join();
      else
         parse_bim();
      if (bim_error) std::rethrow_exception(bim_error);
      if (N == 0) throw std::runtime_error("no samples found in " + fam_file);
      phase(".fam / .bim");

      fpca_ctx *ctx = nullptr;
      uint64_t nsnps = 0; // SNPs in the file (all shards)
      uint64_t P_file_all = 0; // the same, from the file size alone (data.cpp:165-170) -- known before any device work
      // the big results live in UNINITIALISED memory: a std::vector would zero 80 + 80 + 16 MB on this thread first (35 ms of
      // page faults at 500,000 x 100,000); the parallel download touches the pages instead
      struct Buf {
         std::unique_ptr<double[]> p;
         size_t n = 0;
         void resize(size_t k)
         {
            p.reset(new double[k]);
            n = k;
         }
         double *data() { return p.get(); }
         bool empty() const { return n == 0; }
         double *begin() { return p.get(); }
         double *end() { return p.get() + n; }
      } U, Px, V;
      std::vector<std::string> rownames, rn_snp; // "FID\tIID" / "SNP\tRefAllele" row labels of the output files
      Multi mg;
      mg.ngpus = ngpus;
      uint64_t snp_begin = 0, snp_count = 0; // this rank's shard (0, 0 = the whole file)
      // everything that can be refused from the file sizes alone is refused here: before any device work, and -- in a
      // --gpus run -- before the fork, so that no rank is left waiting for another
      {
         struct stat st;
         if (stat(geno_file.c_str(), &st) != 0) throw std::runtime_error("[Data::read_bed] Error reading file " + geno_file + ", error " + strerror(errno));
         const uint64_t np = (N + 3) / 4;
         const uint64_t P_file = (uint64_t)st.st_size > 3 ? ((uint64_t)st.st_size - 3) / np : 0; // data.cpp:165-170
         P_file_all = P_file;
         // flashpca.cpp:623-633
         const unsigned max_dim = (unsigned)((std::fmin((double)N, (double)P_file) - 1) / 2.0);
         if ((unsigned)n_dim > max_dim) { // (every mode, like the reference)
            std::cerr << "Error: You asked for " << n_dim << " dimensions, but only " << max_dim << "allowed" << std::endl;
            return EXIT_FAILURE;
         }
         // the loadings / mean-sd files carry one .bim row name per SNP of the .bed
         if ((do_loadings || save_meansd) && snp_ids.size() != P_file)
            throw std::runtime_error("the .bim file has a different number of SNPs (" + std::to_string(snp_ids.size()) + ") than the .bed (" +
                                     std::to_string(P_file) + ")");
      }
      if (ngpus > 1) {
         struct stat st;
         if (stat(geno_file.c_str(), &st) != 0) throw std::runtime_error("[Data::read_bed] Error reading file " + geno_file + ": " + strerror(errno));
         const uint64_t np = (N + 3) / 4;
         const uint64_t P_file = (uint64_t)st.st_size > 3 ? ((uint64_t)st.st_size - 3) / np : 0; // data.cpp:165-170
         if (P_file < (uint64_t)ngpus) throw std::runtime_error("fewer SNPs than GPUs");
         const char *tt = FPCA_TEST_ENV("FPCA_CLI_TEST_TRANSPORT");
         mg.test_transport = tt && (std::string(tt) == "shm" || std::string(tt) == "shm2");
         mg.test_collectives = tt && std::string(tt) == "shm2";
         if (mg.test_transport)
            std::cerr << "[fpca-cli] FPCA_CLI_TEST_TRANSPORT=shm: all ranks share one device and exchange through host memory -- a test "
                         "hook for one-GPU boxes, not a way to run" << std::endl;
         mg.slot_cap = mg.test_transport ? (size_t)(N + 1024 + 512 * (size_t)ngpus) * 64 : 0; // the row-sharded solver's padded blocks
         const size_t head = (sizeof(MultiShared) + 63) / 64 * 64;
         const size_t bytes = head + ((size_t)P_file * (n_dim + 2) + 2 * (size_t)N * n_dim + (size_t)ngpus * mg.slot_cap) * sizeof(double);
         void *mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
         if (mem == MAP_FAILED) throw std::runtime_error(std::string("mmap of the shared region failed: ") + strerror(errno));
         mg.sh = new (mem) MultiShared();
         mg.sh->created = 0;
         mg.sh->failed = 0;
         mg.sh->id_ready = 0;
         mg.sh->bar_count = 0;
         mg.sh->bar_sense = 0;
         mg.sh->msg[0] = 0;
         mg.V = reinterpret_cast<double *>(static_cast<char *>(mem) + head);
         mg.meansd = mg.V + (size_t)P_file * n_dim;
         mg.U = mg.meansd + (size_t)P_file * 2;
         mg.Px = mg.U + (size_t)N * n_dim;
         mg.slots = mg.Px + (size_t)N * n_dim;
         std::cout.flush();
         std::fflush(nullptr);
         g_shared = mg.sh;
         {
            struct sigaction sa;
            std::memset(&sa, 0, sizeof(sa));
            sa.sa_handler = on_sigchld;
            sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
            sigaction(SIGCHLD, &sa, nullptr);
            sa.sa_handler = on_sigalrm;
            sigaction(SIGALRM, &sa, nullptr);
         }
         const pid_t parent = getpid();
         // SIGCHLD stays blocked until every child pid is registered: a child that dies at once is then still found by the
         // handler (a pending SIGCHLD is delivered on unblocking; the handler polls every registered child)
         sigset_t chld, oldmask;
         sigemptyset(&chld);
         sigaddset(&chld, SIGCHLD);
         sigprocmask(SIG_BLOCK, &chld, &oldmask);
         for (int r = 1; r < ngpus; r++) { // nothing has touched HIP yet: the children initialise their own runtime
            const pid_t pid = fork();
            if (pid < 0) {
               multi_fail(mg, std::string("fork failed: ") + strerror(errno));
               break;
            }
            if (pid == 0) {
               // a child never outlives rank 0 (it may sit in an RCCL collective that will never complete)
               (void)prctl(PR_SET_PDEATHSIG, SIGKILL);
               if (getppid() != parent) _exit(1);
               signal(SIGCHLD, SIG_DFL);
               signal(SIGALRM, SIG_DFL);
               sigprocmask(SIG_SETMASK, &oldmask, nullptr);
               mg.rank = r;
               g_nchildren = 0;
               g_child_rank = r;
               quiet = true;
               std::cout.setstate(std::ios::failbit); // progress lines come from rank 0 only
               break;
            }
            g_child_done[g_nchildren] = 0;
            g_children[g_nchildren++] = pid;
         }
         if (mg.rank == 0) sigprocmask(SIG_SETMASK, &oldmask, nullptr);
         snp_begin = P_file * (uint64_t)mg.rank / (uint64_t)ngpus;
         snp_count = P_file * (uint64_t)(mg.rank + 1) / (uint64_t)ngpus - snp_begin;
      }
      // a rank that cannot go on says so in the shared region and everybody leaves at the next rendezvous
      auto multi_abort = [&](void) -> int {
         if (mg.rank > 0) {
            if (ctx) fpca_destroy(ctx);
            _exit(1);
         }
         abandon_children(); // (two seconds for the orderly exit, then SIGKILL: a rank may be inside a collective)
         alarm(0);
         std::cerr << timestamp() << "Exception: " << mg.sh->msg << std::endl << timestamp() << "Terminating" << std::endl;
         if (ctx) fpca_destroy(ctx);
         return EXIT_FAILURE;
      };
      const int my_device = device + ((ngpus > 1 && !mg.test_transport) ? mg.rank : 0);
      // One GPU, PCA: while the .bed streams to the device (that is the copy engine's and the reader threads' business), a
      // helper thread gets the host side of the results ready -- it touches the pages of the 80 + 80 + 16 MB result buffers
      // (first-touch page faults are 35 ms of a one-threaded pass, and the download would otherwise pay them) and builds the
      // row labels of the output files.
      size_t prep_idx = (size_t)-1;
      if (ngpus == 1 && mode == MODE_PCA && P_file_all > 0) {
         U.resize((size_t)N * n_dim);
         Px.resize((size_t)N * n_dim);
         if (do_loadings) V.resize((size_t)P_file_all * n_dim);
         prep_idx = helpers.th.size();
         helpers.th.emplace_back([&] {
            for (Buf *bf : {&U, &Px, &V})
               for (size_t i = 0; i < bf->n; i += 512) bf->p[i] = 0.0; // one write per 4 KB page
            rownames.resize(N);
            for (uint64_t i = 0; i < N; i++) rownames[i] = fam_ids[i] + "\t" + indiv_ids[i];
            if ((do_loadings || save_meansd) && snp_ids.size() == P_file_all) {
               rn_snp.resize(snp_ids.size());
               for (size_t i = 0; i < rn_snp.size(); i++) rn_snp[i] = snp_ids[i] + "\t" + ref_alleles[i];
            }
         });
      }
      struct JoinOne { // joins the helper before the buffers it writes go out of scope, whichever way this scope is left
         std::thread *t;
         ~JoinOne()
         {
            if (t && t->joinable()) t->join();
         }
      } join_prep{prep_idx != (size_t)-1 ? &helpers.th[prep_idx] : nullptr};
      if (ngpus == 1) {
         fpca_ok(fpca_create_from_bed(&ctx, geno_file.c_str(), N, 0, 0, stand_method_x, device, accum, &nsnps));
         if (join_prep.t && join_prep.t->joinable()) join_prep.t->join();
      } else {
         if (mg.sh->failed.load() == 0) {
            if (fpca_create_from_bed(&ctx, geno_file.c_str(), N, snp_begin, snp_count, stand_method_x, my_device, accum, &nsnps) != FPCA_OK)
               multi_fail(mg, fpca_last_error());
            else if (fpca_set_total_snps(ctx, nsnps) != FPCA_OK)
               multi_fail(mg, fpca_last_error());
         }
         if (!multi_barrier(mg)) return multi_abort();
#ifdef FPCA_TEST_HOOKS
         if (mg.test_transport) {
            // failure injection for tests/test_cli.py (test transport only): rank R kills itself / rank 0 throws after the
            // fork -- the run must end with a message and a non-zero status instead of hanging
            if (const char *kr = FPCA_TEST_ENV("FPCA_CLI_TEST_KILL_RANK")) {
               if (atoi(kr) == mg.rank && mg.rank > 0) raise(SIGKILL);
               if (atoi(kr) == 0 && mg.rank == 0) throw std::runtime_error("injected failure of rank 0 after the fork");
            }
            if (fpca_set_allreduce(ctx, shm_allreduce, &mg) != FPCA_OK || fpca_set_rank(ctx, ngpus, mg.rank) != FPCA_OK)
               multi_fail(mg, fpca_last_error());
            if (mg.test_collectives && fpca_set_collectives(ctx, shm_allgather, shm_reducescatter, &mg) != FPCA_OK) multi_fail(mg, fpca_last_error());
         } else
#endif
         {
            if (mg.rank == 0) {
               if (fpca_comm_unique_id(mg.sh->id) != FPCA_OK) multi_fail(mg, fpca_last_error());
               mg.sh->id_ready.store(1);
            } else
               while (!mg.sh->id_ready.load() && !mg.sh->failed.load()) sched_yield();
            if (mg.sh->failed.load() == 0 && fpca_comm_init_rank(ctx, ngpus, mg.rank, mg.sh->id) != FPCA_OK) multi_fail(mg, fpca_last_error());
         }
         if (!multi_barrier(mg)) return multi_abort();
      }
      phase("device init + .bed upload");
      verbose && std::cout << timestamp() << "Detected BED file: " << geno_file << " with " << N << " samples, " << nsnps << " SNPs." << std::endl;
      if (verbose) {
         char name[256];
         if (fpca_device_name(my_device, name, sizeof(name)) == FPCA_OK) std::cout << timestamp() << "Device " << my_device << ": " << name << std::endl;
         if (ngpus > 1)
            std::cout << timestamp() << ngpus << " GPUs, " << snp_count << " SNPs on this one; transport: "
                      << (mg.test_transport ? "host shared memory (test)" : "RCCL") << std::endl;
      }

      // the reference prints its dense block geometry here (flashpca.cpp:688-690); the whole packed matrix is one resident block
      std::cout << timestamp() << "blocksize: " << nsnps << " (" << (long long)((N + 3) / 4) * (long long)nsnps << " bytes per block)" << std::endl;

      std::vector<double> d, pve, meansd;
      int k_out = n_dim;
      if (mode == MODE_PCA) {
         std::cout << timestamp() << "PCA begin" << std::endl;
         fpca_pca_opts o;
         FPCA_PCA_OPTS_INIT(&o);
         o.ndim = n_dim;
         o.blockvec = blockvec;
         o.maxiter = maxiter;
         o.tol = tol;
         o.divisor = divisor;
         o.do_loadings = do_loadings ? 1 : 0;
         o.max_blocks = maxblocks;
         o.mixed = mixed;
         o.replicated_solver = replicated_solver;
         o.verbose = verbose ? 1 : 0;
         o.seed = (uint64_t)seed;
         d.resize(n_dim);
         pve.resize(n_dim);
         fpca_pca_info info;
         int rc;
         if (ngpus == 1) {
            if (U.empty()) U.resize((size_t)N * n_dim);
            if (Px.empty()) Px.resize((size_t)N * n_dim);
            if (do_loadings && (V.empty() || nsnps != P_file_all)) V.resize((size_t)nsnps * n_dim);
            meansd.resize((size_t)nsnps * 2);
            rc = fpca_pca(ctx, &o, U.data(), d.data(), Px.data(), pve.data(), do_loadings ? V.data() : nullptr, meansd.data(), &info);
         } else {
            // Eigenvectors / PCs: every rank downloads ITS OWN ROWS (its slice of the row-sharded basis, or an even share of the
            // replicated one) straight into the shared region -- no gather of the Ritz blocks, no funnel through rank 0's PCIe
            // link; loadings and mean/sd are this shard's rows and go into the shared region at their place
            const uint64_t P_loc = fpca_nsnps(ctx);
            std::vector<double> Vloc, msloc((size_t)P_loc * 2);
            if (do_loadings) Vloc.resize((size_t)P_loc * n_dim);
            if (mg.rank > 0) o.verbose = 0;
            o.partial_rows = 1;
            rc = fpca_pca(ctx, &o, mg.U, d.data(), mg.Px, pve.data(), do_loadings ? Vloc.data() : nullptr, msloc.data(), &info);
            if (rc != FPCA_OK && rc != FPCA_ENOTCONVERGED) multi_fail(mg, fpca_last_error());
            for (int c = 0; c < n_dim && do_loadings; c++)
               std::memcpy(mg.V + (size_t)c * nsnps + snp_begin, Vloc.data() + (size_t)c * P_loc, P_loc * sizeof(double));
            for (int c = 0; c < 2; c++)
               std::memcpy(mg.meansd + (size_t)c * nsnps + snp_begin, msloc.data() + (size_t)c * P_loc, P_loc * sizeof(double));
            const bool all_ok = multi_barrier(mg);
            if (mg.rank > 0) {
               fpca_destroy(ctx);
               _exit(all_ok ? 0 : 1); // (rank 0 reports "not converged": every rank got the same rc)
            }
            if (!all_ok) return multi_abort();
            wait_children();
            if (do_loadings) {
               V.resize((size_t)nsnps * n_dim);
               std::memcpy(V.data(), mg.V, (size_t)nsnps * n_dim * sizeof(double));
            }
            meansd.assign(mg.meansd, mg.meansd + (size_t)nsnps * 2);
         }
         if (rc == FPCA_ENOTCONVERGED) // randompca.cpp:210-217
            throw std::runtime_error("Spectra eigen-decomposition was not successful, status: not converging");
         fpca_ok(rc);
         verbose && std::cout << timestamp() << "GRM trace: " << info.trace << std::endl;
         verbose && std::cout << timestamp() << info.block_applies << " block applies of width " << info.blockvec << " (" << info.vector_ops
                              << " vector operations), " << info.restarts << " restarts, device " << info.seconds_apply + info.seconds_ortho
                              << " s, host " << info.seconds_host << " s" << std::endl;
         if (verbose && info.cheap_applies > 0)
            std::cout << timestamp() << info.cheap_applies << " of the block applies on " << info.cheap_slices
                      << " byte slices of the operand, verified by exact passes" << std::endl;
         if (verbose && ngpus > 1) {
            static const char *const names[] = {"single", "row-sharded", "replicated", "replicated (the self-test of the row-sharded exchange failed)",
                                                "replicated (a collective of the row-sharded solve failed; started over)"};
            std::cout << timestamp() << "eigensolver layout over " << ngpus << " GPUs: " << names[info.solver_path >= 0 && info.solver_path <= 4 ? info.solver_path : 0]
                      << std::endl;
            uint64_t ccalls = 0, cbytes = 0;
            if (fpca_collective_stats(ctx, &ccalls, &cbytes) == FPCA_OK)
               std::cout << timestamp() << "collectives on the data path: " << ccalls << " calls, " << cbytes << " bytes sent per rank" << std::endl;
         }
         std::cout << timestamp() << "PCA done" << std::endl;
      } else if (mode == MODE_CHECK) {
         // RandomPCA::check(Data&, block_size, evec_file, eval_file) (randompca.cpp:627-661)
         fpca::TextMatrix ev = fpca::read_text(eigvalfile, 1, -1, 0);
         if (ev.rows == 0) throw std::runtime_error("No eigenvalues found in file");
         fpca::TextMatrix evec = fpca::read_text(eigvecfile, 3, -1, 1);
         if (evec.rows != N)
            throw std::runtime_error("Eigenvector dimension doesn't match data dimension (evec.rows = " + std::to_string(evec.rows) +
                                     "; dat.N = " + std::to_string(N) + ")");
         if (ev.rows != evec.cols) throw std::runtime_error("Eigenvector dimension doesn't match the number of eigenvalues");
         const int K = (int)evec.cols;
         std::vector<double> err(K);
         double mse = 0, rmse = 0;
         fpca_ok(fpca_check(ctx, evec.v.data(), (int64_t)N, ev.v.data(), K, divisor, err.data(), &mse, &rmse));
         // printed under --verbose only, like the reference (randompca.cpp:670-700)
         verbose && std::cout << timestamp() << "Checking mean square error between (X X' U) / div and (U D^2) for " << K << " dimensions" << std::endl;
         for (int j = 0; j < K; j++)
            verbose && std::cout << timestamp() << "eval(" << (j + 1) << "): " << ev.v[j] << ", sum squared error: " << err[j] << std::endl;
         verbose && std::cout << timestamp() << "Mean squared error: " << mse << ", Root mean squared error: " << rmse << " (n=" << N << ")" << std::endl;
      } else { // MODE_PROJECT: RandomPCA::project (randompca.cpp:745-820)
         fpca::TextMatrix L = fpca::read_text(in_load_file, 3, -1, 1);
         if (L.rows != nsnps) throw std::runtime_error("number of SNPs in the loadings file doesn't match the data");
         std::vector<double> ms((size_t)nsnps * 2);
         if (!in_maf_file.empty()) {
            std::vector<double> maf = fpca::read_maf(in_maf_file, snp_ids);
            if (maf.size() != nsnps) throw std::runtime_error("number of SNPs in the MAF file doesn't match the data");
            for (uint64_t j = 0; j < nsnps; j++) { // maf2meansd (randompca.cpp:737-743), including its missing sqrt
               ms[j] = maf[j] * 2.0;
               ms[nsnps + j] = maf[j] * 2.0 * (1.0 - maf[j]);
            }
         } else {
            fpca::TextMatrix M2 = fpca::read_text(in_meansd_file, 3, -1, 1);
            if (M2.rows != nsnps || M2.cols < 2) throw std::runtime_error("mean/sd file doesn't match the data");
            for (uint64_t j = 0; j < nsnps; j++) {
               ms[j] = M2.at(j, 0);
               ms[nsnps + j] = M2.at(j, 1);
            }
         }
         fpca_ok(fpca_set_meansd(ctx, ms.data()));
         k_out = (int)L.cols;
         Px.resize((size_t)N * k_out);
         fpca_ok(fpca_apply_x(ctx, L.v.data(), (int64_t)nsnps, k_out, Px.data(), (int64_t)N));
         double div = 1;
         if (divisor == FPCA_DIVISOR_N1) div = (double)N - 1;
         else if (divisor == FPCA_DIVISOR_P) div = (double)L.rows;
         const double s = std::sqrt(div);
         for (double *x = Px.begin(); x != Px.end(); ++x) *x /= s; // randompca.cpp:818
      }
      phase("compute");

      // ---- write out results (flashpca.cpp:755-878) --------------------------------------------------------
      // The files -- eigenvectors, PCs and loadings are 140 + 140 + 28 MB of text at 500,000 x 100,000 -- are written one
      // after the other, each by an in-order writer fed by every CPU this process may use (plink_io.cpp save_text: formatting
      // 22 million numbers IS the output phase; three files at once on a third of the CPUs each measured no faster), while
      // the device context (25 GB to give back) is torn down on another thread.
      const std::vector<std::string> none;
      const unsigned cpus = fpca::usable_cpus();
      if (phase_timing && !quiet) std::fprintf(stderr, "[fpca-cli] usable CPUs: %u\n", cpus);
      std::vector<std::thread> writers;
      std::exception_ptr write_error;
      auto launch = [&](auto fn) {
         writers.emplace_back([&write_error, fn] {
            try {
               fn();
            } catch (...) {
               write_error = std::current_exception();
            }
         });
      };
      auto finish_writers = [&] {
         for (auto &w : writers)
            if (w.joinable()) w.join();
         writers.clear();
         if (write_error) std::rethrow_exception(write_error);
      };
      std::vector<std::string> colnames_u, colnames_pc, cn_load, cn_ms;
      auto sample_rownames = [&] {
         if (rownames.size() == N) return; // (built beside the upload)
         rownames.resize(N);
         const unsigned nt = std::max(1u, std::min(cpus, 8u));
         std::vector<std::thread> th;
         for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
               for (uint64_t i = N * t / nt; i < N * (t + 1) / nt; i++) rownames[i] = fam_ids[i] + "\t" + indiv_ids[i];
            });
         for (auto &x : th) x.join();
      };
      auto snp_rownames = [&] {
         if (!rn_snp.empty()) return;
         rn_snp.resize(snp_ids.size());
         for (size_t i = 0; i < rn_snp.size(); i++) rn_snp[i] = snp_ids[i] + "\t" + ref_alleles[i];
         if (rn_snp.size() != nsnps) throw std::runtime_error("the .bim file has a different number of SNPs than the .bed");
      };
      if (save_meansd && meansd.empty()) { // (--project / --check: the statistics are still on the device)
         meansd.resize((size_t)nsnps * 2);
         fpca_ok(fpca_stats(ctx, meansd.data(), nullptr));
      }
      launch([&] { fpca_destroy(ctx); }); // nothing below needs the device
      try {
         if (mode == MODE_PCA) {
            const unsigned share = cpus;
            std::cout << timestamp() << "Writing " << n_dim << " eigenvalues to file " << eigvalfile << std::endl;
            fpca::save_text(d.data(), n_dim, 1, none, none, eigvalfile, precision);

            std::cout << timestamp() << "Writing " << n_dim << " eigenvectors to file " << eigvecfile << std::endl;
            sample_rownames();
            colnames_u.assign(n_dim + 1, "FID\tIID");
            colnames_pc = colnames_u;
            for (int i = 0; i < n_dim; i++) {
               colnames_u[i + 1] = "U" + std::to_string(i + 1);
               colnames_pc[i + 1] = "PC" + std::to_string(i + 1);
            }
            fpca::save_text(ngpus > 1 ? mg.U : U.data(), N, n_dim, colnames_u, rownames, eigvecfile, precision, share); // (--gpus: the shared region)

            std::cout << timestamp() << "Writing " << n_dim << " PCs to file " << pcfile << std::endl;
            fpca::save_text(ngpus > 1 ? mg.Px : Px.data(), N, n_dim, colnames_pc, rownames, pcfile, precision, share);

            std::cout << timestamp() << "Writing " << n_dim << " proportion variance explained to file " << eigpvefile << std::endl;
            fpca::save_text(pve.data(), n_dim, 1, none, none, eigpvefile, precision);

            if (do_loadings) {
               std::cout << timestamp() << "Writing SNP loadings to file " << loadingsfile << std::endl;
               cn_load = {"SNP\tRefAllele"};
               for (int i = 0; i < n_dim; i++) cn_load.push_back("V" + std::to_string(i + 1));
               snp_rownames();
               fpca::save_text(V.data(), nsnps, n_dim, cn_load, rn_snp, loadingsfile, precision, share);
            }
         } else if (mode == MODE_PROJECT) {
            sample_rownames();
            colnames_pc.assign(k_out + 1, "FID\tIID");
            for (int i = 0; i < k_out; i++) colnames_pc[i + 1] = "PC" + std::to_string(i + 1);
            fpca::save_text(Px.data(), N, k_out, colnames_pc, rownames, projfile, precision);
         }
         if (save_meansd) {
            std::cout << timestamp() << "Writing mean + sd file " << meansdfile << std::endl;
            cn_ms = {"SNP\tRefAllele", "Mean", "SD"};
            snp_rownames();
            fpca::save_text(meansd.data(), nsnps, 2, cn_ms, rn_snp, meansdfile, precision);
         }
      } catch (...) {
         for (auto &w : writers)
            if (w.joinable()) w.join(); // (they hold references to this scope)
         throw;
      }
      finish_writers();
      phase("output files + teardown");
      std::cout << timestamp() << "Goodbye!" << std::endl;
      // every file is closed and the context destroyed: leave without running the HIP runtime's static destructors
      // (tens of milliseconds of unloading code objects and tearing down queues that nobody waits for)
      std::cout.flush();
      std::fflush(nullptr);
      _exit(EXIT_SUCCESS);
   } catch (std::exception &e) {
      std::cerr << timestamp() << "Exception: " << e.what() << std::endl;
      std::cerr << timestamp() << "Terminating" << std::endl;
      if (g_child_rank > 0) {
         if (g_shared) g_shared->failed.fetch_add(1);
         _exit(1);
      }
      abandon_children(); // rank 0 of a --gpus run: the others must not wait for it
      return EXIT_FAILURE;
   } catch (...) {
      std::cerr << timestamp() << "Caught unknown exception, terminating " << std::endl;
      if (g_child_rank > 0) {
         if (g_shared) g_shared->failed.fetch_add(1);
         _exit(1);
      }
      abandon_children();
      return EXIT_FAILURE;
   }
   return EXIT_SUCCESS;
}
