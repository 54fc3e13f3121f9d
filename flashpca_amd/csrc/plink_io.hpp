// plink_io.hpp -- text-side I/O of the flashpca drop-in CLI: .fam/.bim readers, the whitespace matrix reader used by
// --check / --project, and the tab-separated writers (byte-compatible with the reference's save_text).
#pragma once
#include "common.hpp"
#include <cstdint>
#include <string>
#include <vector>

namespace fpca {

// column-major rows x cols matrix
struct TextMatrix {
   uint64_t rows = 0, cols = 0;
   std::vector<double> v;
   double &at(uint64_t i, uint64_t j) { return v[i + j * rows]; }
   double at(uint64_t i, uint64_t j) const { return v[i + j * rows]; }
};

// read_text (data.cpp:504-586): numeric columns firstcol.. (one-based) of a whitespace-separated file, skipping
// `skip` lines; a last line without '\n' is dropped; throws std::runtime_error with the reference's messages.
TextMatrix read_text(const std::string &filename, unsigned firstcol, long nrows = -1, unsigned skip = 0);

// read_plink_fam (data.cpp:639-672) / read_plink_bim (data.cpp:589-637)
void read_plink_fam(const std::string &filename, std::vector<std::string> &fam_ids, std::vector<std::string> &indiv_ids);
void read_plink_bim(const std::string &filename, std::vector<std::string> &snp_ids, std::vector<std::string> &ref_alleles,
                    std::vector<std::string> &alt_alleles);
// read_MAF (data.cpp:419-500): MAF column of a plink .frq whose SNP ids must match snp_ids
std::vector<double> read_maf(const std::string &filename, const std::vector<std::string> &snp_ids);

// What the CLI needs from the .fam in ONE pass over the file: read_text(fam, 6) -- N = its row count, every field from the
// 6th on must parse as a number, every line must have the same number of fields (flashpca.cpp:589 -> data.cpp:408-413,
// 504-586) -- and read_plink_fam's two id columns (data.cpp:639-672); same errors as the two calls in that order.
uint64_t read_fam(const std::string &filename, std::vector<std::string> &fam_ids, std::vector<std::string> &indiv_ids);

// (usable_cpus(): common.hpp -- the library sizes its own helper threads with it too)

// save_text (util.h:69-108): header line (if colnames non-empty), then per row  [rowname TAB] v1 TAB v2 ...
// numbers through operator<< with std::setprecision(precision).  M is column-major rows x cols (ld = rows).
// max_threads: formatting threads (0 = usable_cpus()); the bytes do not depend on it.
bool save_text(const double *M, uint64_t rows, uint64_t cols, const std::vector<std::string> &colnames,
               const std::vector<std::string> &rownames, const std::string &filename, unsigned precision = 7,
               unsigned max_threads = 0);

} // namespace fpca
