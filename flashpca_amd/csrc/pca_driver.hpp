// pca_driver.hpp -- RandomPCA::pca_fast(Data&, ...) (randompca.cpp:168-218) on top of a BlockBackend:
// eigensolve, then d = lambda/div, pve = d/trace, Px = U diag(sqrt(d)), optional loadings.
#pragma once
#include <vector>

#include "../../include/fpca.h"
#include "backend.hpp"

namespace fpca {

// block width for ndim wanted components: requested (validated) or automatic: 16 (32 / 64 for ndim > 64 / > 128)
int choose_blockvec(int ndim, int requested);

struct PcaOutputs {
   double *U = nullptr, *d = nullptr, *Px = nullptr, *pve = nullptr;
   bool partial_rows = false; // several ranks: each writes only its own rows of U / Px (fpca_pca_opts.partial_rows)
};

// Runs the solver on `be` (whose width must equal choose_blockvec(...)).  N_div/P_div are the N and the TOTAL
// SNP count used by the divisor (randompca.cpp:180-184).  On return *ritz_blocks (if non-null) holds the backend
// blocks with the eigenvectors, b per block (caller frees them; used for the loadings), otherwise they are freed here.
// Returns FPCA_OK or FPCA_ENOTCONVERGED (outputs are still filled with the current Ritz pairs).
int run_pca(BlockBackend &be, const fpca_pca_opts &o, uint64_t P_div, const PcaOutputs &out, fpca_pca_info *info,
            std::vector<int> *ritz_blocks, double *div_out);

} // namespace fpca
