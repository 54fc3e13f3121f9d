"""flashpca_amd -- MI355X-native implementation of flashpca's PCA hot path.

The product is the C-ABI library flashpca_amd/_build/libfpca.so (include/fpca.h; hand-written gfx950 HIP kernels
+ C++ host eigensolver) and the drop-in `flashpca` CLI next to it.  This package is the thin Python mirror used
by tests and bench.py.  There is no CPU fallback.
"""
from ._lib import LIB_PATH, CLI_PATH, HOOKS_LIB_PATH, HOOKS_CLI_PATH, build, lib, test_hooks, FpcaError  # noqa: F401
from .api import Context, flashpca, project, count_fam_rows  # noqa: F401
from .api import check_pca as check  # noqa: F401  (R: check())

__version__ = "0.1.0"
