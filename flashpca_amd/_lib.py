"""ctypes binding of libfpca.so (include/fpca.h).  Plumbing only: every call goes straight through the C ABI.

The library is built in-tree (flashpca_amd/_build/libfpca.so) by `make -C flashpca_amd/csrc` (see
__graft_entry__.build()).  There is no fallback: if the library is missing this module raises, and if no
gfx950 device is usable every fpca_create* call fails with FPCA_ENODEVICE.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPCA_LIB") or os.path.join(HERE, "_build", "libfpca.so")  # FPCA_LIB: A/B-test another build
if LIB_PATH == "testhooks":  # shorthand used by the scripts that drive an environment switch
    LIB_PATH = os.path.join(HERE, "_build", "testhooks", "libfpca.so")
CLI_PATH = os.path.join(HERE, "_build", "flashpca")
# the same sources compiled with -DFPCA_TEST_HOOKS: the only binaries in which the FPCA_I8_MODE / FPCA_AR_CHUNKS /
# FPCA_DEBUG_* / FPCA_CLI_TEST_* environment switches exist (csrc/common.hpp).  Tests that need a switch load these.
HOOKS_LIB_PATH = os.path.join(HERE, "_build", "testhooks", "libfpca.so")
HOOKS_CLI_PATH = os.path.join(HERE, "_build", "testhooks", "flashpca")
CSRC = os.path.join(HERE, "csrc")

STANDARDISE = {"binom": 2, "binom2": 3}
STANDARDISE_DENSE = {"none": 0, "sd": 1, "binom": 2, "binom2": 3, "center": 4}
DIVISOR = {"none": 0, "n1": 1, "p": 2}
UNIQUE_ID_BYTES = 128

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)  # all-gather / reduce-scatter


class PcaOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("info_size", C.c_uint32),
        ("ndim", C.c_int),
        ("blockvec", C.c_int),
        ("maxiter", C.c_int),
        ("tol", C.c_double),
        ("divisor", C.c_int),
        ("do_loadings", C.c_int),
        ("max_blocks", C.c_int),
        ("verbose", C.c_int),
        ("seed", C.c_uint64),
        ("replicated_solver", C.c_int),
        ("max_applies", C.c_int),
        ("mixed", C.c_int),
        ("cheap_slices", C.c_int),
        ("partial_rows", C.c_int),
    ]


class SynthModel(C.Structure):
    _fields_ = [("n_pop", C.c_int), ("fst", C.c_double), ("missing_rate", C.c_double), ("maf_model", C.c_int), ("missing_model", C.c_int),
                ("conc_frac", C.c_double), ("lognormal_sigma", C.c_double)]


class PcaInfo(C.Structure):
    _fields_ = [
        ("converged", C.c_int),
        ("block_applies", C.c_int),
        ("vector_ops", C.c_int),
        ("restarts", C.c_int),
        ("blockvec", C.c_int),
        ("trace", C.c_double),
        ("max_residual", C.c_double),
        ("seconds_apply", C.c_double),
        ("seconds_ortho", C.c_double),
        ("seconds_host", C.c_double),
        ("seconds_total", C.c_double),
        ("seconds_download", C.c_double),
        ("seconds_post", C.c_double),
        ("cheap_applies", C.c_int),
        ("cheap_slices", C.c_int),
        ("seconds_exact", C.c_double),
        ("solver_path", C.c_int),
    ]


SOLVER_PATH = {0: "single", 1: "rowshard", 2: "replicated", 3: "replicated (self-test of the row-sharded exchange failed)",
               4: "replicated (a collective of the row-sharded solve failed)"}


class BenchResult(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_xt", C.c_double),
        ("ms_x", C.c_double),
        ("ms_allreduce", C.c_double),
        ("flops_per_step", C.c_double),
        ("packed_bytes_per_step", C.c_double),
        ("ms_gemm_xt", C.c_double),
        ("ms_gemm_x", C.c_double),
    ]


# every function include/fpca.h declares: name -> (restype, argtypes)
_P, _U64, _I, _D = C.c_void_p, C.c_uint64, C.c_int, C.c_double
SIGNATURES = {
    "fpca_last_error": (C.c_char_p, []),
    "fpca_version": (C.c_char_p, []),
    "fpca_abi_version": (_I, []),
    "fpca_device_count": (_I, []),
    "fpca_device_name": (_I, [_I, C.c_char_p, _I]),
    "fpca_warmup": (_I, [_I]),
    "fpca_create": (_I, [C.POINTER(_P), _P, _U64, _U64, _I, _I, _I]),
    "fpca_create_from_bed": (_I, [C.POINTER(_P), C.c_char_p, _U64, _U64, _U64, _I, _I, _I, C.POINTER(_U64)]),
    "fpca_create_synthetic": (_I, [C.POINTER(_P), _U64, _U64, _U64, _U64, _I, _D, _D, _I, _I, _I]),
    "fpca_create_synthetic_model": (_I, [C.POINTER(_P), _U64, _U64, _U64, _U64, C.POINTER(SynthModel), _I, _I, _I]),
    "fpca_create_dense": (_I, [C.POINTER(_P), _P, C.c_int64, _U64, _U64, _I, _I]),
    "fpca_destroy": (None, [_P]),
    "fpca_nsamples": (_U64, [_P]),
    "fpca_nsnps": (_U64, [_P]),
    "fpca_accum": (_I, [_P]),
    "fpca_missing_mode": (_I, [_P, _I]),
    "fpca_allreduce_chunks": (_I, [_P]),
    "fpca_download_packed": (_I, [_P, _P]),
    "fpca_stats": (_I, [_P, _P, C.POINTER(_D)]),
    "fpca_set_meansd": (_I, [_P, _P]),
    "fpca_apply_xxt": (_I, [_P, _P, C.c_int64, _I, _P, C.c_int64]),
    "fpca_apply_xt": (_I, [_P, _P, C.c_int64, _I, _P, C.c_int64]),
    "fpca_apply_x": (_I, [_P, _P, C.c_int64, _I, _P, C.c_int64]),
    "fpca_block_rows": (_U64, [_P]),
    "fpca_apply_xxt_dev": (_I, [_P, _P, _I, _P, _P]),
    "fpca_stream": (_P, [_P]),
    "fpca_synchronize": (_I, [_P]),
    "fpca_comm_unique_id": (_I, [_P]),
    "fpca_comm_init_rank": (_I, [_P, _I, _I, _P]),
    "fpca_set_allreduce": (_I, [_P, ALLREDUCE_FN, _P]),
    "fpca_set_collectives": (_I, [_P, COLLECTIVE_FN, COLLECTIVE_FN, _P]),
    "fpca_set_total_snps": (_I, [_P, _U64]),
    "fpca_set_rank": (_I, [_P, _I, _I]),
    "fpca_collective_stats": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "fpca_pca_init_opts": (None, [C.POINTER(PcaOpts), C.c_size_t, C.c_size_t]),
    "fpca_pca_row_ranges": (_I, [_P, C.POINTER(PcaOpts), C.POINTER(_U64), _I]),
    "fpca_pca": (_I, [_P, C.POINTER(PcaOpts), _P, _P, _P, _P, _P, _P, C.POINTER(PcaInfo)]),
    "fpca_check": (_I, [_P, _P, C.c_int64, _P, _I, _I, _P, C.POINTER(_D), C.POINTER(_D)]),
    "fpca_bench_apply": (_I, [_P, _I, _I, _I, C.POINTER(BenchResult)]),
    "fpca_profile_begin": (_I, [_P, _I]),
    "fpca_profile_sample_every": (_I, [_P, _I]),
    "fpca_profile_end": (_I, [_P, _I, C.POINTER(BenchResult), C.POINTER(_I)]),
    "fpca_bench_stats": (_I, [_P, _I, C.POINTER(_D), C.POINTER(_D)]),
    "fpca_debug_mfma_probe": (_I, [_P, _P, _P]),
    "fpca_debug_mfma_i8_probe": (_I, [_P, _P, _P]),
    "fpca_debug_mfma_peak": (_I, [_I, _I, _I, C.POINTER(_D)]),
    "fpca_debug_census": (_I, [_I, _U64, _P]),
    "fpca_debug_k4": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "fpca_debug_variant": (_I, [_I, _I]),
    "fpca_debug_k4_bench": (_I, [_P, _I, _I, _I, C.POINTER(_D), C.POINTER(_D)]),
    "fpca_debug_k4_fused": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "fpca_debug_k4_fused_bench": (_I, [_P, _I, _I, _I, C.POINTER(_D)]),
}

ABI_VERSION = 4  # FPCA_ABI_VERSION of the include/fpca.h the structures above mirror

_lib = None
_loaded = {}


def _load(path):
    if path not in _loaded:
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found -- build it with `make -C flashpca_amd/csrc` (or __graft_entry__.build()); this package has "
                "no CPU fallback" % path
            )
        L = C.CDLL(path)
        try:
            L.fpca_abi_version.restype = C.c_int
            abi = L.fpca_abi_version()
        except AttributeError:
            abi = 1
        if abi != ABI_VERSION:  # (an older or newer build behind FPCA_LIB: its structs have other sizes)
            raise RuntimeError("%s speaks ABI version %d, this binding %d -- rebuild it (make -C flashpca_amd/csrc)" % (path, abi, ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _loaded[path] = L
    return _loaded[path]


class test_hooks:
    """Context manager: inside it lib() -- and with it every Context / flashpca() call -- goes to the -DFPCA_TEST_HOOKS build
    of the library, where the environment test switches exist.  Contexts must be created AND closed inside the block."""

    def __enter__(self):
        global _lib
        self._prev = _lib
        _lib = _load(HOOKS_LIB_PATH)
        return _lib

    def __exit__(self, *a):
        global _lib
        _lib = self._prev


def build(verbose=False):
    """Compile libfpca.so and the flashpca CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load the C-ABI library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


class FpcaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fpca error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise FpcaError(rc, lib().fpca_last_error().decode(errors="replace"))
