"""numpy-level mirror of the C ABI (include/fpca.h) -- used by the tests, bench.py and smoke().

`Context` is one SNP shard of a genotype matrix resident on one MI355X.  `flashpca()` mirrors the reference's
scripting entry point (the R function flashpca(), flashpcaR/R/flashpca.R:99-204: same argument names and the same
result fields values / vectors / projection / loadings / center / scale / pve) because no R toolchain exists here;
the compiled drop-in is the `flashpca` CLI built from flashpca_amd/csrc/cli_main.cpp.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DIVISOR, STANDARDISE, BenchResult, FpcaError, PcaInfo, PcaOpts, check, lib  # noqa: F401


ACCUM = {"auto": 0, 0: 0, "fp64": 64, "fp32": 32, 64: 64, 32: 32, "i8": 808}
ACCUM.update({"i8x%d" % s: 800 + s for s in range(2, 10)})
ACCUM.update({800 + s: 800 + s for s in range(2, 10)})


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def count_fam_rows(path):
    """N = number of newline-terminated lines of the .fam (the reference drops an unterminated last line,
    data.cpp:526)."""
    with open(path, "rb") as f:
        return f.read().count(b"\n")


class Context:
    def __init__(self, handle):
        self.h = handle
        L = lib()
        self.N = int(L.fpca_nsamples(handle))
        self.P = int(L.fpca_nsnps(handle))
        self.P_total = self.P
        self._keep = []

    @property
    def accum(self):
        """The arithmetic mode in effect: 'fp64', 'fp32' or 'i8xS' ("auto" resolved)."""
        code = int(lib().fpca_accum(self.h))
        return {64: "fp64", 32: "fp32"}.get(code, "i8x%d" % (code - 800))

    # ---- constructors -----------------------------------------------------------------------------
    @classmethod
    def from_packed(cls, packed, N, P, stand="binom2", device=0, accum="fp64"):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        assert packed.size >= ((N + 3) // 4) * P
        h = C.c_void_p()
        check(lib().fpca_create(C.byref(h), _p(packed), N, P, STANDARDISE[stand], device, ACCUM[accum]))
        return cls(h)

    @classmethod
    def from_bed(cls, bed_path, N, snp_begin=0, P=0, stand="binom2", device=0, accum="fp64"):
        h = C.c_void_p()
        ptot = C.c_uint64(0)
        check(lib().fpca_create_from_bed(C.byref(h), bed_path.encode(), N, snp_begin, P, STANDARDISE[stand], device, ACCUM[accum],
                                         C.byref(ptot)))
        c = cls(h)
        c.P_total = int(ptot.value)
        check(lib().fpca_set_total_snps(h, c.P_total))
        return c

    @classmethod
    def from_dense(cls, X, stand="binom2", device=0):
        """In-memory N x P fp64 matrix (NaN = missing), standardised on the GPU like standardise() (util.cpp:24-192)."""
        X = np.asfortranarray(X, dtype=np.float64)
        h = C.c_void_p()
        check(lib().fpca_create_dense(C.byref(h), _p(X), X.shape[0], X.shape[0], X.shape[1], _lib.STANDARDISE_DENSE[stand], device))
        return cls(h)

    @classmethod
    def synthetic(cls, N, P, snp_begin=0, seed=20260928, n_pop=40, fst=0.05, missing_rate=0.001, stand="binom2", device=0,
                  accum="fp64"):
        h = C.c_void_p()
        check(lib().fpca_create_synthetic(C.byref(h), N, snp_begin, P, seed, n_pop, fst, missing_rate, STANDARDISE[stand],
                                          device, ACCUM[accum]))
        return cls(h)

    def close(self):
        if self.h:
            lib().fpca_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data / statistics --------------------------------------------------------------------------
    def download_packed(self):
        out = np.empty(((self.N + 3) // 4) * self.P, dtype=np.uint8)
        check(lib().fpca_download_packed(self.h, _p(out)))
        return out

    def stats(self):
        ms = np.empty((self.P, 2), order="F")
        tr = C.c_double(0)
        check(lib().fpca_stats(self.h, _p(ms), C.byref(tr)))
        return ms, tr.value

    def set_meansd(self, meansd):
        m = np.asfortranarray(meansd, dtype=np.float64)
        check(lib().fpca_set_meansd(self.h, _p(m)))

    def set_total_snps(self, P_total):
        check(lib().fpca_set_total_snps(self.h, int(P_total)))
        self.P_total = int(P_total)

    # ---- operator -------------------------------------------------------------------------------------
    def apply_xxt(self, B):
        B = np.asfortranarray(np.atleast_2d(B.T).T if B.ndim == 1 else B, dtype=np.float64)
        Y = np.empty_like(B, order="F")
        check(lib().fpca_apply_xxt(self.h, _p(B), B.shape[0], B.shape[1], _p(Y), Y.shape[0]))
        return Y

    def apply_xt(self, B):
        B = np.asfortranarray(B.reshape(self.N, -1), dtype=np.float64)
        T = np.empty((self.P, B.shape[1]), order="F")
        check(lib().fpca_apply_xt(self.h, _p(B), B.shape[0], B.shape[1], _p(T), self.P))
        return T

    def apply_x(self, T):
        T = np.asfortranarray(T.reshape(self.P, -1), dtype=np.float64)
        Y = np.empty((self.N, T.shape[1]), order="F")
        check(lib().fpca_apply_x(self.h, _p(T), self.P, T.shape[1], _p(Y), self.N))
        return Y

    # ---- multi-GPU -------------------------------------------------------------------------------------
    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, count:int, stream:int) -> 0 on success; sums `count` fp64 in place across ranks."""
        cb = _lib.ALLREDUCE_FN(lambda user, ptr, count, stream: int(pyfunc(ptr, count, stream) or 0))
        self._keep.append(cb)
        check(lib().fpca_set_allreduce(self.h, cb, None))

    def comm_init_rank(self, nranks, rank, unique_id):
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(lib().fpca_comm_init_rank(self.h, nranks, rank, buf))

    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        check(lib().fpca_comm_unique_id(buf))
        return bytes(buf)

    # ---- driver ---------------------------------------------------------------------------------------
    def pca(self, ndim=10, tol=1e-6, maxiter=500, div="p", do_loadings=False, blockvec=0, max_blocks=0, seed=1, verbose=0,
            allow_unconverged=False):
        o = PcaOpts()
        lib().fpca_pca_default_opts(C.byref(o))
        o.ndim, o.tol, o.maxiter, o.divisor = ndim, tol, maxiter, DIVISOR[div]
        o.do_loadings, o.blockvec, o.max_blocks, o.seed, o.verbose = int(do_loadings), blockvec, max_blocks, seed, verbose
        U = np.empty((self.N, ndim), order="F")
        d = np.empty(ndim)
        Px = np.empty((self.N, ndim), order="F")
        pve = np.empty(ndim)
        V = np.empty((self.P, ndim), order="F") if do_loadings else None
        ms = np.empty((self.P, 2), order="F")
        info = PcaInfo()
        rc = lib().fpca_pca(self.h, C.byref(o), _p(U), _p(d), _p(Px), _p(pve), _p(V), _p(ms), C.byref(info))
        if rc != 0 and not (allow_unconverged and rc == -5):
            check(rc)
        return dict(U=U, d=d, Px=Px, pve=pve, V=V, meansd=ms, info={f[0]: getattr(info, f[0]) for f in PcaInfo._fields_})

    def check(self, evec, evals, div="p"):
        evec = np.asfortranarray(evec, dtype=np.float64)
        evals = np.ascontiguousarray(evals, dtype=np.float64)
        k = evec.shape[1]
        err = np.empty(k)
        mse, rmse = C.c_double(0), C.c_double(0)
        check(lib().fpca_check(self.h, _p(evec), evec.shape[0], _p(evals), k, DIVISOR[div], _p(err), C.byref(mse), C.byref(rmse)))
        return err, mse.value, rmse.value

    # ---- measurement -----------------------------------------------------------------------------------
    def bench_apply(self, b=32, steps=10, warmup=2):
        r = BenchResult()
        check(lib().fpca_bench_apply(self.h, b, steps, warmup, C.byref(r)))
        return {f[0]: getattr(r, f[0]) for f in BenchResult._fields_}

    def block_rows(self):
        return int(lib().fpca_block_rows(self.h))

    def apply_xxt_dev(self, dB_ptr, b, dY_ptr, stream=None):
        """Device-resident operator on row-major [block_rows][b] fp64 blocks given by raw device pointers."""
        check(lib().fpca_apply_xxt_dev(self.h, C.c_void_p(dB_ptr), b, C.c_void_p(dY_ptr), C.c_void_p(stream) if stream else None))

    def synchronize(self):
        check(lib().fpca_synchronize(self.h))

    def profile_begin(self, max_steps):
        check(lib().fpca_profile_begin(self.h, max_steps))

    def profile_end(self, b):
        r = BenchResult()
        n = C.c_int(0)
        check(lib().fpca_profile_end(self.h, b, C.byref(r), C.byref(n)))
        out = {f[0]: getattr(r, f[0]) for f in BenchResult._fields_}
        out["nsteps"] = n.value
        return out

    def bench_stats(self, reps=5):
        ms, by = C.c_double(0), C.c_double(0)
        check(lib().fpca_bench_stats(self.h, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value


def flashpca(X, ndim=10, stand="binom2", divisor="p", maxiter=500, tol=1e-6, do_loadings=False, return_scale=True,
             device=0, verbose=False, accum="auto", **solver_kw):
    """PCA of a PLINK fileset; mirrors flashpca() of the reference's R package for the PLINK-prefix input
    (flashpcaR/R/flashpca.R:99-204 -> flashpca_plink_internal, flashpcaR/src/flashpca.cpp:96-197).

    X: PLINK root name (X.bed / X.bim / X.fam), or a numeric N x P matrix (NaN = missing; the R function's matrix
    input, flashpcaR/src/flashpca.cpp:17-93, which also accepts stand = "sd" / "center" / "none").
    Returns values, vectors, projection, loadings, center, scale, pve.
    """
    if divisor not in DIVISOR:
        raise ValueError("divisor must be one of %s" % sorted(DIVISOR))
    if isinstance(X, str):
        if stand not in STANDARDISE:
            raise ValueError("stand must be one of %s" % sorted(STANDARDISE))  # R: match.arg
        N = count_fam_rows(X + ".fam")
        ctx = Context.from_bed(X + ".bed", N, stand=stand, device=device, accum=accum)
    else:
        if stand not in _lib.STANDARDISE_DENSE:
            raise ValueError("stand must be one of %s" % sorted(_lib.STANDARDISE_DENSE))
        ctx = Context.from_dense(np.asarray(X, dtype=np.float64), stand=stand, device=device)
    with ctx:
        r = ctx.pca(ndim=ndim, tol=tol, maxiter=maxiter, div=divisor, do_loadings=do_loadings, verbose=int(verbose),
                    **solver_kw)
    res = dict(values=r["d"], vectors=r["U"], projection=r["Px"], loadings=r["V"], pve=r["pve"], info=r["info"])
    if return_scale:
        res["center"] = r["meansd"][:, 0]
        res["scale"] = r["meansd"][:, 1]
    return res
