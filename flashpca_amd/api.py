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


ACCUM = {"auto": 0, 0: 0, "fp64": 64, "fp32": 32, 64: 64, 32: 32, "i8": 807}
ACCUM.update({"i8x%d" % s: 800 + s for s in range(2, 9)})
ACCUM.update({800 + s: 800 + s for s in range(2, 9)})


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

    def missing_mode(self, b=32):
        """How the exact-integer path treats the missing-call indicator (fpca_missing_mode): 0 full, 1 skip empty blocks,
        2 nothing missing, 3 sparse gathers, 4 hybrid (sparse gathers + a dense sub-matrix for the SNPs that hold most of the
        missing calls); -1 for the fp64 / fp32 kernels."""
        return int(lib().fpca_missing_mode(self.h, b))

    def allreduce_chunks(self):
        """Row chunks of Y whose all-reduce overlaps the computation of the next chunk (fpca_allreduce_chunks)."""
        return int(lib().fpca_allreduce_chunks(self.h))

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
                  accum="fp64", realistic=False, maf_model=None, missing_model=None, conc_frac=0.05, lognormal_sigma=0.0):
        """realistic=True: the round-4 profile (synth.hpp) -- rare-variant allele-frequency spectrum, missing calls concentrated in
        5 % of the SNPs (maf_model / missing_model = 1; either can be chosen alone)."""
        h = C.c_void_p()
        m = _lib.SynthModel(n_pop, fst, missing_rate, int(realistic if maf_model is None else maf_model),
                            int(realistic if missing_model is None else missing_model), conc_frac, lognormal_sigma)
        check(lib().fpca_create_synthetic_model(C.byref(h), N, snp_begin, P, seed, C.byref(m), STANDARDISE[stand], device, ACCUM[accum]))
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

    def set_collectives(self, allgather, reducescatter):
        """allgather(send_ptr, recv_ptr, count_per_rank, stream) / reducescatter(send_ptr, recv_ptr, count_per_rank, stream) -> 0:
        fp64 collectives on raw device pointers beside set_allreduce (fpca_set_collectives); the row-sharded solver then issues
        the call sequence it issues over RCCL."""
        ag = _lib.COLLECTIVE_FN(lambda user, snd, rcv, count, stream: int(allgather(snd, rcv, count, stream) or 0))
        rs = _lib.COLLECTIVE_FN(lambda user, snd, rcv, count, stream: int(reducescatter(snd, rcv, count, stream) or 0))
        self._keep += [ag, rs]
        check(lib().fpca_set_collectives(self.h, ag, rs, None))

    def set_rank(self, nranks, rank):
        """rank / size beside a caller-supplied all-reduce: lets fpca_pca row-shard the solver (fpca_set_rank)."""
        check(lib().fpca_set_rank(self.h, nranks, rank))

    def collective_stats(self):
        calls, nbytes = C.c_uint64(0), C.c_uint64(0)
        check(lib().fpca_collective_stats(self.h, C.byref(calls), C.byref(nbytes)))
        return int(calls.value), int(nbytes.value)

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
            allow_unconverged=False, max_applies=0, replicated_solver=False, mixed=0, cheap_slices=0, partial_rows=False):
        """fpca_pca.  allow_unconverged: FPCA_ENOTCONVERGED comes back as a result (info["converged"] == 0) whose U / d / Px / pve hold
        the current Rayleigh-Ritz pairs (pca_driver.hpp) instead of raising.  partial_rows (several ranks): U / Px carry only this
        rank's rows (result["row_ranges"]), NaN elsewhere."""
        o = PcaOpts()
        lib().fpca_pca_init_opts(C.byref(o), C.sizeof(PcaOpts), C.sizeof(PcaInfo))
        o.ndim, o.tol, o.maxiter, o.divisor = ndim, tol, maxiter, DIVISOR[div]
        o.do_loadings, o.blockvec, o.max_blocks, o.seed, o.verbose = int(do_loadings), blockvec, max_blocks, seed, verbose
        o.max_applies = max_applies
        o.replicated_solver = int(replicated_solver)
        o.mixed, o.cheap_slices = int(mixed), int(cheap_slices)  # mixed: 0 automatic (on), 1 on, -1 off (every pass exact)
        o.partial_rows = int(partial_rows)
        # (the small outputs are NaN-filled: what the library does not write can never pass for a result; the N x ndim ones only
        #  under partial_rows, where another rank's rows stay unwritten -- filling 160 MB costs a timed solve 8 ms of page faults,
        #  and a call that fails raises, FPCA_ENOTCONVERGED fills everything)
        mk = (lambda shape: np.full(shape, np.nan, order="F")) if partial_rows else (lambda shape: np.empty(shape, order="F"))
        U = mk((self.N, ndim))
        d = np.full(ndim, np.nan)
        Px = mk((self.N, ndim))
        pve = np.full(ndim, np.nan)
        V = np.empty((self.P, ndim), order="F") if do_loadings else None
        ms = np.empty((self.P, 2), order="F")
        info = PcaInfo()
        rc = lib().fpca_pca(self.h, C.byref(o), _p(U), _p(d), _p(Px), _p(pve), _p(V), _p(ms), C.byref(info))
        if rc != 0 and not (allow_unconverged and rc == -5):
            check(rc)
        out = dict(U=U, d=d, Px=Px, pve=pve, V=V, meansd=ms, info={f[0]: getattr(info, f[0]) for f in PcaInfo._fields_})
        if partial_rows:
            rg = (C.c_uint64 * 16)()
            n = lib().fpca_pca_row_ranges(self.h, C.byref(o), rg, 8)
            if n < 0:
                check(n)
            out["row_ranges"] = [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(n)]
        return out

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

    def profile_begin(self, max_steps, sample_every=1):
        check(lib().fpca_profile_sample_every(self.h, sample_every))
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


def _read_bim(prefix):
    rows = [l.split() for l in open(prefix + ".bim").read().splitlines() if l.strip()]
    return [r[1] for r in rows], [r[4] for r in rows]


def project(X, loadings, orig_mean=None, orig_sd=None, ref_alleles=None, divisor="p", device=0, check_bim=True):
    """Project samples onto existing principal components; mirrors project() of the reference's R package
    (flashpcaR/R/project.R:56-163): same arguments, same input checks (raised as ValueError), same result
    `{"projection": Z V / sqrt(div)}` with Z standardised by orig_mean / orig_sd and missing -> 0.

    X: PLINK root name, or a numeric N x P matrix (NaN = missing).  ref_alleles: mapping SNP name -> reference allele in
    .bim order (the R function's named character vector); required for the PLINK input unless check_bim=False.
    """
    if divisor not in DIVISOR:
        raise ValueError("divisor must be one of %s" % sorted(DIVISOR))
    if orig_mean is None:
        raise ValueError("The vector of means used for standardising the data must be provided via 'orig_mean'")
    if orig_sd is None:
        raise ValueError("The vector of standard deviations used for standardising the data must be provided via 'orig_sd'")
    loadings = np.asarray(loadings, dtype=np.float64)
    orig_mean = np.asarray(orig_mean, dtype=np.float64).ravel()
    orig_sd = np.asarray(orig_sd, dtype=np.float64).ravel()
    if isinstance(X, str):
        snp, ref = _read_bim(X)
        p = len(snp)
        if check_bim:
            if loadings.ndim != 2 or loadings.shape[0] != p:
                raise ValueError("The number of rows in %s.bim and the number of columns in the loadings don't match" % X)
            if ref_alleles is None or list(ref_alleles.keys()) != snp:
                raise ValueError("The SNP names in %s.bim do not match the names of the ref_alleles vector" % X)
            if list(ref_alleles.values()) != ref:
                raise ValueError("The reference alleles in %s.bim do not match the ref_alleles vector" % X)
            if orig_mean.size != p:
                raise ValueError("The number of rows in %s.bim and the length of orig_mean don't match" % X)
            if orig_sd.size != p:
                raise ValueError("The number of rows in %s.bim and the length of orig_sd don't match" % X)
            if np.any(orig_sd <= 0):
                raise ValueError("orig_sd cannot be zero or negative")
        n = count_fam_rows(X + ".fam")
        ctx = Context.from_bed(X + ".bed", n, device=device, accum="auto")
        with ctx:
            ctx.set_meansd(np.column_stack([orig_mean, orig_sd]))
            Z = ctx.apply_x(np.asfortranarray(loadings))
    else:
        Xm = np.asarray(X, dtype=np.float64)
        if loadings.ndim != 2 or loadings.shape[0] != Xm.shape[1]:
            raise ValueError("The number of rows in X and number of columns of the loadings don't match")
        if orig_mean.size != Xm.shape[1]:
            raise ValueError("The number of rows in X and length of orig_mean don't match")
        if orig_sd.size != Xm.shape[1]:
            raise ValueError("The number of rows in X and length of orig_sd don't match")
        if np.any(orig_sd <= 0):
            raise ValueError("orig_sd cannot be zero or negative")
        if np.isnan(Xm).any():
            import warnings

            warnings.warn("X contains missing values, will be mean imputed")
        n = Xm.shape[0]
        S = (Xm - orig_mean) / orig_sd  # R: scale(X, center, scale); the product itself runs on the GPU
        S[np.isnan(S)] = 0.0
        with Context.from_dense(S, stand="none", device=device) as ctx:
            Z = ctx.apply_x(np.asfortranarray(loadings))
    div_val = {"p": loadings.shape[0], "n1": n, "none": 1}[divisor]  # project.R:137-142 (as written there)
    return dict(projection=Z / np.sqrt(div_val))


def check_pca(X, evec, eval, stand="binom2", divisor="p", device=0, check_fam=True):  # noqa: A002 (R's argument name)
    """Check the accuracy of an eigen-decomposition; mirrors check() of the reference's R package
    (flashpcaR/R/check.R): err_j = || X X' u_j / div - d_j u_j ||^2, mse, rmse (RandomPCA::check, randompca.cpp:663-703)."""
    if divisor not in DIVISOR:
        raise ValueError("divisor must be one of %s" % sorted(DIVISOR))
    evec = np.asarray(evec, dtype=np.float64)
    evals = np.asarray(eval, dtype=np.float64).ravel()
    if isinstance(X, str):
        if stand not in STANDARDISE:
            raise ValueError("When using PLINK data, you must use stand='binom' or 'binom2'")
        n = count_fam_rows(X + ".fam")
        if check_fam and n != evec.shape[0]:
            raise ValueError("The number of rows in %s.fam and evec don't match" % X)
        ctx = Context.from_bed(X + ".bed", n, stand=stand, device=device, accum="auto")
    else:
        Xm = np.asarray(X, dtype=np.float64)
        if stand not in _lib.STANDARDISE_DENSE:
            raise ValueError("stand must be one of %s" % sorted(_lib.STANDARDISE_DENSE))
        if stand in ("binom", "binom2") and not np.all(np.isin(Xm[~np.isnan(Xm)], (0.0, 1.0, 2.0))):
            raise ValueError("Your data contains values other than {0, 1, 2}, stand='binom'/'binom2' can't be used here")
        if evec.shape[0] != Xm.shape[0]:
            raise ValueError("The number of rows in X and evec don't match")
        ctx = Context.from_dense(Xm, stand=stand, device=device)
    if evec.ndim != 2 or evec.shape[1] != evals.size:
        ctx.close()
        raise ValueError("The number of columns of evec doesn't match the number of eigenvalues eval")
    with ctx:
        err, mse, rmse = ctx.check(evec, evals, div=divisor)
    return dict(err=err, mse=mse, rmse=rmse)
