"""GPU parity tests of the whole path (fpca_pca through the C ABI) against the goldens and the CPU oracle.

Tolerances (fp64 path): eigenvalues 1e-9 relative (north_star asks 1e-6), eigenvectors compared up to sign like
every reference test (flashpcaR/tests/testthat/test_pca.R:29-31, HapMap3/test_pca.R:154-165).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fp(built_lib):
    import flashpca_amd

    return flashpca_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as O

    O.build()
    return O


@pytest.mark.parametrize("accum", ["fp64", "auto"])  # auto = the exact-integer path (FPCA_ACCUM_I8(8)): same tolerances
@pytest.mark.parametrize("name,k,stand", [("hapmap3_data", 10, "binom2"), ("data_chr1", 50, "binom2"),
                                          ("data_chr1", 10, "binom"), ("data_chr1", 3, "binom2")])
def test_pca_matches_golden(golden_dir, name, k, stand, accum, fp):
    g = json.load(open(os.path.join(golden_dir, "golden_%s_%s.json" % (name, stand))))
    r = fp.flashpca(os.path.join(golden_dir, name), ndim=k, stand=stand, do_loadings=True, tol=1e-8, accum=accum)
    ev = np.array(g["eigenvalues_div_p"])[:k]
    assert r["info"]["converged"] == 1
    assert np.max(np.abs(r["values"] - ev) / ev) < 1e-9
    assert np.max(np.abs(r["pve"] - np.array(g["pve"])[:k])) < 1e-11
    U = r["vectors"]
    assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-10
    U5 = np.array(g["U_first5"]).T
    for c in range(min(5, k)):
        assert abs(abs(U5[:, c] @ U[:, c]) - 1.0) < 1e-8
    # projection = U sqrt(d) (randompca.cpp:207)
    assert np.allclose(r["projection"], U * np.sqrt(r["values"]), rtol=1e-14, atol=0)
    # loadings are unit-norm right singular vectors: V = X'U / sqrt(d P) (randompca.cpp:191-204)
    V = r["loadings"]
    assert np.max(np.abs(np.sum(V * V, axis=0) - 1.0)) < 1e-8
    assert np.allclose(r["center"][:8], g["mean_first8"], rtol=0, atol=0)
    assert np.allclose(r["scale"][:8], g["sd_first8"], rtol=0, atol=0)


def test_pca_vs_oracle_reference_path(golden_dir, fp, orc):
    """Same answers as the restated reference path (Spectra-style IRLM on the CPU) at the reference's defaults."""
    name, k = "hapmap3_data", 10
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    od = orc.OracleData(os.path.join(golden_dir, name + ".bed"), N, "binom2")
    ref = orc.pca_fast(od, k, do_loadings=True)
    r = fp.flashpca(os.path.join(golden_dir, name), ndim=k, do_loadings=True, accum="fp64")
    assert np.max(np.abs(r["values"] - ref["d"]) / ref["d"]) < 1e-6  # both converged to tol 1e-6: agree far better
    assert np.max(np.abs(r["pve"] - ref["pve"])) < 1e-8
    for c in range(5):  # well-separated components; up to sign
        s = np.sign(ref["U"][:, c] @ r["vectors"][:, c])
        assert np.max(np.abs(ref["U"][:, c] * s - r["vectors"][:, c])) < 1e-5
        assert np.max(np.abs(ref["V"][:, c] * s - r["loadings"][:, c])) < 1e-5
    assert np.array_equal(r["center"], ref["meansd"][:, 0])
    assert np.array_equal(r["scale"], ref["meansd"][:, 1])


def test_check_mode(golden_dir, fp, orc):
    """fpca_check == RandomPCA::check (randompca.cpp:663-703) == the oracle's restatement."""
    name, k = "data_chr1", 10
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    bed = os.path.join(golden_dir, name + ".bed")
    with fp.Context.from_bed(bed, N) as ctx:
        r = ctx.pca(ndim=k, tol=1e-8)
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert mse < 1e-8  # README.md:207 "should be low (e.g., <1e-8)"
        od = orc.OracleData(bed, N, "binom2")
        oerr, omse, ormse = orc.check(od, r["U"], r["d"], block_size=400)
        assert abs(mse - omse) <= 1e-6 * max(omse, 1e-30) + 1e-22
        # a deliberately wrong eigenvalue must show up identically in both
        bad = r["d"].copy()
        bad[0] *= 1.01
        e2, m2, _ = ctx.check(r["U"], bad)
        o2, om2, _ = orc.check(od, r["U"], bad, block_size=400)
        assert np.allclose(e2, o2, rtol=1e-9, atol=1e-20)


def test_restart_path_and_blockvec(golden_dir, fp):
    """Small basis cap forces thick restarts; result must not change."""
    g = json.load(open(os.path.join(golden_dir, "golden_hapmap3_data_binom2.json")))
    N = fp.count_fam_rows(os.path.join(golden_dir, "hapmap3_data.fam"))
    with fp.Context.from_bed(os.path.join(golden_dir, "hapmap3_data.bed"), N) as ctx:
        ev = np.array(g["eigenvalues_div_p"])
        for kw in (dict(max_blocks=4), dict(blockvec=32), dict(blockvec=48, max_blocks=3), dict(blockvec=64)):
            r = ctx.pca(ndim=10, tol=1e-8, **kw)
            assert r["info"]["converged"] == 1
            assert np.max(np.abs(r["d"] - ev) / ev) < 1e-9
        r = ctx.pca(ndim=10, tol=1e-8, max_blocks=4)
        assert r["info"]["restarts"] >= 1
        # solves reuse the basis blocks of earlier ones (pool in the context, other widths in between): same bits
        a = ctx.pca(ndim=10, blockvec=32)
        ctx.pca(ndim=5, blockvec=16)
        b2 = ctx.pca(ndim=10, blockvec=32)
        assert np.array_equal(a["d"], b2["d"]) and np.array_equal(a["U"], b2["U"])


def test_ndim_limit_and_errors(golden_dir, fp):
    N = fp.count_fam_rows(os.path.join(golden_dir, "data_chr1.fam"))
    with fp.Context.from_bed(os.path.join(golden_dir, "data_chr1.bed"), N) as ctx:
        with pytest.raises(fp.FpcaError):
            ctx.pca(ndim=0)
        with pytest.raises(fp.FpcaError):
            ctx.pca(ndim=10, blockvec=24)
    with pytest.raises(fp.FpcaError):
        fp.Context.from_bed("/nonexistent/file.bed", 10)
    # ndim > (min(N,P)-1)/2 is refused like the CLI does (flashpca.cpp:623-633)
    rng = np.random.default_rng(0)
    packed = rng.integers(0, 256, size=(40, 25), dtype=np.uint8)
    with fp.Context.from_packed(packed, 100, 40) as ctx:
        with pytest.raises(fp.FpcaError):
            ctx.pca(ndim=20)


def test_config2_pca_end_to_end(fp, orc):
    """BASELINE config 2 (50,000 x 20,000, k=20) on the GPU; verified with the reference's own --check quantity
    computed on the GPU for all pairs and by the CPU oracle for the operator on a probe."""
    N, P, k = 50000, 20000, 20
    with fp.Context.synthetic(N, P) as ctx:
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1e-6 * r["d"] * 1.01)  # ||A u/P - d u|| <= tol * d  (Spectra's rule, tol 1e-6)
        assert np.all(np.diff(r["d"]) < 0)
        # oracle leg: one operator application of the top eigenvector on a 2000-SNP shard of the same matrix
        with fp.Context.synthetic(N, 2000, snp_begin=0) as sh:
            packed = sh.download_packed()
            od = orc.OracleData(packed=packed, N=N, P=2000, stand="binom2")
            op = orc.OracleOp(od, 500)
            y_ref = op.perform_op(r["U"][:, 0])
            y = sh.apply_xxt(r["U"][:, :1])[:, 0]
            assert np.max(np.abs(y - y_ref)) <= 1e-11 * np.max(np.abs(y_ref))


@pytest.mark.parametrize("accum", ["auto", "fp64"])
def test_config2_full_oracle_solve(fp, orc, accum):
    """the WHOLE 50,000 x 20,000 / k = 20 problem solved twice -- by the restated reference
    path on the CPU (orc_pca_fast: Spectra-style IRLM, ncv = 2k+1, tol 1e-6, one decode -> LUT -> 2 GEMV pass over all
    20,000 SNPs per Lanczos column; SNP sub-blocks dealt over the host cores) and by the GPU product in both arithmetic
    modes.  Eigenvalues <= 1e-6 relative (north_star), eigenvectors up to sign, pve, trace, mean/sd bit-equal."""
    N, P, k = 50000, 20000, 20
    with fp.Context.synthetic(N, P, accum=accum) as ctx:
        packed = ctx.download_packed()
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    ref = orc.pca_fast(od, k, tol=1e-6, nthreads=orc.host_threads())
    assert ref["nops"] >= 2 * k + 1  # at least one full Lanczos factorisation: this is the reference's cost structure
    rel = np.abs(r["d"] - ref["d"]) / ref["d"]
    assert np.max(rel) < 1e-6, rel
    assert np.max(np.abs(r["pve"] - ref["pve"])) < 1e-9
    assert abs(r["info"]["trace"] - ref["trace"]) <= 1e-12 * ref["trace"]
    assert np.array_equal(r["meansd"], ref["meansd"])
    # eigenvectors up to sign (test_pca.R:29-31); 39 structured eigenvalues, k = 20 of them: every wanted pair is isolated.
    # Both solvers stop at ||A u - theta u|| < 1e-6 theta, so vectors agree to ~tol x theta / gap
    for c in range(k):
        sgn = np.sign(ref["U"][:, c] @ r["U"][:, c])
        assert np.max(np.abs(ref["U"][:, c] * sgn - r["U"][:, c])) < 1e-5, c
        assert abs(abs(ref["U"][:, c] @ r["U"][:, c]) - 1.0) < 1e-8, c


def test_config3_oracle_residual_at_full_size(fp, orc):
    """at 500,000 x 100,000 the CPU oracle cannot run a whole solve, but it can apply the
    reference operator (svdwide.cpp:21-68) to single vectors over ALL 100,000 SNPs: the residual || X X' u / P - d u ||
    of the top two and of the k-th converged pair is computed BY THE ORACLE from the GPU's u and d (randompca.cpp:663-703
    with the oracle as the operator), and one column of the loadings by the oracle's crossprod (svdwide.cpp:122-153).
    Anything that goes wrong only at full size -- index width past 2^16 rows / 2^32 bytes, split-K plans, the XCD-aware
    grid, two-phase split rows -- shows up here, in the default arithmetic mode."""
    N, P, k = 500000, 100000, 20
    nt = orc.host_threads()
    with fp.Context.synthetic(N, P, accum="auto") as ctx:
        packed = ctx.download_packed()
        r = ctx.pca(ndim=k, do_loadings=True)
        assert r["info"]["converged"] == 1
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        op = orc.OracleOp(od, 1000, nthreads=nt)
        for c in (0, 1, k - 1):
            u = np.ascontiguousarray(r["U"][:, c])
            y = op.perform_op(u) / P
            res = np.linalg.norm(y - r["d"][c] * u)
            assert res <= 1e-6 * r["d"][c], (c, res, r["d"][c])  # Spectra's rule at tol 1e-6, judged by the oracle
            # and the GPU operator itself against the oracle's on the same vector, entry by entry
            if c == 0:
                yg = ctx.apply_xxt(r["U"][:, :1])[:, 0] / P
                assert np.max(np.abs(yg - y)) <= 1e-11 * np.max(np.abs(y))
        # statistics of all 100,000 SNPs: bit-equal (K1 vs data.cpp:257-322)
        assert np.array_equal(r["meansd"], od.meansd())
        # loadings column k-1: V = X' u / sqrt(d) / sqrt(P) (randompca.cpp:191-204) with the oracle's crossprod
        c = k - 1
        v = op.crossprod(np.ascontiguousarray(r["U"][:, c])) / np.sqrt(r["d"][c]) / np.sqrt(P)
        assert np.max(np.abs(v - r["V"][:, c])) <= 1e-11 * np.max(np.abs(v))
        assert abs(op.trace / P - r["info"]["trace"]) <= 1e-12 * r["info"]["trace"]
        # the same solve on 32-column blocks (the width bench.py's `apply_at_b32` side block times): a different kernel
        # instantiation (7 column tiles per wave), judged by the oracle the same way
        r32 = ctx.pca(ndim=k, blockvec=32)
        assert r32["info"]["converged"] == 1 and r32["info"]["blockvec"] == 32
        assert np.max(np.abs(r32["d"] - r["d"]) / r["d"]) < 1e-9
        for c in (0, k - 1):
            u = np.ascontiguousarray(r32["U"][:, c])
            res = np.linalg.norm(op.perform_op(u) / P - r32["d"][c] * u)
            assert res <= 1e-6 * r32["d"][c], (c, res)


def test_config3_wide_blocks_vs_oracle(fp, orc):
    """svdwide.cpp:71-118 (perform_op_mat) at 500,000 x 100,000 on 32-COLUMN blocks, in both arithmetics.  fpca_apply_xxt
    pads to a multiple of 16 columns, so 1-2 column probes only ever reach the 16-column kernels; bench.py also times the
    32-column instantiation (I8Cfg with 7 column tiles, k_xt_b<double,2,4>), which must meet the oracle at size too:
    oracle perform_op on columns 0, 17 and 31 entry by entry, every column against the 16-column path, and the two halves
    of the operator alone (crossprod svdwide.cpp:122-153, prod svdwide.cpp:193-226)."""
    N, P, b = 500000, 100000, 32
    nt = orc.host_threads()
    rng = np.random.default_rng(32)
    B = rng.standard_normal((N, b))
    Tin = rng.standard_normal((P, b))
    got = {}
    packed = None
    for accum in ("auto", "fp64"):
        with fp.Context.synthetic(N, P, accum=accum) as ctx:
            if packed is None:
                packed = ctx.download_packed()
            Z = ctx.apply_xxt(B)
            Z16 = np.hstack([ctx.apply_xxt(B[:, :16]), ctx.apply_xxt(B[:, 16:])])
            assert np.max(np.abs(Z - Z16)) <= 1e-12 * np.max(np.abs(Z)), accum  # 32-column kernels == 16-column kernels
            got[accum] = (Z, ctx.apply_xt(B), ctx.apply_x(Tin))
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    op = orc.OracleOp(od, 1000, nthreads=nt)
    for c in (0, 17, 31):
        y = op.perform_op(np.ascontiguousarray(B[:, c]))
        for accum in got:
            assert np.max(np.abs(got[accum][0][:, c] - y)) <= 1e-11 * np.max(np.abs(y)), (accum, c)
    t = op.crossprod(np.ascontiguousarray(B[:, 17]))
    y = op.prod(np.ascontiguousarray(Tin[:, 17]))
    for accum in got:
        assert np.max(np.abs(got[accum][1][:, 17] - t)) <= 1e-11 * np.max(np.abs(t)), accum
        assert np.max(np.abs(got[accum][2][:, 17] - y)) <= 1e-11 * np.max(np.abs(y)), accum
    # all 32 columns of the two arithmetics against each other
    for i in range(3):
        a, f = got["auto"][i], got["fp64"][i]
        assert np.max(np.abs(a - f)) <= 1e-12 * np.max(np.abs(f)), i


def test_config5_shard_widest_blocks_vs_oracle(fp, orc):
    """One GPU's shard of BASELINE configs[4] (1,000,000 samples x 25,000 of 200,000 SNPs) on 64-COLUMN blocks: the K2 plan
    with two column blocks per row tile and the 64-wide partial planes (`bench.py --workload cfg5 / cfg5shard` times them)
    meet the oracle at N = 10^6: one column entry by entry through perform_op / crossprod / prod, all 64 columns of the
    exact-integer path against the fp64 kernels and against the 16-column path."""
    N, Pg, b = 1000000, 25000, 64
    nt = orc.host_threads()
    rng = np.random.default_rng(64)
    B = rng.standard_normal((N, b))
    Tin = rng.standard_normal((Pg, b))
    got = {}
    packed = None
    for accum in ("auto", "fp64"):
        with fp.Context.synthetic(N, Pg, snp_begin=3 * Pg, n_pop=64, accum=accum) as ctx:
            if packed is None:
                packed = ctx.download_packed()
            Z = ctx.apply_xxt(B)
            Z16 = np.hstack([ctx.apply_xxt(B[:, j:j + 16]) for j in range(0, b, 16)])
            assert np.max(np.abs(Z - Z16)) <= 1e-12 * np.max(np.abs(Z)), accum
            got[accum] = (Z, ctx.apply_xt(B), ctx.apply_x(Tin))
    od = orc.OracleData(packed=packed, N=N, P=Pg, stand="binom2")
    op = orc.OracleOp(od, 1000, nthreads=nt)
    for c in (0, 37, 63):
        y = op.perform_op(np.ascontiguousarray(B[:, c]))
        for accum in got:
            assert np.max(np.abs(got[accum][0][:, c] - y)) <= 1e-11 * np.max(np.abs(y)), (accum, c)
    t = op.crossprod(np.ascontiguousarray(B[:, 37]))
    y = op.prod(np.ascontiguousarray(Tin[:, 37]))
    for accum in got:
        assert np.max(np.abs(got[accum][1][:, 37] - t)) <= 1e-11 * np.max(np.abs(t)), accum
        assert np.max(np.abs(got[accum][2][:, 37] - y)) <= 1e-11 * np.max(np.abs(y)), accum
    for i in range(3):
        a, f = got["auto"][i], got["fp64"][i]
        assert np.max(np.abs(a - f)) <= 1e-12 * np.max(np.abs(f)), i


@pytest.mark.parametrize("k,accum", [(100, "auto"), (200, "fp64"), (478, "auto")])
def test_more_components_than_the_block_width(golden_dir, fp, k, accum):
    """ndim > 64 up to the reference's limit (min(N,P)-1)/2 = 478 on HapMap3 (flashpca.cpp:623-633): against the dense
    eigendecomposition to 1e-9, eigenvectors through orthonormality and the per-pair residual, loadings unit-norm."""
    name = "hapmap3_data"
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    with fp.Context.from_bed(os.path.join(golden_dir, name + ".bed"), N, accum=accum) as ctx:
        packed = ctx.download_packed()
        P = ctx.P
        r = ctx.pca(ndim=k, do_loadings=True)
        assert r["info"]["converged"] == 1 and r["info"]["blockvec"] == (32 if k <= 128 else 64)  # automatic width
        with pytest.raises(fp.FpcaError):
            ctx.pca(ndim=479)
    from oracle import oracle as O

    X = O.OracleData(packed=packed, N=N, P=P, stand="binom2").dense()
    w = np.linalg.eigvalsh(X @ X.T)[::-1][:k] / P
    assert np.max(np.abs(r["d"] - w) / w) < 1e-9
    assert np.max(np.abs(r["U"].T @ r["U"] - np.eye(k))) < 1e-9
    res = np.linalg.norm(X @ (X.T @ r["U"]) / P - r["U"] * r["d"], axis=0)
    assert np.max(res / r["d"]) < 2e-6
    assert np.max(np.abs(r["Px"] - r["U"] * np.sqrt(r["d"]))) < 1e-12
    Vref = X.T @ r["U"] / np.sqrt(r["d"]) / np.sqrt(P)
    assert np.max(np.abs(r["V"] - Vref)) < 1e-10
    assert np.max(np.abs(np.sum(r["V"] ** 2, axis=0) - 1.0)) < 1e-5


def test_fp32_mode_tolerance_study(golden_dir, fp):
    """Config-5 style tolerance study at fixture size: eigenvalues of the mixed fp32 path vs the dense golden.
    north_star: eigenvalues within 1e-6 relative."""
    for name, k in (("hapmap3_data", 10), ("data_chr1", 20)):
        g = json.load(open(os.path.join(golden_dir, "golden_%s_binom2.json" % name)))
        r = fp.flashpca(os.path.join(golden_dir, name), ndim=k, accum="fp32")
        ev = np.array(g["eigenvalues_div_p"])[:k]
        err = np.max(np.abs(r["values"] - ev) / ev)
        assert r["info"]["converged"] == 1
        assert err < 1e-6, err
        U5 = np.array(g["U_first5"]).T
        for c in range(5):
            assert abs(abs(U5[:, c] @ r["vectors"][:, c]) - 1.0) < 1e-5


@pytest.mark.parametrize("mode,tol", [("i8", 1e-9), ("i8x6", 1e-9), ("i8x4", 1e-6)])
def test_i8_mode_pca_vs_golden(golden_dir, fp, mode, tol):
    """FPCA_ACCUM_I8(S): S = 8 and 6 reproduce the fp64 eigenvalues to the golden's own accuracy; S = 4 (28-bit operand)
    still meets north_star's 1e-6."""
    for name, k in (("hapmap3_data", 10), ("data_chr1", 20)):
        g = json.load(open(os.path.join(golden_dir, "golden_%s_binom2.json" % name)))
        r = fp.flashpca(os.path.join(golden_dir, name), ndim=k, accum=mode)
        ev = np.array(g["eigenvalues_div_p"])[:k]
        assert r["info"]["converged"] == 1
        assert np.max(np.abs(r["values"] - ev) / ev) < tol
        U5 = np.array(g["U_first5"]).T
        for c in range(5):
            assert abs(abs(U5[:, c] @ r["vectors"][:, c]) - 1.0) < 1e-5


def test_config3_full_size_properties(fp):
    """BASELINE config 3 (500,000 x 100,000, k=20) at full size, where no CPU oracle run is possible: size-independent
    properties of the operator (symmetry, linearity, shard additivity) and the reference's own --check quantity
    (randompca.cpp:663-703) for the converged pairs; plus fp64 vs mixed-fp32 agreement (tolerance study)."""
    N, P, k = 500000, 100000, 20
    rng = np.random.default_rng(3)
    u = rng.standard_normal((N, 2))
    with fp.Context.synthetic(N, P) as ctx:
        Au = ctx.apply_xxt(u)
        lin = ctx.apply_xxt(u @ np.array([[2.0], [-0.5]]))
        assert np.max(np.abs(lin[:, 0] - (2.0 * Au[:, 0] - 0.5 * Au[:, 1]))) <= 1e-11 * np.max(np.abs(lin))
        s1, s2 = u[:, 0] @ Au[:, 1], Au[:, 0] @ u[:, 1]
        assert abs(s1 - s2) <= 1e-10 * max(abs(s1), np.linalg.norm(Au[:, 0]) * np.linalg.norm(u[:, 1]) * 1e-3)
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1 and r["info"]["block_applies"] <= 12
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-6 * r["d"])
        d64 = r["d"]
    # the same matrix in two SNP shards: partial products add up (svdwide.cpp:48-62), i.e. the multi-GPU identity
    with fp.Context.synthetic(N, 60000, snp_begin=0) as a, fp.Context.synthetic(N, 40000, snp_begin=60000) as b:
        s = a.apply_xxt(u[:, :1]) + b.apply_xxt(u[:, :1])
        assert np.max(np.abs(s[:, 0] - Au[:, 0])) <= 1e-11 * np.max(np.abs(Au[:, 0]))
    with fp.Context.synthetic(N, P, accum="fp32") as c32:
        r32 = c32.pca(ndim=k)
        assert np.max(np.abs(r32["d"] - d64) / d64) < 1e-7
    with fp.Context.synthetic(N, P, accum="i8") as c8:
        z = c8.apply_xxt(u[:, :1])
        assert np.max(np.abs(z[:, 0] - Au[:, 0])) <= 1e-12 * np.max(np.abs(Au[:, 0]))  # exact-integer path == fp64 path
        r8 = c8.pca(ndim=k)
        assert r8["info"]["converged"] == 1
        assert np.max(np.abs(r8["d"] - d64) / d64) < 1e-10


def test_config4_eight_way_shards_vs_oracle(fp, orc):
    """BASELINE configs[3]: 500,000 x 100,000 SNP-sharded across 8 GPUs = 12,500 SNPs per rank.  Every one of the eight shard
    contexts (the plan / instantiation `bench.py --workload cfg4shard` times: a 12,500-row K2, a 12,500-deep K3) on one GPU:
    the eight partial products of a 16-column block add up to the whole matrix's (svdwide.cpp:48-62, what the all-reduce of the
    8-rank run sums), shard 5's product and its X'B / X T halves meet the oracle on the same 12,500 SNPs entry by entry, and
    its statistics are bit-equal."""
    N, P, G, b = 500000, 100000, 8, 16
    per = P // G
    rng = np.random.default_rng(44)
    B = rng.standard_normal((N, b))
    with fp.Context.synthetic(N, P, accum="auto") as whole:
        Z = whole.apply_xxt(B)
    acc = np.zeros_like(Z)
    for g in range(G):
        with fp.Context.synthetic(N, per, snp_begin=g * per, accum="auto") as sh:
            sh.set_total_snps(P)
            Zg = sh.apply_xxt(B)
            acc += Zg
            if g == 5:
                packed = sh.download_packed()
                od = orc.OracleData(packed=packed, N=N, P=per, stand="binom2")
                op = orc.OracleOp(od, 1000, nthreads=orc.host_threads())
                for c in (0, 9, 15):
                    y = op.perform_op(np.ascontiguousarray(B[:, c]))
                    assert np.max(np.abs(Zg[:, c] - y)) <= 1e-11 * np.max(np.abs(y)), c
                t = op.crossprod(np.ascontiguousarray(B[:, 9]))
                assert np.max(np.abs(sh.apply_xt(B[:, 9:10])[:, 0] - t)) <= 1e-11 * np.max(np.abs(t))
                Tin = rng.standard_normal((per, 1))
                y = op.prod(np.ascontiguousarray(Tin[:, 0]))
                assert np.max(np.abs(sh.apply_x(Tin)[:, 0] - y)) <= 1e-11 * np.max(np.abs(y))
                assert np.array_equal(sh.stats()[0], od.meansd())
    assert np.max(np.abs(acc - Z)) <= 1e-12 * np.max(np.abs(Z))


def test_config5_full_size_properties(fp):
    """BASELINE config 5 (1,000,000 x 200,000, k=50 -> b=64) at full size on ONE GPU (50 GB packed + the sample-major
    copy): the default exact-integer path against the fp64 kernels on the same block, operator symmetry, the multi-GPU
    shard identity with the config's own 8-way SNP split taken two shards at a time, and a converged k=50 solve whose
    pairs pass the reference's --check quantity.  64 sub-populations as in bench.py: with fewer than k structured
    eigenvalues the tail of the spectrum sits in the bulk and ANY Krylov method needs hundreds of passes."""
    N, P, k = 1000000, 200000, 50
    rng = np.random.default_rng(5)
    u = rng.standard_normal((N, 64))  # 64 columns: the b = 64 kernels `bench.py --workload cfg5` times, at full size
    with fp.Context.synthetic(N, P, n_pop=64, accum="fp64") as c64:
        A64 = c64.apply_xxt(u)
    with fp.Context.synthetic(N, P, n_pop=64, accum="auto") as ctx:
        assert ctx.accum == "i8x7"
        Au = ctx.apply_xxt(u)
        assert np.max(np.abs(Au - A64)) <= 1e-12 * np.max(np.abs(A64))
        A2 = ctx.apply_xxt(u[:, :2])  # the 16-column kernels on the same vectors
        assert np.max(np.abs(A2 - Au[:, :2])) <= 1e-12 * np.max(np.abs(Au[:, :2]))
        r64 = ctx.pca(ndim=k, blockvec=64)
        assert r64["info"]["converged"] == 1 and r64["info"]["blockvec"] == 64
        s1, s2 = u[:, 0] @ Au[:, 1], Au[:, 0] @ u[:, 1]
        assert abs(s1 - s2) <= 1e-10 * np.linalg.norm(Au[:, 0]) * np.linalg.norm(u[:, 1])
        r = ctx.pca(ndim=k)
        # automatic block width: 16 columns, the k = 50 Ritz vectors span four blocks
        assert r["info"]["converged"] == 1 and r["info"]["blockvec"] == 16 and r["info"]["block_applies"] <= 16
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-6 * r["d"])
        d8, U8 = r["d"], r["U"]
        assert np.max(np.abs(r64["d"] - d8) / d8) < 1e-9  # 5 passes of 64 columns == 9 passes of 16
    # configs[4] names "fp32 accumulate (tolerance study)": the mixed fp32 mode at FULL size -- eigenvalues against the
    # exact-integer result (north_star: 1e-6 relative), the operator's entry-wise error on a probe, and the --check quantity
    with fp.Context.synthetic(N, P, n_pop=64, accum="fp32") as c32:
        A32 = c32.apply_xxt(u[:, :1])
        op_err = np.max(np.abs(A32[:, 0] - A64[:, 0])) / np.max(np.abs(A64[:, 0]))
        assert op_err < 5e-6, op_err
        r32 = c32.pca(ndim=k)
        assert r32["info"]["converged"] == 1
        ev_err = np.max(np.abs(r32["d"] - d8) / d8)
        assert ev_err < 1e-6, ev_err
        print("config 5 fp32 tolerance study: operator max rel err %.3e, eigenvalue max rel err %.3e" % (op_err, ev_err))
        for c in range(k):
            assert abs(abs(U8[:, c] @ r32["U"][:, c]) - 1.0) < 1e-6, c
    per = P // 8  # config 5's shard: 25,000 SNPs per GPU
    acc = np.zeros(N)
    for g in range(8):
        with fp.Context.synthetic(N, per, snp_begin=g * per, n_pop=64, accum="auto") as sh:
            acc += sh.apply_xxt(u[:, :1])[:, 0]
    assert np.max(np.abs(acc - Au[:, 0])) <= 1e-11 * np.max(np.abs(Au[:, 0]))


@pytest.mark.parametrize("accum", ["auto", "fp64"])
def test_product_reproduces_survey_known_answers(golden_dir, fp, accum):
    """The known answers printed in SURVEY.md 8(c) (a third, independent computation): eigenvalues, trace, pve, and the
    text the CLI would write at precision 7."""
    from test_oracle_golden import SURVEY_KAT

    for name, kat in SURVEY_KAT.items():
        k = 50 if "eig50" in kat else 10
        r = fp.flashpca(os.path.join(golden_dir, name), ndim=k, tol=1e-8, accum=accum)
        n = len(kat["eig"])
        assert np.max(np.abs(r["values"][:n] - kat["eig"]) / np.array(kat["eig"])) < 1e-10
        if "eig50" in kat:
            assert abs(r["values"][49] - kat["eig50"]) < 1e-10 * kat["eig50"]
        if "pve1" in kat:
            assert abs(r["pve"][0] - kat["pve1"]) < 1e-11
            assert ["%.7g" % v for v in r["values"]] == kat["eigenvalues_txt"]


def test_randomised_end_to_end_sweep(built_lib):
    """scripts/fuzz_pca.py: 40 random small problems (N, P, k up to the reference's limit, standardisation, divisor,
    arithmetic mode, duplicated samples, fewer samples than three blocks) against numpy's dense eigendecomposition:
    eigenvalues, orthonormality, per-pair residual, Px, pve, loadings."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_pca.py"), "40", "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 40 cases ok" in r.stdout


def test_config3_dense_missing_route_at_full_size(fp, orc):
    """2 % missing calls at 500,000 x 100,000: above 0.5 % the exact-integer path multiplies BOTH integer matrices (dosage and
    missing indicator) on the matrix cores -- the two-matrix kernels with the 16-column remainder that `bench.py` times as
    `apply_at_missing_2pct` -- instead of gathering sparse rows.  16 columns against the fp64 kernels, one of them and a
    crossprod against the oracle, and the padded / masked rows stay exactly zero."""
    N, P, b = 500000, 100000, 16
    rng = np.random.default_rng(2)
    B = rng.standard_normal((N, b))
    with fp.Context.synthetic(N, P, missing_rate=0.02, accum="auto") as ctx:
        assert ctx.accum == "i8x7" and ctx.missing_mode(b) == 0  # dense indicator route
        packed = ctx.download_packed()
        Z = ctx.apply_xxt(B)
        T = ctx.apply_xt(B[:, :1])
    with fp.Context.synthetic(N, P, missing_rate=0.02, accum="fp64") as c64:
        Z64 = c64.apply_xxt(B)
    assert np.max(np.abs(Z - Z64)) <= 1e-12 * np.max(np.abs(Z64))
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    op = orc.OracleOp(od, 1000, nthreads=orc.host_threads())
    y = op.perform_op(np.ascontiguousarray(B[:, 5]))
    assert np.max(np.abs(Z[:, 5] - y)) <= 1e-11 * np.max(np.abs(y))
    t = op.crossprod(np.ascontiguousarray(B[:, 0]))
    assert np.max(np.abs(T[:, 0] - t)) <= 1e-11 * np.max(np.abs(t))


def test_config2_slow_spectrum_oracle_solve(fp, orc):
    """BASELINE configs[1] sizes with only 4 sub-populations: 17 of the 20 wanted eigenvalues sit in the bulk, the restated
    reference path (Spectra-style IRLM, ncv = 41) needs several implicit restarts and the GPU solver several thick restarts with
    its rate-based test placement -- the two must still agree on the eigenvalues (north_star: 1e-6 relative; both stop at a
    residual of 1e-6 theta, so they agree far better), on pve and on the trace."""
    N, P, k = 50000, 20000, 20
    with fp.Context.synthetic(N, P, n_pop=4, accum="auto") as ctx:
        packed = ctx.download_packed()
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1 and r["info"]["restarts"] >= 1
        # the default solver is the mixed-precision one: most passes on 4 byte slices (the 2-tile int8 kernels), the verdict on
        # exact residuals -- fpca_check recomputes them with the exact operator
        assert 0 < r["info"]["cheap_applies"] < r["info"]["block_applies"] and r["info"]["cheap_slices"] == 4
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-6 * r["d"])
        rex = ctx.pca(ndim=k, mixed=-1)
        assert rex["info"]["cheap_applies"] == 0 and np.max(np.abs(r["d"] - rex["d"]) / rex["d"]) < 1e-9
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    ref = orc.pca_fast(od, k, tol=1e-6, nthreads=orc.host_threads())
    assert ref["nops"] > 3 * (2 * k + 1)  # (the reference restarts too: this is the slow case for it as well)
    rel = np.abs(r["d"] - ref["d"]) / ref["d"]
    assert np.max(rel) < 1e-6, rel
    assert np.max(np.abs(r["pve"] - ref["pve"])) < 1e-9
    assert abs(r["info"]["trace"] - ref["trace"]) <= 1e-12 * ref["trace"]
    for c in range(3):  # the three structured pairs are isolated: eigenvectors up to sign
        assert abs(abs(ref["U"][:, c] @ r["U"][:, c]) - 1.0) < 1e-8, c


def test_config2_realistic_data_oracle_solve(fp, orc):
    """50,000 x 20,000, k = 20 on the REALISTIC profile (synth.hpp: rare-variant allele-frequency spectrum -- per-SNP sd over a
    16x range --, missing calls concentrated in 5 % of the SNPs at 10-30 %, 10 sub-populations = 9 structured eigenvalues and 11
    in the bulk): the whole problem solved by the restated reference (data.cpp:257-322 statistics with the mean over non-missing
    calls, Spectra-style IRLM) and by the GPU in the default arithmetic with its default mixed-precision solver.  Eigenvalues,
    pve, trace, the per-SNP statistics bit for bit, the reference's --check quantity on the GPU's pairs."""
    N, P, k = 50000, 20000, 20
    with fp.Context.synthetic(N, P, n_pop=10, realistic=True, accum="auto") as ctx:
        packed = ctx.download_packed()
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1
        err, mse, rmse = ctx.check(r["U"], r["d"])
        assert np.all(np.sqrt(err) <= 1.01e-6 * r["d"])
        rex = ctx.pca(ndim=k, mixed=-1)  # every pass exact: the cheap passes must not have moved anything
        assert np.max(np.abs(r["d"] - rex["d"]) / rex["d"]) < 1e-9
    od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
    ref = orc.pca_fast(od, k, tol=1e-6, nthreads=orc.host_threads())
    assert np.array_equal(r["meansd"], od.meansd(), equal_nan=True)  # (the oracle's statistics exist after its first pass)
    sd = od.meansd()[:, 1]
    assert np.nanmin(sd[sd > 1e-9]) < 0.06 and np.mean(sd < 0.3) > 0.3  # (the spectrum the profile promises)
    rel = np.abs(r["d"] - ref["d"]) / ref["d"]
    assert np.max(rel) < 1e-6, rel
    assert np.max(np.abs(r["pve"] - ref["pve"])) < 1e-9
    assert abs(r["info"]["trace"] - ref["trace"]) <= 1e-12 * ref["trace"]
    for c in range(5):
        assert abs(abs(ref["U"][:, c] @ r["U"][:, c]) - 1.0) < 1e-7, c


def test_config3_realistic_data_oracle_columns(fp, orc):
    """500,000 x 100,000 on the realistic profile: one operator column, X'b and X t against the oracle over all 100,000 SNPs
    (rare variants: K3's row scales 1/sd up to ~30; concentrated missingness: 5 % of the SNPs lose 10-30 % of their calls),
    statistics bit-equal, the default solve judged by the oracle's residual on the first and the k-th pair."""
    N, P, k = 500000, 100000, 20
    nt = orc.host_threads()
    rng = np.random.default_rng(7)
    with fp.Context.synthetic(N, P, n_pop=10, realistic=True, accum="auto") as ctx:
        packed = ctx.download_packed()
        od = orc.OracleData(packed=packed, N=N, P=P, stand="binom2")
        op = orc.OracleOp(od, 1000, nthreads=nt)
        b1 = rng.standard_normal((N, 1))
        t1 = rng.standard_normal((P, 1))
        y = op.perform_op(np.ascontiguousarray(b1[:, 0]))
        assert np.max(np.abs(ctx.apply_xxt(b1)[:, 0] - y)) <= 1e-11 * np.max(np.abs(y))
        t = op.crossprod(np.ascontiguousarray(b1[:, 0]))
        assert np.max(np.abs(ctx.apply_xt(b1)[:, 0] - t)) <= 1e-11 * np.max(np.abs(t))
        y = op.prod(np.ascontiguousarray(t1[:, 0]))
        assert np.max(np.abs(ctx.apply_x(t1)[:, 0] - y)) <= 1e-11 * np.max(np.abs(y))
        r = ctx.pca(ndim=k)
        assert r["info"]["converged"] == 1
        assert np.array_equal(r["meansd"], od.meansd(), equal_nan=True)
        for c in (0, k - 1):
            u = np.ascontiguousarray(r["U"][:, c])
            res = np.linalg.norm(op.perform_op(u) / P - r["d"][c] * u)
            assert res <= 1e-6 * r["d"][c], (c, res, r["d"][c])

