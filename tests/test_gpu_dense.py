"""GPU parity tests of the in-memory matrix path (fpca_create_dense): RandomPCA::pca_fast(MatrixXd&) + standardise()
(randompca.cpp:121-166, util.cpp:24-192), the path the R function flashpca(X) takes for a numeric matrix.  Mirrors
flashpcaR/tests/testthat/test_pca.R:108-163 (sd / none / center on rnorm data vs a dense eigendecomposition) and
test_standardisation.R:4-87 (all methods with and without NA)."""
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


def _dosage_matrix(golden_dir, name, fp):
    """N x P dosage matrix of a fileset with NaN for missing (what plink2R / read_bed hand to the matrix path)."""
    N = fp.count_fam_rows(os.path.join(golden_dir, name + ".fam"))
    raw = np.fromfile(os.path.join(golden_dir, name + ".bed"), dtype=np.uint8)[3:]
    npk = (N + 3) // 4
    P = raw.size // npk
    body = raw[:P * npk].reshape(P, npk)
    codes = np.empty((P, npk * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (body >> (2 * s)) & 3
    codes = codes[:, :N]
    X = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, 0.0))
    X[codes == 1] = np.nan
    return np.asfortranarray(X.T), N, P


@pytest.mark.parametrize("stand", ["none", "sd", "binom", "binom2", "center"])
@pytest.mark.parametrize("with_na", [False, True])
def test_standardise_and_operator(stand, with_na, fp, orc):
    rng = np.random.default_rng(11)
    N, P = 700, 300
    X = rng.integers(0, 3, size=(N, P)).astype(np.float64) if stand.startswith("binom") else rng.standard_normal((N, P)) * 2 + 0.5
    X[:, 7] = 1.0  # zero-variance column: util.cpp:111-113 sets it to its mean
    if with_na:
        X[rng.random((N, P)) < 0.02] = np.nan
    Xs, ms = orc.standardise(X, stand)
    with fp.Context.from_dense(X, stand=stand) as ctx:
        gms, trace = ctx.stats()
        assert np.allclose(gms, ms, rtol=1e-12, atol=1e-12)
        assert abs(trace - (Xs * Xs).sum()) <= 1e-11 * (Xs * Xs).sum()
        B = rng.standard_normal((N, 21))
        T = ctx.apply_xt(B)
        assert np.max(np.abs(T - Xs.T @ B)) <= 1e-11 * np.max(np.abs(Xs.T @ B))
        Tin = rng.standard_normal((P, 5))
        Y = ctx.apply_x(Tin)
        assert np.max(np.abs(Y - Xs @ Tin)) <= 1e-11 * np.max(np.abs(Xs @ Tin))
        Z = ctx.apply_xxt(B)
        Zr = Xs @ (Xs.T @ B)
        assert np.max(np.abs(Z - Zr)) <= 1e-11 * np.max(np.abs(Zr))


@pytest.mark.parametrize("stand", ["sd", "none", "center"])
def test_matrix_pca_vs_dense_eigen(stand, fp, orc):
    """test_pca.R:108-163: rnorm 500 x 1000, ndim 50... here ndim 20, against eigen(tcrossprod(S)/ncol(S))."""
    rng = np.random.default_rng(5)
    N, P, k = 500, 1000, 20
    X = rng.standard_normal((N, P)) * np.linspace(0.5, 3.0, P)  # separated spectrum
    X[:, :30] += 4.0 * rng.standard_normal((N, 1))
    Xs, ms = orc.standardise(X, stand)
    w, v = np.linalg.eigh(Xs @ Xs.T / P)
    w, v = w[::-1], v[:, ::-1]
    r = fp.flashpca(X, ndim=k, stand=stand, tol=1e-9)
    assert np.max(np.abs(r["values"] - w[:k]) / w[:k]) < 1e-9
    for c in range(3):
        assert abs(abs(v[:, c] @ r["vectors"][:, c]) - 1) < 1e-7
    assert np.allclose(r["pve"], w[:k] / ((Xs * Xs).sum() / P), rtol=1e-10)
    assert np.allclose(r["center"], ms[:, 0], rtol=1e-12, atol=1e-12)


def test_matrix_path_equals_plink_path(golden_dir, fp):
    """test_pca.R:45-105: the PLINK path (f3) and the in-memory path (f2) give the same PCA on the same genotypes."""
    X, N, P = _dosage_matrix(golden_dir, "data_chr1", fp)
    k = 10
    a = fp.flashpca(os.path.join(golden_dir, "data_chr1"), ndim=k, tol=1e-9, do_loadings=True)
    b = fp.flashpca(X, ndim=k, stand="binom2", tol=1e-9, do_loadings=True)
    assert np.max(np.abs(a["values"] - b["values"]) / a["values"]) < 1e-10
    assert np.allclose(a["center"], b["center"], rtol=0, atol=1e-14)
    assert np.allclose(a["scale"], b["scale"], rtol=1e-14, atol=0)
    for c in range(k):
        s = np.sign(a["vectors"][:, c] @ b["vectors"][:, c])
        assert np.max(np.abs(a["vectors"][:, c] - s * b["vectors"][:, c])) < 1e-6
        assert np.max(np.abs(a["loadings"][:, c] - s * b["loadings"][:, c])) < 1e-6


def test_dense_errors(fp):
    X = np.zeros((20, 10))
    with pytest.raises(ValueError):
        fp.flashpca(X, stand="bogus")
    import ctypes as C

    h = C.c_void_p()
    rc = fp.lib().fpca_create_dense(C.byref(h), X.ctypes.data_as(C.c_void_p), 20, 20, 10, 9, 0)
    assert rc == -1 and b"unknown standardization method" in fp.lib().fpca_last_error()
