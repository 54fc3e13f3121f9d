"""flashpcaR/tests/testthat/test_check.R and test_project.R restated against flashpca_amd.check() / project() (GPU),
with the scripts' tolerances (1e-3 and 1e-5)."""
import os
import warnings

import numpy as np
import pytest

from test_reference_testthat_pca import BEDF, hm3_chr1_bed, scale2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fp(built_lib):
    import flashpca_amd

    return flashpca_amd


def bim_ref_alleles():
    rows = [l.split() for l in open(BEDF + ".bim").read().splitlines()]
    return {r[1]: r[4] for r in rows}  # names(refallele) <- bim[,2]; refallele <- bim[,5]


def test_check_with_stand_binom(fp):  # test_check.R:14-33
    ndim, tol = 50, 1e-3
    S, _, _ = scale2(hm3_chr1_bed(), "1")
    f2 = fp.flashpca(S, ndim=ndim, stand="none")
    fp.flashpca(BEDF, ndim=ndim, stand="binom")
    XXU = S @ (S.T @ f2["vectors"]) / S.shape[1]
    err = np.sum((XXU - f2["vectors"] * f2["values"]) ** 2, axis=0)
    mse = np.sum(err) / (S.shape[0] * ndim)
    c1 = fp.check(S, stand="none", evec=f2["vectors"], eval=f2["values"])
    c2 = fp.check(BEDF, stand="binom", evec=f2["vectors"], eval=f2["values"])
    assert np.allclose(err, c1["err"], atol=tol) and np.allclose(err, c2["err"], atol=tol)
    assert np.allclose(np.zeros(ndim), c1["err"], atol=tol)
    assert abs(mse - c1["mse"]) < tol and abs(mse - c2["mse"]) < tol


def test_check_input_checking(fp):  # test_check.R:35-60
    X, _, _ = scale2(hm3_chr1_bed(), "1")
    rng = np.random.default_rng(0)
    evec = rng.standard_normal((X.shape[0] + 3, 5))
    evals = rng.standard_normal(5) ** 2
    with pytest.raises(ValueError):
        fp.check(X, stand="none", evec=evec, eval=evals)
    with pytest.raises(ValueError):
        fp.check(BEDF, stand="none", evec=evec, eval=evals)
    with pytest.raises(ValueError):
        fp.check(X, stand="none", evec=evec, eval=evals[:3])
    with pytest.raises(ValueError):
        fp.check(BEDF, stand="none", evec=evec, eval=evals[:3])


def test_projection(fp):  # test_project.R:12-47
    ndim, tol = 10, 1e-5
    bed = hm3_chr1_bed()
    X1, c1, s1 = scale2(bed, "2")
    f = fp.flashpca(X1, ndim=ndim, stand="none", do_loadings=True)
    ref = bim_ref_alleles()
    pr1 = fp.project(BEDF, loadings=f["loadings"], ref_alleles=ref, orig_mean=c1, orig_sd=s1)
    assert np.allclose(f["projection"], pr1["projection"], atol=tol)
    with warnings.catch_warnings(record=True) as w:  # expect_warning: X contains missing values
        warnings.simplefilter("always")
        pr2 = fp.project(bed, loadings=f["loadings"], ref_alleles=ref, orig_mean=c1, orig_sd=s1)
        assert any("missing values" in str(x.message) for x in w)
    assert np.allclose(f["projection"], pr2["projection"], atol=tol)
    # PCA on a random half, projection of everyone
    keep = np.random.default_rng(1).random(bed.shape[0]) < 0.5
    X2, c2, s2 = scale2(bed[~keep], "2")
    ok = s2 > 0  # a SNP can be monomorphic in the half sample; R would produce NaN there, keep the comparison well defined
    f2 = fp.flashpca(X2[:, ok], ndim=ndim, stand="none", do_loadings=True)
    load = np.zeros((bed.shape[1], ndim))
    load[ok] = f2["loadings"]
    s2c = np.where(ok, s2, 1.0)
    pr3 = fp.project(BEDF, loadings=load, ref_alleles=ref, orig_mean=c2, orig_sd=s2c)
    X1s = (bed - c2) / s2c
    X1s[np.isnan(X1s)] = 0
    P2 = X1s @ load / np.sqrt(bed.shape[1])
    assert np.allclose(P2, pr3["projection"], atol=tol)


def test_projection_input_checking(fp):  # test_project.R:49-95
    X1, c1, s1 = scale2(hm3_chr1_bed(), "2")
    f = fp.flashpca(X1, ndim=10, stand="none", do_loadings=True)
    ref = bim_ref_alleles()
    names = list(ref)
    shuffled = dict(zip(names, np.random.default_rng(2).permutation(list(ref.values())).tolist()))
    with pytest.raises(ValueError):
        fp.project(BEDF, loadings=f["loadings"], ref_alleles=shuffled, orig_mean=c1, orig_sd=s1)
    with pytest.raises(ValueError):
        fp.project(BEDF, loadings=f["loadings"][:10], ref_alleles=ref, orig_mean=c1, orig_sd=s1)
    with pytest.raises(ValueError):
        fp.project(BEDF, loadings=f["loadings"], ref_alleles=ref, orig_mean=c1[:10], orig_sd=s1)
    with pytest.raises(ValueError):
        fp.project(BEDF, loadings=f["loadings"], ref_alleles=ref, orig_mean=c1, orig_sd=s1[:10])
    osd = s1.copy()
    osd[0], osd[1] = 0, -1
    with pytest.raises(ValueError):
        fp.project(BEDF, loadings=f["loadings"], ref_alleles=ref, orig_mean=c1, orig_sd=osd)


def test_standardisation_and_mean_imputation(fp):  # test_standardisation.R:4-87
    """standardise_impute(X, method) is the matrix the dense path holds after fpca_create_dense; it is read back as X_std I."""
    n, m = 50, 10
    rng = np.random.default_rng(3)
    X = rng.binomial(2, 0.3, size=(n, m)).astype(float)

    def standardise_impute(A, stand):
        with fp.Context.from_dense(A, stand=stand) as c:
            return c.apply_x(np.eye(m))

    def r_scale(A, center=True, scale=True):
        mu = np.nanmean(A, axis=0) if center else np.zeros(m)
        sd = np.nanstd(A, axis=0, ddof=1) if scale else np.ones(m)
        return (A - mu) / sd

    tol = 1.5e-8  # expect_equal
    assert np.allclose(standardise_impute(X, "none"), X, atol=tol)
    s2 = standardise_impute(X, "sd")
    assert np.allclose(s2.mean(axis=0), 0, atol=tol) and np.allclose(s2.std(axis=0, ddof=1), 1, atol=tol)
    for stand, t in (("binom", "1"), ("binom2", "2")):
        s = standardise_impute(X, stand)
        assert np.allclose(s.mean(axis=0), 0, atol=tol)
        assert np.allclose(s, scale2(X, t)[0], atol=tol)
    assert np.allclose(standardise_impute(X, "center"), r_scale(X, True, False), atol=tol)
    # missing values: imputed to the column mean
    X2 = X.copy()
    X2[np.arange(m), np.arange(m)] = np.nan
    xmean = np.nanmean(X2, axis=0)
    s7 = standardise_impute(X2, "none")
    assert np.allclose(xmean, np.diag(s7[:m]), atol=tol) and np.allclose(xmean, s7.mean(axis=0), atol=tol)
    assert not np.isnan(s7).any()
    e = r_scale(X2)
    e[np.isnan(e)] = 0
    s8 = standardise_impute(X2, "sd")
    assert np.allclose(s8.mean(axis=0), 0, atol=tol) and np.allclose(s8, e, atol=tol)
    for stand, t in (("binom", "1"), ("binom2", "2")):
        s = standardise_impute(X2, stand)
        assert np.allclose(s.mean(axis=0), 0, atol=tol)
        assert np.allclose(s, scale2(X2, t)[0], atol=tol)
    e = r_scale(X2, True, False)
    e[np.isnan(e)] = 0
    s10 = standardise_impute(X2, "center")
    assert np.allclose(s10.mean(axis=0), 0, atol=tol) and np.allclose(s10, e, atol=tol)
