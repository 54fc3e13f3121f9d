"""flashpcaR/tests/testthat/test_pca.R restated against flashpca_amd.flashpca() (GPU): the five test_that blocks, with the
script's own comparison helpers (compare_scales, compare_eigenvecs) and its tolerance tol = 1e-4.

f1 = dense eigen(tcrossprod(S)/ncol(S)) [numpy eigh plays R's eigen()], f2 = flashpca(matrix S, stand="none"),
f3 = flashpca(PLINK fileset, stand=...).  hm3.chr1$bed of the R package is the same genotype matrix as the
inst/extdata/data_chr1 fileset (tests/golden/data_chr1.*).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BEDF = os.path.join(GOLD, "data_chr1")
N, P, NDIM, TOL = 500, 1000, 50, 1e-4  # test_pca.R:3-6


def hm3_chr1_bed():
    fam = open(BEDF + ".fam").read().splitlines()
    n = len(fam)
    raw = np.fromfile(BEDF + ".bed", dtype=np.uint8)[3:]
    p = raw.size // ((n + 3) // 4)
    raw = raw.reshape(p, -1)
    codes = np.empty((p, raw.shape[1] * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (raw >> (2 * s)) & 3
    codes = codes[:, :n].T
    return np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, np.where(codes == 3, 0.0, np.nan)))


def scale2(X, type_):
    """flashpcaR::scale2 (R/scale2.R): type "1" = binom (sd sqrt(p(1-p))), "2" = binom2 (sd sqrt(2p(1-p)))."""
    p = np.nansum(X, axis=0) / (2 * np.sum(~np.isnan(X), axis=0))
    center = 2 * p
    scale = np.sqrt(p * (1 - p)) if type_ == "1" else np.sqrt(2 * p * (1 - p))
    S = (X - center) / scale
    S[np.isnan(S)] = 0
    return S, center, scale


def dense_eigen(S):
    w, v = np.linalg.eigh(S @ S.T / S.shape[1])
    w, v = w[::-1], v[:, ::-1]
    return dict(values=w, vectors=v, projection=v[:, :NDIM] * np.sqrt(w[:NDIM]))


def compare_scales(center, scale, *fits):  # test_pca.R:13-22
    for f in fits:
        assert np.allclose(center, f["center"], rtol=TOL, atol=0)
        assert np.allclose(scale, f["scale"], rtol=TOL, atol=0)


def compare_eigenvecs(*l):  # test_pca.R:24-43
    for x in l[1:]:
        cor = np.array([abs(np.corrcoef(l[0][:, j], x[:, j])[0, 1]) for j in range(l[0].shape[1])])
        assert np.allclose(cor, 1.0, atol=TOL)
    r = np.stack([np.sum(x * x, axis=0) for x in l], axis=1)
    assert np.allclose(np.var(r, axis=1, ddof=1), 0.0, atol=TOL)


@pytest.fixture(scope="module")
def fp(built_lib):
    import flashpca_amd

    return flashpca_amd


@pytest.mark.parametrize("stand,type_", [("binom", "1"), ("binom2", "2")])
def test_pca_with_stand_binom(fp, stand, type_):  # test_pca.R:45-105
    S, center, scale = scale2(hm3_chr1_bed(), type_)
    f1 = dense_eigen(S)
    f2 = fp.flashpca(S, ndim=NDIM, stand="none")
    f3 = fp.flashpca(BEDF, ndim=NDIM, stand=stand)
    compare_scales(center, scale, f3)
    compare_eigenvecs(f1["vectors"][:, :NDIM], f2["vectors"], f3["vectors"])
    compare_eigenvecs(f1["projection"], f2["projection"], f3["projection"])
    pve = f1["values"] / np.sum(f1["values"])
    assert np.allclose(pve[:NDIM], f2["pve"], rtol=1.5e-8, atol=1.5e-8)  # expect_equal's default tolerance
    assert np.allclose(pve[:NDIM], f3["pve"], rtol=1.5e-8, atol=1.5e-8)


@pytest.mark.parametrize("stand", ["sd", "none", "center"])
def test_pca_with_matrix_standardisations(fp, stand):  # test_pca.R:108-165
    X = np.random.default_rng({"sd": 1, "none": 2, "center": 3}[stand]).standard_normal((N, P))
    center = X.mean(axis=0) if stand != "none" else np.zeros(P)
    scale = X.std(axis=0, ddof=1) if stand == "sd" else np.ones(P)
    S = (X - center) / scale
    f1 = dense_eigen(S)
    f2 = fp.flashpca(X, ndim=NDIM, stand=stand)
    if stand != "none":
        compare_scales(center, scale, f2)
    compare_eigenvecs(f1["projection"], f2["projection"])
    pve = f1["values"] / np.sum(f1["values"])
    assert np.allclose(pve[:NDIM], f2["pve"], rtol=1.5e-8, atol=1.5e-8)
