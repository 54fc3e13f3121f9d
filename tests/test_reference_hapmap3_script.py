"""The reference's own end-to-end test, HapMap3/test_pca.R, restated step for step against this CLI (GPU).

The R script (reference HapMap3/test_pca.R:1-246) runs `flashpca` five times -- PCA with loadings and mean/sd at
--precision 20, projection onto the same data, projection of a DIFFERENT fileset (1000 Genomes) onto the HapMap3 axes,
projection from a MAF file, --check -- and compares every output with a dense `svd` of the scaled matrix, all with
err.tol = 1e-6 and up to sign.  The same filesets are committed as tests/golden/hm3_thinned.* and kg_thinned.*; numpy's
dense SVD plays R's `svd` (line 34), and the acceptance formulas are the script's (lines 121-246).
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ERR_TOL = 1e-6  # test_pca.R:120
K = 10  # :32
TOL = 1e-6  # :33


def read_plink(prefix):
    """plink2R::read_plink(impute="none") (test_pca.R:7-11): N x P dosage of A1 with NaN for missing, fam, bim."""
    fam = [l.split() for l in open(prefix + ".fam").read().splitlines()]
    bim = [l.split() for l in open(prefix + ".bim").read().splitlines()]
    n, p = len(fam), len(bim)
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)[3:].reshape(p, (n + 3) // 4)
    codes = np.empty((p, raw.shape[1] * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (raw >> (2 * s)) & 3
    codes = codes[:, :n].T
    bed = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, np.where(codes == 3, 0.0, np.nan)))
    return bed, fam, bim


def scale2(X):
    """test_pca.R:13-24."""
    p = np.nansum(X, axis=0) / (2 * np.sum(~np.isnan(X), axis=0))
    center, scale = 2 * p, np.sqrt(2 * p * (1 - p))
    S = (X - center) / scale
    S[np.isnan(S)] = 0
    return S, center, scale


def sign_rmse(A, B):
    """sqrt(sum_m min(mean(a_m - b_m)^2, mean(a_m + b_m)^2)) -- the script's up-to-sign measure (e.g. :151-160)."""
    r = [min(np.mean(A[:, m] - B[:, m]) ** 2, np.mean(A[:, m] + B[:, m]) ** 2) for m in range(A.shape[1])]
    return np.sqrt(np.sum(r))


def table(path):
    lines = open(path).read().splitlines()
    head = lines[0].split("\t")
    rows = [l.split("\t") for l in lines[1:]]
    return head, [r[:2] for r in rows], np.array([r[2:] for r in rows], dtype=float)


def test_hapmap3_test_pca_script(tmp_path, built_lib):
    import flashpca_amd as fp

    cli = fp.CLI_PATH
    hm3, kg = os.path.join(GOLD, "hm3_thinned"), os.path.join(GOLD, "kg_thinned")

    def run(args):
        r = subprocess.run([cli] + args, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stdout

    bed1, fam1, bim1 = read_plink(hm3)
    bed2, fam2, bim2 = read_plink(kg)
    X, center, scale = scale2(bed1)
    X = X / np.sqrt(X.shape[1])  # :28
    U, d, Vt = np.linalg.svd(X, full_matrices=False)  # :34  s1 <- svd(X)
    U, d, V = U[:, :K], d[:K], Vt[:K].T
    maf = np.nanmean(bed1, axis=0) / 2  # :37
    # The script writes a 2-column "SNP MAF" table (:38-39), but the code it drives reads PLINK's 6-column .frq layout
    # (read_MAF, data.cpp:419-499: CHR SNP A1 A2 MAF NCHROBS, SNP ids checked against the .bim) and would reject the
    # script's file with "inconsistent number of columns" -- as this CLI does.  The .frq layout is written here.
    with open(tmp_path / "maf.txt", "w") as f:
        f.write(" CHR SNP A1 A2 MAF NCHROBS\n")
        for b, m in zip(bim1, maf):
            f.write("%s %s %s %s %.20g %d\n" % (b[0], b[1], b[4], b[5], m, 2 * bed1.shape[0]))
    with open(tmp_path / "maf2col.txt", "w") as f:
        f.write("SNP MAF\n")
        for b, m in zip(bim1, maf):
            f.write("%s %.20g\n" % (b[1], m))

    # :41-45  PCA
    run(["--bfile", hm3, "--ndim", str(K), "--tol", str(TOL), "--outload", "loadings.txt", "--outmeansd", "meansd.txt",
         "--precision", "20"])
    # :47-53  projection onto the same data
    run(["--bfile", hm3, "--project", "--inmeansd", "meansd.txt", "--outproj", "projections.txt", "--inload", "loadings.txt",
         "-v", "--precision", "20"])
    # :55-61  projection of the other fileset
    run(["--bfile", kg, "--project", "--inmeansd", "meansd.txt", "--outproj", "projections.1kg.txt", "--inload", "loadings.txt",
         "-v", "--precision", "20"])
    # :63-69  projection from MAF
    run(["--bfile", hm3, "--project", "--inmaf", "maf.txt", "--outproj", "projections.maf.txt", "--inload", "loadings.txt",
         "-v", "--precision", "20"])
    # :71-78  checking mode; the awk picks "eval: <v>, ... sum squared error: <sse>" per dimension
    out = run(["--bfile", hm3, "--check", "--outval", "eigenvalues.txt", "--outvec", "eigenvectors.txt", "-v", "--precision", "20",
               "--notime"])
    chk = [l for l in out.splitlines() if l.startswith("eval")]
    assert len(chk) == K
    sse_obs = np.array([float(l.replace(",", " ").split()[5]) for l in chk])  # awk -F", | " $2 = value, $7 = sse
    eval_chk = np.array([float(l.replace(",", " ").split()[1]) for l in chk])

    _, ids_vec, evec = table(tmp_path / "eigenvectors.txt")
    evals = np.loadtxt(tmp_path / "eigenvalues.txt")
    _, ids_load, loadings = table(tmp_path / "loadings.txt")
    _, ids_pcs, pcs = table(tmp_path / "pcs.txt")
    pve = np.loadtxt(tmp_path / "pve.txt")
    _, ids_msd, msd = table(tmp_path / "meansd.txt")
    _, ids_proj, proj = table(tmp_path / "projections.txt")
    _, ids_kg, proj_kg = table(tmp_path / "projections.1kg.txt")
    _, _, proj_maf = table(tmp_path / "projections.maf.txt")

    # :98-111  identifiers
    assert ids_vec == [f[:2] for f in fam1] and ids_pcs == ids_vec and ids_proj == ids_vec
    assert ids_kg == [f[:2] for f in fam2]
    assert ids_msd == [[b[1], b[4]] for b in bim1] and ids_load == ids_msd

    # :113-118  expected --check values
    XXU = X @ (X.T @ evec)
    sse_exp = np.sum((XXU - evec * evals) ** 2, axis=0)
    # :122-139  scaling
    assert np.sqrt(np.mean((center - msd[:, 0]) ** 2)) < ERR_TOL
    assert np.sqrt(np.std((scale - msd[:, 1]) ** 2, ddof=1)) < ERR_TOL
    # :141-150  eigenvalues
    assert np.sqrt(np.mean((d ** 2 - evals) ** 2)) < ERR_TOL
    assert np.allclose(eval_chk, evals, rtol=1e-5)  # printed at the stream default of 6 significant digits
    # :152-166  eigenvectors, :168-183 PCs, :196-211 loadings (all up to sign)
    assert sign_rmse(U, evec) < ERR_TOL
    assert sign_rmse(X @ V, pcs) < ERR_TOL
    assert sign_rmse(V, loadings) < ERR_TOL
    # :185-194  pve
    assert np.sqrt(np.mean((d ** 2 / np.sum(X ** 2) - pve) ** 2)) < ERR_TOL
    # :213-228  projection of the training samples == PCs
    assert sign_rmse(X @ V, proj) < ERR_TOL and sign_rmse(pcs, proj) < ERR_TOL
    # projections.maf.txt is produced by the script but never compared; the code's maf2meansd (randompca.cpp:745-751)
    # sets the scale to 2 maf (1 - maf) WITHOUT the square root, so it is not the PCs -- check it against that formula
    Sm = (bed1 - 2 * maf) / (2 * maf * (1 - maf))
    Sm[np.isnan(Sm)] = 0
    assert sign_rmse(Sm @ V / np.sqrt(bed1.shape[1]), proj_maf) < ERR_TOL
    r = subprocess.run([cli, "--bfile", hm3, "--project", "--inmaf", "maf2col.txt", "--inload", "loadings.txt"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "inconsistent number of columns" in r.stderr
    # :119-121, :230-244  1000 Genomes samples on the HapMap3 axes
    S2 = (bed2 - center) / scale
    S2[np.isnan(S2)] = 0  # the CLI treats missing as 0 after scaling (data.cpp:300-320); R's scale() keeps NA -> none here
    assert sign_rmse(S2 @ V / np.sqrt(bed2.shape[1]), proj_kg) < ERR_TOL
    # :246-249  the PCA-checking output
    assert np.all((sse_obs - sse_exp) ** 2 < ERR_TOL)
