"""Randomised sweep of the in-memory-matrix path (GPU): flashpca(matrix, stand=...) and project() on random matrices with
NaNs, constant columns and every standardisation against numpy (standardise() semantics of util.cpp:24-192).
python scripts/fuzz_dense.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def standardise(X, stand):
    """util.cpp:24-192: column mean / sd over the non-missing entries (sd with n-1), binom/binom2 from the mean,
    NaN -> 0 after scaling; a column whose sd is 0 / not finite is set to zero."""
    X = X.copy()
    n_good = np.sum(~np.isnan(X), axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.nansum(X, axis=0) / n_good
        if stand == "none":  # util.cpp:34-58: no scaling, missing entries take the column mean
            Z, center, scale = np.where(np.isnan(X), mean, X), np.zeros(X.shape[1]), np.ones(X.shape[1])
        elif stand == "center":
            Z, center, scale = X - mean, mean, np.ones(X.shape[1])
        else:
            if stand == "sd":
                sd = np.sqrt(np.nansum((X - mean) ** 2, axis=0) / (n_good - 1))
            elif stand == "binom":
                sd = np.sqrt(mean / 2 * (1 - mean / 2))
            else:
                sd = np.sqrt(2 * (mean / 2) * (1 - mean / 2))
            Z, center, scale = (X - mean) / sd, mean, sd
    Z[np.isnan(Z)] = 0.0
    return Z, center, scale


t0 = time.time()
for case in range(ncases):
    N = int(rng.integers(40, 1200))
    P = int(rng.integers(20, 1500))
    k = int(min((min(N, P) - 1) // 2, rng.choice([1, 3, 10, 20])))
    stand = str(rng.choice(["none", "center", "sd", "binom", "binom2"]))
    div = str(rng.choice(["p", "n1", "none"]))
    npop = int(rng.integers(2, 8))
    pop = rng.integers(0, npop, size=N)
    if stand in ("binom", "binom2"):
        f = np.clip(rng.uniform(0.1, 0.9, size=(1, P)) + 0.2 * rng.standard_normal((npop, P)), 0.05, 0.95)
        X = rng.binomial(2, f[pop]).astype(float)
    else:
        X = rng.standard_normal((N, P)) * rng.uniform(0.5, 3, size=P) + rng.standard_normal((npop, P))[pop] * 2 + rng.uniform(-5, 5, size=P)
    X[rng.random(X.shape) < float(rng.choice([0.0, 0.01]))] = np.nan
    Z, center, scale = standardise(X, stand)
    ok_cols = np.isfinite(scale) & (scale > 1e-9) & np.all(np.isfinite(Z), axis=0)
    if not np.all(ok_cols):
        continue  # degenerate columns: covered by the unit tests, conventions differ between util.cpp branches
    dv = {"p": P, "n1": N - 1, "none": 1}[div]
    w, v = np.linalg.eigh(Z @ Z.T / dv)
    w = w[::-1]
    desc = dict(N=N, P=P, k=k, stand=stand, div=div)
    try:
        r = fp.flashpca(X, ndim=k, stand=stand, divisor=div, tol=1e-8, maxiter=3000, do_loadings=True)
    except Exception as e:
        print("case", case, desc, "EXCEPTION", e, flush=True)
        raise
    d, U, V = r["values"], r["vectors"], r["loadings"]
    scale_w = max(w[0], 1e-300)
    e_val = float(np.max(np.abs(d - w[:k])) / scale_w)
    e_res = float(np.max(np.linalg.norm(Z @ (Z.T @ U) / dv - U * d, axis=0)) / scale_w)
    e_c = float(np.max(np.abs(r["center"] - center) / np.maximum(1.0, np.abs(center)))) if stand != "none" else 0.0
    e_s = float(np.max(np.abs(r["scale"] - scale) / scale)) if stand in ("sd", "binom", "binom2") else 0.0
    proj = fp.project(X, V, orig_mean=center, orig_sd=scale, divisor=div)["projection"]
    pdiv = {"p": P, "n1": N, "none": 1}[div]  # project.R:137-142 divides by n for "n1" (flashpca itself by n - 1)
    e_p = float(np.max(np.abs(proj * np.sqrt(pdiv / dv) - r["projection"])) / np.sqrt(scale_w))
    if stand == "none" and np.isnan(X).any():
        e_p = 0.0  # project.R imputes to orig_mean (0 here), the "none" standardisation to the column mean: not comparable
    ok = r["info"]["converged"] == 1 and e_val < 1e-7 and e_res < 1e-6 and e_c < 1e-12 and e_s < 1e-12 and e_p < 1e-6
    if not ok or case % 10 == 0:
        print("case %3d %s applies %d eval %.1e resid %.1e center %.1e scale %.1e project %.1e %s" % (
            case, desc, r["info"]["block_applies"], e_val, e_res, e_c, e_s, e_p, "OK" if ok else "FAIL"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases ok, %.0f s" % (ncases, time.time() - t0))
