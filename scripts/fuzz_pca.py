"""Randomised end-to-end sweep (GPU): fpca_pca on small random genotype matrices against numpy's dense eigendecomposition of
X X'/div -- random N, P, k (up to the reference's limit, beyond the block width of 64 included), standardisation, divisor, block width, rank-deficient inputs
(duplicated samples, few SNPs).  python scripts/fuzz_pca.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def standardised(packed, N, stand):
    P = packed.shape[0]
    codes = np.empty((P, packed.shape[1] * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (packed >> (2 * s)) & 3
    codes = codes[:, :N]
    G = np.select([codes == 0, codes == 2, codes == 3], [2.0, 1.0, 0.0], np.nan).T
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.nansum(G, axis=0) / np.sum(~np.isnan(G), axis=0)
        p = mean / 2
        sd = np.sqrt(2 * p * (1 - p)) if stand == "binom2" else np.sqrt(p * (1 - p))
        X = (G - mean) / sd
    X[:, ~(sd > 1e-9)] = 0.0
    X[np.isnan(X)] = 0.0
    return X


t0 = time.time()
for case in range(ncases):
    N = int(rng.integers(8, 1500))
    P = int(rng.integers(8, 2500))
    kmax = (min(N, P) - 1) // 2
    # (70 .. 300: more components than the block width -- several blocks of Ritz vectors, or the dense route when N is small)
    k = int(min(kmax, rng.choice([1, 2, 5, 10, 20, 40, 70, 130, 300])))
    if k < 1:
        continue
    stand = str(rng.choice(["binom2", "binom"]))
    div = str(rng.choice(["p", "n1", "none"]))
    accum = str(rng.choice(["auto", "fp64", "i8x6"]))
    npop = int(rng.integers(1, 12))
    # population-structured frequencies so that the leading eigenvalues are separated
    pop = rng.integers(0, npop, size=N)
    f = np.clip(rng.uniform(0.05, 0.95, size=(P, 1)) + 0.15 * rng.standard_normal((P, npop)), 0.02, 0.98)
    g = rng.binomial(2, f[:, pop])  # P x N
    codes = np.select([g == 2, g == 1], [0, 2], 3).astype(np.uint8)
    codes[rng.random(codes.shape) < float(rng.choice([0.0, 0.002, 0.05]))] = 1
    conc = rng.random() < 0.3  # missing calls concentrated in a few SNPs (the hybrid route), rare variants among the rest
    if conc:
        for j in rng.choice(P, size=max(1, P // 15), replace=False):
            codes[j, rng.random(codes.shape[1]) < rng.uniform(0.05, 0.5)] = 1
        rare = rng.choice(P, size=max(1, P // 5), replace=False)
        codes[rare] = np.where(rng.random((len(rare), codes.shape[1])) < 0.01, 2, 3).astype(np.uint8)
    mixed = int(rng.choice([0, 0, 1, -1]))  # 1: 4-slice passes from the start wherever the basis allows, verified by exact ones
    if rng.random() < 0.3 and N > 20:
        codes[:, N // 2:N // 2 + 5] = codes[:, :5]  # duplicated samples -> rank deficiency
    pad = (-N) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros((P, pad), dtype=np.uint8)], axis=1)
    packed = (codes[:, 0::4] | (codes[:, 1::4] << 2) | (codes[:, 2::4] << 4) | (codes[:, 3::4] << 6)).astype(np.uint8)
    X = standardised(packed, N, stand)
    dv = {"p": P, "n1": N - 1, "none": 1}[div]
    w, v = np.linalg.eigh(X @ X.T / dv)
    w, v = w[::-1], v[:, ::-1]
    desc = dict(N=N, P=P, k=k, stand=stand, div=div, accum=accum, npop=npop, conc=conc, mixed=mixed)
    try:
        with fp.Context.from_packed(packed, N, P, stand=stand, accum=accum) as c:
            r = c.pca(ndim=k, div=div, tol=1e-8, maxiter=2000, do_loadings=True, allow_unconverged=True, mixed=mixed)
    except Exception as e:
        print("case", case, desc, "EXCEPTION", e, flush=True)
        raise
    d, U, Px, V = r["d"], r["U"], r["Px"], r["V"]
    scale = max(w[0], 1e-300)
    e_val = float(np.max(np.abs(d - w[:k])) / scale)
    e_orth = float(np.max(np.abs(U.T @ U - np.eye(k))))
    # residual of each pair against the dense operator (sign- and degeneracy-proof)
    e_res = float(np.max(np.linalg.norm(X @ (X.T @ U) / dv - U * d, axis=0)) / scale)
    e_px = float(np.max(np.abs(Px - U * np.sqrt(np.maximum(d, 0)))))
    e_pve = float(np.max(np.abs(r["pve"] - d / (np.sum(X * X) / dv))))
    with np.errstate(invalid="ignore", divide="ignore"):
        Vref = X.T @ U / np.sqrt(d) / np.sqrt(dv)
    good = d > 1e-9 * scale
    e_v = float(np.max(np.abs(V[:, good] - Vref[:, good]))) if np.any(good) else 0.0
    ok = r["info"]["converged"] == 1 and e_val < 1e-7 and e_orth < 1e-9 and e_res < 1e-6 and e_px < 1e-9 * np.sqrt(scale) + 1e-12 and e_pve < 1e-9 and e_v < 1e-6
    if not ok or case % 10 == 0:
        print("case %3d %s applies %d (%d cheap)  eval %.1e orth %.1e resid %.1e Px %.1e pve %.1e V %.1e %s" % (
            case, desc, r["info"]["block_applies"], r["info"]["cheap_applies"], e_val, e_orth, e_res, e_px, e_pve, e_v, "OK" if ok else "FAIL"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases ok, %.0f s" % (ncases, time.time() - t0))
