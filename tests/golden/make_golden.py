#!/usr/bin/env python3
"""Generate known-answer vectors for the PCA hot path with an INDEPENDENT numpy restatement.

This is deliberately not the C oracle: it is a dense, vectorised numpy computation of the
mathematical definition the reference's own tests use (flashpcaR/tests/testthat/test_pca.R:24-43
compare flashpca against R's dense eigen(tcrossprod(S)/ncol(S)); HapMap3/test_pca.R:121-246 against
svd()).  Standardisation rules follow data.cpp:215-335 of the reference:

  * sample 4*i+s of a SNP lives in bits 2s..2s+1 of byte i (data.cpp:128-148, data.h:42-45)
  * code 00 -> dosage 2, 10 -> 1, 11 -> 0, 01 -> missing           (data.cpp:65-126)
  * mean over non-missing, P = mean/2, sd = sqrt(2P(1-P)) [binom2] or sqrt(P(1-P)) [binom]
  * missing -> 0 AFTER standardisation; sd <= 1e-9 -> whole column zero (data.cpp:299-320)
  * nsnps = (filesize-3) / ceil(N/4)                                (data.cpp:150-176)
  * eigenvalues d = lambda(XX')/div, div = P by default; pve = d / (sum(X^2)/div)
                                                                     (randompca.cpp:180-207)

Run from the repo root:  python tests/golden/make_golden.py
Writes tests/golden/golden_<name>_<stand>.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def count_lines(path):
    # the reference drops a last line that has no trailing '\n' (data.cpp:526,605,655)
    with open(path, "rb") as f:
        return f.read().count(b"\n")


def load_standardised(prefix, stand):
    n = count_lines(prefix + ".fam")
    raw = np.fromfile(prefix + ".bed", dtype=np.uint8)
    magic = raw[:3].tolist()
    body = raw[3:]
    npk = (n + 3) // 4
    p = body.size // npk
    body = body[: p * npk].reshape(p, npk)
    codes = np.empty((p, npk * 4), dtype=np.uint8)
    for s in range(4):
        codes[:, s::4] = (body >> (2 * s)) & 3
    codes = codes[:, :n]  # pad bits ignored
    dosage = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, 0.0))
    miss = codes == 1
    ngood = (~miss).sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.where(miss, 0.0, dosage).sum(axis=1) / ngood
        pp = mean / 2.0
        sd = np.sqrt(2.0 * pp * (1.0 - pp)) if stand == "binom2" else np.sqrt(pp * (1.0 - pp))
        x = (dosage - mean[:, None]) / sd[:, None]
    x[miss] = 0.0
    x[~(sd > 1e-9), :] = 0.0
    counts = [int((codes == c).sum()) for c in range(4)]
    return x.T.copy(), mean, sd, n, p, magic, counts  # X is N x P


def main():
    sets = [("data_chr1", 50), ("hapmap3_data", 10)]
    for name, k in sets:
        for stand in ("binom2", "binom"):
            prefix = os.path.join(HERE, name)
            X, mean, sd, n, p, magic, counts = load_standardised(prefix, stand)
            G = X @ X.T
            w, v = np.linalg.eigh(G)
            order = np.argsort(w)[::-1]
            w = w[order]
            v = v[:, order]
            trace = float((X * X).sum())
            U = v[:, :k].copy()
            # sign convention for the stored vectors only: largest-|.| entry positive
            for c in range(k):
                if U[np.argmax(np.abs(U[:, c])), c] < 0:
                    U[:, c] = -U[:, c]
            out = {
                "name": name,
                "stand": stand,
                "N": n,
                "P": p,
                "magic": magic,
                "code_counts_00_01_10_11": counts,
                "n_monomorphic": int((~(sd > 1e-9)).sum()),
                "k": k,
                "eigenvalues_div_p": (w[:k] / p).tolist(),
                "eigenvalues_div_n1": (w[:k] / (n - 1)).tolist(),
                "eigenvalues_raw": w[:k].tolist(),
                "next_eigenvalue_raw": float(w[k]),
                "trace_raw": trace,
                "pve": (w[:k] / trace).tolist(),
                "mean_first8": mean[:8].tolist(),
                "sd_first8": sd[:8].tolist(),
                "mean_sum": float(np.nansum(mean)),
                "sd_sum": float(np.nansum(sd)),
                # eigenvectors (sign-normalised) for the first 5 components, all N rows
                "U_first5": U[:, :5].T.tolist(),
                # X X' applied to a fixed probe vector: pins the operator itself
                "probe_y_first8": None,
            }
            probe = np.cos(0.37 * np.arange(n) + 0.11) + 0.25
            y = X @ (X.T @ probe)
            out["probe_y_first8"] = y[:8].tolist()
            out["probe_y_norm"] = float(np.linalg.norm(y))
            path = os.path.join(HERE, "golden_%s_%s.json" % (name, stand))
            with open(path, "w") as f:
                json.dump(out, f, indent=0)
            print(name, stand, "N", n, "P", p, "top3/P", (w[:3] / p), "trace/P", trace / p, file=sys.stderr)


if __name__ == "__main__":
    main()
