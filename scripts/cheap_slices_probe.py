"""Round 6: the eigensolver's cheap passes on 3 byte slices (48 slice-columns = 1.5 column tiles: one 32-column tile + the 16-column
remainder on v_mfma_i32_16x16x64_i8, I8Cfg<false,2,2,...,HALF>) against 4 (64 = 2 tiles).
  1. parity of the 3-slice operator against numpy on a small matrix (error <= 2^-(8S-1) of the column maximum per element)
  2. GEMM kernel and block-apply times at 500,000 x 100,000, 16 columns, S = 3 / 4 / 7
  3. whole k = 20 solves (easy / slow / realistic) with cheap_slices = 4 and 3: wall, passes, eigenvalues against the all-exact solve
usage: python scripts/cheap_slices_probe.py [reps]"""
import sys
import time

import numpy as np

import flashpca_amd as fp
from oracle import oracle as O

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, P, b = 3001, 2000, 16
for kw, name in ((dict(n_pop=5), "sparse route"), (dict(n_pop=5, missing_rate=0.0), "nothing missing"), (dict(n_pop=5, realistic=True), "hybrid route")):
    with fp.Context.synthetic(N, P, accum="i8x3", **kw) as c:
        od = O.OracleData(packed=c.download_packed(), N=N, P=P, stand="binom2")
        X = od.dense()
        B = np.random.default_rng(3).standard_normal((N, b)) * np.logspace(-3, 3, b)[None, :]
        T, Tr = c.apply_xt(B), X.T @ B
        Tin = np.random.default_rng(4).standard_normal((P, b))
        Y, Yr = c.apply_x(Tin), X @ Tin
        print("S=3 parity (%s, mode %d): K2 %.2e  K3 %.2e  (2^-23 = %.2e)" % (
            name, c.missing_mode(b), np.max(np.abs(T - Tr) / np.max(np.abs(Tr), axis=0)), np.max(np.abs(Y - Yr) / np.max(np.abs(Yr), axis=0)), 2.0 ** -23), flush=True)

N, P, k = 500000, 100000, 20
for S in (3, 4, 7):
    with fp.Context.synthetic(N, P, n_pop=40, accum="i8x%d" % S) as c:
        c.bench_apply(b=16, steps=3, warmup=2)
        r = c.bench_apply(b=16, steps=20, warmup=3)
        print("S=%d  apply %.3f ms  K2 stage %.3f (GEMM %.3f)  K3 stage %.3f (GEMM %.3f)" % (
            S, r["ms_total"], r["ms_xt"], r["ms_gemm_xt"], r["ms_x"], r["ms_gemm_x"]), flush=True)

for name, kw in (("easy", dict(n_pop=40)), ("slow", dict(n_pop=4)), ("realistic", dict(n_pop=10, realistic=True))):
    with fp.Context.synthetic(N, P, accum="auto", **kw) as c:
        c.pca(ndim=k, max_applies=3, allow_unconverged=True)
        ref = c.pca(ndim=k, mixed=-1)
        print("%-9s all exact: passes %3d  apply %.4f s" % (name, ref["info"]["block_applies"], ref["info"]["seconds_apply"]), flush=True)
        for cs in (4, 3):
            c.pca(ndim=k, cheap_slices=cs)
            for rep in range(reps):
                t0 = time.time()
                r = c.pca(ndim=k, cheap_slices=cs)
                wall = time.time() - t0
                i = r["info"]
                print("%-9s cheap_slices %d: wall %.4f s  passes %3d (%3d cheap)  apply %.4f  ortho %.4f  host %.4f  eig vs exact %.2e  max resid %.2e" % (
                    name, cs, wall, i["block_applies"], i["cheap_applies"], i["seconds_apply"], i["seconds_ortho"], i["seconds_host"],
                    np.max(np.abs(r["d"] - ref["d"]) / ref["d"]), i["max_residual"]), flush=True)
