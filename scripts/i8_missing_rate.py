"""int8 GEMMs vs the missing-call rate (blocks of the missing indicator E without any missing genotype are skipped)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp
N, P = (500000, 100000) if "cfg3" in sys.argv else (50000, 20000)
steps = 3 if N > 100000 else 20
B = np.random.default_rng(0).standard_normal((N, 32))
for mr in (0.0, 0.0002, 0.001, 0.005, 0.02):
    with fp.Context.synthetic(N, P, n_pop=40, missing_rate=mr, accum="fp64") as ref:
        Z0 = ref.apply_xxt(B)
    with fp.Context.synthetic(N, P, n_pop=40, missing_rate=mr, accum="i8") as c:
        Z = c.apply_xxt(B)
        err = float(np.max(np.abs(Z - Z0) / np.max(np.abs(Z0), axis=0)))
        r = c.bench_apply(b=32, steps=steps, warmup=2)
    print("missing %.4f  err_vs_fp64 %.2e  K2 %.3f ms  K3 %.3f ms" % (mr, err, r["ms_xt"], r["ms_x"]), flush=True)
