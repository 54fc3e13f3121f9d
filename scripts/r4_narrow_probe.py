"""Round 4 probe: the narrow column blocks of the one-matrix int8 GEMM (2 / 3 tiles: S = 4 or 5 slices of a 16-column block) and
the number of co-resident workgroups per CU.  One process per setting (the switches are read once):
   FPCA_LIB=testhooks [FPCA_I8_MT4=1] [FPCA_I8_LDS_PAD=bytes] python scripts/r4_narrow_probe.py S [b]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tag = "S=%d b=%d MT4=%s LDS_PAD=%s" % (S, b, os.environ.get("FPCA_I8_MT4", "-"), os.environ.get("FPCA_I8_LDS_PAD", "-"))
# correctness first, on a small ragged problem against the fp64 kernels
N, P = 5003, 3001
with fp.Context.synthetic(N, P, n_pop=6, accum="i8x%d" % S) as c, fp.Context.synthetic(N, P, n_pop=6, accum="fp64") as r:
    B = np.random.default_rng(1).standard_normal((N, b))
    Y, Yr = c.apply_xxt(B), r.apply_xxt(B)
    err = np.max(np.abs(Y - Yr)) / np.max(np.abs(Yr))
    print("%s: small-problem operator error vs fp64 kernels %.2e (missing mode %d)" % (tag, err, c.missing_mode(b)), flush=True)
    assert err < 10.0 ** (-2.2 * S + 1), err
N, P = 500000, 100000
with fp.Context.synthetic(N, P, n_pop=40, accum="i8x%d" % S) as c:
    c.stats()
    c.bench_apply(b=b, steps=3, warmup=2)
    r = c.bench_apply(b=b, steps=20, warmup=2)
    print("%s: apply %.3f ms (stages %.3f / %.3f, GEMM kernels %.3f / %.3f)" % (
        tag, r["ms_total"] / 20, r["ms_xt"], r["ms_x"], r["ms_gemm_xt"], r["ms_gemm_x"]), flush=True)
