"""Exact-integer mode: agreement with the fp64 kernels and K2/K3 timing (run on the GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashpca_amd as fp

configs = [("cfg2", 50000, 20000, 32, 10), ("cfg3", 500000, 100000, 32, 3)]
if len(sys.argv) > 1:
    configs = [c for c in configs if c[0] in sys.argv[1:]]
for name, N, P, b, steps in configs:
    ref = fp.Context.synthetic(N, P, n_pop=40)
    rng = np.random.default_rng(0)
    B = rng.standard_normal((N, b))
    Z64 = ref.apply_xxt(B)
    r64 = ref.bench_apply(b=b, steps=steps, warmup=2)
    ref.close()
    for mode in ("i8x8", "i8", "i8x6", "i8x5", "i8x4"):
        if mode == "i8x4" and b != 64:
            pass
        ctx = fp.Context.synthetic(N, P, n_pop=40, accum=mode)
        Z = ctx.apply_xxt(B)
        err = float(np.max(np.abs(Z - Z64) / np.max(np.abs(Z64), axis=0)))
        r = ctx.bench_apply(b=b, steps=steps, warmup=2)
        ctx.close()
        print(name, mode, json.dumps(dict(err_vs_fp64=err, ms_xt=r["ms_xt"], ms_x=r["ms_x"], fp64_ms_xt=r64["ms_xt"], fp64_ms_x=r64["ms_x"],
                                          speedup=(r64["ms_xt"] + r64["ms_x"]) / (r["ms_xt"] + r["ms_x"]))), flush=True)
