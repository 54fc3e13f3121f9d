"""Block apply through the HOST-pointer entry point (fpca_apply_xxt: column-major fp64 in, out) vs the device-resident one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp
for name, N, P in (("cfg2", 50000, 20000), ("cfg3", 500000, 100000)):
    b = 32
    with fp.Context.synthetic(N, P, n_pop=40, accum="auto") as c:
        B = np.asfortranarray(np.random.default_rng(0).standard_normal((N, b)))
        c.apply_xxt(B)
        reps = 20 if N < 100000 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            c.apply_xxt(B)
        th = (time.perf_counter() - t0) / reps
        r = c.bench_apply(b=b, steps=reps, warmup=1)
        td = (r["ms_xt"] + r["ms_x"]) * 1e-3
        print("%s host-pointer apply %.2f ms (%.3e cells/s) | device-resident %.2f ms (%.3e cells/s) | %d MB each way" % (
            name, th * 1e3, N * P * b / th, td * 1e3, N * P * b / td, N * b * 8 // 1000000), flush=True)
