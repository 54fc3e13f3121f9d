"""Round 4 probe: cost of a block apply for every (slice count S, width b), and how the whole solve behaves when EVERY pass
runs at a reduced slice count (easy and slowly converging spectrum) -- the data behind the mixed-precision solver."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

N, P, k = 500000, 100000, 20
ref = {}
for npop in (40, 4):
    for S in (7, 6, 5, 4, 3):
        with fp.Context.synthetic(N, P, n_pop=npop, accum="i8x%d" % S) as c:
            c.stats()
            if npop == 40:
                for b in (16, 32):
                    c.bench_apply(b=b, steps=3, warmup=2)
                    r = c.bench_apply(b=b, steps=10, warmup=2)
                    print("S=%d b=%d: apply %.3f ms (stages %.3f / %.3f, GEMM kernels %.3f / %.3f)" % (
                        S, b, r["ms_total"] / 10, r["ms_xt"], r["ms_x"], r["ms_gemm_xt"], r["ms_gemm_x"]), flush=True)
            for bv in (16, 32):
                c.pca(ndim=k, blockvec=bv, allow_unconverged=True, max_applies=3)
                t0 = time.perf_counter()
                r = c.pca(ndim=k, blockvec=bv, allow_unconverged=True, max_applies=260)
                w = time.perf_counter() - t0
                i = r["info"]
                key = (npop, bv)
                if S == 7:
                    ref[key] = r["d"].copy()
                dd = np.max(np.abs(r["d"] - ref[key]) / ref[key]) if key in ref else float("nan")
                print("n_pop=%d S=%d b=%d: wall %.3f s, %d applies, %d restarts, conv %d, resid %.2e, apply %.3f ortho %.3f host %.3f, d1/dk %.1f, max rel dev of d from S=7: %.2e" % (
                    npop, S, bv, w, i["block_applies"], i["restarts"], i["converged"], i["max_residual"], i["seconds_apply"], i["seconds_ortho"],
                    i["seconds_host"], r["d"][0] / r["d"][-1], dd), flush=True)
                del r
