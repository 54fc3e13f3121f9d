"""Round 4: the mixed-precision solver at 500,000 x 100,000 on the easy and the slowly converging spectrum (k = 20)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

N, P, k = 500000, 100000, 20
V = int(os.environ.get("V", "0"))
for npop in (40, 4, 10):
    with fp.Context.synthetic(N, P, n_pop=npop, accum="auto") as c:
        c.stats()
        c.pca(ndim=k, allow_unconverged=True, max_applies=3)
        ref = None
        for label, kw in (("exact", dict(mixed=-1)), ("mixed S=4", dict()), ("mixed S=3", dict(cheap_slices=3)), ("mixed S=5", dict(cheap_slices=5))):
            for rep in range(2):
                t0 = time.perf_counter()
                r = c.pca(ndim=k, verbose=V if rep == 0 else 0, **kw)
                c.synchronize()
                w = time.perf_counter() - t0
            i = r["info"]
            if ref is None:
                ref = r["d"].copy()
            e, m, rm = c.check(r["U"], r["d"])
            print("n_pop=%d %-10s wall %.3f s  applies %d (cheap %d)  restarts %d  conv %d  resid %.2e  apply %.3f (exact %.3f) ortho %.3f host %.3f  dev of d %.2e  exact max resid/d %.2e" % (
                npop, label, w, i["block_applies"], i["cheap_applies"], i["restarts"], i["converged"], i["max_residual"], i["seconds_apply"],
                i["seconds_exact"], i["seconds_ortho"], i["seconds_host"], np.max(np.abs(r["d"] - ref) / ref), np.max(np.sqrt(e) / r["d"])), flush=True)
            del r
