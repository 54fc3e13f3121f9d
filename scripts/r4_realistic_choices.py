"""Round 4: the solver's automatic choices (block width 16, basis cap 24 blocks, mixed-precision passes), which were tuned on the
survey's easy generator, re-measured on the REALISTIC profile (rare-variant spectrum, concentrated missingness, 10 sub-populations)
at 500,000 x 100,000: wall of fpca_pca for k = 10 / 20 against block width and basis cap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flashpca_amd as fp

N, P = 500000, 100000
with fp.Context.synthetic(N, P, n_pop=10, realistic=True, accum="auto") as c:
    c.stats()
    ref = {}
    for k in (20, 10):
        for kw in (dict(), dict(mixed=-1), dict(blockvec=32), dict(blockvec=32, mixed=-1), dict(max_blocks=16), dict(max_blocks=32), dict(max_blocks=40), dict(blockvec=64)):
            c.pca(ndim=k, allow_unconverged=True, max_applies=3, **{a: b for a, b in kw.items() if a == "blockvec"})
            t0 = time.perf_counter()
            r = c.pca(ndim=k, allow_unconverged=True, **kw)
            c.synchronize()
            w = time.perf_counter() - t0
            i = r["info"]
            if k not in ref:
                ref[k] = r["d"].copy()
            print("k=%d %-32s wall %.3f s  passes %d (cheap %d) of width %d  restarts %d  conv %d  apply %.3f ortho %.3f host %.3f  dev of d %.1e" % (
                k, kw or "default", w, i["block_applies"], i["cheap_applies"], i["blockvec"], i["restarts"], i["converged"], i["seconds_apply"],
                i["seconds_ortho"], i["seconds_host"], np.max(np.abs(r["d"] - ref[k]) / ref[k])), flush=True)
            del r
