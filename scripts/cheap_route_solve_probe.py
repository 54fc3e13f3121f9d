import sys, time
sys.path.insert(0, ".")
import numpy as np
import flashpca_amd as fp
for rate in (0.003, 0.005):
    with fp.Context.synthetic(500000, 100000, n_pop=4, missing_rate=rate, accum="auto") as c:
        c.pca(ndim=20, max_applies=3, allow_unconverged=True)
        ref = c.pca(ndim=20, mixed=-1)
        t0 = time.time(); r = c.pca(ndim=20); w = time.time() - t0
        print("missing %.1f %%: route %d  slow solve %.3f s, %d passes (%d cheap), apply %.3f s; eig vs all-exact %.2e; all-exact apply %.3f s" % (
            100 * rate, c.missing_mode(16), w, r["info"]["block_applies"], r["info"]["cheap_applies"], r["info"]["seconds_apply"],
            np.max(np.abs(r["d"] - ref["d"]) / ref["d"]), ref["info"]["seconds_apply"]), flush=True)
