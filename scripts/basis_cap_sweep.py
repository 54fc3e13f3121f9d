"""Thick-restart basis cap (fpca_pca_opts.max_blocks) against wall-clock of the slow and the realistic solve at 500,000 x 100,000,
k = 20: fewer restarts and passes with a deeper basis, dearer orthogonalisation and Rayleigh-Ritz per pass.
usage: PYTHONPATH=. python scripts/basis_cap_sweep.py [caps ...]   (default 20 24 28 32 40)"""
import sys
import time

import flashpca_amd as fp

caps = [int(a) for a in sys.argv[1:]] or [20, 24, 28, 32, 40]
N, P, k = 500000, 100000, 20
for name, kw in (("slow", dict(n_pop=4)), ("realistic", dict(n_pop=10, realistic=True))):
    with fp.Context.synthetic(N, P, accum="auto", **kw) as c:
        c.pca(ndim=k, max_applies=3, allow_unconverged=True)  # set-up of the arithmetic, buffers
        for cap in caps:
            best = None
            for rep in range(2):
                t0 = time.time()
                r = c.pca(ndim=k, max_blocks=cap)
                wall = time.time() - t0
                if best is None or wall < best[0]:
                    best = (wall, r["info"])
            wall, i = best
            print("%-9s cap %2d  wall %.4f s  passes %3d (%3d cheap)  apply %.4f  ortho %.4f  host %.4f  restarts %d  converged %d" % (
                name, cap, wall, i["block_applies"], i["cheap_applies"], i["seconds_apply"], i["seconds_ortho"], i["seconds_host"],
                i["restarts"], i["converged"]), flush=True)
