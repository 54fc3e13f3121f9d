"""Whole solves at 500,000 x 100,000, k = 20, default options: the easy spectrum (40 sub-populations), the slow one (4) and the
realistic profile; wall and the phases of fpca_pca_info.  K4=0 runs round 4's orthogonalisation kernels.
usage: python scripts/solve_profiles.py [K4 variant 0|1] [reps]"""
import sys
import time

import flashpca_amd as fp

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fp.lib().fpca_debug_variant(0, variant)
N, P, k = 500000, 100000, 20
for name, kw in (("easy", dict(n_pop=40)), ("slow", dict(n_pop=4)), ("realistic", dict(n_pop=10, realistic=True))):
    with fp.Context.synthetic(N, P, accum="auto", **kw) as c:
        c.pca(ndim=k, max_applies=3, allow_unconverged=True)  # set-up of the arithmetic, buffers
        for rep in range(reps):
            t0 = time.time()
            r = c.pca(ndim=k)
            wall = time.time() - t0
            i = r["info"]
            print("K4=%d %-9s wall %.4f s  passes %3d (%3d cheap)  apply %.4f  ortho %.4f  host %.4f  download %.4f  restarts %d" % (
                variant, name, wall, i["block_applies"], i["cheap_applies"], i["seconds_apply"], i["seconds_ortho"], i["seconds_host"],
                i["seconds_download"], i["restarts"]), flush=True)
