"""Solver work between two applies as a function of the rows a rank keeps (row-sharded solver): seconds_ortho and
seconds_host per block apply at N = 500,000 (one GPU / replicated solver) and N = 62,500 (the slice of one of 8 ranks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashpca_amd as fp
for N in (500000, 250000, 125000, 62500):
    with fp.Context.synthetic(N, 4096, n_pop=40, accum="auto") as ctx:
        ctx.pca(ndim=20, allow_unconverged=True, max_applies=7)
        r = ctx.pca(ndim=20, allow_unconverged=True, max_applies=7)
        i = r["info"]
        print("rows %7d  b %d  applies %d: orthogonalisation %.3f ms / apply, host Rayleigh-Ritz %.3f ms / apply" % (
            N, i["blockvec"], i["block_applies"], 1e3 * i["seconds_ortho"] / i["block_applies"], 1e3 * i["seconds_host"] / i["block_applies"]))
