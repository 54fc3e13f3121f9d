"""One solve of the slowly converging test spectrum (4 sub-populations, k = 20) at 500,000 x 100,000: python scripts/hard_spectrum_once.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashpca_amd as fp
N, P, k = 500000, 100000, 20
with fp.Context.synthetic(N, P, n_pop=4, accum="auto") as ctx:
    ctx.stats()
    ctx.pca(ndim=k, allow_unconverged=True, max_applies=3)
    for i in range(2):
        t = time.perf_counter()
        r = ctx.pca(ndim=k)
        ctx.synchronize()
        w = time.perf_counter() - t
        i_ = r["info"]
        print("wall %.3f s  applies %d  restarts %d  apply %.3f  ortho %.3f  host %.3f  converged %d  resid %.2e  d_k %.6f" % (
            w, i_["block_applies"], i_["restarts"], i_["seconds_apply"], i_["seconds_ortho"], i_["seconds_host"], i_["converged"], i_["max_residual"], r["d"][-1]))
        del r
