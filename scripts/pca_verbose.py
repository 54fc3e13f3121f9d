"""Residual history of one solve (fpca_pca verbose): python scripts/pca_verbose.py N P k [blockvec] [n_pop]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashpca_amd as fp
N, P, k = (int(a) for a in sys.argv[1:4])
bv = int(sys.argv[4]) if len(sys.argv) > 4 else 0
npop = int(sys.argv[5]) if len(sys.argv) > 5 else min(2 * k, 64)
with fp.Context.synthetic(N, P, n_pop=npop, accum="auto") as ctx:
    r = ctx.pca(ndim=k, blockvec=bv, verbose=1)
    print(r["info"])
