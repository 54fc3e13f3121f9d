"""Block width against wall-clock on a slowly converging spectrum (4 sub-populations, k = 20)."""
import sys, time
sys.path.insert(0, ".")
import flashpca_amd as fp
cfgs = {"cfg2": (50000, 20000), "cfg3": (500000, 100000)}
for name in (sys.argv[1:] or ["cfg3"]):
    N, P = cfgs[name]
    with fp.Context.synthetic(N, P, n_pop=4, accum="auto") as ctx:
        ctx.stats()
        for bv in [int(x) for x in __import__("os").environ.get("BVS", "32,64,48").split(",")]:
            ctx.pca(ndim=20, allow_unconverged=True, max_applies=4, blockvec=bv)
            t0 = time.perf_counter()
            r = ctx.pca(ndim=20, allow_unconverged=True, blockvec=bv)
            dt = time.perf_counter() - t0
            i = r["info"]
            print("%s blockvec %2d: wall %.3f s  applies %3d (%d vector ops) restarts %2d  apply %.3f ortho %.3f host %.3f  converged %d" % (
                name, bv, dt, i["block_applies"], i["vector_ops"], i["restarts"], i["seconds_apply"], i["seconds_ortho"], i["seconds_host"], i["converged"]), flush=True)
