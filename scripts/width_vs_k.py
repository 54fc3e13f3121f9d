"""Which block width gives the shortest solve?  k in {10, 20, 50} x b in {16, 32, 64} on an easy (2k sub-populations) and a slowly
converging (4 sub-populations) spectrum."""
import os, sys, time
sys.path.insert(0, ".")
import flashpca_amd as fp
cfgs = {"cfg2": (50000, 20000), "cfg3": (500000, 100000)}
for name in (sys.argv[1:] or ["cfg3"]):
    N, P = cfgs[name]
    for npop in (0, 4):
        for k in (10, 20, 50):
            with fp.Context.synthetic(N, P, n_pop=(min(2 * k, 64) if npop == 0 else npop), accum="auto") as ctx:
                ctx.stats()
                for bv in (16, 32, 64):
                    ctx.pca(ndim=k, allow_unconverged=True, max_applies=4, blockvec=bv)
                    t0 = time.perf_counter()
                    r = ctx.pca(ndim=k, allow_unconverged=True, blockvec=bv)
                    dt = time.perf_counter() - t0
                    i = r["info"]
                    print("%s %s k=%2d b=%2d: wall %.3f s  applies %3d restarts %2d  apply %.3f ortho %.3f host %.3f conv %d" % (
                        name, "easy" if npop == 0 else "hard", k, bv, dt, i["block_applies"], i["restarts"], i["seconds_apply"], i["seconds_ortho"],
                        i["seconds_host"], i["converged"]), flush=True)
