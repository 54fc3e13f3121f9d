"""Basis cap (blocks before a thick restart) against wall-clock on a slowly converging spectrum (4 sub-populations, k = 20)."""
import sys, time
sys.path.insert(0, ".")
import flashpca_amd as fp
cfgs = {"cfg2": (50000, 20000), "cfg3": (500000, 100000)}
for name in (sys.argv[1:] or ["cfg3"]):
    N, P = cfgs[name]
    with fp.Context.synthetic(N, P, n_pop=4, accum="auto") as ctx:
        ctx.stats()
        ctx.pca(ndim=20, allow_unconverged=True, max_applies=4)
        for mb in [int(x) for x in __import__("os").environ.get("MBS", "0,8,12,16,24,32,48").split(",")]:
            t0 = time.perf_counter()
            r = ctx.pca(ndim=20, allow_unconverged=True, max_blocks=mb)
            dt = time.perf_counter() - t0
            i = r["info"]
            print("%s max_blocks %2d: wall %.3f s  applies %3d restarts %2d  apply %.3f ortho %.3f host %.3f  converged %d" % (
                name, mb, dt, i["block_applies"], i["restarts"], i["seconds_apply"], i["seconds_ortho"], i["seconds_host"], i["converged"]), flush=True)
