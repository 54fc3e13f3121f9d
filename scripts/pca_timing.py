import sys, time
sys.path.insert(0, ".")
import flashpca_amd as fp
cfgs = {"cfg2": (50000, 20000, 20), "cfg3": (500000, 100000, 20)}
for name in (sys.argv[1:] or ["cfg2"]):
    N, P, k = cfgs[name]
    for accum in ("fp64", "auto"):
        ctx = fp.Context.synthetic(N, P, n_pop=40, accum=accum)
        ctx.stats()
        for rep in range(3):
            t0 = time.perf_counter()
            r = ctx.pca(ndim=k)
            t1 = time.perf_counter()
            i = r["info"]
            print("%s %s rep %d wall %.1f ms | inside run_pca: total %.1f apply %.1f ortho %.1f host %.1f | applies %d" % (
                name, ctx.accum, rep, (t1 - t0) * 1e3, i["seconds_total"] * 1e3, i["seconds_apply"] * 1e3, i["seconds_ortho"] * 1e3, i["seconds_host"] * 1e3, i["block_applies"]), flush=True)
        ctx.close()
