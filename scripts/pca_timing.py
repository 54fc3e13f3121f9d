import sys, time
sys.path.insert(0, ".")
import flashpca_amd as fp
N, P, k = 50000, 20000, 20
for accum in ("fp64", "i8"):
    ctx = fp.Context.synthetic(N, P, n_pop=40, accum=accum)
    ctx.stats()
    for rep in range(3):
        t0 = time.perf_counter()
        r = ctx.pca(ndim=k)
        t1 = time.perf_counter()
        i = r["info"]
        print("%s rep %d wall %.1f ms | inside: total %.1f apply %.1f ortho %.1f host %.1f | applies %d" % (
            accum, rep, (t1 - t0) * 1e3, i["seconds_total"] * 1e3, i["seconds_apply"] * 1e3, i["seconds_ortho"] * 1e3, i["seconds_host"] * 1e3, i["block_applies"]))
    ctx.close()
