"""k = 10 (the reference's default --ndim): block width 16 (smallest multiple of 16 >= k + 4) against 32."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashpca_amd as fp
for name, N, P in (("cfg2", 50000, 20000), ("cfg3", 500000, 100000)):
    with fp.Context.synthetic(N, P, n_pop=20, accum="auto") as c:
        c.stats()
        for k in (10, 5):
            for bv in (16, 32, 48):
                c.pca(ndim=k, blockvec=bv)
                t0 = time.perf_counter(); r = c.pca(ndim=k, blockvec=bv); t = time.perf_counter() - t0
                i = r["info"]
                print("%s k=%d b=%d: %.1f ms, %d applies (apply %.1f ortho %.1f host %.1f)" % (name, k, bv, t * 1e3, i["block_applies"], i["seconds_apply"] * 1e3, i["seconds_ortho"] * 1e3, i["seconds_host"] * 1e3), flush=True)
        for bv in (16, 32, 48, 64):
            r = c.bench_apply(b=bv, steps=5, warmup=1)
            print("%s apply b=%d: %.3f ms = %.3f ms per column" % (name, bv, r["ms_xt"] + r["ms_x"], (r["ms_xt"] + r["ms_x"]) / bv), flush=True)
