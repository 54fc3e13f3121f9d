import sys, time
sys.path.insert(0, ".")
import numpy as np, flashpca_amd as fp
for N, P in ((50000, 20000), (500000, 100000)):
    c = fp.Context.synthetic(N, P, n_pop=40, accum="i8")
    B = np.random.default_rng(0).standard_normal((N, 32))
    t0 = time.perf_counter(); c.stats(); t1 = time.perf_counter()
    Z = c.apply_xxt(B); t2 = time.perf_counter()
    Z = c.apply_xxt(B); t3 = time.perf_counter()
    print(N, P, "stats %.3f first apply (transposed copy, lists, allocations) %.3f second %.3f s" % (t1 - t0, t2 - t1, t3 - t2))
    c.close()
