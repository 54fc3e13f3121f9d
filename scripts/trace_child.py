import sys, os
sys.path.insert(0, os.getcwd())
import flashpca_amd as fp
acc = sys.argv[1] if len(sys.argv) > 1 else "i8x4"
with fp.Context.synthetic(500000, 100000, n_pop=40, accum=acc) as c:
    c.bench_apply(b=16, steps=20, warmup=3)
