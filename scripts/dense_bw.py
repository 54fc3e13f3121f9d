import sys, time
sys.path.insert(0, ".")
import numpy as np
import flashpca_amd as fp
N, P = 50000, 8000
rng = np.random.default_rng(0)
X = rng.integers(0, 3, size=(N, P)).astype(np.float64)
t = time.time()
ctx = fp.Context.from_dense(X, stand="binom2")
print("create (upload %.1f GB + standardise) %.2f s" % (N * P * 8 / 1e9, time.time() - t))
for b in (16, 32, 64):
    r = ctx.bench_apply(b=b, steps=10, warmup=2)
    by = N * P * 8.0
    print("b=%d: xt %.3f ms (%.2f TB/s, %.1f TFLOP/s)  x %.3f ms (%.2f TB/s, %.1f TFLOP/s)" % (
        b, r["ms_xt"], by / r["ms_xt"] / 1e9, 2.0 * N * P * b / r["ms_xt"] / 1e9, r["ms_x"], by / r["ms_x"] / 1e9, 2.0 * N * P * b / r["ms_x"] / 1e9))
