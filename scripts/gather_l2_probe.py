"""How fast is the sparse gather when the gathered matrix is L2-resident?  N = 8192 samples (B = 2 MB), P = 1,000,000 SNPs, 1 % missing,
sparse route forced: the K2 gather reads rows of B (L2-resident), the K3 gather rows of T (256 MB: Infinity Cache / HBM).  Run under
rocprofv3 --kernel-trace --stats and compare the two k_sparse_rows_sum averages (same number of gathered rows each)."""
import os, sys
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
os.environ["FPCA_I8_MODE"] = "3"
sys.path.insert(0, ".")
import flashpca_amd as fp
N, P, b = 8192, 1000000, 32
with fp.Context.synthetic(N, P, n_pop=8, accum="i8", missing_rate=0.01) as ctx:
    ctx.stats()
    r = ctx.bench_apply(b=b, steps=6, warmup=2)
    print({k: round(v, 3) for k, v in r.items() if k.startswith("ms")}, "gathered bytes per stage ~ %.2f GB" % (N * P * 0.01 * b * 8 / 1e9))
