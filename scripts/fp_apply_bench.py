"""The FP64- / FP32-MFMA GEMM kernels (--accum fp64 | fp32) at 500,000 x 100,000: per-kernel launch time (HIP events around the launch)
and the fraction of the matrix peak (78.6 / 157.3 TFLOP/s), at 16 and 32 columns.
usage: python scripts/fp_apply_bench.py [N] [P]"""
import sys

import flashpca_amd as fp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
for accum, peak in (("fp64", 78.6), ("fp32", 157.3)):
    with fp.Context.synthetic(N, P, n_pop=40, accum=accum) as c:
        for b, big in ((16, 0), (32, 0), (64, 0)):
            r = c.bench_apply(b=b, steps=6, warmup=2)
            fl = 2.0 * N * P * b
            print("%s b=%2d big=%d  K2 %.2f ms = %.1f TF (%.3f)  K3 %.2f ms = %.1f TF (%.3f)  stages %.2f + %.2f ms" % (
                accum, b, big, r["ms_gemm_xt"], fl / r["ms_gemm_xt"] / 1e9, fl / r["ms_gemm_xt"] / 1e9 / peak, r["ms_gemm_x"], fl / r["ms_gemm_x"] / 1e9,
                fl / r["ms_gemm_x"] / 1e9 / peak, r["ms_xt"], r["ms_x"]), flush=True)
