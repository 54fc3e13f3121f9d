"""Split-K plan of the FP64 GEMM kernels at 500,000 x 100,000, 16 columns: launch time against the number of splits (test build:
FPCA_XT_SPLITS / FPCA_X_SPLITS are read once per process, so every setting is its own process).
usage: python scripts/fp_splits_probe.py            (driver)
       python scripts/fp_splits_probe.py one b      (one setting, from the environment)"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "one":
    import flashpca_amd as fp

    b = int(sys.argv[2])
    with fp.test_hooks(), fp.Context.synthetic(500000, 100000, n_pop=40, accum=os.environ.get("ACC", "fp64")) as c:
        r = c.bench_apply(b=b, steps=5, warmup=2)
        print("%s b=%d XT_SPLITS=%s X_SPLITS=%s  K2 %.2f ms  K3 %.2f ms  stages %.2f + %.2f" % (
            os.environ.get("ACC", "fp64"), b, os.environ.get("FPCA_XT_SPLITS", "auto"), os.environ.get("FPCA_X_SPLITS", "auto"), r["ms_gemm_xt"],
            r["ms_gemm_x"], r["ms_xt"], r["ms_x"]), flush=True)
else:
    for s in ("auto", "1", "2", "3", "5", "8", "13", "20", "32", "48"):
        env = dict(os.environ, PYTHONPATH=".")
        if s != "auto":
            env["FPCA_XT_SPLITS"] = s
            env["FPCA_X_SPLITS"] = s
        subprocess.run([sys.executable, __file__, "one", "16"], env=env)
