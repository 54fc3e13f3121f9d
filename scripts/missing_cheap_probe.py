"""Round 6: the eigensolver's 4-slice passes on data whose missing calls take the dense (two-matrix) route -- uniform 0.5 / 1 / 2 % --
with round 5's smallest instantiated column blocks (4 tiles for K2, 3 for K3: zero padding multiplied) and round 6's (2 tiles); and
the exact apply for reference.  Needs the test-hook build (FPCA_I8_LO_R5).  500,000 x 100,000, 16 columns.
usage: python scripts/missing_cheap_probe.py"""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import numpy as np

    import flashpca_amd as fp

    rate = float(sys.argv[2])
    out = {}
    with fp.test_hooks():
        for S in (4, 7):
            with fp.Context.synthetic(500000, 100000, n_pop=40, missing_rate=rate, accum="i8x%d" % S) as c:
                c.bench_apply(b=16, steps=2, warmup=1)
                r = c.bench_apply(b=16, steps=8, warmup=2)
                out["S%d" % S] = dict(mode=c.missing_mode(16), apply=round(r["ms_xt"] + r["ms_x"], 3), k2=round(r["ms_gemm_xt"], 3), k3=round(r["ms_gemm_x"], 3))
        if len(sys.argv) > 3:
            with fp.Context.synthetic(500000, 100000, n_pop=4, missing_rate=rate, accum="auto") as c:
                c.pca(ndim=20, max_applies=3, allow_unconverged=True)
                import time
                t0 = time.time()
                r = c.pca(ndim=20)
                out["pca_slow"] = dict(wall=round(time.time() - t0, 4), passes=r["info"]["block_applies"], cheap=r["info"]["cheap_applies"])
    print(json.dumps(out))
    sys.exit(0)
for rate in (0.005, 0.02):
    for lo in ("r5", "r6 32-row waves", "r6"):
        env = dict(os.environ)
        if lo == "r5":
            env["FPCA_I8_LO_R5"] = "1"
        if lo == "r6 32-row waves":
            env["FPCA_I8_NARROW_MT1"] = "1"
        a = [sys.executable, os.path.abspath(__file__), "child", str(rate)] + (["pca"] if rate == 0.02 else [])
        o = subprocess.run(a, env=env, capture_output=True, text=True)
        print("missing %.1f %%  shapes %s: %s" % (100 * rate, lo, o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:]), flush=True)
