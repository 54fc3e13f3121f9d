"""Round 6: 4-slice block apply (the eigensolver's cheap passes), sparse gathers against the two-matrix kernels for the missing-call
indicator, around their break-even rate (uniform missing calls, 500,000 x 100,000, 16 columns).  Test-hook build (FPCA_I8_MODE)."""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp
    with fp.test_hooks(), fp.Context.synthetic(500000, 100000, n_pop=40, missing_rate=float(sys.argv[2]), accum="i8x4") as c:
        c.bench_apply(b=16, steps=2, warmup=1)
        r = c.bench_apply(b=16, steps=8, warmup=2)
        print(json.dumps(dict(mode=c.missing_mode(16), apply=round(r["ms_xt"] + r["ms_x"], 3))))
    sys.exit(0)
for rate in (0.001, 0.0015, 0.002, 0.003, 0.005):
    row = []
    for mode in ("3", "0"):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(rate)], env=dict(os.environ, FPCA_I8_MODE=mode), capture_output=True, text=True)
        row.append(o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-200:])
    print("missing %.2f %%: sparse %s   two-matrix %s" % (100 * rate, row[0], row[1]), flush=True)
