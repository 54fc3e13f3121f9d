"""Where does the sparse missing-indicator route stop paying?  Block apply at cfg3 for b = 16 and 32, missing-call rates
0.1 ... 4 %, route forced sparse (FPCA_I8_MODE=3) and dense (0); each point in its own process (the mode is read per call,
the context set-up is not)."""
import os, subprocess, sys
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import flashpca_amd as fp
    mr, b = float(sys.argv[2]), int(sys.argv[3])
    with fp.Context.synthetic(500000, 100000, n_pop=40, missing_rate=mr, accum="i8") as c:
        c.stats()
        c.bench_apply(b=b, steps=2, warmup=1)
        r = c.bench_apply(b=b, steps=6, warmup=1)
    print("RESULT %.3f %.3f" % (r["ms_xt"], r["ms_x"]))
    sys.exit(0)
for b in (16, 32):
    for mr in (0.001, 0.003, 0.005, 0.01, 0.02, 0.04):
        row = []
        for mode in ("3", "0"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(mr), str(b)], env=dict(os.environ, FPCA_I8_MODE=mode),
                                 capture_output=True, text=True)
            l = [x for x in out.stdout.splitlines() if x.startswith("RESULT")]
            row.append(l[-1].split()[1:] if l else ["err", out.stderr[-200:]])
        print("b=%2d missing %.3f   sparse K2/K3 %s ms   dense K2/K3 %s ms" % (b, mr, "/".join(row[0]), "/".join(row[1])), flush=True)
