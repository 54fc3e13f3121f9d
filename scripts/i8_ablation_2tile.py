"""The 2-tile int8 GEMM of the 4-slice passes with parts of its main loop compiled out (library built with -DFPCA_TEST_HOOKS
-DFPCA_I8_ABLATION into flashpca_amd/_build/abl/, as for scripts/i8_ablation.py; results wrong by construction): what the packed-word
loads, the operand staging and the decode cost at 500,000 x 100,000, 16 columns, nothing missing.  python scripts/i8_ablation_2tile.py"""
import os, subprocess, sys, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp
    with fp.Context.synthetic(500000, 100000, n_pop=40, missing_rate=0.0, accum="i8x4") as c:
        c.bench_apply(b=16, steps=2, warmup=1)
        r = c.bench_apply(b=16, steps=10, warmup=2)
    print(json.dumps(dict(ms_gemm_xt=round(r["ms_gemm_xt"], 4), ms_gemm_x=round(r["ms_gemm_x"], 4))))
    sys.exit(0)
names = {0: "full loop", 1: "- operand staging", 2: "- decode", 8: "- packed-word loads", 15: "MFMAs + barrier only", 16: "operand stream L2-resident", 32: "packed words L2-resident", 48: "both L2-resident", 64: "no per-chunk barrier"}
for ab in [0, 32, 16, 48, 8, 1, 2, 15, 64, 0]:
    env = dict(os.environ, FPCA_I8_ABL=str(ab), FPCA_LIB=os.path.abspath("flashpca_amd/_build/abl/libfpca.so"))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    print("%2d %-34s %s" % (ab, names[ab], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
