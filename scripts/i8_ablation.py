"""What each part of the int8 GEMM main loop costs: the same launch with parts of the loop compiled out (library built
with -DFPCA_I8_ABLATION, results wrong by construction).  FPCA_LIB=flashpca_amd/_build/abl/libfpca.so python scripts/i8_ablation.py"""
import os, subprocess, sys, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import flashpca_amd as fp
    N, P = (500000, 100000) if os.environ.get("WL", "cfg3") == "cfg3" else (50000, 20000)
    with fp.Context.synthetic(N, P, n_pop=40, missing_rate=0.0, accum="i8") as c:
        r = c.bench_apply(b=32, steps=3 if N > 100000 else 20, warmup=2)
    print(json.dumps(dict(ms_gemm_xt=round(r["ms_gemm_xt"], 4), ms_gemm_x=round(r["ms_gemm_x"], 4))))
    sys.exit(0)
names = {0: "full loop", 1: "- operand staging (global->LDS)", 2: "- genotype decode", 4: "- LDS fragment reads", 8: "- packed-word loads",
         9: "- staging - packed loads", 6: "- decode - LDS reads", 15: "MFMAs + barrier only",
         16: "operand stream L2-resident", 32: "packed words L2-resident", 48: "both L2-resident",
         64: "no per-chunk barrier", 112: "no barrier, both streams L2-resident"}
for ab in [int(a) for a in os.environ.get("ABLS", "0,1,2,4,8,9,6,15,16,32,48").split(",")]:
    env = dict(os.environ, FPCA_I8_ABL=str(ab))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    print("%2d %-34s %s" % (ab, names[ab], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
