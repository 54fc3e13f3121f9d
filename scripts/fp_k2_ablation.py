"""Round 6: the fp32 K2 kernel (k_xt_b<float,1,4,128>) with parts of its inner loop compiled out (libraries built with
-DFPCA_XTB_ABL=bits into flashpca_amd/_build/ablN/; results wrong by construction): bit 0 no table gathers (with bit 1: the address
arithmetic stays), bit 2 no B-fragment reads, bit 3 no per-chunk staging / barriers, bit 4 no global loads after the first chunk.
500,000 x 100,000, 16 columns; K2 GEMM kernel ms (HIP events)."""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp
    with fp.Context.synthetic(500000, 100000, n_pop=40, accum="fp32") as c:
        c.bench_apply(b=16, steps=2, warmup=1)
        r = c.bench_apply(b=16, steps=8, warmup=2)
    print(json.dumps(dict(K2=round(r["ms_gemm_xt"], 3), K3=round(r["ms_gemm_x"], 3))))
    sys.exit(0)
names = {501: "K2 B fragment first, split groups", 502: "K2 B fragment first, whole groups", 503: "K2 B fragment last, whole groups", 401: "K3 with the fetch groups of round 5", 301: "K2 fetch group split: bursts of 4 MFMAs", 201: "K3 k-steps per group: bursts of 4 MFMAs", 2016: "K3 bursts of 16 MFMAs", 1008: "K3 with 8 m-tiles per wave", 32: "- B tile loads after the first chunk", 64: "- packed-word loads after the first chunk", 96: "- both global loads", 0: "full", 1: "- gathers and their addresses", 3: "- gathers (addresses stay)", 4: "- B fragment reads", 5: "- gathers, - B reads (MFMA + VALU-free loop + staging)",
         8: "- staging / barriers", 24: "- staging / barriers / global loads", 29: "MFMAs only"}
for ab in ([int(x) for x in sys.argv[1:]] or (0, 1, 3, 4, 5, 8, 24, 29, 0)):
    env = dict(os.environ)
    if ab:
        env["FPCA_LIB"] = os.path.abspath("flashpca_amd/_build/abl%d/libfpca.so" % ab)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    print("%2d %-58s %s" % (ab, names[ab], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
