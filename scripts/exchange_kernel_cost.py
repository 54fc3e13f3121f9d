"""Round 6: what the row-sharded operand exchange costs on the device -- per-kernel durations (rocprofv3 --kernel-trace) of a few
passes of the row-sharded solver forced on ONE rank over RCCL (test-hook build) at 500,000 x 100,000: k_colmax / k_slice_rows over
the rows a rank owns (here: all of them; a rank of G owns 1 / G), k_unpack_slices and k_dequant_rows over all rows (replicated on
every rank), against k_colmax + k_slice of the one-GPU path (what every rank ran over all rows in rounds 3-5).
usage: python scripts/exchange_kernel_cost.py"""
import collections, csv, glob, os, subprocess, sys, tempfile
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp
    with fp.test_hooks(), fp.Context.synthetic(500000, 100000, n_pop=4, accum="auto") as c:
        if sys.argv[2] == "sharded":
            c.comm_init_rank(1, 0, fp.Context.comm_unique_id())
        c.pca(ndim=20, max_applies=24, allow_unconverged=True)
    sys.exit(0)
for which in ("sharded", "plain"):
    env = dict(os.environ, TMPDIR="/tmp")
    if which == "sharded":
        env["FPCA_FORCE_ROWSHARD"] = "1"
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "child", which],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True, env=env)
        rows = list(csv.DictReader(open(glob.glob(tmp + "/**/*kernel_trace.csv", recursive=True)[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        for key in ("k_colmax", "k_slice_rows", "k_slice<", "k_unpack_slices", "k_dequant_rows", "k_maxbits", "AllGather", "ReduceScatter", "k_gemm_i8", "k_i8_combine", "k_sparse_rows_sum"):
            if key in n:
                agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%s solver, 24 passes (us per launch; the 24 passes are ~10 exact ones on 7 slices and ~14 on 4):" % which)
    for k, v in sorted(agg.items()):
        v.sort()
        print("   %-20s median %9.1f us   10 %% %9.1f   90 %% %9.1f   x %d" % (k, v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10], len(v)), flush=True)
