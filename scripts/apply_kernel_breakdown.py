"""Per-kernel time inside one block apply (rocprofv3 --kernel-trace of a child that runs block applies only): which launches make up the
stage times around the two GEMMs.  usage: python scripts/apply_kernel_breakdown.py [accum=i8x4] [b=16]"""
import collections
import csv
import glob
import os
import subprocess
import sys
import tempfile

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp

    with fp.Context.synthetic(500000, 100000, n_pop=40, accum=sys.argv[2]) as c:
        c.bench_apply(b=int(sys.argv[3]), steps=20, warmup=3)
    sys.exit(0)
accum = sys.argv[1] if len(sys.argv) > 1 else "i8x4"
b = sys.argv[2] if len(sys.argv) > 2 else "16"
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "child", accum, b],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200, check=True, env=dict(os.environ, TMPDIR="/tmp"))
    fs = glob.glob(tmp + "/**/*kernel_trace.csv", recursive=True)
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
gemm = [i for i, r in enumerate(rows) if "k_gemm_i8" in r["Kernel_Name"] or "k_xt_b" in r["Kernel_Name"] or "k_x_t" in r["Kernel_Name"]]
rows = rows[gemm[-20]:]  # the last 10 applies (two GEMMs each)
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-70:]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("%s b=%s: 10 applies span %.3f ms each (kernel time may overlap: side stream)" % (accum, b, (t1 - t0) / 10e6))
for n, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-72s %3d launches  %8.3f ms per apply" % (n, cnt, ns / 10e6))
